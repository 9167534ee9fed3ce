"""The engine's FAST arithmetic mode (hd_engine_config.arith = HD_ARITH_FAST: fused multiply-add in every FIR of the chain, kernels/arith.h) against the same
CPU oracle as the exact mode, through the C ABI.  north_star's bar: decoded sentences bit-identical, intermediate floats within 1e-5 relative -- evaluated
norm-wise per call (max|gpu - oracle| / max|oracle|; a FIR's output relative to the larger of its output's and its input's peak --
bench.fir_normwise: a low-pass that rejects an off-tune carrier leaves an output far below the terms it sums; reference arithmetic:
code/Decoder/Decimator.h:128-138, FirFilter.h:155-161):

  a. call by call on small batches through every launch path and stage plan: decimated and filtered samples <= 1e-5 norm-wise, the discriminator output
     within the bound those 1e-5 propagate to (bench.demod_excess), bits / backlog / characters / sentences IDENTICAL to the oracle's;
  b. the golden chain (tests/golden/chain_small: expectations recorded from the reference's own compiled stage classes): sentences and text identical;
  c. at bench size, free running as bench.py drives it: symbols produced, characters and sentences identical on EVERY stream of the headline workload
     (1024 streams x ~60 steps through k_step_cu) and of the other four workloads, plus the floats of sampled streams inside the step kernel.

The exact mode stays the default and is what every other test file runs."""
import json
from pathlib import Path

import numpy as np
import pytest

from habdec_amd import synth

pytestmark = pytest.mark.gpu
C = 65536
TOL = 1e-5


@pytest.fixture(scope="module")
def hd():
    import habdec_amd
    habdec_amd.lib()
    return habdec_amd


def run_fast(hd, iq, fs, *, factor, baud, bits, stops, lowpass_bw=None, lowpass_trans=None, dc_remove=False, lookup=1, ungated=False, check_every=1, chunk=C,
             pipeline=0, expect_path=None):
    """Engine in fast mode and one oracle per stream over the same pushes; returns the worst norm-wise differences seen."""
    import bench
    from oracle import pyoracle
    S, N = iq.shape
    kw = {}
    if lowpass_trans is not None:
        kw["lowpass_trans"] = lowpass_trans
    eng = hd.Engine(n_streams=S, max_chunk=chunk, sampling_rate=fs, decimation=factor, baud=baud, rtty_bits=bits, rtty_stops=stops,
                    lowpass_bw_hz=lowpass_bw if lowpass_bw is not None else 1500.0, dc_remove=dc_remove, lookup_mode=lookup, ungated=ungated, keep_filtered=True,
                    pipeline=pipeline, arith=1, **kw)
    orcs = [pyoracle.Decoder("oracle", factor=factor, baud=baud, bits=bits, stops=stops, lowpass_bw=lowpass_bw, dc_remove=dc_remove, mathh_context=lookup,
                             ungated=ungated, **kw) for _ in range(S)]
    worst = {"decimated": 0.0, "filtered": 0.0, "demod_over_bound": 0.0, "demod_abs": 0.0, "filtered_n": 0, "identical_floats": 0, "floats": 0}
    prev = [None] * S
    for k in range(N // chunk):
        piece = np.ascontiguousarray(iq[:, k * chunk:(k + 1) * chunk])
        eng.process_host(piece)
        for s in range(S):
            orcs[s](piece[s], fs)
        for s in range(S):
            o = orcs[s]
            fo = o.array("last_filtered")
            if k % check_every == 0:
                gdec, odec = eng.decimated(s), o.array("last_decimated")
                worst["decimated"] = max(worst["decimated"], bench.normwise(gdec, odec))
                gf = eng.filtered(s)
                worst["filtered"] = max(worst["filtered"], bench.fir_normwise(gf, fo, odec))
                ex, ab = bench.demod_excess(eng.demodulated(s), o.array("last_demod"), fo, prev[s], TOL, scale=float(np.max(np.abs(odec))) if odec.size else None)
                worst["demod_over_bound"] = max(worst["demod_over_bound"], ex); worst["demod_abs"] = max(worst["demod_abs"], ab)
                worst["filtered_n"] += int(fo.size)
                worst["floats"] += int(odec.size); worst["identical_floats"] += int(np.count_nonzero(gdec.view(np.uint64) == odec.view(np.uint64))) if gdec.shape == odec.shape else 0
                assert worst["decimated"] <= TOL and worst["filtered"] <= TOL, ("floats", k, s, worst)
                assert worst["demod_over_bound"] <= 1.0, ("discriminator output beyond the propagated bound", k, s, worst)
                assert np.array_equal(eng.bits(s), o.bits()), ("bits", k, s)
                assert eng.symbol_backlog(s) == o.symex_held(), ("backlog", k, s)
            if fo.size:
                prev[s] = fo[-1]
    if expect_path is not None:
        assert eng.timing()["path"] == expect_path, eng.timing()
    for s in range(S):
        o = orcs[s]
        assert eng.rtty(s) == o.text("rtty_stream"), ("rtty", s)
        assert eng.take_sentences(s) == o.sentences(), ("sentences", s)
        assert eng.take_chars(s) == o.text("chars_log"), ("chars", s)
    return eng, orcs, worst


CASES = {
    # /64 = /32 212 taps + /2 69 taps, 161-tap low-pass: the stream tail (k_tail<256,...> for a handful of streams) behind k_decimate<32,212,64>
    "D64_300_8N2": dict(fs=2.048e6, factor=64, baud=300, bits=8, stops=2, S=3, path=2),
    # /16 = /8 54 taps + /2, 3 kHz low-pass: the separate kernels (k_decimate x 2, k_fir_demod, k_symbols)
    "D16_2p5M_lp3k": dict(fs=2.5e6, factor=16, baud=300, bits=8, stops=2, S=2, lowpass_bw=3000.0, path=0),
    "D64_dc_remove": dict(fs=2.048e6, factor=64, baud=300, bits=8, stops=2, S=2, dc_remove=True, path=0),
    "D64_lookup0": dict(fs=2.048e6, factor=64, baud=300, bits=8, stops=2, S=2, lookup=0, path=2),
    "D32_600_7N1": dict(fs=1.024e6, factor=32, baud=600, bits=7, stops=1, S=2, path=2),
    "D4_ungated": dict(fs=2.048e6, factor=4, baud=300, bits=8, stops=2, S=2, ungated=True, path=0, nchunks=6),
    "D64_50_7N2": dict(fs=2.048e6, factor=64, baud=50, bits=7, stops=2, S=4, path=2, f0=[0.0, 120.0, -200.0, 1500.0], check_every=3),
}


@pytest.mark.parametrize("name", list(CASES))
def test_fast_mode_call_by_call(hd, name):
    from test_gpu_parity import make_streams
    c = dict(CASES[name])
    S, path, nchunks, f0 = c.pop("S"), c.pop("path"), c.pop("nchunks", None), c.pop("f0", None)
    fs = c.pop("fs")
    iq, sent = make_streams(S, fs, c["baud"], c["bits"], c["stops"], nchunks, seed0=100, f0=f0)
    eng, orcs, worst = run_fast(hd, iq, fs, expect_path=path, **c)
    assert worst["filtered_n"] > 0 or name == "D64_50_7N2"
    # the mode really is another arithmetic: most decimated samples differ from the oracle's in their last bits (an exact-mode engine would match all of them)
    assert worst["identical_floats"] < worst["floats"], worst
    if name != "D4_ungated":
        assert sum(len(o.sentences()) for o in orcs) >= 2


@pytest.mark.parametrize("factor", [2, 8, 128, 256])
def test_fast_mode_every_decimation_plan(hd, factor):
    """All stage designs (filtercoef.h's eight tables) on noise, rate gate lifted; 256 needs a push long enough for its second stage."""
    fs = 10e6 if factor >= 128 else 0.4e6
    r = np.random.default_rng(factor)
    iq = (0.4 * (r.standard_normal((2, 4 * C)) + 1j * r.standard_normal((2, 4 * C)))).astype(np.complex64)
    run_fast(hd, iq, fs, factor=factor, baud=300, bits=8, stops=2, ungated=True)


@pytest.mark.parametrize("route", ["transforms", "direct"])
def test_fast_mode_4097_tap_lowpass(hd, monkeypatch, route):
    """configs[4] shape: 10 MS/s, /256, lp_trans = 4/4096 -> 4097 taps (2^20-sample pushes): the longest sums of the chain -- in fast mode through N-point
    transforms (the default from 1024 taps on; every run but the one that starts the filter from zeros), or as direct fused multiply-add sums (HD_NO_LP_FFT=1)."""
    if route == "direct":
        monkeypatch.setenv("HD_NO_LP_FFT", "1")
    S, fs, big = 2, 10e6, 1 << 20
    b = synth.rtty_bits(synth.make_sentence("BIG", "1,2,3") * 2, 8, 2, 4, 4)
    iq = np.stack([synth.fsk_iq(b, fs, 300, sigma=0.05, seed=s, n_samples=6 * big) for s in range(S)])
    eng, orcs, worst = run_fast(hd, iq, fs, factor=256, baud=300, bits=8, stops=2, lowpass_trans=4.0 / 4096, chunk=big)
    assert len(eng.fir_taps(0)) == 4097 and worst["filtered_n"] >= 6 * 2 * 4096 - 2 * 4352 and sum(len(o.text("chars_log")) for o in orcs) > 0
    assert eng.timing()["lowpass_fft_calls"] == (5 if route == "transforms" else 0)     # (six runs; the first starts from zeros: direct)


def test_fast_mode_long_lowpass_through_a_tap_count_change(hd):
    """The transform route of the long low-pass across a runtime change of the tap count (4097 -> 2049 -> 4097: the reference's stale-buffer transient, FirHistory,
    rebuilt by k_lp_gather exactly as k_fir_demod's tile loader rebuilds it) and with a DC blocker switched on in between (another launch sequence, same route):
    floats within tolerance of the oracle's, bits and text identical."""
    import bench
    from oracle import pyoracle
    S, fs, big = 2, 10e6, 1 << 20
    b = synth.rtty_bits(synth.make_sentence("LONG", "4,5,6") * 2, 8, 2, 4, 4)
    iq = np.stack([synth.fsk_iq(b, fs, 300, sigma=0.05, seed=40 + s, n_samples=8 * big) for s in range(S)])
    eng = hd.Engine(n_streams=S, max_chunk=big, sampling_rate=fs, decimation=256, lowpass_trans=4.0 / 4096, keep_filtered=True, arith=1)
    orcs = [pyoracle.Decoder("oracle", factor=256, lowpass_trans=4.0 / 4096) for _ in range(S)]
    ntaps = set()
    for k in range(8):
        if k in (3, 6):
            t = 8.0 / 4096 if k == 3 else 4.0 / 4096
            for s in range(S):
                hd.capi.check(eng.L.hd_stream_set_lowpass_trans(eng.h, s, t)); orcs[s].lowpass_trans(t)
        piece = np.ascontiguousarray(iq[:, k * big:(k + 1) * big])
        eng.process_host(piece)
        for s in range(S):
            o = orcs[s]
            o(piece[s], fs)
            ntaps.add(len(eng.fir_taps(s)))
            assert np.array_equal(eng.fir_taps(s).view(np.uint32), o.array("fir_taps").view(np.uint32)), (k, s)
            fo, do = o.array("last_filtered"), o.array("last_decimated")
            assert bench.normwise(eng.decimated(s), do) <= TOL, (k, s)
            assert bench.fir_normwise(eng.filtered(s), fo, do) <= TOL, ("filtered", k, s, bench.fir_normwise(eng.filtered(s), fo, do))
            assert np.array_equal(eng.bits(s), o.bits()), ("bits", k, s)
    assert len(ntaps) == 2 and min(ntaps) >= 1024, ntaps
    assert eng.timing()["lowpass_fft_calls"] >= 7
    for s in range(S):
        assert eng.take_chars(s) == orcs[s].text("chars_log") and eng.take_sentences(s) == orcs[s].sentences()


@pytest.mark.parametrize("ctx", ["mathh", "cmath"])
def test_fast_mode_golden_chain(hd, ctx):
    """tests/golden/chain_small (input on disk, expectations recorded from the reference's own compiled stage classes): in fast mode the bits of every call,
    the AFC peaks, the sentences, the characters and the text stream are the RECORDED ones (the floats' SHA-1s are the exact mode's business)."""
    gold = Path(__file__).resolve().parent / "golden"
    meta = json.loads((gold / "chain_small.json").read_text())
    q = np.load(gold / "chain_small_input.npz")["iq_int16"]
    x = (q[:, 0].astype(np.float32) / np.float32(meta["scale"]) + 1j * (q[:, 1].astype(np.float32) / np.float32(meta["scale"]))).astype(np.complex64)
    want = meta["expected"][ctx]
    Cn = meta["chunk"]
    eng = hd.Engine(n_streams=1, max_chunk=Cn, sampling_rate=meta["fs"], decimation=meta["factor"], baud=meta["baud"], rtty_bits=8, rtty_stops=2,
                    lookup_mode=1 if ctx == "mathh" else 0, arith=1)
    for k, i in enumerate(range(0, len(x), Cn)):
        eng.process_host(x[None, i:i + Cn])
        w = want["per_call"][k]
        assert eng.bits(0).tolist() == w[3], k
        a = eng.afc(0)
        assert (a["peak_l"], a["peak_r"]) == (w[4]["peak_l"], w[4]["peak_r"]), k
    assert eng.take_sentences(0) == want["sentences"] and eng.take_chars(0) == want["chars"]
    assert eng.rtty(0) == want["rtty"] and eng.last_sentence(0) == want["last"]


@pytest.fixture(scope="module")
def headline_ring():
    torch = pytest.importorskip("torch")
    import bench
    w = dict(bench.WORKLOADS["cfg4"])
    ring, ring_chunks, _ = bench.generate_ring(torch, torch.device("cuda", 0), w, w["S"], 0, seed=78)
    yield w, ring, ring_chunks
    del ring
    torch.cuda.empty_cache()


def test_fast_mode_inside_the_step_kernel(headline_ring):
    """The headline workload in batch mode (k_step_cu: worker waves with the two-chain systolic tap loop, the tails beside them), fast mode: the floats of 16
    sampled streams call by call, then a free-running stretch whose symbols, characters and backlog must be the oracle's."""
    import bench
    import habdec_amd
    from oracle import pyoracle
    w, ring, ring_chunks = headline_ring
    S, fs = w["S"], w["fs"]
    check = [0, 1, 2, 127, 128, 500, 511, 512, 640, 1000, 1022, 7, 15, 263, 775, 1023]
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                            lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], pipeline=1, keep_filtered=True, arith=1)
    orcs = {s: pyoracle.Decoder("oracle", factor=w["D"], baud=w["baud"], bits=w["bits"], stops=w["stops"], lowpass_bw=w["lp_bw"]) for s in check}
    host = {s: ring[:, s].cpu().numpy().view(np.complex64).reshape(ring_chunks, C) for s in check}
    prev = {s: None for s in check}
    obits = {s: 0 for s in check}
    worst = [0.0, 0.0, 0.0]
    for k in range(10):
        eng.process_device(ring[k % ring_chunks].data_ptr(), C, C)
        for s, o in orcs.items():
            o(host[s][k % ring_chunks], fs)
            fo, do = o.array("last_filtered"), o.array("last_decimated")
            worst[0] = max(worst[0], bench.normwise(eng.decimated(s), do))
            worst[1] = max(worst[1], bench.fir_normwise(eng.filtered(s), fo, do))
            worst[2] = max(worst[2], bench.demod_excess(eng.demodulated(s), o.array("last_demod"), fo, prev[s], TOL, scale=float(np.max(np.abs(do))))[0])
            if fo.size:
                prev[s] = fo[-1]
            assert worst[0] <= TOL and worst[1] <= TOL and worst[2] <= 1.0, (k, s, worst)
            assert np.array_equal(eng.bits(s), o.bits()), ("bits", k, s)
            obits[s] += len(o.bits())
            assert eng.symbol_backlog(s) == o.symex_held(), ("backlog", k, s)
    assert eng.timing()["path"] == 3
    for k in range(10, 46):
        eng.process_device(ring[k % ring_chunks].data_ptr(), C, C)
        for s, o in orcs.items():
            o(host[s][k % ring_chunks], fs)
            obits[s] += len(o.bits())
    assert eng.timing()["path"] == 3 and eng.timing()["step_variant"] == 1
    eng.flush()
    for s, o in orcs.items():
        assert eng.take_chars(s) == o.text("chars_log"), ("chars", s)
        assert eng.bits_total(s) == obits[s], ("symbols produced", s)
        assert eng.symbol_backlog(s) == o.symex_held(), ("backlog", s)
        assert bench.normwise(eng.decimated(s), o.array("last_decimated")) <= TOL
    eng.close()


@pytest.mark.parametrize("name,steps,S", [("cfg4", 40, 0), ("cfg1", 24, 0), ("cfg2", 12, 0), ("cfg3", 12, 0), ("cfg5", 4, 0)])
def test_fast_mode_every_stream_at_bench_size(name, steps, S):
    """bench.py's own loop in fast mode at the workload's full stream count: symbols produced, characters and sentences of EVERY stream equal the oracle's over
    every step taken, and the float probe of the same run stays within 1e-5."""
    torch = pytest.importorskip("torch")
    import bench
    r = bench.run_workload(torch, None, torch.device("cuda", 0), 0, 0, 1, name, steps, 2, S, False, cpu_leg="check_all", prewarm=0, arith=1, min_kernel_samples=4)
    cb = r["cpu_baseline"]
    assert r["arith"] == "fast" and cb["all_streams_of_the_shard"] is True and cb["streams_in_sample"] == bench.WORKLOADS[name]["S"], cb
    assert cb["gpu_matches_oracle_on_sample"] is True, cb
    assert cb["bits_in_sample"] > 0 and (name in ("cfg3",) or cb["chars_in_sample"] > 0)
    assert r["float_parity"]["within_tolerance"] is True, r["float_parity"]
