"""GPU parity: the HIP path (through the C ABI, libhabdec_amd.so) against the CPU oracle on the same seeded
inputs.  Integer results (flip points, bits, characters, sentences) and the float stages computed with exact
arithmetic (decimated, filtered, demodulated) must be IDENTICAL; the spectrum side (rocFFT vs a double DFT,
device log10f) is compared norm-wise at 1e-5 as BASELINE.json's north_star states."""
import numpy as np
import pytest

from habdec_amd import synth

pytestmark = pytest.mark.gpu

C = 65536


@pytest.fixture(scope="module")
def hd():
    import habdec_amd
    habdec_amd.lib()
    return habdec_amd


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def normwise(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return np.inf
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def power_close(gp, op):
    """dB power arrays agree: P = 40*log10|X| + const, so weak bins magnify the (unpinned, FFT-dependent) relative
    error of |X|.  Compare the implied bin magnitudes norm-wise (1e-5 of the strongest bin, the north_star tolerance,
    with 2x headroom for the log/exp round trip) and the dB values themselves on every bin within 40 dB of the peak."""
    if gp.shape != op.shape:
        return False
    top = float(np.max(op))
    mg, mo = 10.0 ** ((gp.astype(np.float64) - top) / 40.0), 10.0 ** ((op.astype(np.float64) - top) / 40.0)
    strong = op > top - 40.0
    return bool(np.max(np.abs(mg - mo)) <= 2e-5 and np.max(np.abs(gp[strong] - op[strong])) <= 5e-3)


@pytest.mark.parametrize("baud", [50, 110])
def test_long_averaging_windows_at_512_kHz(hd, baud):
    """/4 with the gate lifted and slow keying: symbols of 10240 / 4655 decimated samples, averaging half-windows R = 2560 / 1163 -- longer than a
    sweep of k_symbols covers with its batched loads (the plain-loop staging), and a backlog that lives right under the reference's 30000-sample vent
    (three symbols at 50 baud are 30720 samples: the vent fires on the way, SymbolExtractor.h:116-124)."""
    from oracle import pyoracle
    S, fs = 2, 2.048e6
    iq, _ = make_streams(S, fs, baud, 7, 2, 40, seed0=61, texts=["$$A,1*", "$$B,2*"])
    eng, orcs, stats = run_both(hd, pyoracle, iq, fs, factor=4, baud=baud, bits=7, stops=2, ungated=True, check_every=5, spectrum=False)
    assert stats["demod_total"] > 0 and stats["demod_mismatch"] == 0


def make_streams(S, fs, baud, bits, stops, nchunks=None, *, sigma=0.08, f0=None, seed0=0, texts=None, repeat=2):
    """S continuous streams; stream s carries its own short sentence `repeat` times.  nchunks=None sizes the
    streams so that every sentence (plus trailing idle) fits."""
    if texts is None:
        texts = [synth.make_sentence(f"HAB{s}", f"{s + 1},52.{100 + s},21.{400 + s}") * repeat for s in range(S)]
    frame = 1 + bits + int(stops)
    if nchunks is None:
        longest = max(len(t) for t in texts)
        nchunks = int(np.ceil((longest * frame + 6 + 3 * S + 12) * (fs / baud) / C)) + 1
    out = np.zeros((S, nchunks * C), np.complex64)
    for s in range(S):
        b = synth.rtty_bits(texts[s], bits, stops, 6 + 3 * s, 10)
        out[s] = synth.fsk_iq(b, fs, baud, sigma=sigma, seed=seed0 + s, n_samples=nchunks * C, f0=(f0[s] if f0 is not None else 0.0))
    return out, texts


def run_both(hd, pyoracle, iq, fs, *, factor, baud, bits, stops, lowpass_bw=None, dc_remove=False, lookup=1, ungated=False,
             check_every=1, spectrum=True):
    S, N = iq.shape
    eng = hd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=factor, baud=baud, rtty_bits=bits, rtty_stops=stops,
                    lowpass_bw_hz=lowpass_bw if lowpass_bw is not None else 1500.0, dc_remove=dc_remove, lookup_mode=lookup,
                    ungated=ungated, keep_filtered=True, enable_spectrum=spectrum)
    orcs = [pyoracle.Decoder("oracle", factor=factor, baud=baud, bits=bits, stops=stops, lowpass_bw=lowpass_bw, dc_remove=dc_remove,
                             mathh_context=lookup, ungated=ungated, with_fft=spectrum) for _ in range(S)]
    stats = {"demod_mismatch": 0, "demod_total": 0, "power_worst": 0.0}
    for k in range(N // C):
        chunk = np.ascontiguousarray(iq[:, k * C:(k + 1) * C])
        eng.process_host(chunk)
        for s in range(S):
            orcs[s](chunk[s], fs)
        if k % check_every:
            continue
        for s in range(S):
            o = orcs[s]
            assert same_bits(eng.decimated(s), o.array("last_decimated")), ("decimated", k, s)
            assert same_bits(eng.filtered(s), o.array("last_filtered")), ("filtered", k, s)
            gd, od = eng.demodulated(s), o.array("last_demod")
            assert normwise(gd, od) <= 1e-5, ("demod", k, s)
            stats["demod_total"] += od.size
            stats["demod_mismatch"] += int(np.count_nonzero(gd.view(np.uint32) != od.view(np.uint32))) if od.size else 0
            assert np.array_equal(eng.bits(s), o.bits()), ("bits", k, s)
            assert eng.symbol_backlog(s) == o.symex_held(), ("backlog", k, s)
            if spectrum:
                gp, op = eng.power(s), o.array("power")
                if op.size:
                    stats["power_worst"] = max(stats["power_worst"], normwise(gp, op))
                    assert power_close(gp, op), ("power", k, s, normwise(gp, op))
                    assert normwise(eng.spectrum(s), o.array("spectrum")) <= 1e-5, ("spectrum", k, s)
                ga, oa = eng.afc(s), o.afc()
                assert (ga["peak_l"], ga["peak_r"]) == (oa["peak_l"], oa["peak_r"]), ("peaks", k, s, ga, oa)
                for key in ("correction", "shift_hz"):
                    assert ga[key] == pytest.approx(oa[key], rel=1e-9, abs=1e-9), (key, k, s)
                for key in ("noise_floor", "noise_var"):
                    assert ga[key] == pytest.approx(oa[key], rel=1e-5, abs=1e-4), (key, k, s)
    for s in range(S):
        o = orcs[s]
        assert eng.rtty(s) == o.text("rtty_stream"), ("rtty", s)
        assert eng.last_sentence(s) == o.text("last_sentence"), ("last", s)
        assert eng.take_sentences(s) == o.sentences(), ("sentences", s)
        assert eng.take_chars(s) == o.text("chars_log"), ("chars", s)
    return eng, orcs, stats


def test_device_is_gfx950_and_library_is_the_hip_one(hd):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    assert hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value >= 1
    eng = hd.Engine(n_streams=1)
    assert eng.L.hd_engine_decimation(eng.h) == 64
    eng.close()


CASES = {
    # BASELINE.json configs[0]/[1]-like: 2.048 MS/s, /64, 300 baud 8N2
    "cfg1_D64_300_8N2": dict(fs=2.048e6, factor=64, baud=300, bits=8, stops=2, S=3),
    # configs[1]: 2.5 MS/s, dec=4 (/16), lowpass 3 kHz, 300 baud 8N2
    "cfg2_D16_2p5M_lp3k": dict(fs=2.5e6, factor=16, baud=300, bits=8, stops=2, S=2, lowpass_bw=3000.0),
    # DC blocker on
    "cfg1_dc_remove": dict(fs=2.048e6, factor=64, baud=300, bits=8, stops=2, S=2, dc_remove=True),
    # <cmath>-only lookup context (double-trig taps, integer flip weights)
    "cfg1_lookup0": dict(fs=2.048e6, factor=64, baud=300, bits=8, stops=2, S=2, lookup=0),
    # 7N1 at 600 baud through /32
    "D32_600_7N1": dict(fs=1.024e6, factor=32, baud=600, bits=7, stops=1, S=2),
}


@pytest.mark.parametrize("name", list(CASES))
def test_chain_identical_to_oracle(hd, name):
    from oracle import pyoracle
    c = dict(CASES[name])
    S = c.pop("S")
    iq, sent = make_streams(S, c["fs"], c["baud"], c["bits"], c["stops"], seed0=100)
    fs = c.pop("fs")
    eng, orcs, stats = run_both(hd, pyoracle, iq, fs, **c)
    # the exact-arithmetic discriminator is expected to be bit-identical, not merely within tolerance
    assert stats["demod_mismatch"] == 0, stats
    got = orcs[0].sentences()
    assert len(got) == 2 and all(("$$" + g + "\n") in sent[0] for g in got)
    assert eng.sentences_ok() == sum(len(o.sentences()) for o in orcs)


@pytest.mark.parametrize("path", ["tail", "separate"])
@pytest.mark.parametrize("baud", [2000, 1300, 1200, 900, 750])
def test_short_symbols_small_averaging_windows(hd, monkeypatch, baud, path):
    """Symbols of 16 .. 43 decimated samples (32 kHz): averaging half-windows R = 4, 6, 6, 8, 10 -- below and around the eight positions a lane of
    the window-sum code owns (its plain-loop branch for R < 8, one-chunk interiors, R not a multiple of four), through the stream tail and
    through k_symbols.  The 5 kHz low-pass lets the keying through; what is decoded is compared, whatever it is worth."""
    from oracle import pyoracle
    if path == "separate":
        monkeypatch.setenv("HD_NO_TAIL", "1")
        monkeypatch.setenv("HD_NO_FUSE", "1")
    fs = 2.048e6
    iq, sent = make_streams(2, fs, baud, 8, 2, seed0=300 + baud, repeat=3)
    eng, orcs, stats = run_both(hd, pyoracle, iq, fs, factor=64, baud=baud, bits=8, stops=2, lowpass_bw=5000.0, spectrum=False)
    assert stats["demod_mismatch"] == 0, stats
    assert eng.timing()["path"] == (2 if path == "tail" else 0)
    assert sum(len(o.text("chars_log")) for o in orcs) > 20


@pytest.mark.parametrize("spectrum_by", ["tail", "rocfft", "wave_launch"])
def test_cfg4_50baud_7N2_with_offsets_and_afc(hd, monkeypatch, spectrum_by):
    """configs[3] shape on few streams: /64, 50 baud 7N2, per-stream carrier offsets; AFC outputs every call.  The spectrum of a completed
    buffer comes from the stream tail itself (default), from rocFFT + the commit kernel (HD_ROCFFT=1), or from the single-wave kernel as a
    launch of its own (HD_ROCFFT=1 HD_OWN_FFT=1): the same tolerances against the exact DFT for all three."""
    from oracle import pyoracle
    if spectrum_by != "tail":
        monkeypatch.setenv("HD_ROCFFT", "1")
    if spectrum_by == "wave_launch":
        monkeypatch.setenv("HD_OWN_FFT", "1")
    S, fs = 4, 2.048e6
    texts = [synth.make_sentence("A", str(s)) * 3 for s in range(S)]
    iq, _ = make_streams(S, fs, 50, 7, 2, f0=[0.0, 120.0, -200.0, 1500.0], seed0=7, texts=texts)
    eng, orcs, stats = run_both(hd, pyoracle, iq, fs, factor=64, baud=50, bits=7, stops=2, check_every=3)
    assert stats["demod_mismatch"] == 0
    assert len(orcs[0].sentences()) >= 2 and len(orcs[1].sentences()) >= 2
    assert abs(eng.afc(3)["correction"] - 1500.0) < 60.0      # the far-off stream latches a correction near its offset


@pytest.mark.parametrize("ungated", [False, True])
def test_cfg3_D4_gate_and_stage_level(hd, ungated):
    """configs[2]: /4 -> 512 kHz is above the reference's 160 kHz gate: only decimation+spectrum+AFC run and nothing
    decodes; with ungated=True the FIR/demod/symbol kernels run at that rate and must still match the oracle."""
    from oracle import pyoracle
    S, fs = 2, 2.048e6
    iq, _ = make_streams(S, fs, 300, 8, 2, 6, seed0=21)
    eng, orcs, stats = run_both(hd, pyoracle, iq, fs, factor=4, baud=300, bits=8, stops=2, ungated=ungated)
    if not ungated:
        assert stats["demod_total"] == 0 and eng.take_chars(0) == ""
    else:
        assert stats["demod_total"] == 6 * S * (C // 4) and stats["demod_mismatch"] == 0


@pytest.mark.parametrize("factor", [1, 2, 8, 128, 256])
def test_every_decimation_plan(hd, factor):
    """All stage plans of setupDecimationStagesFactor; 256 needs a push long enough for its second stage."""
    from oracle import pyoracle
    fs = 10e6 if factor >= 128 else 0.4e6
    S = 2
    r = np.random.default_rng(factor)
    iq = (0.4 * (r.standard_normal((S, 4 * C)) + 1j * r.standard_normal((S, 4 * C)))).astype(np.complex64)
    run_both(hd, pyoracle, iq, fs, factor=factor, baud=300, bits=8, stops=2, ungated=True, spectrum=(factor == 8))


def test_cfg5_4097_tap_lowpass(hd):
    """configs[4] shape: 10 MS/s, /256, lp_trans = 4/4096 -> 4097 taps; needs 2^20-sample pushes."""
    import habdec_amd
    from oracle import pyoracle
    S, fs, big = 2, 10e6, 1 << 20
    b = synth.rtty_bits(synth.make_sentence("BIG", "1,2,3") * 2, 8, 2, 4, 4)
    iq = np.stack([synth.fsk_iq(b, fs, 300, sigma=0.05, seed=s, n_samples=6 * big) for s in range(S)])
    eng = habdec_amd.Engine(n_streams=S, max_chunk=big, sampling_rate=fs, decimation=256, lowpass_trans=4.0 / 4096, keep_filtered=True)
    orcs = [pyoracle.Decoder("oracle", factor=256, lowpass_trans=4.0 / 4096) for _ in range(S)]
    for k in range(6):
        chunk = np.ascontiguousarray(iq[:, k * big:(k + 1) * big])
        eng.process_host(chunk)
        for s in range(S):
            orcs[s](chunk[s], fs)
            assert len(eng.fir_taps(s)) == 4097
            assert same_bits(eng.fir_taps(s), orcs[s].array("fir_taps"))
            assert same_bits(eng.decimated(s), orcs[s].array("last_decimated"))
            assert same_bits(eng.filtered(s), orcs[s].array("last_filtered"))
            assert same_bits(eng.demodulated(s), orcs[s].array("last_demod"))
            assert np.array_equal(eng.bits(s), orcs[s].bits())
    for s in range(S):
        assert eng.take_sentences(s) == orcs[s].sentences()


def test_ragged_and_idle_streams(hd):
    """Per-stream sample counts, including streams that hand over nothing in a call."""
    import ctypes
    import habdec_amd
    from oracle import pyoracle
    S, fs = 4, 2.048e6
    iq, _ = make_streams(S, fs, 300, 8, 2, seed0=55)
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, keep_filtered=True)
    orcs = [pyoracle.Decoder("oracle", factor=64) for _ in range(S)]
    pos = [0] * S
    r = np.random.default_rng(1)
    for k in range(80):
        n = np.array([C, 0 if k % 3 == 1 else C, 32768 if k >= 2 else C, (16384 * int(r.integers(1, 5)))], np.uint32)
        n = np.minimum(n, [iq.shape[1] - p for p in pos]).astype(np.uint32)
        buf = np.zeros((S, C), np.complex64)
        for s in range(S):
            buf[s, :n[s]] = iq[s, pos[s]:pos[s] + n[s]]
        habdec_amd.capi.check(eng.L.hd_process_host(eng.h, buf.ctypes.data, C, n.ctypes.data, 0))
        for s in range(S):
            if n[s]:
                orcs[s](iq[s, pos[s]:pos[s] + n[s]], fs)
                assert same_bits(eng.decimated(s), orcs[s].array("last_decimated")), (k, s)
                assert same_bits(eng.demodulated(s), orcs[s].array("last_demod")), (k, s)
                assert np.array_equal(eng.bits(s), orcs[s].bits()), (k, s)
            pos[s] += int(n[s])
    for s in range(S):
        assert eng.take_chars(s) == orcs[s].text("chars_log")


def test_ragged_and_idle_pushes_free_running_through_the_unfused_two_stage_path(hd):
    """/16 with pushes of up to 65536 samples: 4096 decimated samples per call, too many for the one-wave stream tail -- the separate kernels on two queues
    (launch path 0).  Round 5 took the event waits off its front queue (three low-pass buffers take turns, the parameter copy rides behind stage 1), so
    calls really overlap: ragged sizes, streams that skip calls, and NO getter between the calls (a getter drains the pipeline); every call's discriminator
    checksum is compared afterwards, then text, and the last call's buffers."""
    import habdec_amd
    from oracle import pyoracle
    S, fs = 6, 2.5e6
    iq, _ = make_streams(S, fs, 300, 8, 2, seed0=77, repeat=3)
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=16, baud=300, rtty_bits=8, rtty_stops=2, lowpass_bw_hz=3000.0, pipeline=1)
    orcs = [pyoracle.Decoder("oracle", factor=16, baud=300, bits=8, stops=2, lowpass_bw=3000.0) for _ in range(S)]
    pos = [0] * S
    r = np.random.default_rng(5)
    ncalls = 0
    while min(pos) < iq.shape[1] and ncalls < 90:
        n = np.array([C, 0 if ncalls % 3 == 1 else C, 32768 if ncalls >= 2 else C, 2048 * int(r.integers(1, 33)), C if ncalls % 5 else 4096, 2048 * int(r.integers(8, 33))], np.uint32)
        n = np.minimum(n, [iq.shape[1] - p for p in pos]).astype(np.uint32)
        buf = np.zeros((S, C), np.complex64)
        for s in range(S):
            buf[s, :n[s]] = iq[s, pos[s]:pos[s] + n[s]]
        habdec_amd.capi.check(eng.L.hd_process_host(eng.h, buf.ctypes.data, C, n.ctypes.data, 0))
        for s in range(S):
            if n[s]: orcs[s](iq[s, pos[s]:pos[s] + n[s]], fs)
            pos[s] += int(n[s])
        last_n = n
        ncalls += 1
    eng.flush()
    assert eng.timing()["path"] == 0
    for s in range(S):
        assert eng.take_chars(s) == orcs[s].text("chars_log"), s
        assert eng.take_sentences(s) == orcs[s].sentences(), s
        if last_n[s]:
            assert same_bits(eng.demodulated(s), orcs[s].array("last_demod")), s
            assert same_bits(eng.decimated(s), orcs[s].array("last_decimated")), s
    assert sum(len(o.sentences()) for o in orcs) >= 2 and all(len(o.text("chars_log")) > 10 for o in orcs)


@pytest.mark.parametrize("factor,fs,lowpass", [(64, 2.048e6, None), (16, 2.5e6, 3000.0)])
def test_silence_between_transmissions_takes_the_general_paths(hd, factor, fs, lowpass):
    """Samples that are exactly zero (a muted receiver, a file padded with zeros): the discriminator sees arg(0 * conj(0)), the window sums are exactly 0 -- the
    cases the wave-uniform fast paths (exact_math.h discriminate, sym_common.h sign_flags: no special operands, no sum inside the quotient's underflow) hand to
    the general code.  Through the stream tail (/64) and through k_fir_demod + k_symbols (/16), call by call against the oracle."""
    import habdec_amd
    from oracle import pyoracle
    S = 3
    iq, _ = make_streams(S, fs, 300, 8, 2, seed0=91, repeat=2)
    iq = iq.copy()
    nch = iq.shape[1] // C
    iq[1, :3 * C] = 0                                        # silence first, then the transmission
    iq[2, 2 * C:5 * C] = 0                                   # silence in the middle of it
    iq[2, 5 * C + 1000:5 * C + 1007] = 0                     # ... and a few zero samples inside live signal
    kw = dict(lowpass_bw_hz=lowpass) if lowpass else {}
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=factor, baud=300, rtty_bits=8, rtty_stops=2, **kw)
    orcs = [pyoracle.Decoder("oracle", factor=factor, baud=300, bits=8, stops=2, **({"lowpass_bw": lowpass} if lowpass else {})) for _ in range(S)]
    for k in range(nch):
        eng.process_host(np.ascontiguousarray(iq[:, k * C:(k + 1) * C]), C)
        for s in range(S):
            orcs[s](iq[s, k * C:(k + 1) * C], fs)
            assert same_bits(eng.demodulated(s), orcs[s].array("last_demod")), (k, s)
            assert np.array_equal(eng.bits(s), orcs[s].bits()), (k, s)
    for s in range(S):
        assert eng.take_chars(s) == orcs[s].text("chars_log"), s
        assert eng.take_sentences(s) == orcs[s].sentences(), s
    assert len(orcs[0].sentences()) >= 1


def test_short_chunk_history_quirk_and_rejects(hd):
    """Q4: a chunk so short that the in-place history overlaps the outputs still matches; a chunk shorter than the
    history is undefined in the reference and is rejected."""
    import habdec_amd
    from oracle import pyoracle
    fs = 48000.0
    eng = habdec_amd.Engine(n_streams=1, max_chunk=4096, sampling_rate=fs, decimation=2, ungated=True, enable_spectrum=False)
    o = pyoracle.Decoder("oracle", factor=2, ungated=True, with_fft=False)
    r = np.random.default_rng(2)
    for n in [4096, 100, 80, 70, 68, 4096, 512]:
        x = (0.3 * (r.standard_normal(n) + 1j * r.standard_normal(n))).astype(np.complex64)
        buf = np.zeros((1, 4096), np.complex64)
        buf[0, :n] = x
        eng.process_host(buf, n)
        o(x, fs)
        assert same_bits(eng.decimated(0), o.array("last_decimated")), n
    buf = np.zeros((1, 4096), np.complex64)
    with pytest.raises(habdec_amd.HabdecError):
        eng.process_host(buf, 66)        # 66 < 68 = taps-1
    with pytest.raises(habdec_amd.HabdecError):
        eng.process_host(buf, 4097 - 1 + 2 * 4096)   # more than max_chunk
    with pytest.raises(habdec_amd.HabdecError):
        habdec_amd.Engine(n_streams=1, decimation=3)


def test_device_resident_batch_and_callbacks(hd):
    """hd_process_device on an HBM-resident slab (torch only allocates it) + the sentence callback."""
    torch = pytest.importorskip("torch")
    import habdec_amd
    from oracle import pyoracle
    S, fs = 8, 2.048e6
    iq, _ = make_streams(S, fs, 300, 8, 2, seed0=900)
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64)
    seen = []
    eng.on_sentence(lambda s, call, data, crc: seen.append((s, f"{call},{data}*{crc}")))
    orcs = [pyoracle.Decoder("oracle", factor=64) for _ in range(S)]
    for k in range(iq.shape[1] // C):
        eng.process_device(dev.data_ptr() + k * C * 8, iq.shape[1], C)
        for s in range(S):
            orcs[s](iq[s, k * C:(k + 1) * C], fs)
    want = [(s, x) for s in range(S) for x in orcs[s].sentences()]
    assert sorted(seen) == sorted(want) and len(want) >= 2 * S
    t = eng.timing()
    assert t["samples"] == S * C and t["ms_front"] > 0 and t["front_bytes"] == S * C * 8 + S * C // 32 * 8


def test_full_size_batch_properties(hd):
    """BASELINE size (1024 streams x 65536): properties that need no oracle at that size -- every stream fed the
    SAME signal yields identical results, and matches the oracle run once on that signal."""
    torch = pytest.importorskip("torch")
    import habdec_amd
    from oracle import pyoracle
    S, fs = 1024, 2.048e6
    iq1, _ = make_streams(1, fs, 300, 8, 2, seed0=3)
    nch = iq1.shape[1] // C
    dev = torch.from_numpy(iq1.view(np.float32)).cuda()
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64)
    o = pyoracle.Decoder("oracle", factor=64)
    for k in range(nch):
        # stride 0: every stream reads the same HBM-resident chunk
        eng.process_device(dev.data_ptr() + k * C * 8, 0, C)
        o(iq1[0, k * C:(k + 1) * C], fs)
        assert same_bits(eng.demodulated(S - 1), o.array("last_demod"))
        assert same_bits(eng.demodulated(517), eng.demodulated(0))
    want = o.text("chars_log")
    assert len(o.sentences()) >= 1
    for s in (0, 1, 255, 256, 1000, 1023):
        assert eng.take_chars(s) == want
    assert eng.sentences_ok() == S * len(o.sentences())


@pytest.mark.parametrize("wgs_per_cu", [5, 6, 7])
def test_full_size_linear_split_stage1(hd, monkeypatch, wgs_per_cu):
    """Batch mode at BASELINE size runs stage 1 as a linear split: k workgroups per CU share all 32768 tiles evenly and walk
    across stream seams (history in, history carry out, in the middle of a workgroup's range).  Every stream gets its own
    delayed copy of one signal, so a seam handled wrongly shows; decimated/demodulated samples must equal the oracle's bit
    for bit and the text must be complete."""
    torch = pytest.importorskip("torch")
    import habdec_amd
    from oracle import pyoracle
    monkeypatch.setenv("HD_DEC_WGS_PER_CU", str(wgs_per_cu))
    S, fs = 1024, 2.048e6
    iq1, _ = make_streams(1, fs, 300, 8, 2, seed0=5)
    nch = iq1.shape[1] // C - 1
    base = torch.from_numpy(iq1.view(np.float32)).cuda()                     # [1, L, 2]
    shifts = (np.arange(S) * 37) % 4096                                       # stream s starts 'shifts[s]' samples into the signal
    slab = torch.empty((nch, S, C, 2), dtype=torch.float32, device="cuda")
    L = base.shape[1] if base.dim() == 3 else base.numel() // 2
    flat = base.reshape(-1, 2)
    for s in range(S):
        seg = flat[int(shifts[s]):int(shifts[s]) + nch * C]
        slab[:, s] = seg.view(nch, C, 2)
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, pipeline=True)
    check = (0, 1, 7, 8, 511, 512, 1023)
    orcs = {s: pyoracle.Decoder("oracle", factor=64) for s in check}
    for k in range(nch):
        eng.process_device(slab[k].data_ptr(), C, C)
        eng.flush()
        for s, o in orcs.items():
            x = iq1[0, int(shifts[s]) + k * C: int(shifts[s]) + (k + 1) * C]
            o(x, fs)
            assert same_bits(eng.decimated(s).view(np.float32), o.array("last_decimated").view(np.float32)), (k, s)
            assert same_bits(eng.demodulated(s), o.array("last_demod")), (k, s)
    for s, o in orcs.items():
        assert eng.take_chars(s) == o.text("chars_log")
    assert eng.sentences_ok() >= S * (len(orcs[0].sentences()) - 1)


@pytest.mark.parametrize("base0", ["0xFFFFF000", "0xFFFFFF80", "0x7FFFFE00"])
def test_symbol_ring_positions_wrap_around(hd, monkeypatch, base0):
    """Positions in the symbol rings are monotonic 32-bit counters (a 32 kHz stream wraps them after 37 hours).  Started just
    below 2^32 (and 2^31, for the signed comparisons) the decoder must produce the same bits, flips and text as the oracle."""
    from oracle import pyoracle
    monkeypatch.setenv("HD_SYM_BASE0", base0)
    S, fs = 4, 2.048e6
    iq, _ = make_streams(S, fs, 300, 8, 2, seed0=1200)
    eng, orcs, stats = run_both(hd, pyoracle, iq, fs, factor=64, baud=300, bits=8, stops=2)
    assert stats["demod_mismatch"] == 0                  # run_both has compared bits, backlog and text call by call
    assert all(len(o.sentences()) >= 1 for o in orcs)


@pytest.mark.parametrize("dc", [False, True])
def test_control_plane_changes_mid_stream(hd, dc):
    """baud / framing / low-pass bandwidth and transition set while samples flow (websocket clients do that): the symbol
    extractor restarts its caches, the framer its bit queue, a bandwidth change with an unchanged tap count does NOT redesign
    the filter (Q8), and a changed tap count reuses the filter's one buffer the way FirFilter does (stale-history transient)
    -- all exactly as in the reference, call by call.  With the unfused back end (DC blocker on) as well."""
    import habdec_amd
    from oracle import pyoracle
    S, fs = 3, 2.048e6
    iq, _ = make_streams(S, fs, 300, 8, 2, seed0=1500, repeat=5)
    nch = iq.shape[1] // C
    assert nch >= 13
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, baud=300, rtty_bits=8, rtty_stops=2, keep_filtered=True, dc_remove=dc)
    orcs = [pyoracle.Decoder("oracle", factor=64, baud=300, bits=8, stops=2, dc_remove=dc) for _ in range(S)]
    plan = {2: ("baud", 100.0), 3: ("baud", 300.0), 4: ("lp_bw", 2000.0), 5: ("rtty", (7, 1.0)), 6: ("rtty", (8, 2.0)), 7: ("lp_bw", 1500.0),
            8: ("lp_trans", 0.05), 9: ("lp_trans", 0.01), 10: ("lp_trans", 0.025), 11: ("lp_trans", 0.004)}     # 161 -> 81 -> 401 -> 161 -> 1001 taps
    for k in range(nch):
        if k in plan:
            what, v = plan[k]
            for s in range(S):
                if what == "baud": eng.set_baud(s, v); orcs[s].set_baud(v)
                elif what == "lp_bw": eng.set_lowpass_bw(s, v); orcs[s].lowpass_bw(v)
                elif what == "lp_trans": eng.set_lowpass_trans(s, v); orcs[s].lowpass_trans(v)
                else: eng.set_rtty(s, *v); orcs[s].set_rtty(int(v[0]), v[1])
        eng.process_host(np.ascontiguousarray(iq[:, k * C:(k + 1) * C]))
        for s in range(S):
            o = orcs[s]
            o(iq[s, k * C:(k + 1) * C], fs)
            assert same_bits(eng.filtered(s), o.array("last_filtered")), ("filtered", k, s)
            assert same_bits(eng.demodulated(s), o.array("last_demod")), ("demod", k, s)
            assert np.array_equal(eng.bits(s), o.bits()), ("bits", k, s)
            assert eng.symbol_backlog(s) == o.symex_held(), ("backlog", k, s)
    for s in range(S):
        assert eng.rtty(s) == orcs[s].text("rtty_stream") and eng.take_chars(s) == orcs[s].text("chars_log")
        assert eng.take_sentences(s) == orcs[s].sentences()


@pytest.mark.parametrize("pipeline", [False, True])
def test_switching_between_fused_and_unfused_back_end(hd, pipeline):
    """The DC blocker forces the unfused kernels (stage 2, DC, FIR separately, front half on the other queue); switching it
    on and off mid-stream makes the engine change paths from call to call -- histories, pending samples, carries and the symbol
    ring must flow across the switch exactly as in the reference."""
    import habdec_amd
    from oracle import pyoracle
    S, fs = 3, 2.048e6
    iq, _ = make_streams(S, fs, 300, 8, 2, seed0=900)
    nch = iq.shape[1] // C
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, pipeline=pipeline)
    orcs = [pyoracle.Decoder("oracle", factor=64) for _ in range(S)]
    for k in range(nch):
        on = k % 3 == 1                                  # off, ON, off, off, ON, ...
        for s in range(S):
            eng.set_dc_remove(s, on)
            orcs[s].set_dc_remove(on)
        eng.process_host(np.ascontiguousarray(iq[:, k * C:(k + 1) * C]))
        eng.flush()
        for s in range(S):
            orcs[s](iq[s, k * C:(k + 1) * C], fs)
            assert same_bits(eng.decimated(s).view(np.float32), orcs[s].array("last_decimated").view(np.float32)), (k, s)
            assert same_bits(eng.demodulated(s), orcs[s].array("last_demod")), (k, s)
    for s in range(S):
        assert eng.take_chars(s) == orcs[s].text("chars_log")


def test_pipelined_mode_delivers_identical_text(hd):
    """pipeline=1: a call returns the PREVIOUS call's text while its own symbol kernels overlap the next call's
    decimation on a second HIP stream; after hd_flush() everything must equal the synchronous result and the oracle."""
    torch = pytest.importorskip("torch")
    import habdec_amd
    from oracle import pyoracle
    S, fs = 32, 2.048e6
    iq, _ = make_streams(S, fs, 300, 8, 2, seed0=4000, repeat=2)
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    nch = iq.shape[1] // C
    engs = {p: habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, pipeline=p) for p in (False, True)}
    seen = {False: [], True: []}
    for p, eng in engs.items():
        eng.on_sentence(lambda s, call, data, crc, p=p: seen[p].append((s, f"{call},{data}*{crc}")))
        for k in range(nch):
            eng.process_device(dev.data_ptr() + k * C * 8, iq.shape[1], C)
        eng.flush()
    orcs = [pyoracle.Decoder("oracle", factor=64) for _ in range(4)]
    for s, o in enumerate(orcs):
        for k in range(nch):
            o(iq[s, k * C:(k + 1) * C], fs)
    assert seen[True] == seen[False] and len(seen[True]) >= 2 * S
    for s in range(S):
        assert engs[True].take_chars(s) == engs[False].take_chars(s)
        assert engs[True].rtty(s) == engs[False].rtty(s)
        assert same_bits(engs[True].demodulated(s), engs[False].demodulated(s))
        assert engs[True].symbol_backlog(s) == engs[False].symbol_backlog(s)
    for s, o in enumerate(orcs):
        assert [x for (ss, x) in seen[True] if ss == s] == o.sentences()


@pytest.mark.parametrize("ctx", ["mathh", "cmath"])
def test_golden_chain_through_the_gpu(hd, ctx):
    """tests/golden/chain_small: input stored on disk, expectations recorded from the reference's own stage classes.  The HIP
    path must reproduce the per-call SHA-1 of the decimated / filtered / demodulated floats, the bits and the text."""
    import hashlib
    import json
    from pathlib import Path
    gold = Path(__file__).resolve().parent / "golden"
    meta = json.loads((gold / "chain_small.json").read_text())
    q = np.load(gold / "chain_small_input.npz")["iq_int16"]
    x = (q[:, 0].astype(np.float32) / np.float32(meta["scale"]) + 1j * (q[:, 1].astype(np.float32) / np.float32(meta["scale"]))).astype(np.complex64)
    want = meta["expected"][ctx]
    Cn = meta["chunk"]
    eng = hd.Engine(n_streams=1, max_chunk=Cn, sampling_rate=meta["fs"], decimation=meta["factor"], baud=meta["baud"], rtty_bits=8, rtty_stops=2,
                    lookup_mode=1 if ctx == "mathh" else 0, keep_filtered=True)
    sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
    for k, i in enumerate(range(0, len(x), Cn)):
        eng.process_host(x[None, i:i + Cn])
        w = want["per_call"][k]
        assert [sha(eng.decimated(0)), sha(eng.filtered(0)), sha(eng.demodulated(0))] == w[:3], k
        assert eng.bits(0).tolist() == w[3], k
        a = eng.afc(0)
        assert (a["peak_l"], a["peak_r"]) == (w[4]["peak_l"], w[4]["peak_r"]), k
    assert eng.take_sentences(0) == want["sentences"] and eng.take_chars(0) == want["chars"]
    assert eng.rtty(0) == want["rtty"] and eng.last_sentence(0) == want["last"]


def test_flip_list_bound_delays_bits_but_loses_none(hd):
    """More flip points in one call than the device's flip list holds (512): noise through four-sample windows at 16384 decimated samples per call
    (/4, the reference's 160 kHz gate lifted).  The reference has no such bound; the engine stops at it and resumes in the next call (not
    an error any more).  The stream of bits must be the reference's -- per call the engine may lag behind, never differ: its cumulative bits are a
    prefix of the oracle's at every call and equal once some short calls (few new flips each, so the backlog shrinks instead of running into the
    30 000-sample vent) have let it catch up."""
    import habdec_amd
    from oracle import pyoracle
    fs, D, CH, S = 2.048e6, 4, 65536, 2
    baud = fs / D / 16.0                                   # 16 decimated samples per bit: R = 4
    r = np.random.default_rng(5)
    noisy, quiet, QC = 2, 40, 4096                         # two full pushes of noise, then short pushes of a clean alternating-bit signal
    n0 = noisy * CH
    iq = np.zeros((S, n0 + quiet * QC), np.complex64)
    for s in range(S):
        iq[s, :n0] = (0.3 * (r.standard_normal(n0) + 1j * r.standard_normal(n0))).astype(np.complex64)
        iq[s, n0:] = synth.fsk_iq(np.tile(np.array([0, 1], np.uint8), quiet * QC // 128 + 2), fs, baud, sigma=0.01, seed=70 + s, n_samples=quiet * QC)
    eng = habdec_amd.Engine(n_streams=S, max_chunk=CH, sampling_rate=fs, decimation=D, baud=baud, lowpass_bw_hz=200000.0, ungated=True)
    orcs = [pyoracle.Decoder("oracle", factor=D, baud=baud, lowpass_bw=200000.0, ungated=True) for _ in range(S)]
    gb, ob = [[] for _ in range(S)], [[] for _ in range(S)]
    lag, pos = 0, 0
    for k in range(noisy + quiet):
        n = CH if k < noisy else QC
        buf = np.zeros((S, CH), np.complex64)
        buf[:, :n] = iq[:, pos:pos + n]
        eng.process_host(buf, n)
        for s in range(S):
            orcs[s](iq[s, pos:pos + n], fs)
            gb[s] += list(eng.bits(s)); ob[s] += list(orcs[s].bits())
            assert len(gb[s]) <= len(ob[s]) and gb[s] == ob[s][:len(gb[s])], ("the engine's bits left the oracle's", k, s)
            lag = max(lag, len(ob[s]) - len(gb[s]))
        pos += n
    assert all(eng.flip_list_full(s) >= 1 for s in range(S)) and lag > 0, ([eng.flip_list_full(s) for s in range(S)], lag)     # the bound was reached at all
    for s in range(S):
        assert len(ob[s]) > 1000 and gb[s] == ob[s], (s, len(gb[s]), len(ob[s]))
        assert eng.symbol_backlog(s) == orcs[s].symex_held()
    eng.close()


def test_host_pushes_from_page_locked_memory_are_read_in_place(hd):
    """hd_pinned_alloc + hd_process_host on a synchronous engine (what the Decoder facade's input queue does since round 6): stage 1 reads the caller's
    buffer over PCIe, no staging copy (hd_timing.host_calls_in_place counts such calls) -- and the results are those of the copying path and of the oracle.
    In batch mode, and from a base the 16-byte loads cannot take, the same call copies as before."""
    from oracle import pyoracle
    S, fs = 3, 2.048e6
    iq, _ = make_streams(S, fs, 300, 8, 2, seed0=500)
    nch = iq.shape[1] // C
    eng = hd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, keep_filtered=True)
    pin = eng.pinned_array((S, C + 2))
    orcs = [pyoracle.Decoder("oracle", factor=64) for _ in range(S)]
    for k in range(nch):
        pin[:, :C] = iq[:, k * C:(k + 1) * C]
        hd.capi.check(eng.L.hd_process_host(eng.h, pin.ctypes.data, C + 2, None, C))
        for s in range(S):
            orcs[s](iq[s, k * C:(k + 1) * C], fs)
            assert same_bits(eng.decimated(s), orcs[s].array("last_decimated")), (k, s)
            assert same_bits(eng.demodulated(s), orcs[s].array("last_demod")), (k, s)
    assert eng.timing()["host_calls_in_place"] == nch
    for s in range(S):
        assert eng.take_sentences(s) == orcs[s].sentences() and eng.take_chars(s) == orcs[s].text("chars_log")
    # a base at an odd sample: the copying path
    hd.capi.check(eng.L.hd_process_host(eng.h, pin.ctypes.data + 8, C + 2, None, C))
    assert eng.timing()["host_calls_in_place"] == nch
    eng.close()
    # batch mode never reads in place: the caller may reuse the buffer while the call is still queued
    eng = hd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, pipeline=1)
    pin = eng.pinned_array((S, C))
    for k in range(4):
        pin[:] = iq[:, k * C:(k + 1) * C]
        eng.process_host(pin)
    eng.flush()
    assert eng.timing()["host_calls_in_place"] == 0
    eng.close()


@pytest.mark.parametrize("dc", [False, True])
@pytest.mark.parametrize("pipeline", [0, 1])
def test_tap_count_increase_after_idle_calls(hd, dc, pipeline):
    """The FirHistory head, lazily (dev_types.h, round 6): no run writes the first samples of its input aside any more -- a later run with MORE taps finds them in the
    previous call's low-pass buffer, or, when the stream sat out one or more calls in between, in the side buffer the first of those calls copied them to.  Streams
    that run every call, skip exactly one call, skip three calls, and skip the call in which the others change -- then 161 -> 1001 taps (840 head samples), back, and
    up again with other gaps -- against the oracle bit for bit, through the stream tail and (DC blocker on) through the separate kernels, synchronous and batch."""
    import habdec_amd
    from oracle import pyoracle
    S, fs = 4, 2.048e6
    iq, _ = make_streams(S, fs, 300, 8, 2, seed0=1700, repeat=6)
    nch = iq.shape[1] // C
    assert nch >= 16
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, baud=300, rtty_bits=8, rtty_stops=2, keep_filtered=True, dc_remove=dc, pipeline=pipeline)
    orcs = [pyoracle.Decoder("oracle", factor=64, baud=300, bits=8, stops=2, dc_remove=dc) for _ in range(S)]
    idle = {1: {5}, 2: {3, 4, 5}, 3: {6, 11, 12}}                # stream -> calls it sits out
    plan = {6: 0.004, 9: 0.025, 13: 0.004}                         # 161 -> 1001 -> 161 -> 1001 taps
    pos = [0] * S
    for k in range(16):
        if k in plan:
            for s in range(S):
                eng.set_lowpass_trans(s, plan[k]); orcs[s].lowpass_trans(plan[k])
        n = np.array([0 if k in idle.get(s, ()) else C for s in range(S)], np.uint32)
        buf = np.zeros((S, C), np.complex64)
        for s in range(S):
            buf[s, :n[s]] = iq[s, pos[s]:pos[s] + n[s]]
        habdec_amd.capi.check(eng.L.hd_process_host(eng.h, buf.ctypes.data, C, n.ctypes.data, 0))
        for s in range(S):
            if n[s]:
                orcs[s](iq[s, pos[s]:pos[s] + n[s]], fs)
                assert same_bits(eng.filtered(s), orcs[s].array("last_filtered")), ("filtered", k, s)
                assert same_bits(eng.demodulated(s), orcs[s].array("last_demod")), ("demod", k, s)
                assert np.array_equal(eng.bits(s), orcs[s].bits()), ("bits", k, s)
            pos[s] += int(n[s])
    eng.flush()
    taps = {len(eng.fir_taps(s)) for s in range(S)}
    assert taps == {len(orcs[0].array("fir_taps"))} and max(taps) > 900, taps          # the long design is the one in force at the end
    for s in range(S):
        assert eng.take_chars(s) == orcs[s].text("chars_log") and eng.take_sentences(s) == orcs[s].sentences()
