"""GPU parity at the sizes and launch shapes bench.py actually runs (all against the CPU oracle, through the C ABI):

  a. the headline workload -- 1024 streams, 50 baud 7N2, bench.py's carrier-offset distribution (1/8 of the streams far off-tune: the
     busy path of the symbol extractor) -- in batch mode, i.e. through the STEP kernel (stage 1 + the previous call's stream tails in
     one launch), 16+ sampled streams incl. far-off ones: decimated / discriminator output bit-equal, bits, backlog, AFC per call,
     then a stretch of calls with nothing in between (two calls in flight) and the text compared at the end;
  b. the linear-split grid and the step kernels of the other single-wave first stages (/128: <32,174,64>, /256: <64,348,64>), which
     need 2^20-sample pushes to reach the tile count that switches them on;
  c. the classic grids of the 256-lane first stages at 1024 streams (/16: <8,54,256>, /4: <4,139,256>);
  d. two engines in one process, driven by two threads at once, against the same engines run one after the other.
"""
import threading

import numpy as np
import pytest

from habdec_amd import synth

pytestmark = pytest.mark.gpu
C = 65536


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.fixture(scope="module")
def headline_ring():
    torch = pytest.importorskip("torch")
    import bench
    w = dict(bench.WORKLOADS["cfg4"])
    ring, ring_chunks, _ = bench.generate_ring(torch, torch.device("cuda", 0), w, w["S"], 0, seed=77)
    yield w, ring, ring_chunks
    del ring
    torch.cuda.empty_cache()


@pytest.mark.parametrize("shares", ["drawn", "drawn2", "fixed", "deep", "rocfft", "single_wave", "run9", "run12", "short0", "short60"])
def test_headline_workload_through_the_step_kernel(monkeypatch, headline_ring, shares):
    """shares: which step kernel serves the batch and how its stage-1 tiles are handed out -- k_step_cu (one workgroup per CU: stage-1 worker waves
    that load their own tiles with LDS-DMA and sum them with the systolic tap loop, runs of four 57-output tiles drawn from per-XCD counters: the
    default; "run9" / "run12": runs of nine / twelve), or the single-wave
    k_step ("single_wave"; "drawn2": with runs of two -- HD_STEP_RUN does not reach k_step_cu) -- or the fixed shares of HD_NO_CLAIM (the first call of a stream, which restarts its history, always takes fixed shares); "deep" = pipeline 2,
    three calls undelivered in the free-running stretch."""
    import habdec_amd
    from oracle import pyoracle
    if shares == "drawn2":
        monkeypatch.setenv("HD_STEP_RUN", "2")
    if shares == "fixed":
        monkeypatch.setenv("HD_NO_CLAIM", "1")
    if shares == "rocfft":                                   # the spectra as launches of their own (rocFFT + commit) instead of inside the tails
        monkeypatch.setenv("HD_ROCFFT", "1")
    if shares == "single_wave":                              # round 2's step kernel (single-wave workgroups) instead of one workgroup per CU
        monkeypatch.setenv("HD_NO_CU_STEP", "1")
    if shares in ("run9", "run12"):                          # k_step_cu with nine- / twelve-tile runs (fewer run changes; 36 tiles per stream and call)
        monkeypatch.setenv("HD_RING_RUN", shares[3:])
    if shares in ("short0", "short60"):                      # the worker waves' guided hand-out: whole runs to the end of the launch / the last 60 % of an XCD's tiles as single tiles (default 25 %)
        monkeypatch.setenv("HD_RING_SHORT_PCT", shares[5:])
    w, ring, ring_chunks = headline_ring
    S, fs = w["S"], w["fs"]
    # 7/8 of the streams are within +-200 Hz, every 8th is far off: sample both kinds (and the first / last stream of XCD blocks)
    check = [0, 1, 2, 127, 128, 500, 511, 512, 640, 1000, 1022, 7, 15, 263, 775, 1023]
    assert sum(1 for s in check if s % 8 == 7) >= 4 and len(set(check)) >= 16
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                            lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], pipeline=2 if shares == "deep" else 1)
    orcs = {s: pyoracle.Decoder("oracle", factor=w["D"], baud=w["baud"], bits=w["bits"], stops=w["stops"], lowpass_bw=w["lp_bw"]) for s in check}
    host = {s: ring[:, s].cpu().numpy().view(np.complex64).reshape(ring_chunks, C) for s in check}
    n_checked, n_free = 12, 36
    obits = {s: 0 for s in check}
    for k in range(n_checked):                               # per-call comparison (each getter waits for the call: the tails run as their own launch)
        eng.process_device(ring[k % ring_chunks].data_ptr(), C, C)
        for s, o in orcs.items():
            o(host[s][k % ring_chunks], fs)
            assert same_bits(eng.decimated(s), o.array("last_decimated")), ("decimated", k, s)
            assert same_bits(eng.demodulated(s), o.array("last_demod")), ("demod", k, s)
            assert np.array_equal(eng.bits(s), o.bits()), ("bits", k, s)
            obits[s] += len(o.bits())
            assert eng.symbol_backlog(s) == o.symex_held(), ("backlog", k, s)
            ga, oa = eng.afc(s), o.afc()
            assert (ga["peak_l"], ga["peak_r"]) == (oa["peak_l"], oa["peak_r"]), ("peaks", k, s)
            assert ga["correction"] == pytest.approx(oa["correction"], rel=1e-9, abs=1e-9), ("afc", k, s)
    assert eng.timing()["path"] == 3                         # the step kernel is what ran
    # Free running: two (three) calls in flight, every call's tails INSIDE the next call's step launch -- the launch shape bench.py times.
    # Asking for samples would flush the pipeline and turn the pending tails into a launch of their own, so each call is checked through
    # the checksum of its discriminator output that the tail leaves in the result slot (BitsHeader::demod_ck) -- call by call, against the
    # oracle's output of the same call -- plus the symbols it produced.
    want, seen = {}, {s: set() for s in check}
    def compare_delivered():
        for s in check:
            ci, n, c0, c1 = eng.demod_checksum(s)
            if ci in want and ci not in seen[s]:
                assert (n, c0, c1) == want[ci][s], ("discriminator checksum of a call served inside a step launch", ci, s)
                seen[s].add(ci)
    for k in range(n_checked, n_checked + n_free):
        eng.process_device(ring[k % ring_chunks].data_ptr(), C, C)
        want[k] = {}
        for s, o in orcs.items():
            o(host[s][k % ring_chunks], fs)
            obits[s] += len(o.bits())
            d = o.array("last_demod").view(np.uint32).astype(np.uint64)
            want[k][s] = (len(d), int(d.sum() & 0xFFFFFFFF), int((d * np.arange(1, len(d) + 1, dtype=np.uint64)).sum() & 0xFFFFFFFF))
        compare_delivered()
    assert eng.timing()["path"] == 3
    # which step kernel: one workgroup per CU (worker waves, stage1_ring.h) unless switched off or the runs are not drawn
    assert eng.timing()["step_variant"] == (0 if shares in ("fixed", "single_wave") else 1), shares
    in_launch = min(len(v) for v in seen.values())           # calls whose tails rode in a step launch and were compared before the final flush
    assert in_launch >= n_free - 4, in_launch
    eng.flush()
    compare_delivered()
    busy = 0
    for s, o in orcs.items():
        assert eng.take_chars(s) == o.text("chars_log"), ("chars", s)
        assert eng.rtty(s) == o.text("rtty_stream"), ("rtty", s)
        assert eng.bits_total(s) == obits[s], ("symbols produced", s)
        assert eng.symbol_backlog(s) == o.symex_held(), ("backlog", s)
        assert same_bits(eng.demodulated(s), o.array("last_demod")), ("demod at the end", s)
        busy += int(s % 8 == 7)
    assert busy >= 4
    eng.close()


@pytest.mark.parametrize("kernel", ["per_cu", "single_wave"])
def test_stage1_alone_at_full_size(monkeypatch, headline_ring, kernel):
    """Synchronous calls (what the Decoder facade makes): stage 1 as a launch of its own, then the stream tails.  "per_cu": one workgroup per CU of eight
    worker waves, each loading its own tiles with LDS-DMA (k_stage1_cu, the default for a /32 first stage); "single_wave": k_decimate with fixed shares."""
    import habdec_amd
    from oracle import pyoracle
    if not kernel.startswith("per_cu"):
        monkeypatch.setenv("HD_NO_CU_STEP", "1")
    w, ring, ring_chunks = headline_ring
    S, fs = w["S"], w["fs"]
    check = [0, 127, 128, 511, 512, 775, 1023]
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                            lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"])
    orcs = {s: pyoracle.Decoder("oracle", factor=w["D"], baud=w["baud"], bits=w["bits"], stops=w["stops"], lowpass_bw=w["lp_bw"]) for s in check}
    for k in range(7):
        eng.process_device(ring[k].data_ptr(), C, C)
        for s, o in orcs.items():
            o(ring[k, s].cpu().numpy().view(np.complex64).reshape(-1), fs)
            assert same_bits(eng.decimated(s), o.array("last_decimated")), ("decimated", k, s)
            assert same_bits(eng.demodulated(s), o.array("last_demod")), ("demod", k, s)
            assert np.array_equal(eng.bits(s), o.bits()), ("bits", k, s)
    assert eng.timing()["path"] == 2 and eng.timing()["step_variant"] == (1 if kernel.startswith("per_cu") else 0)
    eng.close()


@pytest.mark.parametrize("factor,S,force_tail", [(256, 24, True), (256, 24, False), (128, 12, True), (128, 12, False)])
def test_linear_split_and_step_kernels_of_the_other_single_wave_stages(monkeypatch, factor, S, force_tail):
    """/256 = <64,348,64> + <4,139>, /128 = <32,174,64> + <4,139>: with 2^20-sample pushes the batch reaches the 6144 tiles that switch
    the linear split on.  force_tail lifts the engine's call-size limit for the stream tail, so batch mode runs k_step<64,348,4,139> /
    k_step<32,174,4,139>; without it the split feeds the separate back-half kernels."""
    torch = pytest.importorskip("torch")
    import habdec_amd
    from oracle import pyoracle
    if force_tail:
        monkeypatch.setenv("HD_TAIL_MAX_N2", "1000000")
    fs, CH = 2.048e6, 1 << 20
    text = synth.make_sentence("LIN", "1,52.1,21.4,100")
    iq1 = synth.fsk_iq_for_text(text, fs, 300, 8, 2, sigma=0.08, seed=21, idle_before=8, idle_after=12, chunk=CH)
    nch = max(2, min(3, len(iq1) // CH))
    need = nch * CH + 4096
    if len(iq1) < need:
        iq1 = np.concatenate([iq1, synth.fsk_iq(np.ones(4, np.uint8), fs, 300, sigma=0.08, seed=22, n_samples=need - len(iq1))])
    shifts = (np.arange(S) * 173) % 4096
    base = torch.from_numpy(np.ascontiguousarray(iq1).view(np.float32).reshape(-1, 2)).cuda()
    slab = torch.empty((nch, S, CH, 2), dtype=torch.float32, device="cuda")
    for s in range(S):
        slab[:, s] = base[int(shifts[s]):int(shifts[s]) + nch * CH].view(nch, CH, 2)
    eng = habdec_amd.Engine(n_streams=S, max_chunk=CH, sampling_rate=fs, decimation=factor, pipeline=True)
    check = (0, 1, S // 2, S - 1)
    orcs = {s: pyoracle.Decoder("oracle", factor=factor) for s in check}
    for k in range(nch):
        eng.process_device(slab[k].data_ptr(), CH, CH)
        eng.flush()
        for s, o in orcs.items():
            o(iq1[int(shifts[s]) + k * CH: int(shifts[s]) + (k + 1) * CH], fs)
            assert same_bits(eng.decimated(s), o.array("last_decimated")), (k, s)
            assert same_bits(eng.demodulated(s), o.array("last_demod")), (k, s)
            assert np.array_equal(eng.bits(s), o.bits()), (k, s)
    # (both plans' tails fit their step workgroup since the search phase's LDS is carved per stream -- round 3's /128 tail needed a little more
    # than k_step<32,174,4,139>'s 19.2 KB and ran as a kernel of its own)
    assert eng.timing()["path"] == (3 if force_tail else 0)
    for s, o in orcs.items():
        assert eng.take_chars(s) == o.text("chars_log")
    eng.close()


@pytest.mark.parametrize("factor,fs,ungated,per_cu", [(16, 2.5e6, False, True), (16, 2.5e6, False, False), (4, 2.048e6, True, True), (4, 2.048e6, True, False)])
def test_first_stages_of_the_small_ratios_at_1024_streams(monkeypatch, factor, fs, ungated, per_cu):
    """/16 (/8 54 taps -- as k_stage1_cu<54,8>, one workgroup per CU, and as the classic grid k_decimate<8,54,256> -- + <2,69,256>) and
    /4 (139 taps, the only stage of its plan: k_stage1_cu<139,4> with eight outputs per lane writing behind the low-pass buffer's pending samples
    and feeding the spectrum collection, and the classic grid <4,139,256>) at the batch size the bench uses: 1024 streams x 65536 samples per call, every stream its own delayed copy of one
    signal, sampled streams against the oracle bit for bit."""
    torch = pytest.importorskip("torch")
    if not per_cu:
        monkeypatch.setenv("HD_NO_CU_STEP", "1")
    import habdec_amd
    from oracle import pyoracle
    S = 1024
    text = synth.make_sentence("GRID", "1,52.1,21.4,100")
    iq1 = synth.fsk_iq_for_text(text, fs, 300, 8, 2, sigma=0.08, seed=31, idle_before=8, idle_after=12)
    nch = min(4, len(iq1) // C - 1)
    shifts = (np.arange(S) * 37) % 4096
    base = torch.from_numpy(np.ascontiguousarray(iq1).view(np.float32).reshape(-1, 2)).cuda()
    slab = torch.empty((nch, S, C, 2), dtype=torch.float32, device="cuda")
    for s in range(S):
        slab[:, s] = base[int(shifts[s]):int(shifts[s]) + nch * C].view(nch, C, 2)
    kw = dict(lowpass_bw_hz=3000.0) if factor == 16 else {}
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=factor, ungated=ungated, **kw)
    check = (0, 3, 255, 256, 700, 1023)
    orcs = {s: pyoracle.Decoder("oracle", factor=factor, ungated=ungated, lowpass_bw=(3000.0 if factor == 16 else None)) for s in check}
    for k in range(nch):
        eng.process_device(slab[k].data_ptr(), C, C)
        for s, o in orcs.items():
            o(iq1[int(shifts[s]) + k * C: int(shifts[s]) + (k + 1) * C], fs)
            assert same_bits(eng.decimated(s), o.array("last_decimated")), (k, s)
            assert same_bits(eng.demodulated(s), o.array("last_demod")), (k, s)
            assert np.array_equal(eng.bits(s), o.bits()), (k, s)
            assert eng.symbol_backlog(s) == o.symex_held(), (k, s)
            ga, oa = eng.afc(s), o.afc()                       # (the spectrum collection is fed by the final stage's epilogue: the ring consumer's at /4)
            assert (ga["peak_l"], ga["peak_r"]) == (oa["peak_l"], oa["peak_r"]), ("peaks", k, s)
            assert ga["correction"] == pytest.approx(oa["correction"], rel=1e-9, abs=1e-9), ("afc", k, s)
    assert eng.timing()["step_variant"] == (1 if per_cu else 0)
    eng.close()


@pytest.mark.parametrize("factor,fs,S,CH,lowpass", [(16, 2.5e6, 8, 16384, 3000.0), (16, 2.5e6, 24, 4096, 3000.0), (64, 2.048e6, 8, 16384, None), (64, 2.048e6, 40, 4096, None), (64, 2.048e6, 3, 4096, None),
                                                     (256, 10e6, 8, 16384, None), (256, 10e6, 16, 24576, None), (128, 4.096e6, 16, 8192, None)])
def test_per_cu_first_stage_with_few_tiles(factor, fs, S, CH, lowpass):
    """k_stage1_cu with far fewer tiles than the chip has waves: a handful of streams and short pushes -- nine, five, three tiles per stream and call for the
    worker waves of the /32 and /64 stages (57- and 58-output tiles; a run may run on into the next stream), eight or two for the loader / consumer
    kernel of /8 -- most CUs finding no run at all, and the fall-back to the classic grid where no run length divides among the XCDs.  Synchronous
    calls, EVERY stream compared with the oracle after every call."""
    torch = pytest.importorskip("torch")
    import habdec_amd
    from oracle import pyoracle
    text = synth.make_sentence("FEW", "1,52.1,21.4,100")
    iq1 = synth.fsk_iq_for_text(text, fs, 300, 8, 2, sigma=0.08, seed=41, idle_before=4, idle_after=6)
    nch = min(6, (len(iq1) - 4096) // CH)
    assert nch >= 4
    shifts = (np.arange(S) * 97) % 4096
    base = torch.from_numpy(np.ascontiguousarray(iq1).view(np.float32).reshape(-1, 2)).cuda()
    slab = torch.empty((nch, S, CH, 2), dtype=torch.float32, device="cuda")
    for s in range(S):
        slab[:, s] = base[int(shifts[s]):int(shifts[s]) + nch * CH].view(nch, CH, 2)
    kw = dict(lowpass_bw_hz=lowpass) if lowpass else {}
    eng = habdec_amd.Engine(n_streams=S, max_chunk=CH, sampling_rate=fs, decimation=factor, **kw)
    orcs = [pyoracle.Decoder("oracle", factor=factor, lowpass_bw=lowpass) for _ in range(S)]
    variants = []
    for k in range(nch):
        eng.process_device(slab[k].data_ptr(), CH, CH)
        variants.append(eng.timing()["step_variant"])
        for s, o in enumerate(orcs):
            o(iq1[int(shifts[s]) + k * CH: int(shifts[s]) + (k + 1) * CH], fs)
            assert same_bits(eng.decimated(s), o.array("last_decimated")), (k, s)
            assert same_bits(eng.demodulated(s), o.array("last_demod")), (k, s)
            assert np.array_equal(eng.bits(s), o.bits()), (k, s)
    # the first call of a stream restarts its history (classic grid); after that the per-CU kernel serves every call whose tiles can be cut into
    # runs that divide evenly among the 8 XCDs (engine.cpp: pick_ring_run, make_claim) -- otherwise the classic grid stays
    r1, t1 = {16: (8, 54), 64: (32, 212), 128: (32, 174), 256: (64, 348)}[factor]
    worker = r1 >= 32
    if worker:
        adv = 64 - (t1 - 1 + r1 - 1) // r1
        ntiles = -(-(CH // r1) // adv)
    else:
        ntiles = CH // 2048
    total = S * ntiles
    ok = lambda r: 2 <= r <= ntiles and (total % r == 0 if worker else ntiles % r == 0) and (total // r) % 8 == 0
    want = 4 if worker else 8
    per_cu = any(ok(want + d) or (0 < d < want and ok(want - d)) for d in range(9))
    assert variants[0] == 0 and all(v == (1 if per_cu else 0) for v in variants[1:]), (variants, per_cu)
    eng.close()


@pytest.mark.parametrize("S,CH", [(8, 16384), (13, 16384), (64, 4096), (1040, 4096)])
def test_step_launches_with_few_tiles_or_more_streams_than_tail_waves(S, CH):
    """k_step_cu away from the bench's shape: a handful of streams (most CUs draw no run and have no tail to run), a stream count whose runs
    do not divide among the XCDs (13: the single-wave k_step with fixed shares serves the batch), and 1040 streams -- more than the four tail
    waves of 256 workgroups hold, so the grid grows by four workgroups whose stage-1 waves find the counters empty.  Batch mode, free running;
    every call of EVERY stream is checked through the discriminator checksum its tail writes, and the end state against the oracle."""
    torch = pytest.importorskip("torch")
    import habdec_amd
    from oracle import pyoracle
    fs = 2.048e6
    text = synth.make_sentence("STEP", "1,52.1,21.4,100")
    iq1 = synth.fsk_iq_for_text(text, fs, 300, 8, 2, sigma=0.08, seed=43, idle_before=4, idle_after=6)
    nch = min(10, (len(iq1) - 4096) // CH)
    assert nch >= 6
    shifts = (np.arange(S) * 97) % 4096
    check = list(range(S)) if S <= 64 else [0, 1, 255, 256, 1023, 1024, 1031, 1039]
    base = torch.from_numpy(np.ascontiguousarray(iq1).view(np.float32).reshape(-1, 2)).cuda()
    slab = torch.empty((nch, S, CH, 2), dtype=torch.float32, device="cuda")
    for s in range(S):
        slab[:, s] = base[int(shifts[s]):int(shifts[s]) + nch * CH].view(nch, CH, 2)
    eng = habdec_amd.Engine(n_streams=S, max_chunk=CH, sampling_rate=fs, decimation=64, pipeline=True)
    orcs = {s: pyoracle.Decoder("oracle", factor=64) for s in check}
    want, seen, obits = {}, {s: set() for s in check}, {s: 0 for s in check}
    def compare_delivered():
        for s in check:
            ci, n, c0, c1 = eng.demod_checksum(s)
            if ci in want and ci not in seen[s]:
                # (no checksum: a call without discriminator output; and (0, None) is what the read-out says before anything was delivered)
                if n is None and (want[ci][s][0] == 0 or ci == 0):
                    continue
                assert (n, c0, c1) == want[ci][s], ("discriminator checksum", ci, s)
                seen[s].add(ci)
    variants = []
    for k in range(nch):
        eng.process_device(slab[k].data_ptr(), CH, CH)
        variants.append(eng.timing()["step_variant"])
        want[k] = {}
        for s, o in orcs.items():
            o(iq1[int(shifts[s]) + k * CH: int(shifts[s]) + (k + 1) * CH], fs)
            obits[s] += len(o.bits())
            d = o.array("last_demod").view(np.uint32).astype(np.uint64)
            want[k][s] = (len(d), int(d.sum() & 0xFFFFFFFF), int((d * np.arange(1, len(d) + 1, dtype=np.uint64)).sum() & 0xFFFFFFFF))
        compare_delivered()
    assert eng.timing()["path"] == 3
    ntiles, run = CH // 2048, 8
    while run > 2 and (ntiles % run or (S * ntiles // run) % 8):
        run //= 2
    per_cu = ntiles % run == 0 and (S * ntiles // run) % 8 == 0
    assert variants[0] == 0 and all(v == (1 if per_cu else 0) for v in variants[1:]), (variants, per_cu)
    eng.flush()
    compare_delivered()
    assert min(len(v) for v in seen.values()) >= nch - 3      # (the checksum read-out holds the latest delivered call: the flush delivers the last two at once)
    for s, o in orcs.items():
        assert eng.take_chars(s) == o.text("chars_log"), ("chars", s)
        assert eng.bits_total(s) == obits[s], ("symbols produced", s)
        assert eng.symbol_backlog(s) == o.symex_held(), ("backlog", s)
        assert same_bits(eng.demodulated(s), o.array("last_demod")), ("demod at the end", s)
    eng.close()


def test_two_engines_in_one_process_on_two_threads():
    """"one engine + HIP stream + host thread per device".  Two engines (different plans) fed concurrently by two host
    threads must give what the same engines give when run one after the other -- nothing in the library is shared between engines
    except read-only tables (and the per-thread error string)."""
    import habdec_amd
    fs = 2.048e6
    S = 8
    jobs = []
    for j, (factor, baud, bits) in enumerate([(64, 300, 8), (16, 300, 8)]):
        texts = [synth.make_sentence(f"T{j}S{s}", f"{s},52.{s},21.{j}") * 2 for s in range(S)]
        nchunks = int(np.ceil((max(len(t) for t in texts) * (1 + bits + 2) + 40) * fs / baud / C)) + 1
        iq = np.zeros((S, nchunks * C), np.complex64)
        for s in range(S):
            iq[s] = synth.fsk_iq(synth.rtty_bits(texts[s], bits, 2, 6 + s, 10), fs, baud, sigma=0.08, seed=100 * j + s, n_samples=nchunks * C)
        jobs.append((factor, baud, bits, iq))

    def run(job, out, pipeline):
        factor, baud, bits, iq = job
        eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=factor, baud=baud, rtty_bits=bits, rtty_stops=2, pipeline=pipeline)
        for k in range(iq.shape[1] // C):
            eng.process_host(np.ascontiguousarray(iq[:, k * C:(k + 1) * C]))
        eng.flush()
        out.append([(eng.take_chars(s), eng.take_sentences(s), eng.rtty(s), eng.symbol_backlog(s), eng.demodulated(s).tobytes()) for s in range(S)])
        eng.close()

    for pipeline in (False, True):
        serial = [[], []]
        for j in range(2):
            run(jobs[j], serial[j], pipeline)
        threaded = [[], []]
        th = [threading.Thread(target=run, args=(jobs[j], threaded[j], pipeline)) for j in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert threaded == serial
        assert all(len(x[1]) >= 1 for x in serial[0][0])          # (sentences were decoded at all)


def test_one_engine_per_device_in_one_process():
    """SURVEY section 8(e) as it is written: one engine + host thread PER DEVICE in one process (bench.py's N-GPU driver is one process per GPU).
    Needs two visible GPUs; the boxes the suite usually runs on have one, where this is skipped."""
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible GPU")
    import habdec_amd
    fs, S = 2.048e6, 16
    texts = [synth.make_sentence(f"DEV{s}", f"{s},52.{s},21.0") * 2 for s in range(S)]
    nchunks = int(np.ceil((max(len(t) for t in texts) * 11 + 40) * fs / 300 / C)) + 1
    iq = np.stack([synth.fsk_iq(synth.rtty_bits(texts[s], 8, 2, 6 + s, 10), fs, 300, sigma=0.08, seed=500 + s, n_samples=nchunks * C) for s in range(S)])
    out = {}

    def run(dev):
        eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, device=dev, pipeline=1)
        for k in range(nchunks):
            eng.process_host(np.ascontiguousarray(iq[:, k * C:(k + 1) * C]))
        eng.flush()
        out[dev] = [(eng.take_chars(s), eng.take_sentences(s), eng.rtty(s), eng.demodulated(s).tobytes()) for s in range(S)]
        eng.close()

    th = [threading.Thread(target=run, args=(d,)) for d in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert out[0] == out[1] and all(len(x[1]) >= 1 for x in out[0])


@pytest.mark.parametrize("pipeline", [False, True])
def test_long_idle_backlog_vent_and_parameter_change(pipeline):
    """What a stream does between transmissions: an idle carrier produces no flip points, the symbol extractor's backlog grows call by call
    until the reference's vent drops it (more than 30000 samples held, SymbolExtractor.h:116-120); a baud change in that state makes
    every cached window sum stale (the whole backlog is recomputed in one call).  Bits and backlog per call against the oracle, through
    the stream-tail kernel (synchronous) and through the step kernel (batch mode)."""
    import habdec_amd
    from oracle import pyoracle
    fs, S = 2.048e6, 3
    ncalls = 64
    text = synth.make_sentence("V", "1") * 2                         # 22 characters: 0.8 s at 300 baud
    frame = synth.rtty_bits(text, 8, 2, 4, 4)
    iq = np.zeros((S, ncalls * C), np.complex64)
    idle_bits = int(36 * C * 300 / fs)
    iq[0] = synth.fsk_iq(np.concatenate([np.ones(idle_bits, np.uint8), frame]), fs, 300, sigma=0.02, seed=1, n_samples=ncalls * C)   # idle, then a sentence
    iq[1] = synth.fsk_iq(frame, fs, 300, sigma=0.02, seed=2, n_samples=ncalls * C)                                                  # a sentence, then idle for good
    iq[2] = synth.fsk_iq(np.ones(8, np.uint8), fs, 300, sigma=0.02, seed=3, n_samples=ncalls * C)                                   # idle throughout, baud changed on the way
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, pipeline=pipeline)
    orcs = [pyoracle.Decoder("oracle", factor=64) for _ in range(S)]
    peak = 0
    for k in range(ncalls):
        if k in (20, 31):
            b = 50.0 if k == 20 else 300.0
            eng.set_baud(2, b); orcs[2].set_baud(b)
        eng.process_host(np.ascontiguousarray(iq[:, k * C:(k + 1) * C]))
        eng.flush()
        for s in range(S):
            orcs[s](iq[s, k * C:(k + 1) * C], fs)
            assert np.array_equal(eng.bits(s), orcs[s].bits()), ("bits", k, s)
            assert eng.symbol_backlog(s) == orcs[s].symex_held(), ("backlog", k, s)
            assert same_bits(eng.demodulated(s), orcs[s].array("last_demod")), ("demod", k, s)
        peak = max(peak, eng.symbol_backlog(0))
    assert peak > 30000                                             # stream 0's backlog did reach the vent
    assert eng.take_sentences(0) == orcs[0].sentences() and len(orcs[0].sentences()) >= 1       # decoded after the vent
    assert eng.take_sentences(1) == orcs[1].sentences() and len(orcs[1].sentences()) >= 1
    for s in range(S):
        assert eng.take_chars(s) == orcs[s].text("chars_log")
    eng.close()


@pytest.mark.parametrize("case", ["odd_stream_count", "pushes_grow"])
def test_tails_meet_a_launch_other_than_the_one_they_were_laid_out_for(case):
    """in step mode a call's tails are laid out when the call is enqueued and run inside the NEXT call's launch.  Where the per-CU kernel serves the
    shape they get its 23 KB slice of LDS -- and must not then ride in the single-wave fallback, whose workgroups own 20 KB (reads past the allocation
    return 0, writes are dropped: wrong window sums and flips, no error).  Two ways to get there: a stream count whose tile runs never divide among the XCDs
    (1001: every launch is the fallback -- such batches keep the 20 KB layout), and pushes that grow (16384 -> 32768 -> 65536 samples: the launch that
    carries the last small call's tails restarts the stage-1 histories and cannot be the per-CU kernel -- those tails run as a launch of their own).  50 baud
    with bench.py's carrier offsets: seconds of backlog, the search phase's caches full.  Free running, every call of the sampled streams compared through
    the folded discriminator checksums, then symbols, characters and backlog."""
    torch = pytest.importorskip("torch")
    import bench
    import habdec_amd
    from oracle import pyoracle
    w = dict(bench.WORKLOADS["cfg4"])
    S = 1001 if case == "odd_stream_count" else 1024
    fs = w["fs"]
    ring, ring_chunks, _ = bench.generate_ring(torch, torch.device("cuda", 0), w, S, 0, seed=4242)
    sizes = [C] * 40 if case == "odd_stream_count" else [16384] * 12 + [32768] * 12 + [C] * 12
    check = [0, 1, 7, 15, 127, 500, 503, 775, S - 2, S - 1]
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                            lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], pipeline=1)
    orcs = {s: pyoracle.Decoder("oracle", factor=w["D"], baud=w["baud"], bits=w["bits"], stops=w["stops"], lowpass_bw=w["lp_bw"]) for s in check}
    host = {s: ring[:, s].cpu().numpy().view(np.complex64).reshape(ring_chunks, C) for s in check}
    want = {s: 0xCBF29CE484222325 for s in check}
    obits = {s: 0 for s in check}
    variants = set()
    for k, n in enumerate(sizes):
        eng.process_device(ring[k % ring_chunks].data_ptr(), C, n)
        variants.add((eng.timing()["path"], eng.timing()["step_variant"]))
        for s, o in orcs.items():
            o(host[s][k % ring_chunks][:n], fs)
            obits[s] += len(o.bits())
            d = o.array("last_demod").view(np.uint32).astype(np.uint64)
            for x in (len(d), int(d.sum() & 0xFFFFFFFF), int((d * np.arange(1, len(d) + 1, dtype=np.uint64)).sum() & 0xFFFFFFFF)):
                want[s] = ((want[s] ^ x) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    eng.flush()
    assert all(p == 3 for p, _ in variants), variants
    if case == "odd_stream_count":
        assert variants == {(3, 0)}, variants                  # never the per-CU kernel
    else:
        assert (3, 1) in variants and (3, 0) in variants, variants     # both kernels took turns
    for s, o in orcs.items():
        ncalls, unknown, h = eng.demod_checksum_total(s)
        assert (ncalls, unknown) == (len(sizes), 0), (s, ncalls, unknown)
        assert h == want[s], ("discriminator output of some call differs", s)
        assert eng.bits_total(s) == obits[s], ("symbols produced", s)
        assert eng.take_chars(s) == o.text("chars_log"), ("chars", s)
        assert eng.symbol_backlog(s) == o.symex_held(), ("backlog", s)
    assert sum(obits.values()) > 100
    eng.close()
    del ring
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name,steps", [("cfg4", 40), ("cfg2", 12), ("cfg3", 12), ("cfg5", 4)])
def test_every_stream_of_the_workload_against_the_oracle(name, steps):
    """bench.py's own loop at the workload's FULL stream count inside the suite -- symbols produced, characters, sentences and (through the
    folded checksums the kernels leave in the result slots) every call's discriminator output bit for bit, for every stream of the shard over every step the
    engine took (cfg4: 1024 streams through k_step_cu; cfg2 / cfg3 / cfg5: the separate kernels on two queues)."""
    torch = pytest.importorskip("torch")
    import bench
    r = bench.run_workload(torch, None, torch.device("cuda", 0), 0, 0, 1, name, steps, 2, 0, False, cpu_leg="check_all", prewarm=0, min_kernel_samples=4)
    cb = r["cpu_baseline"]
    S = bench.WORKLOADS[name]["S"]
    assert r["arith"] == "exact" and cb["all_streams_of_the_shard"] is True and cb["streams_in_sample"] == S, cb
    assert cb["gpu_matches_oracle_on_sample"] is True, cb
    assert cb["discriminator_checksums_compared"] == S * cb["steps_checked"] and cb["steps_checked"] >= steps + 2, cb
    assert cb["bits_in_sample"] > 0
    assert r["per_rank"][0]["gpu_matches_oracle"] is True and r["all_ranks_match_oracle"] is True
