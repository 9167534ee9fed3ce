"""Parity AT THE LAUNCH SHAPES bench.py quotes numbers on, all against the CPU oracle through the C ABI:

  a. BASELINE configs[4] at bench size -- 512 streams x 2^20-sample pushes at 10 MS/s, /256 (/64 348 taps on the four-per-CU grid + /4 139
     taps), 4097-tap low-pass, batch mode as bench.py drives it -- per call (decimated / filtered / discriminator output bit-equal, bits,
     backlog) and free running with the end state and the text compared;
  b. the ring protocol of the per-CU step launch hammered instead of sampled: the headline workload free running, a fresh seed and fresh
     engines per iteration, every call of every sampled stream checked through the discriminator checksum its tail writes;
  c. `bench.run_workload`'s own self-check (`gpu_matches_oracle_on_sample`) for every workload the bench can be asked for, on a few steps.
"""
import numpy as np
import pytest

from habdec_amd import synth

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def ck(d):
    d = np.ascontiguousarray(d).view(np.uint32).astype(np.uint64)
    return len(d), int(d.sum() & 0xFFFFFFFF), int((d * np.arange(1, len(d) + 1, dtype=np.uint64)).sum() & 0xFFFFFFFF)


def test_configs4_at_bench_size():
    """512 streams x 2^20 samples per push (4 GiB per slab), 3 pushes: streams are delayed copies of one 300-baud 8N2 signal at 10 MS/s."""
    torch = pytest.importorskip("torch")
    import habdec_amd
    from oracle import pyoracle
    import bench
    w = bench.WORKLOADS["cfg5"]
    S, fs, CH, ncalls = w["S"], w["fs"], w["C"], 3
    assert (S, CH, w["D"]) == (512, 1 << 20, 256)
    frame = synth.rtty_bits(synth.make_sentence("BIG5", "1,52.1,21.4,100"), 8, 2, 3, 3)
    iq1 = synth.fsk_iq(np.concatenate([frame] * 3), fs, 300, sigma=0.06, seed=55, n_samples=ncalls * CH + 4096)
    shifts = (np.arange(S) * 29) % 4096
    base = torch.from_numpy(np.ascontiguousarray(iq1).view(np.float32).reshape(-1, 2)).cuda()
    slab = torch.empty((ncalls, S, CH, 2), dtype=torch.float32, device="cuda")
    for s in range(S):
        slab[:, s] = base[int(shifts[s]):int(shifts[s]) + ncalls * CH].view(ncalls, CH, 2)
    check = (0, 1, 255, 256, 300, 511)
    kw = dict(n_streams=S, max_chunk=CH, sampling_rate=fs, decimation=256, baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
              lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"])

    def oracles():
        return {s: pyoracle.Decoder("oracle", factor=256, baud=w["baud"], bits=w["bits"], stops=w["stops"], lowpass_bw=w["lp_bw"], lowpass_trans=w["lp_trans"]) for s in check}

    # (1) per call: each getter drains the pipeline, so every call's arrays can be compared
    eng = habdec_amd.Engine(pipeline=2, keep_filtered=True, **kw)
    orcs = oracles()
    for k in range(ncalls):
        eng.process_device(slab[k].data_ptr(), CH, CH)
        for s, o in orcs.items():
            o(iq1[int(shifts[s]) + k * CH: int(shifts[s]) + (k + 1) * CH], fs)
            assert len(eng.fir_taps(s)) == 4097 and same_bits(eng.fir_taps(s), o.array("fir_taps")), (k, s)
            assert same_bits(eng.decimated(s), o.array("last_decimated")), ("decimated", k, s)
            assert same_bits(eng.filtered(s), o.array("last_filtered")), ("filtered", k, s)
            assert same_bits(eng.demodulated(s), o.array("last_demod")), ("demod", k, s)
            assert np.array_equal(eng.bits(s), o.bits()), ("bits", k, s)
            assert eng.symbol_backlog(s) == o.symex_held(), ("backlog", k, s)
    assert eng.timing()["path"] == 0                         # the many-workgroup kernels: what bench.py --workload cfg5 times
    per_call_chars = {s: eng.take_chars(s) for s in check}
    for s, o in orcs.items():
        assert per_call_chars[s] == o.text("chars_log"), ("chars", s)
    eng.close()
    # (2) free running, exactly as bench.py drives it (three calls undelivered, nothing read in between), then the end state
    eng = habdec_amd.Engine(pipeline=2, **kw)
    for k in range(ncalls):
        eng.process_device(slab[k].data_ptr(), CH, CH)
    eng.flush()
    for s, o in orcs.items():
        assert same_bits(eng.decimated(s), o.array("last_decimated")), ("decimated, free running", s)
        assert same_bits(eng.demodulated(s), o.array("last_demod")), ("demod, free running", s)
        assert eng.symbol_backlog(s) == o.symex_held(), ("backlog, free running", s)
        assert eng.take_chars(s) == per_call_chars[s], ("chars, free running", s)
    assert sum(len(c) for c in per_call_chars.values()) > 0
    eng.close()
    del slab
    torch.cuda.empty_cache()


def test_step_launch_ring_protocol_repeatedly():
    """Six free-running passes of the headline workload through k_step_cu, each with its own ring (new seed), sampled streams moved around,
    every call's discriminator checksum compared with the oracle's.  The protocol's last bug (round 3: a descriptor entry rewritten under a
    stalled consumer) showed in a few per cent of such runs and in nothing else."""
    torch = pytest.importorskip("torch")
    import habdec_amd
    import bench
    from oracle import pyoracle
    w = dict(bench.WORKLOADS["cfg4"])
    S, fs, C = w["S"], w["fs"], w["C"]
    n_iter, n_calls = 6, 24
    bad = []
    for it in range(n_iter):
        ring, ring_chunks, _ = bench.generate_ring(torch, torch.device("cuda", 0), w, S, 0, seed=9000 + it)
        rs = np.random.default_rng(it)
        check = sorted(set([0, 1023, 7 + 8 * int(rs.integers(0, 128))] + [int(x) for x in rs.integers(0, S, 9)]))
        eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                                lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], pipeline=2 if it % 2 else 1)
        orcs = {s: pyoracle.Decoder("oracle", factor=w["D"], baud=w["baud"], bits=w["bits"], stops=w["stops"], lowpass_bw=w["lp_bw"]) for s in check}
        host = {s: ring[:n_calls, s].cpu().numpy().view(np.complex64).reshape(n_calls, C) for s in check}
        want, seen = {}, {s: set() for s in check}

        def compare_delivered():
            for s in check:
                ci, n, c0, c1 = eng.demod_checksum(s)
                if ci in want and ci not in seen[s]:
                    if n is None and ci == 0:          # (0, None) is what the read-out says before anything was delivered
                        continue
                    if (n, c0, c1) != want[ci][s]:
                        bad.append((it, ci, s))
                    seen[s].add(ci)
        for k in range(n_calls):
            eng.process_device(ring[k].data_ptr(), C, C)
            want[k] = {}
            for s, o in orcs.items():
                o(host[s][k], fs)
                want[k][s] = ck(o.array("last_demod"))
            compare_delivered()
        assert eng.timing()["path"] == 3 and eng.timing()["step_variant"] == 1
        eng.flush()
        compare_delivered()
        assert min(len(v) for v in seen.values()) >= n_calls - 5, (it, {s: len(v) for s, v in seen.items()})
        for s, o in orcs.items():
            if eng.take_chars(s) != o.text("chars_log") or not same_bits(eng.demodulated(s), o.array("last_demod")) or eng.symbol_backlog(s) != o.symex_held():
                bad.append((it, "end state", s))
        eng.close()
        del ring, host
        torch.cuda.empty_cache()
    assert not bad, bad


@pytest.mark.parametrize("name,steps", [("cfg1", 8), ("cfg2", 8), ("cfg3", 6), ("cfg5", 3), ("cfg4", 8)])
def test_bench_self_check_on_every_workload(name, steps):
    """bench.py compares the engine with the oracle on a few streams inside every run (`gpu_matches_oracle_on_sample`); here on a reduced stream
    count so that all five workloads fit the suite.  The workloads' launch paths are the ones the full-size runs take."""
    torch = pytest.importorskip("torch")
    import bench
    S = {"cfg5": 64}.get(name, 256)
    r = bench.run_workload(torch, None, torch.device("cuda", 0), 0, 0, 1, name, steps, 2, S, False, cpu_leg="check")
    cb = r["cpu_baseline"]
    assert cb["gpu_matches_oracle_on_sample"] is True, cb
    assert cb["bits_in_sample"] > 0


def test_bench_threads_mode_two_shards(monkeypatch):
    """`bench.py --gpus 2 --threads`: one process, an engine + host thread per device, no torchrun, no RCCL.  The box has one GPU, so both
    shards sit on device 0 (HD_BENCH_SAME_DEVICE): what is exercised is the threading -- two rings, two engines, two step loops sharing one interpreter,
    the timed region = the slower shard's -- and shard 0's self-check."""
    torch = pytest.importorskip("torch")
    import bench
    monkeypatch.setenv("HD_BENCH_SAME_DEVICE", "1")
    r = bench.run_workload(torch, None, torch.device("cuda", 0), 0, 0, 1, "cfg4", 12, 3, 256, False, cpu_leg="check", threads=2)
    assert r["cpu_baseline"]["gpu_matches_oracle_on_sample"] is True, r["cpu_baseline"]
    assert r["value"] > 0 and r["steps"] == 12
    # value counts both shards' samples over the slower shard's region
    assert r["value"] == pytest.approx(2 * 256 * 65536 * 12 / (r["timed_region_ms"] * 1e-3) / 1e6, rel=1e-2)


def test_bench_two_ranks_over_gloo_on_one_device():
    """The driver's N > 1 launch on hardware, as far as a one-GPU box can take it: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`
    with HD_BENCH_SAME_DEVICE=1 (both ranks put their shard on device 0; the line says so).  What is exercised: two processes with an engine and a ring each,
    the gloo process group, the barriers around the regions, the MAX of the ranks' own region times, every rank's self-check on sampled streams of ITS shard
    (exact leg, fast leg) and the gather of the verdicts on rank 0 -- tests/test_sharding_gloo.py covers the same plumbing without a GPU."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ, HD_BENCH_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(root / "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--prewarm", "0", "--streams", "128", "--no-cpu-baseline", "--no-also"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)          # (a child process: this one's GPU context stays as it is)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "same_device" in line
    assert line["value"] == pytest.approx(2 * 128 * 65536 * 8 / (line["timed_region_ms"] * 1e-3) / 1e6, rel=1e-2)
    assert [r["rank"] for r in line["per_rank"]] == [0, 1]
    assert all(r["gpu_matches_oracle"] is True for r in line["per_rank"]) and line["all_ranks_match_oracle"] is True
    f = line["fast"]
    assert f["parity"]["all_ranks_match_oracle"] is True and f["parity"]["symbols_characters_sentences_equal"] is True
    assert [r["gpu_matches_oracle"] for r in f["per_rank"]] == [True, True]
