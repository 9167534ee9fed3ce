"""The N>1 path of bench.py without GPUs: two processes over gloo.  Streams shard with no data-path collective; the only
cross-rank traffic is the barrier and the MAX all-reduce of the timed region, which must yield the slowest rank's time on
every rank.  (The per-rank decode itself is covered by the GPU parity tests; here rank r decodes its shard with the CPU
oracle to show that shards are disjoint and complete.)"""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    import bench
    from habdec_amd import synth
    from oracle import pyoracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = 2
    ids = list(bench.shard(rank, world, S))
    fs, C = 48000.0, 4096
    decoded = []
    for local, gid in enumerate(ids):
        text = bench.stream_text(rank, local)
        iq = synth.fsk_iq_for_text(text, fs, 600, 8, 2, chunk=C, sigma=0.05, seed=gid, idle_before=6, idle_after=12)
        d = pyoracle.Decoder("oracle", factor=2, baud=600, bits=8, stops=2)
        for i in range(0, len(iq), C):
            d(iq[i:i + C], fs)
        decoded += [(gid, s) for s in d.sentences()]
    dist.barrier()
    dt = bench.job_time(dist, 1.0 + rank)          # rank 1 is "slower"
    gathered = [None] * world
    dist.all_gather_object(gathered, decoded)
    q.put((rank, ids, dt, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_over_gloo():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, ids0, dt0, g0), (r1, ids1, dt1, g1) = res
    assert ids0 == [0, 1] and ids1 == [2, 3]                       # disjoint, complete
    assert dt0 == dt1 == 2.0                                        # every rank sees the slowest rank's time
    flat = [x for part in g0 for x in part]
    assert g0 == g1 and sorted({gid for gid, _ in flat}) == [0, 1, 2, 3]
    # every stream decoded its own rank-tagged sentence
    for gid, sent in flat:
        assert sent.startswith(f"R{gid // 2}S{gid % 2:04d},")


def test_algorithmic_bytes_model():
    sys.path.insert(0, str(ROOT))
    import bench
    assert bench.bytes_per_sample(64) == pytest.approx(8.1875) and bench.bytes_per_sample(4) == 11.0
    assert set(bench.WORKLOADS) == {"cfg1", "cfg2", "cfg3", "cfg4", "cfg5"}


def _bench(*args, env=None):
    import json
    import subprocess
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=300, env=e)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_gpus_n_as_a_plain_command_starts_n_ranks():
    """`python bench.py --gpus 8` without torchrun ran on ONE GPU and said n_gpus: 1.  The plain command now starts the ranks itself (the parent
    never imports torch), relays rank 0's line and fails when the ranks fail; a world size that differs from --gpus is an error.  --dry-run: the launch
    plumbing alone (gloo), no GPU."""
    rc, line, err = _bench("--gpus", "2", "--dry-run", "--steps", "3")
    assert rc == 0 and line is not None, err[-2000:]
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["steps"] == 3
    assert line["shards"] == [[0, 1024], [1024, 2048]]
    assert line["slowest_rank_ms"] == pytest.approx(2.0)            # the MAX over ranks (rank r reports r + 1 ms)
    assert [v["rank"] for v in line["per_rank"]] == [0, 1] and line["process_group"] == "gloo"     # every rank's verdict reaches rank 0's line; no RCCL
    rc1, line1, _ = _bench("--dry-run")
    assert rc1 == 0 and line1["n_gpus"] == 1
    # asked for two, started inside a one-rank job: refused, not a silently smaller run
    rc2, line2, err2 = _bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc2 != 0 and line2 is None and "--gpus 2" in err2
    # without GPUs the real run fails in every rank -- and so does the launcher
    rc3, line3, _ = _bench("--gpus", "2", "--steps", "1")
    assert rc3 != 0 and line3 is None


def test_bench_under_torchrun_without_gpus_flag_adopts_the_world_size():
    """`torchrun --nproc-per-node N bench.py` without --gpus used to stop with an error since round 5; the job's size is what was asked for."""
    import json
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        str(ROOT / "bench.py"), "--dry-run", "--steps", "2"], capture_output=True, text=True, timeout=300, env=e)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr[-2000:]
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 2 and len(line["per_rank"]) == 2


def test_self_check_reads_an_earlier_checks_oracle_logs_only_for_the_same_samples():
    """bench.oracle_sample_check keeps the oracle's per-stream logs of a check (sentences, characters, symbols, the folded discriminator hash) for a later check
    over the SAME samples in the same order in this process (the fast leg behind the exact leg): reused when parameters, chunk sequence and the sums over
    the samples' bit patterns agree, decoded again when one sample differs.  The engine here is a stand-in that answers with the oracle's own first log."""
    import torch
    sys.path.insert(0, str(ROOT))
    import bench
    from habdec_amd import synth
    w = dict(bench.WORKLOADS["cfg4"], baud=300.0)
    C, S, n = 16384, 2, 8
    x = np.stack([synth.fsk_iq(synth.rtty_bits("$$CALL,1,2*ABCD\n", w["bits"], w["stops"], 2, 2), w["fs"], w["baud"], sigma=0.08, seed=s, n_samples=n * C) for s in range(S)])
    ring = torch.from_numpy(np.ascontiguousarray(x.reshape(S, n, C).transpose(1, 0, 2)).view(np.float32).reshape(n, S, C, 2).copy())

    class Eng:
        S = 2
        logs = None
        def take_sentences(self, s): return list(self.logs[s].sentences) if self.logs else []
        def take_chars(self, s): return self.logs[s].chars if self.logs else ""
        def bits_total(self, s): return self.logs[s].bits if self.logs else 0
        def demod_checksum_total(self, s): return (n, 0, self.logs[s].demod_hash if self.logs else 0)
    eng = Eng()
    bench.ORACLE_LOG_CACHE.clear()
    chunks = list(range(n))
    a = bench.oracle_sample_check(w, eng, ring, chunks, C, [0, 1], group=2)
    assert a["oracle_logs_reused"] == 0 and len(bench.ORACLE_LOG_CACHE) == 2
    eng.logs = {k[-1]: v for k, v in bench.ORACLE_LOG_CACHE.items()}
    b = bench.oracle_sample_check(w, eng, ring, chunks, C, [0, 1], group=2)
    assert b["oracle_logs_reused"] == 2 and b["gpu_matches_oracle_on_sample"] in (True, None) and b["bits_in_sample"] == a["bits_in_sample"] and not b["mismatches"]
    assert b["discriminator_checksums_compared"] == 2 * n
    ring2 = ring.clone(); ring2[1, 1, 77, 0] += 1e-3                      # one sample of one stream: nothing is reused
    c = bench.oracle_sample_check(w, eng, ring2, chunks, C, [0, 1], group=2)
    assert c["oracle_logs_reused"] == 0 and len(bench.ORACLE_LOG_CACHE) == 4
    d = bench.oracle_sample_check(w, eng, ring, chunks[:2], C, [0, 1], group=2)      # another chunk sequence: nothing is reused
    assert d["oracle_logs_reused"] == 0
    bench.ORACLE_LOG_CACHE.clear()
