"""Kernel duration (the engine's HIP events, every 3rd call) against the number of steps since the engine's first call, in buckets -- a free-running loop, and the
same loop with a flush + device synchronize every 25 steps (what bench.py's regions do at --steps 20 --warmup 5).  `--engine-first`: allocate the engine before the
ring, as bench.py's Shard does.      python3 tools/micro/r06_ramp.py [--arith 0|1] [--engine-first]"""
import argparse, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np, torch, bench
from habdec_amd import engine

ap = argparse.ArgumentParser(); ap.add_argument("--arith", type=int, default=0); ap.add_argument("--engine-first", action="store_true"); ap.add_argument("--steps", type=int, default=3000)
a = ap.parse_args()
w = dict(bench.WORKLOADS["cfg4"]); S, C = w["S"], w["C"]
dev = torch.device("cuda", 0)
mk = lambda: engine.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                           lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"], pipeline=2, arith=a.arith)
e0 = mk() if a.engine_first else None
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
base = ring.data_ptr(); torch.cuda.synchronize()
for mode in ("free", "flush25", "free"):
    e = e0 if e0 is not None else mk(); e0 = None
    e.set_timing(3)
    seen = e.timing()["timed_calls"]; ks = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        e.process_device(base + (i % rc) * S * C * 8, C, C)
        t = e.timing()
        if t["timed_calls"] != seen: seen = t["timed_calls"]; ks.append((i, t["ms_front"]))
        if mode == "flush25" and i % 25 == 24: e.flush(); torch.cuda.synchronize()
    e.flush(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e.close()
    edges = [0, 30, 60, 100, 150, 250, 400, 700, 1000, 1500, 2000, 3000, 10 ** 9]
    row = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        v = [k for (i, k) in ks if lo <= i < hi]
        if v: row.append(f"[{lo},{min(hi, a.steps)}) {np.mean(v) * 1e3:.1f}")
    print(f"{mode:8s} engine_first={a.engine_first} arith={a.arith}: {dt / a.steps * 1e3:.4f} ms/step; kernel us by step index: " + "  ".join(row))
    time.sleep(0.5)
