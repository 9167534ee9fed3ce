"""What do the HIP-event brackets of hd_engine_set_timing(n) (every n-th call) and the per-step hd_engine_timing() read cost a free-running step loop?
One box, one ring, one engine per variant, 2 s loops taking turns.      python3 tools/micro/r06_timing_cost.py [--arith 0|1]"""
import argparse, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch, bench
from habdec_amd import engine

ap = argparse.ArgumentParser(); ap.add_argument("--arith", type=int, default=0); ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
w = dict(bench.WORKLOADS["cfg4"]); S, C = w["S"], w["C"]
dev = torch.device("cuda", 0)
ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
base = ring.data_ptr(); torch.cuda.synchronize()
variants = [("timing 0", 0, False), ("timing 3", 3, False), ("timing 3 + read", 3, True), ("timing 1 + read", 1, True)]
res = {v[0]: [] for v in variants}; kern = {v[0]: [] for v in variants}
for r in range(a.rounds):
    for name, every, read in variants:
        e = engine.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                          lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"], pipeline=2, arith=a.arith)
        e.set_timing(every)
        i = 0
        for _ in range(1000): e.process_device(base + (i % rc) * S * C * 8, C, C); i += 1
        e.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0; ks = []; seen = e.timing()["timed_calls"]
        while time.perf_counter() - t0 < 2.0:
            for _ in range(50):
                e.process_device(base + (i % rc) * S * C * 8, C, C); i += 1; n += 1
                if read:
                    t = e.timing()
                    if t["timed_calls"] != seen: seen = t["timed_calls"]; ks.append(t["ms_front"])
        e.flush(); torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / n * 1e3); kern[name].append(sum(ks) / len(ks) if ks else float("nan"))
        e.close(); time.sleep(0.3)
print(f"cfg4 arith {a.arith}: ms per step over 2 s loops | kernel average from the engine's HIP events")
for name, _, _ in variants:
    print(f"  {name:18s} " + "  ".join(f"{x:.4f}" for x in res[name]) + "   | " + "  ".join(f"{x:.4f}" for x in kern[name]))
