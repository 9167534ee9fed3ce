"""How long does the step loop take to reach the rate it sustains, and does it matter WHO warmed the GPU?  One box, one ring, fresh engines per measurement:
K = 20 timed steps (barriered by flush + device synchronize, as bench.py's regions) behind
  own:P      P untimed steps of the SAME engine (bench.py's pre-warm pass; P = 0 is `cold`), then 5 warm-up steps;
  other:P    P steps of ANOTHER engine on the same ring (flushed, still alive), then the measured engine's 5 warm-up steps;
  gap:P:ms   own:P, then the host sleeps `ms` milliseconds with the GPU idle before the 5 warm-up steps;
and `loop`: the average over a 3 s loop.      python3 tools/micro/r06_warm.py [--arith 0|1] [--rounds 3]"""
import argparse, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch, bench
from habdec_amd import engine


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--arith", type=int, default=0); ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--workload", default="cfg4")
    a = ap.parse_args()
    w = dict(bench.WORKLOADS[a.workload]); S, C = w["S"], w["C"]
    dev = torch.device("cuda", 0)
    ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
    base = ring.data_ptr(); torch.cuda.synchronize()

    def make():
        e = engine.Engine(n_streams=S, max_chunk=C, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                          lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"], pipeline=2, arith=a.arith)
        e.set_timing(0); e.i = 0
        return e

    def run(e, n):
        for _ in range(n):
            e.process_device(base + (e.i % rc) * S * C * 8, C, C); e.i += 1

    def timed(e, K=20):
        run(e, 5); e.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(e, K); e.flush(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3

    variants = ["own:0", "own:145", "own:500", "own:1500", "own:4000", "other:1500", "other:4000", "gap:1500:2", "gap:1500:20", "gap:1500:200", "loop"]
    res = {v: [] for v in variants}
    for r in range(a.rounds):
        for v in variants:
            kind, *p = v.split(":")
            e = make()
            if kind == "own":
                run(e, int(p[0])); res[v].append(timed(e))
            elif kind == "other":
                o = make(); run(o, int(p[0])); o.flush(); res[v].append(timed(e)); o.close()
            elif kind == "gap":
                run(e, int(p[0])); e.flush(); torch.cuda.synchronize(); time.sleep(int(p[1]) / 1e3); res[v].append(timed(e))
            else:
                run(e, 1000); e.flush(); torch.cuda.synchronize()
                t0 = time.perf_counter(); n = 0
                while time.perf_counter() - t0 < 3.0:
                    run(e, 50); n += 50
                e.flush(); torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / n * 1e3)
            e.close()
            time.sleep(0.5)          # every measurement starts from an idle GPU
    print(f"{a.workload} arith {a.arith}: ms per step over 20 timed steps (rounds: {a.rounds})")
    for v in variants:
        print(f"  {v:14s} " + "  ".join(f"{x:.4f}" for x in res[v]))


if __name__ == "__main__":
    main()
