#!/bin/bash
# Run ON the GPU box (through gpurun): the bench lines that go to profiles/ -- the driver's command, one pass over the ring, synchronous delivery, the other workloads.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lines
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/lines/driver.err | tail -1 > gpurun_out/lines/driver_bench_line.json
timeout 600 python3 bench.py 2>gpurun_out/lines/default.err | tail -1 > gpurun_out/lines/default_bench_line.json
timeout 300 python3 bench.py --sync --no-cpu-baseline --no-also 2>/dev/null | tail -1 > gpurun_out/lines/sync_bench_line.json
rm -f gpurun_out/lines/other_workloads.jsonl
for w in cfg1 cfg2 cfg3 cfg5; do timeout 400 python3 bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 >> gpurun_out/lines/other_workloads.jsonl; done     # (each line: exact `value` + a `fast` block)
python3 - <<P
import json
for f in ("driver", "default", "sync"):
    d = json.load(open(f"gpurun_out/lines/{f}_bench_line.json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], "fast", (d.get("fast") or {}).get("value"), (d.get("fast") or {}).get("ms_per_step"), r["kernel"], r["avg_launch_ms"], r.get("n_samples"), r.get("rocprof_avg_launch_ms"), "e2e", d["pipeline"]["hbm_frac_end_to_end"], "also", (d.get("also") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("gpu_matches_oracle_on_sample"))
for l in open("gpurun_out/lines/other_workloads.jsonl"):
    d = json.loads(l); print(d["config"]["workload"][:6], d["value"], d["ms_per_step"], d["pipeline"]["launch_path"], "fast", (d.get("fast") or {}).get("value"), (d.get("fast") or {}).get("ms_per_step"))
P
