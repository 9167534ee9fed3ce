#!/bin/bash
# Run ON the GPU box: the full GPU suite, the driver's command, the profile passes of the headline shape.
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5n; mkdir -p $o
timeout 700 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $o/pytest.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>$o/driver.err | tail -1 > $o/driver_bench_line.json
bash tools/gpu_profile.sh step > $o/profile_step.log 2>&1
cat $o/pytest.txt
python3 - <<P
import json
d = json.load(open("$o/driver_bench_line.json")); r = d["roofline"]
print("driver", d["value"], d["ms_per_step"], "cold", d["cold"]["value"], d["cold"]["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["n_samples"], r.get("frac"), "e2e", d["pipeline"]["hbm_frac_end_to_end"], "also", (d.get("also") or {}).get("value"), "per_rank", d["per_rank"])
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "threads", "gpu_matches_oracle_on_sample", "streams_in_sample", "bits_in_sample", "check_seconds", "mismatches")})
P
tail -3 $o/driver.err; tail -5 $o/profile_step.log; cat gpurun_out/prof_step/stats/steady_kernel_stats.csv | head -8
