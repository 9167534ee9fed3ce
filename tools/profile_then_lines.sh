#!/bin/bash
# Run ON the GPU box (through gpurun): the headline evidence of ONE session on ONE box -- the step profile (kernel stats + the two PMC passes), its summary
# written into profiles/ (so that the bench lines made next read THIS session's steady-state kernel average as roofline.rocprof_avg_launch_ms), then the
# driver's command and the default command.  Usage: tools/profile_then_lines.sh <round tag, e.g. r05>  -> gpurun_out/headline/
cd $GRAFT_REPO_ROOT
rnd=$1; o=gpurun_out/headline; mkdir -p $o
bash tools/gpu_profile.sh step > $o/profile_step.log 2>&1
python3 tools/collect_profiles.py step ${rnd}_step cfg4 > $o/collect.log 2>&1
bash tools/gpu_profile.sh stepfast --arith fast > $o/profile_stepfast.log 2>&1            # the same three passes on the fast mode's step launch
python3 tools/collect_profiles.py stepfast ${rnd}_step_fast cfg4_fast > $o/collect_fast.log 2>&1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>$o/driver.err | tail -1 > $o/driver_bench_line.json
timeout 600 python3 bench.py 2>$o/default.err | tail -1 > $o/default_bench_line.json
cp profiles/${rnd}_step_* profiles/traffic.json $o/
python3 - <<P
import json
for f in ("driver", "default"):
    d = json.load(open("$o/%s_bench_line.json" % f)); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], "cold", d["cold"]["value"], d["cold"]["ms_per_step"], r["kernel"], r["avg_launch_ms"], r.get("n_samples"), "rocprof", r.get("rocprof_avg_launch_ms"), r.get("frac"), r.get("frac_rocprof"), "traffic", r.get("traffic"))
P
