#!/bin/bash
# Run ON the GPU box: the bench lines that go to profiles/ and the profile passes of the other shapes.
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5o; mkdir -p $o
bash tools/final_lines.sh > $o/final_lines.log 2>&1
cp gpurun_out/lines/* $o/ 2>/dev/null
timeout 300 python3 tools/single_stream.py > $o/single_stream.jsonl 2>$o/single.err
bash tools/gpu_profile.sh sync --sync > $o/profile_sync.log 2>&1
bash tools/gpu_profile.sh cfg2 --workload cfg2 > $o/profile_cfg2.log 2>&1
bash tools/gpu_profile.sh cfg3 --workload cfg3 > $o/profile_cfg3.log 2>&1
bash tools/gpu_profile.sh cfg5 --workload cfg5 > $o/profile_cfg5.log 2>&1
tail -12 $o/final_lines.log
