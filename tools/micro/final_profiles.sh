# Run ON the GPU box: the round's committed evidence (kernel stats + PMC traffic for the headline, synchronous mode and the other configs; bench lines)
cd $GRAFT_REPO_ROOT
tools/gpu_profile.sh c_step > /dev/null 2>&1
tools/gpu_profile.sh c_sync --sync > /dev/null 2>&1
tools/gpu_profile.sh c_cfg3 --workload cfg3 > /dev/null 2>&1
tools/gpu_profile.sh c_cfg2 --workload cfg2 > /dev/null 2>&1
tools/gpu_profile.sh c_cfg5 --workload cfg5 --steps 16 --warmup 4 > /dev/null 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/c_driver_line.json 2> gpurun_out/c_driver_line.err
timeout 400 python bench.py > gpurun_out/c_default_line.json 2> gpurun_out/c_default_line.err
rm -f gpurun_out/c_other.jsonl
for w in cfg1 cfg2 cfg3 cfg5; do timeout 200 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-also >> gpurun_out/c_other.jsonl 2>/dev/null; done
ls gpurun_out
