cd $GRAFT_REPO_ROOT
tools/gpu_profile.sh c_cfg3 --workload cfg3 > /dev/null 2>&1
tools/gpu_profile.sh c_cfg2 --workload cfg2 > /dev/null 2>&1
tools/gpu_profile.sh c_cfg5 --workload cfg5 --steps 16 --warmup 4 > /dev/null 2>&1
rm -f gpurun_out/c_other.jsonl
for w in cfg1 cfg2 cfg3 cfg5; do timeout 200 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-also >> gpurun_out/c_other.jsonl 2>/dev/null; done
ls gpurun_out | tail -3
