#!/bin/bash
# Run ON the GPU box (through gpurun): everything profiles/<round>_* is made from -- the GPU suite, the bench lines, the single-stream lines, the
# rocprofv3 kernel-stats + PMC passes of every shape (exact mode, and the step launch in fast mode), SQ counters, energy per step, the clock / power series.  Usage: tools/collect_round.sh <tag> [round, default r06]
cd $GRAFT_REPO_ROOT
o=gpurun_out/$1; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $o/pytest.txt
# the headline pair first, in this order: step profile -> its summary into profiles/ -> the driver's and the default command (their roofline.rocprof_avg_launch_ms
# then is THIS session's steady-state average); tools/final_lines.sh makes the other lines (its own driver / default lines land in lines/ as a second sample)
bash tools/profile_then_lines.sh ${2:-r06} > $o/headline.log 2>&1
cp gpurun_out/headline/* $o/ 2>/dev/null
bash tools/final_lines.sh > $o/final_lines.log 2>&1
mkdir -p $o/lines_second_sample; cp gpurun_out/lines/driver_bench_line.json gpurun_out/lines/default_bench_line.json $o/lines_second_sample/ 2>/dev/null
cp gpurun_out/lines/sync_bench_line.json gpurun_out/lines/other_workloads.jsonl $o/ 2>/dev/null
timeout 300 python3 tools/single_stream.py > $o/single_stream.jsonl 2>$o/single.err
bash tools/gpu_profile.sh sync --sync > $o/profile_sync.log 2>&1
for w in cfg2 cfg3 cfg5; do bash tools/gpu_kstats.sh ${w}_fast --workload $w --arith fast --steps 40 --warmup 5 --no-also > $o/kstats_${w}_fast.txt 2>&1; done
bash tools/gpu_profile.sh cfg2 --workload cfg2 > $o/profile_cfg2.log 2>&1
bash tools/gpu_profile.sh cfg3 --workload cfg3 > $o/profile_cfg3.log 2>&1
bash tools/gpu_profile.sh cfg5 --workload cfg5 > $o/profile_cfg5.log 2>&1
{
  echo "== SQ counters cfg4 step (bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also; per-kernel medians, device totals per launch)"
  bash tools/gpu_pmc.sh ${1}_sq4 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY" --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids
  for w in cfg2 cfg3 cfg5; do
    echo "== SQ counters $w (sync: every kernel has the GPU to itself)"
    bash tools/gpu_pmc.sh ${1}_sq_$w "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" --workload $w --steps 24 --warmup 3 --sync 2>&1 | grep -v amdgpu.ids
  done
} > $o/pmc_sq.txt 2>&1
{
  echo "== tools/micro/joules.py: board power (hwmon) x time per step over >= 3 s loops, one box"
  HD_BUILD_VARIANT=timingexp python3 -m habdec_amd.build > /dev/null 2>&1
  timeout 300 python3 tools/micro/joules.py idle default default+f:ARITH=1 timingexp+s1:HD_CU_EXP=1 timingexp+tails:HD_CU_EXP=2 2>&1 | grep -v amdgpu.ids
  timeout 100 python3 tools/micro/joules.py --sync default 2>&1 | tail -1
  for w in cfg1 cfg2 cfg3 cfg5; do timeout 100 python3 tools/micro/joules.py --workload $w default default+f:ARITH=1 2>&1 | tail -2 | sed "s/^default/$w/"; done
} > $o/joules.txt 2>&1
bash tools/micro/clock_power_series.sh > $o/clock_power_series.txt 2>&1
cat $o/pytest.txt; tail -12 $o/final_lines.log; cat $o/joules.txt
