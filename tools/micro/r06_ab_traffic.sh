#!/bin/bash
# the library as it is against the previous commit's (gpurun_in/variants/libhd_prev.so), every separate-kernels workload, exact and fast
cd $GRAFT_REPO_ROOT
for w in cfg2 cfg3 cfg5; do
  st=80; [ $w = cfg5 ] && st=24; [ $w = cfg3 ] && st=48
  timeout 600 python3 tools/micro/ab_step.py --workload $w --steps $st --rounds 3 prev default prev+f:ARITH=1 default+f:ARITH=1 2>&1 | grep -v amdgpu.ids
done
