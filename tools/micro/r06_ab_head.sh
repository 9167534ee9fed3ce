#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in cfg4 cfg2 cfg3; do
  st=200; [ $w = cfg3 ] && st=48; [ $w = cfg2 ] && st=80
  timeout 600 python3 tools/micro/ab_step.py --workload $w --steps $st --rounds 3 prev default prev+f:ARITH=1 default+f:ARITH=1 2>&1 | grep -v amdgpu.ids
done
