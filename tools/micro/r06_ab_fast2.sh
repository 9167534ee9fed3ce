#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in cfg3 cfg2 cfg4 cfg5; do
  st=120; [ $w = cfg5 ] && st=24; [ $w = cfg3 ] && st=48; [ $w = cfg2 ] && st=80
  timeout 600 python3 tools/micro/ab_step.py --workload $w --steps $st --rounds 3 default default+f:ARITH=1 2>&1 | grep -v amdgpu.ids
done
