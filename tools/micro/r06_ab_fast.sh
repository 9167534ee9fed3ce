#!/bin/bash
# exact against fast mode, taking turns in one process on one box, every workload of bench.py (tools/micro/ab_step.py)
cd $GRAFT_REPO_ROOT
for w in cfg4 cfg1 cfg2 cfg3 cfg5; do
  st=120; [ $w = cfg5 ] && st=24; [ $w = cfg3 ] && st=48; [ $w = cfg2 ] && st=80
  timeout 600 python3 tools/micro/ab_step.py --workload $w --steps $st --rounds 3 default default+f:ARITH=1 2>&1 | grep -v amdgpu.ids
done
