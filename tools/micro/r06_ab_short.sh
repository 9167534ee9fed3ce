#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 tools/micro/ab_step.py --workload cfg4 --steps 300 --rounds 3 default+a:HD_RING_SHORT_PCT=0 default+b:HD_RING_SHORT_PCT=12 default+c:HD_RING_SHORT_PCT=25 default+d:HD_RING_SHORT_PCT=50 default+e:HD_RING_SHORT_PCT=100 2>&1 | grep -v amdgpu.ids
timeout 900 python3 tools/micro/ab_step.py --workload cfg4 --steps 300 --rounds 3 default+fa:ARITH=1,HD_RING_SHORT_PCT=0 default+fc:ARITH=1,HD_RING_SHORT_PCT=25 default+fd:ARITH=1,HD_RING_SHORT_PCT=50 2>&1 | grep -v amdgpu.ids
timeout 600 python3 tools/micro/ab_step.py --workload cfg5 --steps 24 --rounds 2 default+a:HD_RING_SHORT_PCT=0 default+c:HD_RING_SHORT_PCT=25 2>&1 | grep -v amdgpu.ids
