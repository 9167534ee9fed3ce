#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/micro/ab_step.py --workload cfg5 --steps 24 --rounds 3 default default+f:ARITH=1 2>&1 | grep -v amdgpu.ids
bash tools/gpu_kstats.sh cfg5_fastfft --workload cfg5 --arith fast --steps 24 --warmup 3 --no-also 2>&1 | tail -14
