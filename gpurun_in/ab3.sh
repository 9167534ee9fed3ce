#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { env $1 timeout 200 python3 bench.py --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ', d['ms_per_step'], d['roofline']['avg_launch_ms'])
"; }
build() { HD_EXTRA_FLAGS="$1" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head -3; }
for round in 1 2; do
  build "-DHD_STEP_PRIO=0"; echo "regs prio0:"; run HD_X=0; run HD_X=0
  build "-DHD_RING_DMA -DHD_STEP_PRIO=0"; echo "dma prio0:"; run HD_X=0; run HD_X=0
  build ""; echo "regs prio3:"; run HD_X=0; run HD_X=0; echo "classic:"; run HD_NO_CU_STEP=1; run HD_NO_CU_STEP=1
  build "-DHD_STEP_PRIO=0"; echo "classic prio0:"; run HD_NO_CU_STEP=1; run HD_NO_CU_STEP=1
done
