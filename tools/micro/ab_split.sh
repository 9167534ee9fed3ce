#!/bin/bash
# A/B: the step kernel against stage 1 and the stream tails on disjoint CU sets (two masked queues).
mkdir -p gpurun_out
B="python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-also"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   ', d['value'], 'MS/s', d['ms_per_step'], 'ms/step;', r['kernel'], r['avg_launch_ms'], '| path:', d['pipeline'].get('launch_path'))
"; }
run HD_X=0
run HD_NO_STEP=1
for a in 12 14 16 18 20; do run HD_NO_STEP=1 HD_CU_SPLIT=$a HD_DEC_WGS_PER_CU=8; done
run HD_NO_STEP=1 HD_CU_SPLIT=16 HD_DEC_WGS_PER_CU=6
