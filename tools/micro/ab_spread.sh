#!/bin/bash
# A/B: stage-1 prefetch issued in one burst (default) vs in parts between the tap blocks (-DHD_DEC_SPREAD).
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-also"
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   ', d['value'], 'MS/s', d['ms_per_step'], 'ms/step;', r['kernel'], r['avg_launch_ms'], 'isolated', r.get('isolated', {}).get('avg_launch_ms'), '| path:', d['pipeline'].get('launch_path'))
"; }
for flags in "" "-DHD_DEC_SPREAD" $EXTRA_VARIANTS; do
  echo "=== build [$flags]"
  HD_EXTRA_FLAGS="$flags" python3 -m habdec_amd.build --force 2>&1 | grep -E "error|spill" | head
  for i in 1 2; do $B 2>/dev/null | show; done
  echo "  sync:"; $B --sync 2>/dev/null | show
  echo "  stage 1 on half of every XCD (HD_CU_SPLIT=16, no step):"; HD_NO_STEP=1 HD_CU_SPLIT=16 HD_DEC_WGS_PER_CU=8 $B 2>/dev/null | show
  python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
done
