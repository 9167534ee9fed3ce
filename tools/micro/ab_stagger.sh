#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-also"
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   ', d['value'], 'MS/s', d['ms_per_step'], 'ms/step;', r['kernel'], r['avg_launch_ms'], 'isolated', r.get('isolated', {}).get('avg_launch_ms'), '| path:', d['pipeline'].get('launch_path'))
"; }
IFS='|' read -ra VS <<< "${VARIANTS:-|}"
for flags in "${VS[@]}"; do
  echo "=== build [$flags]"
  HD_EXTRA_FLAGS="$flags" python3 -m habdec_amd.build --force 2>&1 | grep -E "error|spill" | head
  echo "  step default:"; $B 2>/dev/null | show
  echo "  step, 2048 WGs x 16 tiles:"; HD_STEP_WGS=2048 $B 2>/dev/null | show
  echo "  stage 1 on half of every XCD:"; HD_NO_STEP=1 HD_CU_SPLIT=16 HD_DEC_WGS_PER_CU=8 $B 2>/dev/null | show
done
