#!/bin/bash
# A/B over build flags: VARIANTS="flagsA|flagsB|..." tools/micro/ab_flags.sh  (an empty entry = the default build)
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
  for i in 1 2; do timeout 200 $B 2>/dev/null | show; done
  [ -n "$SYNC" ] && { echo "  sync:"; $B --sync 2>/dev/null | show; }
  [ -n "$PARITY" ] && python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
done
