#!/bin/bash
# A/B over environment settings: VARIANTS="A=1|B=2 C=3|HD_X=0" tools/micro/ab_env.sh  (use a dummy like HD_X=0 for the default)
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline --no-also ${WL:+--workload $WL}"
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   ', d['value'], 'MS/s', d['ms_per_step'], 'ms/step;', r['kernel'][:40], r.get('avg_launch_ms'), 'isolated', r.get('isolated', {}).get('avg_launch_ms'), '| path:', d['pipeline'].get('launch_path'), '| match', d.get('cpu_baseline', {}).get('gpu_matches_oracle_on_sample'))
"; }
[ -n "$PARITY" ] && python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -2
IFS='|' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  echo "== $v"
  for i in 1 2; do env $v $B 2>/dev/null | show; done
done
