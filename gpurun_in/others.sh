#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in cfg1 cfg2 cfg3 cfg5; do
  for m in "" "--sync"; do
    timeout 300 python3 bench.py --workload $w --steps ${STEPS:-30} --warmup 4 --no-cpu-baseline --no-also $m 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; r = r.get('stage1_hbm', r)
        print('$w $m', d['value'], 'MS/s', d['ms_per_step'], 'ms/step; front', r.get('avg_launch_ms'), 'isolated', r.get('isolated', {}).get('avg_launch_ms'), '| path:', d['pipeline'].get('launch_path'), '| call latency', d['pipeline'].get('call_latency_ms'))
"
  done
done
