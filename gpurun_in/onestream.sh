#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head -3
for w in cfg5 cfg2 cfg3; do
  for v in HD_X=0 HD_ONE_STREAM=1; do
    env $v timeout 300 python3 bench.py --workload $w --steps 30 --warmup 4 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w $v', d['value'], 'MS/s', d['ms_per_step'], 'ms/step')
"
  done
done
