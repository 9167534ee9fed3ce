#!/bin/bash
# VARIANTS="A=1|B=2 C=3|HD_X=0" gpurun_in/ab.sh
cd $GRAFT_REPO_ROOT
IFS='|' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  echo "== $v"
  for i in $(seq 1 ${REP:-1}); do env $v timeout 200 python3 bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   ', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], d['pipeline'].get('launch_path'), d.get('cpu_baseline', {}).get('gpu_matches_oracle_on_sample'))
"; done
done
