#!/bin/bash
cd $GRAFT_REPO_ROOT
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   ', d['warmup'], 'warm-up,', d['steps'], 'steps:', d['value'], 'MS/s', d['ms_per_step'], 'ms/step;', r['kernel'], r['avg_launch_ms'], 'drain', d['pipeline'].get('drain_ms'), d['pipeline']['host_us_per_step'])
"; }
for w in 5 50 120; do for i in 1 2; do python3 bench.py --steps 20 --warmup $w --no-cpu-baseline --no-also 2>/dev/null | show; done; done
python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | show
python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | show
