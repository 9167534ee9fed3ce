#!/bin/bash
# the whole GPU suite + the driver's bench command + a long run (+ synchronous mode)
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   ', d['steps'], 'steps:', d['value'], 'MS/s', d['ms_per_step'], 'ms/step;', r['kernel'], r['avg_launch_ms'], 'isolated', r.get('isolated', {}).get('avg_launch_ms'), '| path:', d['pipeline'].get('launch_path'))
"; }
for i in 1 2; do timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | show; done
for i in 1 2; do timeout 200 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | show; done
for v in HD_X=1 HD_NO_CLAIM=1 HD_X=1 HD_NO_CLAIM=1; do echo "sync $v"; env $v timeout 200 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-also --sync 2>/dev/null | show; done
