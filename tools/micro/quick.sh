#!/bin/bash
# parity subset + bench lines (20 and 100 steps) + optional tail stamps
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -3
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   ', d['steps'], 'steps:', d['value'], 'MS/s', d['ms_per_step'], 'ms/step;', r['kernel'], r['avg_launch_ms'], 'isolated', r.get('isolated', {}).get('avg_launch_ms'), '| path:', d['pipeline'].get('launch_path'))
"; }
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | show; done
for i in 1 2; do python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | show; done
if [ -n "$STAMPS" ]; then
HD_EXTRA_FLAGS="-DHD_STAMP_TAIL" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
echo "--- tail stand-alone"; python3 tools/micro/tail_stamps.py 2>&1 | tail -21
echo "--- tail inside the step kernel"; PIPE=1 python3 tools/micro/tail_stamps.py 2>&1 | tail -21
fi
