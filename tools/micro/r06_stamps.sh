#!/bin/bash
# In-kernel phase clocks of the step launch's worker waves and stream tails, exact mode against fast mode (diagnostic builds; the product library on the box is rebuilt).
cd $GRAFT_REPO_ROOT
HD_EXTRA_FLAGS="-DHD_STAMP_RING" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
for A in 0 1; do echo "=== ring stamps, ARITH=$A"; ARITH=$A timeout 300 python3 tools/micro/ring_stamps.py 2>&1 | tail -14; done
HD_EXTRA_FLAGS="-DHD_STAMP_TAIL" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
for A in 0 1; do echo "=== tail stamps inside the step launch, ARITH=$A"; PIPE=1 ARITH=$A timeout 300 python3 tools/micro/tail_stamps.py 2>&1 | tail -24; done
python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
