#!/bin/bash
cd $GRAFT_REPO_ROOT
HD_EXTRA_FLAGS="-DHD_STAMP_DEC" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
echo "--- step launch (drawn runs)"; HD_STEP_WGS=2048 timeout 120 python3 tools/micro/step_stamps.py 2>&1 | tail -9
