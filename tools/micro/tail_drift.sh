#!/bin/bash
cd $GRAFT_REPO_ROOT
HD_EXTRA_FLAGS="-DHD_STAMP_TAIL" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
python3 tools/micro/tail_drift.py 2>&1 | tail -16
