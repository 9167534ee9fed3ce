#!/bin/bash
cd $GRAFT_REPO_ROOT
HD_EXTRA_FLAGS="-DHD_STAMP_TAIL" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
true
echo "--- inside the step kernel (batch mode)"; PIPE=1 python3 tools/micro/tail_stamps.py 2>&1 | tail -22
