#!/bin/bash
cd $GRAFT_REPO_ROOT
HD_EXTRA_FLAGS="-DHD_STAMP_RING" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
for v in "HD_X=0"; do echo "--- $v"; env $v timeout 120 python3 tools/micro/ring_stamps.py 2>&1 | tail -6; done
