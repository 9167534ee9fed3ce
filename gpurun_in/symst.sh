#!/bin/bash
cd $GRAFT_REPO_ROOT
HD_EXTRA_FLAGS="-DHD_STAMP" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
for w in cfg3 cfg5 cfg2; do echo "--- $w"; WL=$w NCALLS=24 HD_NO_TAIL=1 timeout 200 python3 tools/micro/sym_stamps.py 2>&1 | tail -5; done
