#!/bin/bash
cd $GRAFT_REPO_ROOT
HD_EXTRA_FLAGS="-DHD_STAMP_RING -DHD_RING_DMA" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
for v in "HD_X=0" "HD_CU_EXP=1"; do echo "--- $v"; env $v timeout 120 python3 tools/micro/ring_stamps.py 2>&1 | grep -v amdgpu.ids | head -8; done
