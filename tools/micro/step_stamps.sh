#!/bin/bash
cd $GRAFT_REPO_ROOT
for flags in "-DHD_STAMP_DEC" "-DHD_STAMP_DEC -DHD_DEC_OLDLOOP"; do
echo "=== [$flags]"
HD_EXTRA_FLAGS="$flags" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
for w in ${WGS:-8192}; do echo "--- HD_STEP_WGS=$w"; HD_STEP_WGS=$w python3 tools/micro/step_stamps.py 2>&1 | tail -8; done
echo "--- stage 1 alone (synchronous)"; python3 tools/micro/dec_stamps.py 2>&1 | tail -8
done
