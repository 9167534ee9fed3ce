#!/bin/bash
# FLAGSETS="-DA=1|-DB=2 -DC=3|" gpurun_in/ab_flags.sh   (empty = default build)
cd $GRAFT_REPO_ROOT
IFS='|' read -ra FS <<< "$FLAGSETS"
for f in "${FS[@]}" ""; do
  echo "=== flags: [$f]"
  HD_EXTRA_FLAGS="$f" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head -3
  VARIANTS="${VARIANTS:-HD_X=0}" REP=${REP:-2} gpurun_in/ab.sh
done
