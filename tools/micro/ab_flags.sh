#!/bin/bash
# A/B over BUILD flags on one box (run through gpurun): FLAGSETS="-DA=1|-DB=2 -DC=3" [VARIANTS="HD_X=0|HD_CU_EXP=1"] tools/micro/ab_flags.sh
# Every flag set is built (HD_EXTRA_FLAGS, habdec_amd/build.py) and measured with tools/micro/ab_env.sh; the default build comes last and stays.
cd $GRAFT_REPO_ROOT
IFS='|' read -ra FS <<< "$FLAGSETS"
for f in "${FS[@]}" ""; do
  echo "=== flags: [$f]"
  HD_EXTRA_FLAGS="$f" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head -3
  VARIANTS="${VARIANTS:-HD_X=0}" tools/micro/ab_env.sh
done
