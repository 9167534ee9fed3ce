#!/bin/bash
cd $GRAFT_REPO_ROOT
IFS='|' read -ra VS <<< "${VARIANTS:-|}"
for flags in "${VS[@]}"; do
  echo "=== build [$flags]"
  HD_EXTRA_FLAGS="$flags" python3 -m habdec_amd.build --force 2>&1 | grep -E "error" | head
  python3 -m pytest $TESTS -m gpu -x -q 2>&1 | tail -4
done
