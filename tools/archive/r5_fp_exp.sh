#!/bin/bash
# compile-time experiment legs of the footprint backward, same box: default build, then one build per argument, then default
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
echo "=== default build"; CFGS="${CFGS:-config2 config3}" bash tools/r5_fp.sh | grep footprint
for f in "$@"; do
  EG_EXTRA_HIPCC_FLAGS="$f" python -m edgegaussians_amd.build --force 2>&1 | tail -1
  echo "=== $f"; CFGS="${CFGS:-config2 config3}" bash tools/r5_fp.sh | grep footprint
done
python -m edgegaussians_amd.build --force 2>&1 | tail -1
