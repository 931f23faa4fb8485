#!/bin/bash
# Builds libpiper_hip.so of another commit into piper_amd/libab_base.so (git-ignored, travels to the GPU box) for
# scripts/gpu_ab.sh. Usage: scripts/build_base.sh <commit>
set -e
C=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/pe_base_$$
rm -rf $W && mkdir -p $W
git -C "$ROOT" archive "$C" | tar -x -C $W
make -C $W -j8 all > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
cp $W/piper_amd/libpiper_hip.so "$ROOT/piper_amd/libab_base.so"
rm -rf $W
echo "piper_amd/libab_base.so = $(git -C "$ROOT" rev-parse --short "$C")"
