#!/bin/bash
# builds scfgp_amd/lib/libscfgp_hip_head.so from the sources of a commit (default HEAD), for same-box A/B runs against the working tree
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/scfgp_head
rm -rf $T && mkdir -p $T
git -C $ROOT archive $REV scfgp_amd/csrc include | tar -x -C $T
make -C $T/scfgp_amd/csrc -j7 2>&1 | grep -i "error" || true
cp $T/scfgp_amd/lib/libscfgp_hip.so $ROOT/scfgp_amd/lib/libscfgp_hip_head.so
ls -la $ROOT/scfgp_amd/lib/libscfgp_hip_head.so
