#!/bin/bash
# on the GPU box: time every built variant of the uniform-geometry kernel (tools/ug_variants.sh)
cd $GRAFT_REPO_ROOT
cp irs_mpc_amd/csrc/libirs_hip.so /tmp/libirs_hip_keep.so
for v in irs_mpc_amd/csrc/variants/libirs_hip_*.so; do
  echo "== $v"
  cp $v irs_mpc_amd/csrc/libirs_hip.so
  timeout -k 10 120 python tests/tools/ug_time.py "$@" 2>&1 | grep "N="
done
cp /tmp/libirs_hip_keep.so irs_mpc_amd/csrc/libirs_hip.so
