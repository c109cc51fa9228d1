#!/bin/bash
# Builds variants of the uniform-geometry kernel for tuning runs: tools/ug_variants.sh TAG "-DIRS_UG_SWEEPS=4 ..." [TAG FLAGS]...
# -> irs_mpc_amd/csrc/variants/libirs_hip_TAG.so (git-ignored; travels to the GPU box).  On the box:
#    for v in irs_mpc_amd/csrc/variants/*.so; do cp $v irs_mpc_amd/csrc/libirs_hip.so; python tests/tools/ug_time.py ...; done
set -e
cd $(dirname $0)/../irs_mpc_amd/csrc
mkdir -p variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function -ffinite-math-only -fno-signed-zeros -fno-slp-vectorize"
while [ $# -ge 2 ]; do
  TAG=$1; EXTRA=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $EXTRA -c smooth_ug.hip -o variants/smooth_ug_$TAG.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libirs_hip_$TAG.so plugin.o smooth.o variants/smooth_ug_$TAG.o tvlqr.o cem.o boxqp.o ctrlbox.o ctrlbox_mfma.o collective.o iterate.o -ldl
  rm variants/smooth_ug_$TAG.o
  echo built $TAG
done
