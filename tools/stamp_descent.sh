#!/bin/bash
# Diagnostic build of the matrix-core descent kernel with in-kernel cycle stamps (s_memtime totals per phase
# of the solver wave), run on the GPU box:   gpurun -- bash tools/stamp_descent.sh [planar_hand|box_pivoting] [T] [N]
# Builds a SEPARATE library (libirs_hip_stamps.so) and points the timing tool at it; the product library is
# untouched and contains no stamp.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/irs_mpc_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -fno-signed-zeros \
    -DIRS_CBM_STAMPS -c ctrlbox_mfma.hip -o /tmp/ctrlbox_mfma_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_out/libirs_hip_stamps.so plugin.o smooth.o smooth_ug.o tvlqr.o cem.o boxqp.o ctrlbox.o collective.o iterate.o /tmp/ctrlbox_mfma_stamps.o -ldl
cd $R
IRS_HIP_LIB=$R/gpurun_out/libirs_hip_stamps.so python tools/time_quasistatic.py "$@" --stamps
# ... and the LAST descent of the benchmark's iLQR loop (a warm-started one, late in an episode)
IRS_PRINT_STAMPS=1 IRS_HIP_LIB=$R/gpurun_out/libirs_hip_stamps.so python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 20 > /dev/null
[ -n "$LOOP_PROFILE" ] && IRS_HIP_LIB=$R/gpurun_out/libirs_hip_stamps.so python tools/descent_loop_profile.py "$1" --stamps
