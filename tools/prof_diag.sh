# phase times inside the diagonal / row tile tasks: builds libdotmi.so with -DDIAG_PROFILE, runs tools/prof_diag.py, rebuilds
# usage (GPU box): bash tools/prof_diag.sh [workload]
set -e
cd "$(dirname "$0")/../dot_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -Wno-unused-value"
/opt/rocm/bin/hipcc $FLAGS -DDIAG_PROFILE -c kernels.hip -o kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libdotmi.so kernels.o dotmi.o -L/opt/rocm/lib -lrocblas -lrccl -Wl,-rpath,/opt/rocm/lib
DOTMI_TILE_FLOW=${PROF_FLOW:-0} python ../../tools/prof_diag.py "${1:-bunny5K_LTSS}" || true
touch kernels.hip && make
