# per-workgroup timeline of the back-solve launch: builds libdotmi.so with -DBS_PROFILE (start / end wall-clock stamps per
# workgroup in a device array, read back through dotmi_debug_bs_prof), prints the timeline, rebuilds the product library.
# usage (GPU box): bash tools/prof_backsolve.sh [workload]
set -e
cd "$(dirname "$0")/../dot_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -Wno-unused-value"
/opt/rocm/bin/hipcc $FLAGS -DBS_PROFILE -c kernels.hip -o kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libdotmi.so kernels.o dotmi.o -L/opt/rocm/lib -lrocblas -lrccl -Wl,-rpath,/opt/rocm/lib
python ../../tools/prof_backsolve.py "${1:-bar17K_twist}" || true
touch kernels.hip && make
