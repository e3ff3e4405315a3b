# per-workgroup phase times of the element pass (library built with -DEP_PROFILE, product library rebuilt afterwards)
# usage (GPU box): bash tools/prof_elem.sh [workload]
set -e
cd "$(dirname "$0")/../dot_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -Wno-unused-value"
/opt/rocm/bin/hipcc $FLAGS -DEP_PROFILE -c kernels.hip -o kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libdotmi.so kernels.o dotmi.o -L/opt/rocm/lib -lrocblas -lrccl -Wl,-rpath,/opt/rocm/lib
python ../../tools/prof_elem.py "${1:-synbar:140x35x35:256}" || true
touch kernels.hip && make
