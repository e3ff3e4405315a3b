# stage times inside elem_vertex_kernel (library built with -DK_PROFILE, product library rebuilt afterwards)
# usage (GPU box): bash tools/prof_elemvert.sh [workload]
set -e
cd "$(dirname "$0")/../dot_amd/csrc"
touch k_elemvert.hip && make -s EXTRA=-DK_PROFILE k_elemvert.o ../libdotmi.so
python ../../tools/prof_elemvert.py "${1:-bar17K_twist}" || true
touch k_elemvert.hip && make -s
