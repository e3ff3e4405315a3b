# phase times inside the diagonal / row tile tasks: builds libdotmi.so with -DDIAG_PROFILE, runs tools/prof_diag.py, rebuilds
# usage (GPU box): bash tools/prof_diag.sh [workload]
set -e
cd "$(dirname "$0")/../dot_amd/csrc"
touch k_tilefactor.hip && make -s EXTRA=-DDIAG_PROFILE k_tilefactor.o ../libdotmi.so
DOTMI_TILE_FLOW=${PROF_FLOW:-0} python ../../tools/prof_diag.py "${1:-bunny5K_LTSS}" || true
touch k_tilefactor.hip && make -s
