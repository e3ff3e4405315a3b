# per-workgroup timeline of the back-solve launch: builds libdotmi.so with -DBS_PROFILE (start / end wall-clock stamps per
# workgroup in a device array, read back through dotmi_debug_bs_prof), prints the timeline, rebuilds the product library.
# usage (GPU box): bash tools/prof_backsolve.sh [workload]      (environment switches apply, e.g. DOTMI_ND_LEVELS=2)
set -e
cd "$(dirname "$0")/../dot_amd/csrc"
touch k_backsolve.hip && make -s EXTRA=-DBS_PROFILE k_backsolve.o ../libdotmi.so
python ../../tools/prof_backsolve.py "${1:-bar17K_twist}" || true
touch k_backsolve.hip && make -s
