# stage times inside the gather and merge_early (library built with -DK_PROFILE, product library rebuilt afterwards)
# usage (GPU box): bash tools/prof_loopkern.sh [workload]
set -e
cd "$(dirname "$0")/../dot_amd/csrc"
touch k_loopvec.hip && make -s EXTRA=-DK_PROFILE k_loopvec.o ../libdotmi.so
python ../../tools/prof_loopkern.py "${1:-bar17K_twist}" || true
touch k_loopvec.hip && make -s
