# per-workgroup timeline of the speculative unit-step launch (library built with -DDS_PROFILE, product library rebuilt afterwards)
# usage (GPU box): bash tools/prof_dirstep.sh [workload]
set -e
cd "$(dirname "$0")/../dot_amd/csrc"
touch k_dirstep.hip && make -s EXTRA="-DDS_PROFILE -DEP_PROFILE" k_dirstep.o ../libdotmi.so
DOTMI_SPEC_STEP=1 python ../../tools/prof_dirstep.py "${1:-bar17K_twist}" || true
touch k_dirstep.hip && make -s
