# per-workgroup phase times of the element pass (library built with -DEP_PROFILE, product library rebuilt afterwards)
# usage (GPU box): bash tools/prof_elem.sh [workload]
set -e
cd "$(dirname "$0")/../dot_amd/csrc"
touch k_element.hip && make -s EXTRA=-DEP_PROFILE k_element.o ../libdotmi.so
python ../../tools/prof_elem.py "${1:-synbar:140x35x35:256}" || true
touch k_element.hip && make -s
