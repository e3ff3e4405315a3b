# HBM traffic of the back-solve kernels from the PMC counters: one rocprofv3 pass per counter, kernel trace only
# (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE do not fit one pass; FETCH_SIZE reports half of a
# wide coalesced streaming read on gfx950 -> doubled; counter unit KiB).
# usage (GPU box): bash tools/pmc_backsolve.sh <workload> <tag>   -> gpurun_out/<tag>_backsolve_pmc_<slug>.json
wl=${1:-bar17K_twist}; tag=${2:-r02}
slug=$(echo $wl | tr ':x' '__')
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out
raw=/root/repo/gpurun_out/${tag}_backsolve_pmc_${slug}_raw.txt
: > $raw
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python /root/repo/tools/bench_backsolve.py $wl > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c >> $raw <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name']==sys.argv[2]:
        n=r['Kernel_Name']
        k='backsolve' if 'backsolve' in n else ('reduce_partial' if 'reduce_partial' in n else None)
        if k: acc[(k, n)].append(float(r['Counter_Value']))
# a back-solve may be two launches (wide and narrow tiles: two template instances, each once per back-solve):
# per back-solve = sum over the distinct kernels of their average
tot=collections.defaultdict(float); cnt=collections.defaultdict(int)
for (k,n),v in acc.items():
    tot[k]+=sum(v)/len(v); cnt[k]=max(cnt[k],len(v))
for k in tot: print(sys.argv[2], k, 'dispatches', cnt[k], 'avg', tot[k])
PY
done
grep "back-solve" /tmp/pmc_FETCH_SIZE.log >> $raw
cat $raw
python - $raw $wl /root/repo/gpurun_out/${tag}_backsolve_pmc_${slug}.json <<'PY'
import json, sys
v = {}
alg = None
for l in open(sys.argv[1]):
    t = l.split()
    if len(t) >= 6 and t[0] in ("FETCH_SIZE", "WRITE_SIZE"):
        v[(t[0], t[1])] = float(t[5])
    if "algorithmic" in l and "bytes" in l:
        alg = int(l.split("bytes")[1].split()[0])
hbm = 2 * 1024 * v[("FETCH_SIZE", "backsolve")] + 1024 * v[("WRITE_SIZE", "backsolve")]
rec = {"_what": "rocprofv3 PMC passes (separate runs: --pmc FETCH_SIZE, then --pmc WRITE_SIZE, each with --kernel-trace only; "
       "tools/pmc_backsolve.sh) of tools/bench_backsolve.py, one MI355X, averaged over the dispatches (a back-solve split into a wide-tile and a narrow-tile launch counts both). Counter unit KiB; gfx950 "
       "correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE doubled, WRITE_SIZE as is. hbm_bytes_per_backsolve = "
       "backsolve_kernel alone (the kernel bench.py's roofline entry is about).",
       "workload": sys.argv[2],
       "backsolve_kernel": {"FETCH_SIZE_KiB": v[("FETCH_SIZE", "backsolve")], "WRITE_SIZE_KiB": v[("WRITE_SIZE", "backsolve")]},
       "reduce_partial_p_kernel": {"FETCH_SIZE_KiB": v.get(("FETCH_SIZE", "reduce_partial")), "WRITE_SIZE_KiB": v.get(("WRITE_SIZE", "reduce_partial"))},
       "hbm_bytes_per_backsolve": int(hbm), "algorithmic_bytes_per_backsolve": alg,
       "ratio": round(hbm / alg, 3) if alg else None}
json.dump(rec, open(sys.argv[3], "w"), indent=1)
print(json.dumps(rec)[:400])
PY
