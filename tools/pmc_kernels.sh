# HBM traffic per kernel class from the PMC counters: one rocprofv3 pass per counter, kernel trace only
# (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE do not fit one pass; counter unit KiB; on gfx950
# FETCH_SIZE reports half of a wide coalesced streaming read -> doubled, WRITE_SIZE as is).
# usage (GPU box): bash tools/pmc_kernels.sh <workload> <tag>   -> gpurun_out/<tag>_pmc_<slug>.json
wl=${1:-bar17K_twist}; tag=${2:-r03}
slug=$(echo $wl | tr ':x@' '___')
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python /root/repo/tools/kernel_bench.py $wl 1 > /tmp/pmc_$c.log 2>&1
done
# (round 6) FP64 vector instructions per launch, a third pass: the ALU side of the roofline for the kernels that are bound by
# arithmetic rather than by bytes (the fixed-corotational element pass with its SVD)
rm -rf /tmp/pmc_F64
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d /tmp/pmc_F64 -- python /root/repo/tools/kernel_bench.py $wl 1 > /tmp/pmc_F64.log 2>&1
python - "$wl" /root/repo/gpurun_out/${tag}_pmc_${slug}.json <<'PY'
import csv, glob, json, sys, collections
sys.path.insert(0, "/root/repo")
from bench import source_id   # sha256 over dot_amd/csrc/*.hip, *.hpp: bench.py reports a traffic figure only for the build it runs
CLASSES = [("dirstep_kernel", "dirstep"), ("elem_vertex_kernel", "elem_vertex"), ("spmv_zp_wide_kernel", "spmv_zp"), ("elem_patch_kernel", "elem_pass"), ("vertex_gather_kernel", "vertex_gather"), ("spmv_dots_kernel", "spmv_dots"),
           ("backsolve_kernel", "backsolve"), ("merge_tiles_kernel", "merge"), ("merge_tiles_early_kernel", "merge_early"), ("merge_kernel", "merge_split"), ("reduce_partial_p_kernel", "reduce_partial"),
           ("spmv_zp_kernel", "spmv_zp"), ("build_qpad_kernel", "build_qpad"),
           ("build_p_kernel", "build_p"), ("step_forward_kernel", "step_forward"), ("elem_hessian_kernel", "elem_hessian"),
           ("assemble_kernel", "assemble"), ("tile_task_kernel", "tile_task"), ("tile_flow_kernel", "tile_flow"), ("tile_gemm_kernel", "tile_gemm"),
           ("dense_fill_kernel", "dense_fill"), ("clear_tiles_kernel", "clear_tiles"),
           ("twolevel_forward_kernel", "twolevel_forward"), ("twolevel_backward_kernel", "twolevel_backward"),
           ("twolevel_rhs_kernel", "twolevel_rhs"), ("twolevel_pack_kernel", "twolevel_pack")]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        for pat, name in CLASSES:
            if pat in r["Kernel_Name"]:
                # template instances (energy-only / energy+gradient, wide / narrow tiles) are kept apart
                acc[(name, r["Kernel_Name"][:160])].append(float(r["Counter_Value"]))
    for (name, kn), v in acc.items():
        out.setdefault(name, {}).setdefault(kn, {})[c + "_KiB"] = sum(v) / len(v)
        out[name][kn]["dispatches"] = len(v)
f = glob.glob("/tmp/pmc_F64/**/*counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if not r["Counter_Name"].startswith("SQ_INSTS_VALU_"):
            continue
        for pat, name in CLASSES:
            if pat in r["Kernel_Name"]:
                acc[(name, r["Kernel_Name"][:160])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
    for (name, kn), cs in acc.items():
        g = {c: sum(v) / len(v) for c, v in cs.items()}
        d = out.setdefault(name, {}).setdefault(kn, {})
        d["fp64_wave_instructions"] = {c.replace("SQ_INSTS_VALU_", ""): g[c] for c in g}
        # a wave instruction = 64 lanes; an FMA two operations (the profiler's own TotalFlops expression)
        d["fp64_flop_per_launch"] = int(64 * (g.get("SQ_INSTS_VALU_ADD_F64", 0) + g.get("SQ_INSTS_VALU_MUL_F64", 0) +
                                             2 * g.get("SQ_INSTS_VALU_FMA_F64", 0) + g.get("SQ_INSTS_VALU_TRANS_F64", 0)))
for name, ks in out.items():
    for kn, d in ks.items():
        if "FETCH_SIZE_KiB" in d and "WRITE_SIZE_KiB" in d:
            d["hbm_bytes_per_launch"] = int(2 * 1024 * d["FETCH_SIZE_KiB"] + 1024 * d["WRITE_SIZE_KiB"])
rec = {"_what": "rocprofv3 PMC passes (separate runs: --pmc FETCH_SIZE, then --pmc WRITE_SIZE, each with --kernel-trace only; "
       "tools/pmc_kernels.sh) of tools/kernel_bench.py (every kernel class launched 31 times back to back on the workload's resident "
       "state), one MI355X, averaged over the dispatches of a kernel instance.  Counter unit KiB; gfx950 correction per "
       "MI355X_MICROARCH.md section HBM: FETCH_SIZE doubled, WRITE_SIZE as is; Infinity-Cache hits are counted, not excluded.  "
       "fp64_flop_per_launch (a third pass): 64 x (ADD_F64 + MUL_F64 + 2 FMA_F64 + TRANS_F64) wave instructions.",
       "workload": sys.argv[1], "source_id": source_id(), "kernels": out}
json.dump(rec, open(sys.argv[2], "w"), indent=1)
for name, ks in out.items():
    for kn, d in ks.items():
        print(name, kn[:60], d.get("hbm_bytes_per_launch"), d.get("fp64_flop_per_launch"))
PY
