# FETCH_SIZE / WRITE_SIZE calibration on known access patterns (tools/bench_gather.hip): bash tools/pmc_gather.sh <tag>
tag=${1:-r04}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/bench_gather tools/bench_gather.hip || exit 1
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$c -- /root/repo/tools/bench_gather > /tmp/cal_$c.log 2>&1
done
python - /root/repo/gpurun_out/${tag}_pmc_calibration.json <<'PY'
import csv, glob, json, sys, collections
useful = {"cal_stream16": 2 << 30, "cal_gather8_line": (1 << 24) * 8, "cal_gather24": (1 << 24) * 24, "cal_gather72": (1 << 24) * 72,
          "cal_wstream16": 2 << 30, "cal_scatter8": (1 << 24) * 8, "cal_scatter24": (1 << 24) * 24}
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/cal_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            for k in useful:
                if k in r["Kernel_Name"]:
                    acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[1:] if len(v) > 1 else v      # first launch: cold
        kib = sum(v) / len(v)
        out.setdefault(k, {"useful_bytes": useful[k]})[c + "_KiB"] = kib
        out[k][c + "_bytes_per_useful_byte"] = round(1024 * kib / useful[k], 4)
        out[k][c + "_bytes_per_access"] = round(1024 * kib / (useful[k] / {"cal_stream16": 16, "cal_gather8_line": 8, "cal_gather24": 24,
                                                 "cal_gather72": 72, "cal_wstream16": 16, "cal_scatter8": 8, "cal_scatter24": 24}[k]), 2)
json.dump({"_what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only) of tools/bench_gather.hip on one MI355X: "
           "counter value (KiB x 1024, uncorrected) against the useful bytes each launch touches in a 2 GB buffer", "kernels": out},
          open(sys.argv[1], "w"), indent=1)
for k, d in out.items():
    print(k, d)
PY
