cd /tmp && export TMPDIR=/tmp
for lv in 1 2; do
DOTMI_ND_LEVELS=$lv rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$lv -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/b$lv.log 2>&1
f=$(find /tmp/prof$lv -name "*kernel_stats.csv" | head -1)
echo "== levels $lv"; head -30 $f | cut -c1-200
cp $f /root/repo/gpurun_out/nd_l${lv}_kernel_stats.csv
done
