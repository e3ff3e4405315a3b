# refresh the judged artifacts: bench JSON line + rocprofv3 kernel stats of the same command
# usage (GPU box): bash tools/make_profiles.sh r01e
tag=${1:-r01x}
cd /tmp && export TMPDIR=/tmp
out=/root/repo/gpurun_out
mkdir -p $out
python /root/repo/bench.py > $out/${tag}_bench_bar17K.log 2>&1
grep '^{' $out/${tag}_bench_bar17K.log | tail -1 > $out/${tag}_bench_bar17K.json
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python /root/repo/bench.py --no-cpu-baseline > /tmp/prof_$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f $out/${tag}_bench_bar17K_kernel_stats.csv
grep '^{' /tmp/prof_$tag.log | tail -1 > $out/${tag}_bench_bar17K_under_rocprof.json
head -12 $out/${tag}_bench_bar17K_kernel_stats.csv | cut -c1-160
cat $out/${tag}_bench_bar17K.json | cut -c1-1500
