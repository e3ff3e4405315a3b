# refresh the judged artifacts: bench JSON line + rocprofv3 kernel stats of the same command (+ per-workload stats)
# usage (GPU box): bash tools/make_profiles.sh r02a
tag=${1:-r02x}
cd /tmp && export TMPDIR=/tmp
out=/root/repo/gpurun_out
mkdir -p $out
timeout 900 python /root/repo/bench.py > $out/${tag}_bench.log 2>&1
grep '^{' $out/${tag}_bench.log | tail -1 > $out/${tag}_bench.json
cp bench_detail.json $out/${tag}_bench_detail.json 2>/dev/null || cp /root/repo/bench_detail.json $out/${tag}_bench_detail.json
for wl in bar17K_twist bunny5K_LTSS horse7K_stretch@r1:64 monkey18K_stiff synbar:140x35x35:256; do
  slug=$(echo $wl | tr ':x@' '___')
  steps=20; [ "$slug" != "${slug#synbar}" ] && steps=6; [ "$slug" != "${slug#horse}" ] && steps=8; [ "$slug" != "${slug#monkey}" ] && steps=5
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python /root/repo/bench.py --workload $wl --steps $steps --no-cpu-baseline --extra-workloads none > /tmp/prof_$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  cp $f $out/${tag}_bench_${slug}_kernel_stats.csv
  grep '^{' /tmp/prof_$tag.log | tail -1 > $out/${tag}_bench_${slug}_under_rocprof.json
  echo "== $wl"; head -6 $out/${tag}_bench_${slug}_kernel_stats.csv | cut -c1-100
done
cut -c1-1800 $out/${tag}_bench.json
