# where a time step's wall time goes between its kernels: idle gaps of the stream around the end of the L-BFGS loop and the refresh
# (UNDER rocprofv3 every host call is slower: the gaps that are host round trips come out 3-4 x their size in a plain run --
# profiles/r05_same_box.txt, last section)
# usage (GPU box): bash tools/step_gaps.sh [--workload bar17K_twist]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --extra-workloads none "$@" > /tmp/b.log 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
K=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].split('::')[-1][:28]) for r in rows]
# a step = from one init_x_kernel (start of dotmi_step's loop) to the next
hs=[i for i,k in enumerate(K) if 'init_x' in k[2]]
for a,b in list(zip(hs[:-1],hs[1:]))[-3:]:
    seg=K[a:b]
    wall=(K[b][0]-seg[0][0])/1e3
    busy=sum(e-s for s,e,_ in seg)/1e3
    gaps=[((seg[i+1][0]-seg[i][1])/1e3, seg[i][2], seg[i+1][2]) for i in range(len(seg)-1)]+[((K[b][0]-seg[-1][1])/1e3, seg[-1][2], K[b][2])]
    small=sum(g[0] for g in gaps if g[0]<4)
    print(f"step: {len(seg)} kernels, wall {wall:.1f} us, busy {busy:.1f}, idle {wall-busy:.1f} (of it {small:.1f} in gaps below 4 us); gaps of 4 us and more:")
    for g in gaps:
        if g[0]>=4: print("    %.1f us between %s -> %s"%g)
PY
