# held back-solves (DOTMI_EARLY_HOLD) on / off: ms per step mean / p50 / p95, iterations, halvings -- same box
for wl in ${HOLD_WORKLOADS:-monkey18K_stiff horse7K_stretch bar17K_twist}; do
  for hld in 0 1; do
    DOTMI_EARLY_HOLD=$hld python bench.py --workload $wl --steps ${HOLD_STEPS:-20} --no-cpu-baseline --extra-workloads none 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl hold=$hld ms %.3f p50 %.3f p95 %.3f iters %.2f bs us %.1f stopped %d of %d held %d (rejected %d)' % (d['value'], d['ms_per_step_p50'], d['ms_per_step_p95'], d['iters_per_frame'], 1e3*r['avg_launch_ms'], r['launches_stopped'], r['launches_total'], r.get('launches_held',0), r.get('launches_held_rejected',0)))"
  done
done
