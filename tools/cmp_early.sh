for wl in "horse7K_stretch - 12" "kingkong18K_SS_1K - 8" "monkey18K_stiff - 6" "synbar:140x35x35:256 - 4" "bar17K_twist - 12" "bunny5K_LTSS - 12"; do
  set -- $wl
  for e in 0 1 2; do
    echo "== $1 EARLY=$e"
    DOTMI_EARLY_BACKSOLVE=$e timeout 300 python tools/run_case.py $1 $2 $3 2>&1 | grep iters | awk '{it+=$3; hv+=$5; ms+=$7; lp+=$9; n++} END {printf "steps %d iters %d halv %d ms/step %.3f loop %.3f\n", n, it, hv, ms/n, lp/n}'
  done
done
