# first cut of the n = 4096 factorisation: forks x tile budget (the trailing update after the first block has (n - J)^2 J flops on (n - J)^2 / 8192 tiles)
for F in "60,80" "64,82" "67,84" "70,85" "56,78"; do for B in 1600 2400 3200; do
echo "forks $F budget $B: $(GPLE_CHOL_FORKS=$F GPLE_CHOL_TILE_BUDGET=$B timeout -k 10 150 python probes/fit_timing.py real 4096 2>&1 | grep 'fit ' | awk '{printf "%s %s %s", $4, $5, $6}')"
done; done
