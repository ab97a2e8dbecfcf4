# workgroups of the panel launches behind the first fork (GPLE_CHOL_DAG_LATE_BLOCKS), with the in-launch inverse of their row blocks
for B in 48 64 96 128 192 0; do
echo "late blocks $B: $(GPLE_CHOL_DAG_LATE_BLOCKS=$B timeout -k 10 150 python probes/fit_timing.py real 4096 8192 2>&1 | grep 'fit ' | awk '{printf "%s ", $4}')"
done
