# where the in-kernel inverse stops paying: one outer block (budget 1600) with and without it, against two blocks (budget 800)
for N in 2048 2560 3072 3584; do
echo "n $N: inside $(GPLE_CHOL_TILE_BUDGET=1600 timeout -k 10 150 python probes/dag_check.py $N 2>&1 | grep 'fit ' | awk '{printf "%s ", $2}') tree $(GPLE_CHOL_TILE_BUDGET=1600 GPLE_CHOL_DAG_INVERSE=0 timeout -k 10 150 python probes/dag_check.py $N 2>&1 | grep 'fit ' | awk '{printf "%s ", $2}') budget-800 $(GPLE_CHOL_TILE_BUDGET=800 timeout -k 10 150 python probes/dag_check.py $N 2>&1 | grep 'fit ' | awk '{printf "%s ", $2}')"
done
