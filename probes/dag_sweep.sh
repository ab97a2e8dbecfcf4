for F in "60,80" "25,60,80" "35,60,80" "25,50,70,85" "20,40,60,80" "30,55,75,90"; do
for R in 0 48; do
echo "forks $F reserve $R: $(GPLE_CHOL_FORKS=$F GPLE_CHOL_SIDE_RESERVE=$R GPLE_CHOL_SCHEME=dag timeout -k 10 150 python probes/dag_check.py 4096 8192 2>&1 | grep 'fit ' | tr '\n' ' ')"
done; done
