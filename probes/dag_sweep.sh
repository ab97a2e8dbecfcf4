# fork points of the block-row inverse with the one-launch factorisation (A/B through GPLE_CHOL_FORKS)
for F in "60,80" "60,80,92" "55,75,90" "60,82,94" "50,70,85,95" "65,85" "60,80,90,96"; do
echo "forks $F: $(GPLE_CHOL_FORKS=$F timeout -k 10 150 python probes/dag_check.py 2048 4096 8192 2>&1 | grep 'fit ' | tr '\n' ' ')"
done
