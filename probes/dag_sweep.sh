# tile rule of the triangular / lower-only GEMMs of the fit (trailing updates, block-row products) with the one-launch factorisation
for V in 4096 512 128 64; do
echo "128-tiles from $V tri tiles: $(GPLE_GEMM_128_MIN_TILES_TRI=$V timeout -k 10 150 python probes/fit_timing.py real 4096 8192 2>&1 | grep 'fit ' | awk '{printf "%s ", $4}')"
done
for V in 256 512 1024; do
echo "split-k up to $V tiles: $(GPLE_GEMM_SPLITK_MAX_TILES=$V timeout -k 10 150 python probes/fit_timing.py real 4096 8192 2>&1 | grep 'fit ' | awk '{printf "%s ", $4}')"
done
