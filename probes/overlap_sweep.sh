#!/bin/bash
# sweep of the overlapped predict's knobs on the per-rank proxy (bench.py --emulate-rank 0/8): CUs left to the fit, N-tiles that go early, K* generation early or late
run() { python bench.py --via capi --emulate-rank 0/8 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());p=d['phases_ms'];print('$1', d['value'], 'fit', p['fit_device'], 'predict', p['predict_device'], 'late kernel', p['rownorm_kernel_per_step'])"; }
GPLE_PREDICT_OVERLAP=0 run "off                      "
for cus in 64 128 192; do for tiles in 9 6 4; do for late in 0 1; do
  GPLE_PREDICT_OVERLAP=1 GPLE_PREDICT_OVERLAP_CUS=$cus GPLE_PREDICT_OVERLAP_TILES=$tiles GPLE_PREDICT_OVERLAP_KSTAR_LATE=$late run "cus $cus tiles $tiles kstar_late $late"
done; done; done
