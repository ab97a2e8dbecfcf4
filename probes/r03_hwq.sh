#!/bin/bash
for pre in 0 1 2 3 4 6; do for comm in 0 1; do
  python probes/hwqueue_fit_probe.py $pre $comm 2>&1 | grep pre_streams || exit 1
  GPLE_CHOL_SIDE_PRIORITY=0 python probes/hwqueue_fit_probe.py $pre $comm 2>&1 | grep pre_streams || exit 1
done; done
for pre in 0 3 6; do
  GPU_MAX_HW_QUEUES=8 python probes/hwqueue_fit_probe.py $pre 1 2>&1 | grep pre_streams || exit 1
  GPU_MAX_HW_QUEUES=8 GPLE_CHOL_SIDE_PRIORITY=0 python probes/hwqueue_fit_probe.py $pre 1 2>&1 | grep pre_streams || exit 1
done
