"""Soak of the one-launch factorisation: every tile is summed in a fixed order whatever the timing of its hand-overs, so repeated fits of the same inputs
must agree BIT FOR BIT — any difference is a race (a tile read before it was final, a stale cache line).  Sizes interleaved so that flag words change their
meaning between fits (the flag buffer is never cleared); real and complex.  usage: python probes/dag_soak.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests import parity
from tests.test_gpu_configs import config_inputs, THETA_R, THETA_C

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
api = pkg.open_api(0)
cases = [(False, N) for N in (256, 640, 1024, 1536, 2048, 3072, 3584, 4096)] + [(True, N) for N in (256, 1024, 2048)]
inputs = {k: config_inputs(k[1], 8, 1, cplx=k[0]) for k in cases}
ref = {}
t0 = time.time()
bad = 0
for r in range(rounds):
    order = list(cases)
    np.random.default_rng(r).shuffle(order)
    for k in order:
        X, y, _, _ = inputs[k]
        f = (api.complex_fit if k[0] else api.real_fit)(THETA_C if k[0] else THETA_R, X, y, 3)
        v = f.get(c.C_INVLBL if k[0] else c.R_INVLBL)
        sig = (f.scalars["info"], f.scalars["error"], v.tobytes())
        f.release()
        if k not in ref:
            ref[k] = sig
            assert sig[0] == 0, (k, sig[0])
        elif sig != ref[k]:
            bad += 1
            print("MISMATCH", k, "round", r, "info", sig[0], "error", sig[1], "vs", ref[k][1], flush=True)
    if r % 10 == 9:
        print(f"round {r + 1}: {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
print("soak done:", rounds * len(cases), "fits,", bad, "mismatches")
api.close()
sys.exit(1 if bad else 0)
