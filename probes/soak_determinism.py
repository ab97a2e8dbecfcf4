"""Soak: the same random sequence of fits / predicts / objective evaluations twice; every output must repeat bit for bit
(no data race shows up as run-to-run noise), and the deferred-scalar and pooled paths are exercised under churn."""
import sys, hashlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import kernels as K
import parity


def run(api, iters, seed):
    rng = np.random.default_rng(seed)
    h = hashlib.sha256()
    for it in range(iters):
        N = int(rng.integers(2, 1100)); M = int(rng.choice([1, 50, 900, 5000, 20000]))
        X, y, Xs = parity.synthetic_real(N, M, int(rng.integers(1, 10 ** 6)))
        th = [float(rng.uniform(0.7, 1.4)), float(rng.uniform(0.4, 1.0)), float(rng.uniform(0.4, 1.0)), float(10 ** rng.uniform(-2, -0.7))]
        kind = it % 3
        if kind == 0:
            fit = api.real_fit(th, X, y, 7, defer_scalars=bool(it % 2))
            p = api.real_predict(fit, Xs)
            h.update(p["prediction"].tobytes()); h.update(p["variance"].tobytes())
            h.update(np.asarray([fit.scalars[k] for k in ("error", "population", "purity")]).tobytes())
            h.update(np.asarray(fit.scalars["purity_derivative"]).tobytes())
            fit.release()
        elif kind == 1:
            Nc = min(N, 400)
            yc = 0.5 * y[:Nc] * np.exp(0.5j * (X[:Nc, 0] + 10.0))
            thc = [th[0], 1.0, th[1], th[2], 1.2, th[2], th[1], th[3]]
            fit = api.complex_fit(thc, X[:Nc], yc, 7)
            p = api.complex_predict(fit, Xs[:3000])
            h.update(p["prediction"].tobytes()); h.update(p["variance"].tobytes())
            h.update(np.asarray(fit.scalars["error_derivative"]).tobytes())
            fit.release()
        else:
            v, g = api.loose_function(th, X, y.astype(complex), Xs[:300], np.ones(min(300, M), complex) * 0.01)
            h.update(np.float64(v).tobytes()); h.update(g.tobytes())
    return h.hexdigest()


api = pkg.open_api(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
a = run(api, iters, 1)
print("pass 1", a, flush=True)
b = run(api, iters, 1)
print("pass 2", b, flush=True)
api2 = pkg.open_api(0)
c = run(api2, iters, 1)
print("pass 3 (fresh context)", c, flush=True)
print("DETERMINISTIC" if a == b == c else "MISMATCH")
api.close(); api2.close()
sys.exit(0 if a == b == c else 1)
