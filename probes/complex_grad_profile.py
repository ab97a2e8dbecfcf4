"""One complex objective + gradient evaluation (opt.cpp:441-482) at N = 1024, repeated; run under rocprofv3 --kernel-trace --stats."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
import parity
api = pkg.open_api(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
X, y, Xs = parity.synthetic_real(N, N // 4, 7 + N)
yc = 0.5 * y * np.exp(0.5j * (X[:, 0] + 10.0))
ye = np.interp(Xs[:, 0], np.sort(X[:, 0]), y[np.argsort(X[:, 0])])
yec = 0.5 * ye * np.exp(0.5j * (Xs[:, 0] + 10.0))
thc = [1.0, 1.0, 0.7086, 0.7056, 1.2, 0.8, 0.6, 1e-2]
for _ in range(6):
    api.loose_function(thc, X, yc, Xs, yec, want_grad=True)
api.close()
