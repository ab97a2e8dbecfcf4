"""Which part of the objective's gradient deviates from central differences at large N?  (LOOCV part = fit error derivative,
validation part = predict error derivative on extra points whose cut-off factor is 1.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests.test_gpu_configs import config_inputs, extra_set, THETA_R, THETA_C

api = pkg.open_api(0)
for cplx in (False, True):
    for N in (512, 1024, 2048, 4096):
        X, y, _, _ = config_inputs(N, 8, 20240607 + 3, cplx=cplx)
        Xe, ye = extra_set(X, 77, cplx)
        theta = np.array(THETA_C if cplx else THETA_R); theta[-1] = 0.05
        fitf = api.complex_fit if cplx else api.real_fit
        predf = api.complex_predict if cplx else api.real_predict
        f0 = fitf(theta, X, y, 1)
        pe = predf(f0, Xe)
        keep = np.abs(pe["prediction"]) ** 2 >= 9.0 * pe["variance"]
        Xk, yk = Xe[keep], (ye[keep] if cplx else ye[keep].real)
        fd_ = fitf(theta, X, y, 1 | 4)
        g_fit = fd_.scalars["error_derivative"]
        g_val = predf(fd_, Xk, flags=4, labels=yk)["error_derivative"]
        for ip in ([1, 2] if not cplx else [2, 3, 5]):
            h = 1e-5 * theta[ip]
            tp, tm = theta.copy(), theta.copy(); tp[ip] += h; tm[ip] -= h
            fp, fm = fitf(tp, X, y, 1), fitf(tm, X, y, 1)
            fd_fit = (fp.scalars["error"] - fm.scalars["error"]) / (2 * h)
            fd_val = (predf(fp, Xk, labels=yk)["error"] - predf(fm, Xk, labels=yk)["error"]) / (2 * h)
            print(f"{'complex' if cplx else 'real'} N={N} kept={keep.sum()} ip={ip}: fit grad {g_fit[ip]:.8e} fd {fd_fit:.8e} rel {abs(g_fit[ip]-fd_fit)/abs(fd_fit):.2e} | "
                  f"val grad {g_val[ip]:.8e} fd {fd_val:.8e} rel {abs(g_val[ip]-fd_val)/abs(fd_val):.2e}", flush=True)
api.close()
