#!/usr/bin/env python3
"""gen_golden.py — independent 50-digit restatement of the reference's GPR formulas; writes tests/golden/*.npz.

TEST INFRASTRUCTURE ONLY.  Run in the development container:  python oracle/gen_golden.py

The reference (kaigu1997/gaussian_process_liouville_equation) ships no golden vectors and cannot be built or
imported here (C++ with absent Eigen/xtensor/NLopt/spdlog headers), so these fixtures are produced by restating
the reference's *formulas* (kernel.cpp, complex_kernel.cpp — line numbers cited below, relative to
/root/reference/gaussian_process_liouville_equation/) in mpmath at 50 significant digits, with plain matrix
inversion instead of any factorisation.  They pin both the C++ oracle (oracle/gple_oracle.cpp) and the HIP path:
every value here is the correctly rounded result of the reference's mathematics (including its quirks, which are
listed where they occur), so a fp64 implementation must agree up to conditioning * epsilon.

Inputs are drawn with numpy PCG64 (seeds listed per case) and converted exactly to mpf.
"""
import os
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 50
PI = mp.pi
SQRT2 = mp.sqrt(2)
PURITY_FACTOR = 2 * PI  # stdafx.h:125 with Dim = 1, hbar = 1
RESCALE_MAX = mp.mpf(10)  # kernel.h:37
CONNECTING = mp.mpf(2)  # kernel.h:16


def mpf_list(a):
    return [mp.mpf(float(x)) for x in a]


def points(X):
    """X: numpy (N,2) -> list of [mpf, mpf] (exact)."""
    return [[mp.mpf(float(r[0])), mp.mpf(float(r[1]))] for r in X]


def delta_kernel(XL, XR, same):  # kernel.cpp:8-31
    R, C = len(XL), len(XR)
    D = mp.zeros(R, C)
    for i in range(R):
        for j in range(C):
            if same:
                D[i, j] = 1 if i == j else 0
            else:
                D[i, j] = 1 if (XL[i][0] == XR[j][0] and XL[i][1] == XR[j][1]) else 0
    return D


def gaussian_kernel(l, XL, XR):  # kernel.cpp:38-85
    R, C = len(XL), len(XR)
    G = mp.zeros(R, C)
    for i in range(R):
        for j in range(C):
            d0 = (XL[i][0] - XR[j][0]) / l[0]
            d1 = (XL[i][1] - XR[j][1]) / l[1]
            G[i, j] = mp.exp(-(d0 * d0 + d1 * d1) / 2)
    return G


def kernel_base(p, XL, XR, same, deriv):
    """p = (magnitude, lx, lp, noise). Returns K and the 4 derivative matrices (kernel.cpp:168-242)."""
    m, l, n = p[0], [p[1], p[2]], p[3]
    R, C = len(XL), len(XR)
    G = gaussian_kernel(l, XL, XR)
    D = delta_kernel(XL, XR, same)
    K = m * m * (G + n * n * D)
    dK = None
    if deriv:
        dK = [K * (2 / m)]
        # kernel.cpp:185-198: training set subtracts the noise diagonal; the test set uses K (noise delta included)
        base = (K - (m * n) ** 2 * D) if same else K
        for d in range(2):
            M = mp.zeros(R, C)
            for i in range(R):
                for j in range(C):
                    diff = (XL[i][d] - XR[j][d]) / l[d]
                    M[i, j] = base[i, j] * diff * diff / l[d]
                if same:
                    M[i, i] = 0
            dK.append(M)
        dK.append((2 * m * m * n) * D if same else mp.zeros(R, C))
    return K, dK


def purity_aux(p):  # kernel.h:285-294
    return (p[0] ** 2 * mp.sqrt(p[1] * p[2]), SQRT2 * p[1], SQRT2 * p[2], mp.mpf(0))


def purity_aux_mixed(a, b):  # complex_kernel.cpp:206-219
    prod = 1
    for d in (1, 2):
        prod *= (1 / a[d] ** 2 + 1 / b[d] ** 2) / 2
    return (a[0] * b[0] / mp.sqrt(mp.sqrt(prod)), mp.sqrt(a[1] ** 2 + b[1] ** 2), mp.sqrt(a[2] ** 2 + b[2] ** 2), mp.mpf(0))


def cutoff(pred_sq, pred_abs, var):  # kernel.h:301-332
    if pred_sq >= CONNECTING ** 2 * var:
        return mp.mpf(1)
    if pred_sq <= var:
        return mp.mpf(0)
    a = pred_abs / mp.sqrt(var)
    return (3 * CONNECTING - 2 * a - 1) * (a - 1) ** 2 / (CONNECTING - 1) ** 3


def colvec(v):
    return mp.matrix(v)


def tonp(M):
    """mp.matrix -> numpy float64 array (column vectors become 1-D), complex if any entry is complex."""
    R, C = M.rows, M.cols
    is_c = any(isinstance(M[i, j], mp.mpc) for i in range(R) for j in range(C))
    out = np.zeros((R, C), dtype=np.complex128 if is_c else np.float64)
    for i in range(R):
        for j in range(C):
            z = M[i, j]
            out[i, j] = complex(z) if is_c else float(z)
    return out[:, 0] if C == 1 else out


def sum_all(M):
    return mp.fsum([M[i, 0] for i in range(M.rows)])


def bil(x, A, y):
    """x^T A y (no conjugation)."""
    return (x.T * A * y)[0, 0]


# ------------------------------------------------------------------------------------------------------------------
def real_case(seed, N, M, theta, Mv, coincident):
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.normal([-10.0, 14.112], [0.7086, 0.7056], size=(N, 2))
    rho = np.exp(-0.5 * (((X[:, 0] + 10.0) / 0.7086) ** 2 + ((X[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
    y = rho * (1 + 0.05 * rng.standard_normal(N))
    Xs = rng.normal([-10.0, 14.112], [1.2, 1.2], size=(M, 2))
    if coincident:
        Xs[:coincident] = X[:coincident]  # exercise the delta kernel's exact-equality branch
    Xv = X[rng.integers(0, N, size=Mv)] + rng.normal(0, [0.3, 0.3], size=(Mv, 2))
    tv = np.exp(-0.5 * (((Xv[:, 0] + 10.0) / 0.7086) ** 2 + ((Xv[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)

    p = tuple(mpf_list(theta))
    sf, lx, lp, sn = p
    XL = points(X)
    K, dK = kernel_base(p, XL, XL, True, True)
    s = RESCALE_MAX / max(abs(mp.mpf(float(v))) for v in y)  # kernel.cpp:279
    ys = colvec([mp.mpf(float(v)) * s for v in y])
    W = mp.inverse(K)
    v = W * ys
    out = dict(theta=np.array(theta), X=X, y=y, Xs=Xs, Xv=Xv, tv=tv)
    out.update(K=tonp(K), dK=np.stack([tonp(d) for d in dK]), W=tonp(W), v=tonp(v), rescale=float(s))
    # (2-norm condition: in mpmath for the small cases; for the large ones numpy's on the rounded matrix — it only scales tolerances)
    out["cond"] = float(mp.norm(K, 2) * mp.norm(W, 2)) if N <= 40 else float(np.linalg.cond(tonp(K)))
    out["magnitude"] = float(mp.sqrt(abs((ys.T * v)[0, 0] / N)))  # kernel.h:167-179
    diagW = [W[i, i] for i in range(N)]
    out["error"] = float(mp.fsum([(v[i] / diagW[i]) ** 2 for i in range(N)]))  # :285
    F = 2 * PI * sf * sf * lx * lp
    out["population"] = float(F * sum_all(v) / s)  # :293
    out["first_order"] = np.array([float(F * mp.fsum([XL[i][d] * v[i] for i in range(N)]) / s) for d in range(2)])  # :308
    p1 = purity_aux(p)
    K1, dK1 = kernel_base(p1, XL, XL, True, True)
    G2 = PURITY_FACTOR * PI
    out["purity"] = float(G2 * bil(v, K1, v) / s ** 2)  # :331
    # derivatives :337-379
    dW = [W * (-2 / sf), -(W * dK[1] * W), -(W * dK[2] * W), W * W * (-2 * sf * sf * sn)]
    dv = [d * ys for d in dW]
    out["dv"] = np.stack([tonp(d) for d in dv])
    derr = []
    for ip in range(4):  # :381-400
        t = mp.fsum([(v[i] / diagW[i]) / diagW[i] * (dv[ip][i] - (v[i] / diagW[i]) * dW[ip][i, i]) for i in range(N)])
        derr.append(float(2 * t))
    out["error_derivative"] = np.array(derr)
    l = [lx, lp]
    dpop = [mp.mpf(0)] + [F * (sum_all(v) / l[d] + sum_all(dv[1 + d])) for d in range(2)] + [F * sum_all(dv[3])]  # :401-435 (d/d sf = 0: sic)
    out["population_derivative"] = np.array([float(x / s) for x in dpop])
    dpur = [mp.mpf(0)]
    for d in range(2):  # :454-463
        comb = K1 / l[d] + SQRT2 * dK1[1 + d]
        dpur.append(G2 * (bil(v, comb, v) + 2 * bil(dv[1 + d], K1, v)))
    dpur.append(2 * G2 * bil(dv[3], K1, v))  # :466
    out["purity_derivative"] = np.array([float(x / s ** 2) for x in dpur])

    # ---- predictive kernel on the test points (kernel.cpp:481-544) ----
    def predict(Xt, labels):
        XT = points(Xt)
        Ks, dKs = kernel_base(p, XT, XL, False, True)
        mu = Ks * v
        self_k = sf * sf * (1 + sn * sn)
        var = [self_k - (Ks[i, :] * W * Ks[i, :].T)[0, 0] for i in range(len(XT))]
        cf = [cutoff(mu[i] ** 2, abs(mu[i]), var[i]) for i in range(len(XT))]
        cut = [mu[i] * cf[i] / s for i in range(len(XT))]
        res = dict(mean=np.array([float(x) for x in mu]), var=np.array([float(x) for x in var]),
                   cutoff_factor=np.array([float(x) for x in cf]), cut=np.array([float(x) for x in cut]))
        if labels is not None:
            lab = [mp.mpf(float(t)) * s for t in labels]
            res["error"] = float(mp.fsum([(mu[i] - lab[i]) ** 2 for i in range(len(XT))]))  # :522 (uncut prediction)
            diff = colvec([cut[i] * s - lab[i] for i in range(len(XT))])  # :527 (cut prediction: reference asymmetry)
            res["error_derivative"] = np.array([float(2 * (diff.T * (dKs[ip] * v + Ks * dv[ip]))[0, 0]) for ip in range(4)])
        return res

    pt = predict(Xs, None)
    out.update({"t_" + k: val for k, val in pt.items()})
    pv = predict(Xv, tv)
    out.update({"v_" + k: val for k, val in pv.items()})
    if N > 64:  # the large case pins what crosses tile boundaries (factor, inverse, weights, predictions, derivative members); its 4 N x N
        out.pop("dK")  # derivative Grams are entry-wise formulas already pinned by the small cases
    return out


# ------------------------------------------------------------------------------------------------------------------
def complex_base(p8, XL, XR, same, deriv):
    """complex_kernel.cpp:134-200. p8 = (s, sR, lRx, lRp, sI, lIx, lIp, sn). Returns dict."""
    s0, sn = p8[0], p8[7]
    RP = (p8[1], p8[2], p8[3], mp.mpf(0))
    IP = (p8[4], p8[5], p8[6], mp.mpf(0))
    ss = [RP[1 + d] ** 2 + IP[1 + d] ** 2 for d in range(2)]
    prod = 1
    for d in range(2):
        prod *= 2 * RP[1 + d] * IP[1 + d] / ss[d]
    CP = (mp.sqrt(RP[0] * IP[0] * prod), mp.sqrt(ss[0] / 2), mp.sqrt(ss[1] / 2), mp.mpf(0))  # :144-157
    KR, dKR = kernel_base(RP, XL, XR, same, deriv)
    KI, dKI = kernel_base(IP, XL, XR, same, deriv)
    KC, dKC = kernel_base(CP, XL, XR, same, deriv)
    D = delta_kernel(XL, XR, same)
    K = s0 * s0 * (KR + KI + sn * sn * D)  # :163
    Kt = s0 * s0 * (KR - KI + 2j * KC)  # :164
    res = dict(K=K, Kt=Kt, RP=RP, IP=IP, CP=CP)
    if deriv:
        R, C = len(XL), len(XR)
        dK = [K * (2 / s0)] + [dKR[i] for i in range(3)] + [dKI[i] for i in range(3)]  # :34-47 (no s^2 factor: sic)
        dK.append((2 * sn) * D if same else mp.zeros(R, C))  # :49-56
        dKt = [Kt * (2 / s0)]  # :94
        for (sub, dsub, sign) in ((RP, dKR, 1), (IP, dKI, -1)):
            dKt.append(sign * dsub[0] + (2j / sub[0]) * KC)  # :101, :117
            for d in range(2):
                l, lc = sub[1 + d], CP[1 + d]
                dKt.append(sign * dsub[1 + d] + 2j * (1 / l - l / lc ** 2) * KC + 1j * (l / lc) * dKC[1 + d])  # :106-108
        dKt.append(mp.zeros(R, C))  # :129
        res.update(dK=dK, dKt=dKt)
    return res


def complex_case(seed, N, M, theta, Mv, coincident):
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.normal([-10.0, 14.112], [0.7086, 0.7056], size=(N, 2))
    rho = np.exp(-0.5 * (((X[:, 0] + 10.0) / 0.7086) ** 2 + ((X[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
    y = 0.5 * rho * np.exp(0.5j * (X[:, 0] + 10.0)) * (1 + 0.05 * rng.standard_normal(N))
    Xs = rng.normal([-10.0, 14.112], [1.2, 1.2], size=(M, 2))
    if coincident:
        Xs[:coincident] = X[:coincident]
    Xv = X[rng.integers(0, N, size=Mv)] + rng.normal(0, [0.3, 0.3], size=(Mv, 2))
    rv = np.exp(-0.5 * (((Xv[:, 0] + 10.0) / 0.7086) ** 2 + ((Xv[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
    tv = 0.5 * rv * np.exp(0.5j * (Xv[:, 0] + 10.0))

    p8 = tuple(mpf_list(theta))
    XL = points(X)
    B = complex_base(p8, XL, XL, True, True)
    K, Kt = B["K"], B["Kt"]
    ym = [mp.mpc(complex(v)) for v in y]
    s = RESCALE_MAX / max(abs(v) for v in ym)  # :262
    ys = colvec([v * s for v in ym])
    Kinv = mp.inverse(K)
    A = Kinv * Kt.conjugate()  # :265
    P = mp.inverse(K - Kt * A)  # :266
    Q = -(A * P)  # :267
    v = P * ys + (Q * ys).conjugate()  # :268
    out = dict(theta=np.array(theta), X=X, y=y, Xs=Xs, Xv=Xv, tv=tv, rescale=float(s))
    out.update(K=tonp(K), Kt=tonp(Kt), P=tonp(P), Q=tonp(Q), v=tonp(v))
    out["magnitude"] = float(mp.sqrt(abs(mp.re((ys.H * v)[0, 0]) / N)))  # complex_kernel.h:192-204
    Pd = [P[i, i] for i in range(N)]
    Qd = [Q[i, i] for i in range(N)]
    sqd = [mp.re(Pd[i]) ** 2 - abs(Qd[i]) ** 2 for i in range(N)]
    diff = [(Pd[i] * v[i] - mp.conj(Qd[i] * v[i])) / sqd[i] for i in range(N)]
    out["error"] = float(mp.fsum([abs(d) ** 2 for d in diff]))  # :270-286
    # purity :287-377
    RP, IP, CP = B["RP"], B["IP"], B["CP"]
    pRC, pIC = purity_aux_mixed(RP, CP), purity_aux_mixed(IP, CP)
    KRp, dKRp_ = kernel_base(purity_aux(RP), XL, XL, True, True)
    KIp, dKIp_ = kernel_base(purity_aux(IP), XL, XL, True, True)
    KCp, dKCp_ = kernel_base(purity_aux(CP), XL, XL, True, True)
    KRC, dKRC_ = kernel_base(pRC, XL, XL, True, True)
    KIC, dKIC_ = kernel_base(pIC, XL, XL, True, True)
    GF = PURITY_FACTOR * 2 * PI
    K1 = KRp + KIp + 2 * KCp
    K2 = KRp - KIp - 2j * (KRC + KIC)
    vc = v.conjugate()
    out["purity"] = float(GF * p8[0] ** 4 * (mp.re(bil(vc, K1, v)) + mp.re(bil(v, K2, v))) / s ** 2)  # :370-373
    # derivatives :379-442
    dK, dKt = B["dK"], B["dKt"]
    Qh = Q.H
    dP, dQ, dv = [], [], []
    for ip in range(8):
        r = P * dK[ip] * P + Qh * dK[ip] * Q + P * dKt[ip] * Q + Qh * dKt[ip].conjugate() * P
        r = -(r + r.H) / 2
        dP.append(r)
        dQ.append(-(Kinv * (dK[ip] * Q + dKt[ip].conjugate() * P)) - A * r)
        dv.append(dP[ip] * ys + (dQ[ip] * ys).conjugate())
    out["dv"] = np.stack([tonp(d) for d in dv])
    derr = []
    for ip in range(8):  # :444-474
        tot = mp.mpf(0)
        for i in range(N):
            pdd, qdd, vd = dP[ip][i, i], dQ[ip][i, i], dv[ip][i]
            num = mp.conj(diff[i]) * (pdd * v[i] + Pd[i] * vd - mp.conj(qdd * v[i] + Qd[i] * vd))
            den = -2 * abs(diff[i]) ** 2 * (Pd[i] * pdd - mp.re(mp.conj(Qd[i]) * qdd))
            tot += mp.re((num + den) / sqd[i])
        derr.append(float(2 * tot))
    out["error_derivative"] = np.array(derr)
    # purity derivative :475-590 (GlobalFactor without magnitude^4: reference quirk)
    lR, lI, lC = [RP[1], RP[2]], [IP[1], IP[2]], [CP[1], CP[2]]
    lRC, lIC = [pRC[1], pRC[2]], [pIC[1], pIC[2]]
    Z = mp.zeros(N, N)
    dKRp, dKIp, dKCp, dKRC, dKIC = ([Z] * 8 for _ in range(5))
    dKRp, dKIp, dKCp, dKRC, dKIC = list(dKRp), list(dKIp), list(dKCp), list(dKRC), list(dKIC)
    dKRp[1] = KRp * (4 / RP[0]); dKCp[1] = KCp * (2 / RP[0]); dKRC[1] = KRC * (3 / RP[0]); dKIC[1] = KIC * (1 / RP[0])
    for d in range(2):
        ip = 2 + d
        dKRp[ip] = KRp / lR[d] + SQRT2 * dKRp_[1 + d]
        dKCp[ip] = (2 / lR[d] - 3 * (lR[d] / lC[d] ** 2) / 2) * KCp + (1 / SQRT2) * (lR[d] / lC[d]) * dKCp_[1 + d]
        dKRC[ip] = (2 / lR[d] - (lR[d] / lC[d] ** 2) / 2) * KRC + mp.mpf(1.5) * (lR[d] / lRC[d]) * (dKRC_[1 + d] - KRC / lRC[d])
        dKIC[ip] = (1 / lR[d] - (lR[d] / lC[d] ** 2) / 2) * KIC + (lR[d] / lIC[d]) / 2 * (dKIC_[1 + d] - KIC / lIC[d])
    dKIp[4] = KIp * (4 / IP[0]); dKCp[4] = KCp * (2 / IP[0]); dKRC[4] = KRC * (1 / IP[0]); dKIC[4] = KIC * (3 / IP[0])
    for d in range(2):
        ip = 5 + d
        dKIp[ip] = KIp / lI[d] + SQRT2 * dKIp_[1 + d]
        dKCp[ip] = (2 / lI[d] - 3 * (lI[d] / lC[d] ** 2) / 2) * KCp + (1 / SQRT2) * (lI[d] / lC[d]) * dKCp_[1 + d]
        dKRC[ip] = (1 / lI[d] - (lI[d] / lC[d] ** 2) / 2) * KRC + (lI[d] / lRC[d]) / 2 * (dKRC_[1 + d] - KRC / lRC[d])
        dKIC[ip] = (2 / lI[d] - (lI[d] / lC[d] ** 2) / 2) * KIC + mp.mpf(1.5) * (lI[d] / lIC[d]) * (dKIC_[1 + d] - KIC / lIC[d])
    dpur = []
    for ip in range(8):
        K1d = dKRp[ip] + dKIp[ip] + 2 * dKCp[ip]
        K2d = dKRp[ip] - dKIp[ip] - 2j * (dKRC[ip] + dKIC[ip])
        r = 2 * mp.re(bil(vc, K1, dv[ip])) + mp.re(bil(vc, K1d, v)) + 2 * mp.re(bil(v, K2, dv[ip])) + mp.re(bil(v, K2d, v))
        dpur.append(float(r * GF / s ** 2))
    out["purity_derivative"] = np.array(dpur)

    def predict(Xt, labels):
        XT = points(Xt)
        Bs = complex_base(p8, XT, XL, False, True)
        Ks, Kts = Bs["K"], Bs["Kt"]
        mu = Ks * v + Kts * vc  # :608
        self_k = p8[0] ** 2 * (p8[1] ** 2 + p8[4] ** 2 + p8[7] ** 2)  # :632
        Pc, Qc = P.conjugate(), Q.conjugate()
        var = []
        for i in range(len(XT)):
            kr, pr = Ks[i, :], Kts[i, :]
            val = self_k - (kr * P * kr.T)[0, 0] - (pr * Pc * pr.H)[0, 0] - (pr * Q * kr.T)[0, 0] - (kr * Qc * pr.H)[0, 0]
            var.append(mp.re(val))  # :631-637
        cf = [cutoff(abs(mu[i]) ** 2, abs(mu[i]), var[i]) for i in range(len(XT))]
        cut = [mu[i] * cf[i] / s for i in range(len(XT))]
        res = dict(mean=np.array([complex(x) for x in mu]), var=np.array([float(x) for x in var]),
                   cutoff_factor=np.array([float(x) for x in cf]), cut=np.array([complex(x) for x in cut]))
        if labels is not None:
            lab = [mp.mpc(complex(t)) * s for t in labels]
            res["error"] = float(mp.fsum([abs(mu[i] - lab[i]) ** 2 for i in range(len(XT))]))  # :646
            dif = colvec([cut[i] * s - lab[i] for i in range(len(XT))])
            ed = []
            for ip in range(8):  # :662 (dot conjugates its first argument)
                w = Bs["dK"][ip] * v + Ks * dv[ip] + Bs["dKt"][ip] * vc + Kts * dv[ip].conjugate()
                ed.append(float(2 * mp.re((dif.H * w)[0, 0])))
            res["error_derivative"] = np.array(ed)
        return res

    pt = predict(Xs, None)
    out.update({"t_" + k: val for k, val in pt.items()})
    pv = predict(Xv, tv)
    out.update({"v_" + k: val for k, val in pv.items()})
    return out


REAL_CASES = {
    # name: (seed, N, M, theta(sf, lx, lp, sn), Mv, coincident test points)
    "real_a": (20240607, 24, 40, (1.3, 0.8, 0.6, 0.05), 30, 3),
    "real_b": (20240608, 32, 48, (1.0, 0.7086, 0.7056, 0.01), 40, 0),  # the reference's initial parameters (opt.cpp:25-27)
    "real_c": (20240609, 8, 16, (0.7, 0.3, 1.1, 0.2), 8, 1),
    # beyond one 64-block (SURVEY.md §8c planned N up to 64, M up to 256): N = 136 pads to n = 256 = four panels of the factorisation, three
    # of them with real rows, so the multi-panel factorisation, the in-launch inverse and the trailing updates meet 50-digit values directly;
    # M = 256 fills two 128-row blocks of the contraction
    "real_d": (20240610, 136, 256, (1.0, 0.7086, 0.7056, 0.02), 64, 4),
}
COMPLEX_CASES = {
    # theta = (s, sR, lRx, lRp, sI, lIx, lIp, sn)
    "complex_a": (20240617, 12, 20, (1.0, 1.2, 0.8, 0.6, 0.9, 0.7, 0.9, 0.05), 12, 2),
    "complex_b": (20240618, 16, 24, (1.0, 1.0, 0.7086, 0.7056, 1.0, 0.7086, 0.7056, 0.01), 16, 0),  # opt.cpp:306-332
    "complex_c": (20240619, 8, 12, (1.4, 0.8, 0.5, 0.9, 1.1, 0.8, 0.4, 0.1), 8, 1),  # s != 1 pins the missing-s^2 quirks
    # N = 72: 144 real rows in the [Re; Im] embedding (Re rows 0..71 of block rows 0-1, Im rows 256..327 of block rows 4-5 at n = 512): more
    # than two 64-blocks of real data, the Re-Im coupling blocks span panels
    "complex_d": (20240620, 72, 96, (1.0, 1.2, 0.8, 0.6, 0.9, 0.7, 0.9, 0.05), 32, 2),
}

if __name__ == "__main__":
    outdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    only = set(sys.argv[1:])
    for name, args in REAL_CASES.items():
        if only and name not in only:
            continue
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **real_case(*args))
        print("wrote", name, flush=True)
    for name, args in COMPLEX_CASES.items():
        if only and name not in only:
            continue
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **complex_case(*args))
        print("wrote", name, flush=True)
