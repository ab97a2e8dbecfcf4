"""CPU ORACLE of the N-level step loop ("extend beyond NumPES == 2") — TEST INFRASTRUCTURE ONLY.

The reference stops at two levels: non_adiabatic_evolve_predict asserts "NO INSTANTATION OF MORE THAN TWO LEVEL SYSTEM NOW"
(evolve.cpp:367-371).  Everything around that function is generic in NumPES — pes.cpp:73-155 (SelfAdjointEigenSolver for
NumPES > 2, C^T F C, d_jk = F_jk / (E_j - E_k)), is_coupling (evolve.cpp:64-79), adiabatic_evolve, calculate_omega0,
evolve() — so what has to be DERIVED is the back-propagation alone.  DESIGN.md §10 writes the derivation out; in short, the
reference's two-level code is the symmetric Trotter splitting of the mixed quantum-classical Liouville equation (Dim = 1)

    exp(L dt) ~ A(dt/2) R(dt/2) J(dt) R(dt/2) A(dt/2)

    A: classical motion of element (k, l) on the mean surface (F_kk + F_ll) / 2 with the phase exp(-i (E_k - E_l) t / hbar)
    R: the non-adiabatic-coupling commutator  d rho / dt = -[v D, rho],  D_kl = d_kl  ->  rho <- O rho O^T,  O = exp(-v D t)
       (two levels: the "off-diagonal rotation" by the angle 2 phi t of evolve.cpp:219-235)
    J: the off-diagonal force anticommutator  d rho / dt = -1/2 {F_off, d rho / dp}.  In the eigenbasis of F_off (eigenvalues
       lambda_a, projectors P_a) the component P_a rho P_b is translated in momentum by (lambda_a + lambda_b) / 2 * t
       (two levels: F_off = f01 sigma_x, lambda = +-f01, the three shifts +f01, 0, -f01 = the "3 branches" of evolve.cpp:9-15 and
       the combinations (rho00 +- 2 Re rho10 + rho11) / 4, (rho00 - rho11) / 2, i Im rho10 of :339-367 are P_+- rho P_+-, P_+ rho P_- + h.c.)

For N levels that is N (N + 1) / 2 momentum branches x N (N + 1) / 2 source elements of predicted densities per point
(3 x 3 = 9 at N = 2 — one of which is the exact density the point carries, evolve.cpp:309-313 — 6 x 6 = 36 at N = 3).
This module restates that for any N with numpy's own eigen-solver and a Pade matrix exponential, independently of the
device code (csrc/gple_evolve_n.hip: cyclic Jacobi, Taylor + squaring); tests/test_step_loop_oracle.py checks that N = 2
reproduces oracle/evolve_oracle.py (the restatement of the reference's two-level code) to rounding.

Potentials.  pes.cpp's diabatic_potential fills the upper-left 2 x 2 block only; compiled with NumPES = 3 the reference's own
models are Tully's two surfaces plus an uncoupled third diabat at V = 0 (models 0-2 here do exactly that).  Model 3 (TSAC,
"three-state avoided crossings") is OURS — the reference has no genuinely three-level model:
    V00 = A tanh(B x), V11 = 0, V22 = -A tanh(B x), V01 = V12 = C / cosh(D x), V02 = 0;  A = 0.02, B = 0.8, C = 0.005, D = 0.5
(the couplings decay like 2 C exp(-D |x|), slowly enough that every eigenvector component stays far above rounding over the whole
phase-space box: the sign convention below never has to decide on a component that has underflowed)
Sign convention of the adiabatic states (the reference takes whatever Eigen returns for NumPES > 2 — unpinned): eigenvalues
ascending, every eigenvector with its LAST NON-ZERO component positive.  For two levels this is pes.cpp:73-96 (second
component 1 before normalisation); for a tridiagonal diabatic matrix with positive couplings the first and last components
of an eigenvector never vanish, so the convention is continuous in x.
"""
import numpy as np
from scipy.linalg import expm

hbar = 1.0
SAC, DAC, ECR, TSAC = 0, 1, 2, 3
TSAC_A, TSAC_B, TSAC_C, TSAC_D = 0.02, 0.8, 0.005, 0.5


def elements(num_pes):
    """lower-triangular packing order of the density-matrix elements (storage.h): (0,0), (1,0), (1,1), (2,0), ..."""
    return [(i, j) for i in range(num_pes) for j in range(i + 1)]


def sgn(x):
    return (x > 0).astype(float) - (x < 0).astype(float)


def diabatic(x, model, num_pes):
    """V (m, N, N) and F = -dV/dx (m, N, N) of pes.cpp:25-69 compiled for num_pes levels (models 0-2), or of TSAC (model 3)"""
    x = np.asarray(x, dtype=float).reshape(-1)
    V, F = np.zeros((len(x), num_pes, num_pes)), np.zeros((len(x), num_pes, num_pes))
    if model == SAC:
        e = np.exp(-sgn(x) * 1.6 * x)
        V[:, 0, 0] = sgn(x) * 0.01 * (1.0 - e)
        V[:, 1, 1] = -V[:, 0, 0]
        V[:, 0, 1] = V[:, 1, 0] = 0.005 * np.exp(-1.0 * x ** 2)
        F[:, 0, 0] = -0.01 * 1.6 * e
        F[:, 1, 1] = -F[:, 0, 0]
        F[:, 0, 1] = F[:, 1, 0] = 2.0 * 0.005 * 1.0 * x * np.exp(-1.0 * x ** 2)
    elif model == DAC:
        V[:, 1, 1] = 0.05 - 0.10 * np.exp(-0.28 * x ** 2)
        V[:, 0, 1] = V[:, 1, 0] = 0.015 * np.exp(-0.06 * x ** 2)
        F[:, 1, 1] = -2 * 0.10 * 0.28 * x * np.exp(-0.28 * x ** 2)
        F[:, 0, 1] = F[:, 1, 0] = 2 * 0.015 * 0.06 * x * np.exp(-0.06 * x ** 2)
    elif model == ECR:
        e = np.exp(-sgn(x) * 0.90 * x)
        V[:, 0, 0], V[:, 1, 1] = 6e-4, -6e-4
        V[:, 0, 1] = V[:, 1, 0] = 0.10 * (1 - sgn(x) * (e - 1))
        F[:, 0, 1] = F[:, 1, 0] = -0.10 * 0.90 * e
    elif model == TSAC:
        assert num_pes == 3
        t, g, th = np.tanh(TSAC_B * x), 1.0 / np.cosh(TSAC_D * x), np.tanh(TSAC_D * x)
        V[:, 0, 0], V[:, 2, 2] = TSAC_A * t, -TSAC_A * t
        V[:, 0, 1] = V[:, 1, 0] = V[:, 1, 2] = V[:, 2, 1] = TSAC_C * g
        F[:, 0, 0], F[:, 2, 2] = -TSAC_A * TSAC_B * (1 - t ** 2), TSAC_A * TSAC_B * (1 - t ** 2)
        F[:, 0, 1] = F[:, 1, 0] = F[:, 1, 2] = F[:, 2, 1] = TSAC_C * TSAC_D * g * th  # -d/dx [C sech(D x)] = C D sech tanh
    else:
        raise ValueError(model)
    return V, F


def adiabatic(x, model, num_pes):
    """E (m, N) ascending, C (m, N, N) columns = eigenvectors (last non-zero component positive), F = C^T F_dia C (m, N, N),
    NAC (m, N, N) antisymmetric with NAC[j, k] = F[j, k] / (E[j] - E[k]) for j > k (pes.cpp:137-155; 0 where F[j, k] is exactly 0)"""
    V, Fd = diabatic(x, model, num_pes)
    if model != TSAC and num_pes == 3:
        # Tully's two surfaces + the uncoupled third diabat: the two-level eigenvectors of pes.cpp:73-96 embedded, the spectator e_2, sorted
        # by energy (eigh would return +-1e-17 where exact zeros belong, and F of the spectator must be exactly 0 for the NAC guard below)
        E2, C2, _, _ = adiabatic(x, model, 2)
        E = np.concatenate([E2, V[:, 2, 2][:, None]], axis=1)
        C = np.zeros((len(E), 3, 3))
        C[:, :2, :2], C[:, 2, 2] = C2, 1.0
        order = np.argsort(E, axis=1, kind="stable")
        E = np.take_along_axis(E, order, axis=1)
        C = np.take_along_axis(C, order[:, None, :], axis=2)
    else:
        E, C = np.linalg.eigh(V)
        for m in range(len(E)):
            for k in range(num_pes):
                nz = np.nonzero(C[m, :, k] != 0.0)[0]
                if len(nz) and C[m, nz[-1], k] < 0:
                    C[m, :, k] = -C[m, :, k]
    F = np.einsum("mak,mab,mbl->mkl", C, Fd, C)
    F = 0.5 * (F + np.swapaxes(F, 1, 2))
    NAC = np.zeros_like(F)
    for j in range(1, num_pes):
        for k in range(j):
            with np.errstate(divide="ignore", invalid="ignore"):
                d = np.where(F[:, j, k] == 0.0, 0.0, F[:, j, k] / (E[:, j] - E[:, k]))
            NAC[:, j, k], NAC[:, k, j] = d, -d
    return E, C, F, NAC


def adiabatic_evolve(x, p, mass, dt, drc, row, col, model, num_pes):
    """evolve.cpp:103-128"""
    x = x + drc * dt / 2.0 * (p / mass)
    _, _, F, _ = adiabatic(x, model, num_pes)
    p = p + drc * dt / 2.0 * (F[:, row, row] + F[:, col, col])
    x = x + drc * dt / 2.0 * (p / mass)
    return x, p


def omega(xa, xb, k, l, model, num_pes):
    """calculate_omega0(xa, xb, Forward, l, k) of evolve.cpp:137-151 as the back-propagation calls it for the stored element (k, l), k > l:
    (E_l - E_k) averaged over the two positions"""
    Ea, Eb = adiabatic(xa, model, num_pes)[0], adiabatic(xb, model, num_pes)[0]
    return (Ea[:, l] - Ea[:, k] + Eb[:, l] - Eb[:, k]) / 2.0 / hbar


def rotation(x, p, mass, dt, model, num_pes):
    """O = exp(-v D dt), D = the antisymmetric matrix of non-adiabatic couplings: the N-level form of evolve.cpp:219-235"""
    NAC = adiabatic(x, model, num_pes)[3]
    return np.stack([expm(-(p[m] / mass) * NAC[m] * dt) for m in range(len(x))])


def hermitian(vals, num_pes):
    """lower-packed element values {(k, l): (m,)} -> (m, N, N) Hermitian matrices"""
    m = len(next(iter(vals.values())))
    R = np.zeros((m, num_pes, num_pes), dtype=complex)
    for (k, l), v in vals.items():
        R[:, k, l] = v
        if k != l:
            R[:, l, k] = np.conj(v)
    return R


def non_adiabatic_evolve_predict(r, density, mass, dt, distribution, row, col, model, num_pes):
    """the N-level form of evolve.cpp:184-372 for every row of r (m, 2); density: the exact values of element (row, col) at r, or None"""
    x0, p0 = r[:, 0], r[:, 1]
    m = len(r)
    x2, p1 = adiabatic_evolve(x0, p0, mass, dt / 2.0, -1.0, row, col, model, num_pes)
    _, _, F2, _ = adiabatic(x2, model, num_pes)
    Foff = F2.copy()
    for k in range(num_pes):
        Foff[:, k, k] = 0.0
    lam, W = np.linalg.eigh(Foff)
    scale = np.abs(lam).max(axis=1)
    comb = np.zeros((m, num_pes, num_pes), dtype=complex)
    for a in range(num_pes):
        for b in range(a, num_pes):
            shift = 0.5 * (lam[:, a] + lam[:, b])
            zero = np.abs(shift) <= 1e-13 * scale
            p2 = p1 + dt * shift                                             # backward: p2 = p1 - n dt f01 with n = -+1 <-> shift = +-f01
            x3 = x2 - (dt / 4.0) * p2 / mass
            F3 = adiabatic(x3, model, num_pes)[2]
            vals = {}
            for (k, l) in elements(num_pes):
                p3 = p2 - (dt / 4.0) * (F3[:, k, k] + F3[:, l, l])
                x4 = x3 - (dt / 4.0) * p3 / mass
                v = np.asarray(distribution(np.stack([x4, p3], axis=1), k, l), dtype=complex)
                if density is not None and (k, l) == (row, col):
                    v = np.where(zero, np.asarray(density, dtype=complex), v)  # evolve.cpp:309-313: the branch that retraces the forward step
                if k != l:
                    v = v * np.exp(1j * omega(x2, x4, k, l, model, num_pes) * dt / 2.0)
                vals[(k, l)] = v
            R = hermitian(vals, num_pes)
            O = rotation(x2, p2, mass, dt / 2.0, model, num_pes)
            R = O @ R @ np.swapaxes(O, 1, 2)
            Pa = np.einsum("mi,mj->mij", W[:, :, a], W[:, :, a])
            Pb = np.einsum("mi,mj->mij", W[:, :, b], W[:, :, b])
            comb += Pa @ R @ Pb
            if a != b:
                comb += Pb @ R @ Pa
    O = rotation(x2, p1, mass, dt / 2.0, model, num_pes)
    comb = O @ comb @ np.swapaxes(O, 1, 2)
    res = comb[:, row, col]
    if row != col:
        res = res * np.exp(1j * omega(x0, x2, row, col, model, num_pes) * dt / 2.0)
    return res


def evolve(density, mass, dt, distribution, model, num_pes):
    """evolve.cpp:377-423 for num_pes levels (is_coupling is true for every finite point: CouplingCriterion = 0 with >=, evolve.cpp:60-82)"""
    out = {}
    for (i, j) in elements(num_pes):
        r, rho = density[(i, j)]
        r, rho = np.asarray(r, dtype=float).reshape(-1, 2), np.asarray(rho, dtype=complex)
        if len(r) == 0:
            out[(i, j)] = (r.copy(), rho.copy())
            continue
        x2, p1 = adiabatic_evolve(r[:, 0], r[:, 1], mass, dt / 2, 1.0, i, j, model, num_pes)
        x4, p2 = adiabatic_evolve(x2, p1, mass, dt / 2, 1.0, i, j, model, num_pes)
        r_new = np.stack([x4, p2], axis=1)
        out[(i, j)] = (r_new, non_adiabatic_evolve_predict(r_new, rho, mass, dt, distribution, i, j, model, num_pes))
    return out


def new_point_predict(r, mass, dt, distribution, row, col, model, num_pes):
    """evolve.cpp:425-443"""
    r = np.asarray(r, dtype=float).reshape(-1, 2)
    return non_adiabatic_evolve_predict(r, None, mass, dt, distribution, row, col, model, num_pes)
