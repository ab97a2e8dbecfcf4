"""CPU ORACLE of the per-tick step loop — TEST INFRASTRUCTURE ONLY, like everything under oracle/.

numpy restatement, vectorised over the sample points, of the reference's MQCLE point propagation for the case the reference
itself instantiates (NumPES = 2, Dim = 1; evolve.cpp:367-371 asserts for more levels):
    Tully's models and the adiabatic representation       pes.cpp:8-189, pes.h:8-41
    is_coupling, adiabatic_evolve, calculate_omega0        evolve.cpp:39-170
    non_adiabatic_evolve_predict (3-branch back-propagation) evolve.cpp:184-372
    evolve                                                 evolve.cpp:377-423
    generate_markov_chain (Metropolis)                     mc.cpp:118-165
Paths relative to /root/reference/gaussian_process_liouville_equation/.  The reference seeds its Mersenne twister from the
clock (mc.cpp:17) and shares it between threads, so no random stream of the reference can be reproduced; the Metropolis
restatement here draws from a counter-based Philox4x32-10 stream (Salmon et al., SC'11) that the device kernel reproduces
bit for bit.  `distribution(points (m, 2), iPES, jPES) -> complex (m,)` is the batched form of the reference's
DistributionFunction (stdafx.h:155); the tests plug in the C++ oracle's predictors.  Parity unpinned (no reference vectors).
"""
import numpy as np

hbar = 1.0  # stdafx.h:107
SAC, DAC, ECR = 0, 1, 2  # pes.h:27-32; the reference's default is DAC (pes.h:38-41)
Backward, Forward = -1, 1  # evolve.cpp:18-22
OffDiagonalBranches = (-1, 0, 1)  # evolve.cpp:9
TRI = {(0, 0): 0, (1, 0): 1, (1, 1): 2}  # calculate_lower_triangular_index, evolve.cpp:197-200


def sgn(x):  # pes.h:13-17
    return (x > 0).astype(float) - (x < 0).astype(float)


def diabatic_potential(x, model):
    """pes.cpp:25-47 -> V00, V01, V11 (arrays like x)"""
    x = np.asarray(x, dtype=float)
    z = np.zeros_like(x)
    if model == SAC:
        v00 = sgn(x) * 0.01 * (1.0 - np.exp(-sgn(x) * 1.6 * x))
        return v00, 0.005 * np.exp(-1.0 * x ** 2), -v00
    if model == DAC:
        return z, 0.015 * np.exp(-0.06 * x ** 2), 0.05 - 0.10 * np.exp(-0.28 * x ** 2)
    return z + 6e-4, 0.10 * (1 - sgn(x) * (np.exp(-sgn(x) * 0.90 * x) - 1)), z - 6e-4


def diabatic_force(x, model):
    """pes.cpp:49-69 -> F00, F01, F11 (= -dV/dx)"""
    x = np.asarray(x, dtype=float)
    z = np.zeros_like(x)
    if model == SAC:
        f00 = -0.01 * 1.6 * np.exp(-sgn(x) * 1.6 * x)
        return f00, 2.0 * 0.005 * 1.0 * x * np.exp(-1.0 * x ** 2), -f00
    if model == DAC:
        return z, 2 * 0.015 * 0.06 * x * np.exp(-0.06 * x ** 2), -2 * 0.10 * 0.28 * x * np.exp(-0.28 * x ** 2)
    return z, -0.10 * 0.90 * np.exp(-sgn(x) * 0.90 * x), z


def adiabatic_potential(x, model):
    """pes.cpp:98-120 -> E0, E1"""
    v00, v01, v11 = diabatic_potential(x, model)
    root = np.sqrt((v00 - v11) ** 2 + (2.0 * v01) ** 2)
    return (-root + (v00 + v11)) / 2.0, (root + (v00 + v11)) / 2.0


def diabatic_to_adiabatic_matrix(x, model):
    """pes.cpp:73-96 -> C00, C01, C10, C11 (columns normalised)"""
    v00, v01, v11 = diabatic_potential(x, model)
    root = np.sqrt((v00 - v11) ** 2 + 4.0 * v01 ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        c00 = (-root + (v00 - v11)) / (2.0 * v01)
        c01 = (root + (v00 - v11)) / (2.0 * v01)
    c10, c11 = np.ones_like(c00), np.ones_like(c00)
    n0, n1 = np.sqrt(c00 ** 2 + c10 ** 2), np.sqrt(c01 ** 2 + c11 ** 2)
    return c00 / n0, c01 / n1, c10 / n0, c11 / n1


def adiabatic_force(x, model):
    """pes.cpp:122-135: C^T F C, the lower triangle mirrored (selfadjointView<Lower>) -> F00, F10 (= F01), F11"""
    f00, f01, f11 = diabatic_force(x, model)
    c00, c01, c10, c11 = diabatic_to_adiabatic_matrix(x, model)
    # M = F C (F symmetric), A = C^T M
    m00, m01 = f00 * c00 + f01 * c10, f00 * c01 + f01 * c11
    m10, m11 = f01 * c00 + f11 * c10, f01 * c01 + f11 * c11
    return c00 * m00 + c10 * m10, c01 * m00 + c11 * m10, c01 * m01 + c11 * m11


def adiabatic_coupling_01(x, model):
    """pes.cpp:137-155: NAC(0,1) = -F(1,0) / (E1 - E0)"""
    e0, e1 = adiabatic_potential(x, model)
    _, f10, _ = adiabatic_force(x, model)
    with np.errstate(divide="ignore", invalid="ignore"):
        return -(f10 / (e1 - e0))


def is_coupling(x, p, mass, dt, model):
    """evolve.cpp:39-82 with CouplingCriterion = 0 and `>=`: true wherever either expression is not NaN"""
    f00, f01, f11 = adiabatic_force(x, model)
    nac = adiabatic_coupling_01(x, model)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (np.abs(nac * p / mass) * dt >= 0) | (np.abs(f01 / ((f00 + f11) / 2.0)) >= 0)


def adiabatic_evolve(x, p, mass, dt, drc, row, col, model):
    """evolve.cpp:103-128: half position step, momentum step with (f_row + f_col) / 2 ... written as dt / 2 * (f_row + f_col)"""
    x = x + drc * dt / 2.0 * (p / mass)
    f00, _, f11 = adiabatic_force(x, model)
    f = (f00, f11)
    p = p + drc * dt / 2.0 * (f[row] + f[col])
    x = x + drc * dt / 2.0 * (p / mass)
    return x, p


def calculate_omega0(x0, x2, drc, row, col, model):
    """evolve.cpp:137-151"""
    if row == col:
        return np.zeros_like(np.asarray(x0, dtype=float))
    e0, e2 = adiabatic_potential(x0, model), adiabatic_potential(x2, model)
    return drc * (e0[row] - e0[col] + e2[row] - e2[col]) / 2.0 / hbar


def offdiagonal_rotation(rho, x, p, mass, dt, model):
    """evolve.cpp:219-235; rho: list of three complex arrays (rho00, rho10, rho11), returns the rotated three"""
    phi = p / mass * adiabatic_coupling_01(x, model) * is_coupling(x, p, mass, dt, model).astype(float)
    c, s = np.cos(2.0 * phi * dt), np.sin(2.0 * phi * dt)
    r0, r1, r2 = rho
    return [(1.0 + c) / 2.0 * r0 - s * r1.real + (1.0 - c) / 2.0 * r2,
            s / 2.0 * r0 + c * r1.real + 1j * r1.imag - s / 2.0 * r2,
            (1.0 - c) / 2.0 * r0 + s * r1.real + (1.0 + c) / 2.0 * r2]


def back_propagation_points(r, mass, dt, row, col, model):
    """The 3 x 3 phase-space points (element it comes from, branch) at which non_adiabatic_evolve_predict asks for the
    distribution (evolve.cpp:236-284), plus the intermediates the combination needs."""
    drc = Backward
    x0, p0 = r[:, 0], r[:, 1]
    couple = is_coupling(x0, p0, mass, dt, model).astype(float)
    x2, p1 = adiabatic_evolve(x0, p0, mass, dt / 2.0, drc, row, col, model)
    _, f01, _ = adiabatic_force(x2, model)
    f01 = f01 * couple
    p2 = [p1 + dt * float(drc) * n * f01 for n in OffDiagonalBranches]                     # :244-250
    x3 = [x2 + drc * (dt / 4.0) * p2[b] / mass for b in range(3)]                            # :251
    p3 = [[None] * 3 for _ in range(3)]
    x4 = [[None] * 3 for _ in range(3)]
    for b in range(3):
        f00, _, f11 = adiabatic_force(x3[b], model)
        f = (f00, f11)
        for (i, j), e in TRI.items():
            p3[e][b] = p2[b] + drc * (dt / 2.0) / 2.0 * (f[i] + f[j])                       # :253-281
            x4[e][b] = x3[b] + drc * (dt / 4.0) * p3[e][b] / mass                            # :283
    return dict(x0=x0, p0=p0, x2=x2, p1=p1, p2=p2, x4=x4, p3=p3)


def non_adiabatic_evolve_predict(r, density, mass, dt, distribution, row, col, model):
    """evolve.cpp:184-372 for every row of r (m, 2).  density: the exact element values (m,) complex, or None."""
    g = back_propagation_points(r, mass, dt, row, col, model)
    m = len(r)
    rp = [[None] * 3 for _ in range(3)]
    for (i, j), e in TRI.items():
        for b in range(3):
            if (i, j) == (row, col) and OffDiagonalBranches[b] == 0 and density is not None:
                rp[e][b] = np.asarray(density, dtype=complex).copy()                          # :309-313
            else:
                rp[e][b] = np.asarray(distribution(np.stack([g["x4"][e][b], g["p3"][e][b]], axis=1), i, j), dtype=complex)
    comb = [np.zeros(m, dtype=complex) for _ in range(3)]
    for b, n in enumerate(OffDiagonalBranches):
        rho = [rp[0][b], rp[1][b] * np.exp(calculate_omega0(g["x2"], g["x4"][1][b], Forward, 0, 1, model) * dt / 2 * 1j), rp[2][b]]  # :327-329
        rho = offdiagonal_rotation(rho, g["x2"], g["p2"][b], mass, dt / 2.0, model)          # :331-337
        if n == -1:                                                                           # :339-367
            v = (rho[0] + 2.0 * rho[1].real + rho[2]) / 4.0
            comb = [c + v for c in comb]
        elif n == 0:
            v = (rho[0] - rho[2]) / 2.0
            comb = [comb[0] + v, comb[1] + 1j * rho[1].imag, comb[2] - v]
        else:
            v = (rho[0] - 2.0 * rho[1].real + rho[2]) / 4.0
            comb = [comb[0] + v, comb[1] - v, comb[2] + v]
    comb = offdiagonal_rotation(comb, g["x2"], g["p1"], mass, dt / 2.0, model)               # :369-375
    result = comb[TRI[(row, col)]]
    if row != col:
        result = result * np.exp(calculate_omega0(g["x0"], g["x2"], Forward, 0, 1, model) * dt / 2.0 * 1j)
    return result


def evolve(density, mass, dt, distribution, model):
    """evolve.cpp:377-423.  density: {(iPES, jPES): (r (n, 2), rho (n,) complex)}; returns the same structure one tick later."""
    out = {}
    for (i, j) in TRI:
        r, rho = density[(i, j)]
        r, rho = np.asarray(r, dtype=float).reshape(-1, 2), np.asarray(rho, dtype=complex)
        if len(r) == 0:
            out[(i, j)] = (r.copy(), rho.copy())
            continue
        x0, p0 = r[:, 0], r[:, 1]
        couple = is_coupling(x0, p0, mass, dt, model)
        # coupled points: two half steps forward, then the exact density by back-propagation (:401-409)
        x2, p1 = adiabatic_evolve(x0, p0, mass, dt / 2, Forward, i, j, model)
        x4, p2 = adiabatic_evolve(x2, p1, mass, dt / 2, Forward, i, j, model)
        r_c = np.stack([x4, p2], axis=1)
        rho_c = non_adiabatic_evolve_predict(r_c, rho, mass, dt, distribution, i, j, model)
        # uncoupled points: one adiabatic step and a phase factor (:411-418)
        xa, pa = adiabatic_evolve(x0, p0, mass, dt, Forward, i, j, model)
        rho_a = np.asarray(distribution(r, i, j), dtype=complex) * np.exp(-calculate_omega0(x0, xa, Forward, i, j, model) * dt * 1j)
        out[(i, j)] = (np.where(couple[:, None], r_c, np.stack([xa, pa], axis=1)), np.where(couple, rho_c, rho_a))
    return out


# ---- counter-based random numbers: Philox4x32-10 -------------------------------------------------------------------------
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def new_point_predict(r, mass, dt, distribution, row, col, model):
    """evolve.cpp:425-443 for every row of r: the back-propagated prediction without an exact density where the point couples, else 0"""
    r = np.asarray(r, dtype=float).reshape(-1, 2)
    out = np.zeros(len(r), dtype=complex)
    cpl = is_coupling(r[:, 0], r[:, 1], mass, dt, model)
    if cpl.any():
        out[cpl] = non_adiabatic_evolve_predict(r[cpl], None, mass, dt, distribution, row, col, model)
    return out


def is_very_small(density, mass, dt, distribution, model):
    """evolve.cpp:445-478: an element without points is small when the new-point prediction is below 1e-5 in modulus at every point of
    element (0, 0); an element with points is not small"""
    out = {}
    test = np.asarray(density[(0, 0)][0], dtype=float)
    for e in [(0, 0), (1, 0), (1, 1)]:
        if len(density[e][0]) == 0:
            out[e] = bool(np.all(np.abs(new_point_predict(test, mass, dt, distribution, e[0], e[1], model)) ** 2 < 1e-10))
        else:
            out[e] = False
    return out


def philox4x32(counter, key):
    """counter: (..., 4) uint32, key: (2,) -> (..., 4) uint32 (10 rounds)"""
    c = np.array(counter, dtype=np.uint64) & 0xFFFFFFFF
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c[..., 0], _M1 * c[..., 2]
        hi0, lo0, hi1, lo1 = p0 >> 32, p0 & 0xFFFFFFFF, p1 >> 32, p1 & 0xFFFFFFFF
        c = np.stack([hi1 ^ c[..., 1] ^ k0, lo1, hi0 ^ c[..., 3] ^ k1, lo0], axis=-1) & 0xFFFFFFFF
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def philox_uniform(walker, step, seed):
    """Three uniforms in [0, 1) per (walker, step): 53-bit doubles from counter (walker, step, 0 | 1, 0), key = seed.
    Words (0,1) and (2,3) of block 0 give u0, u1; words (0,1) of block 1 give u2."""
    walker = np.asarray(walker, dtype=np.uint64)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = []
    for blk in (0, 1):
        ctr = np.stack([walker & 0xFFFFFFFF, np.full_like(walker, step), np.full_like(walker, blk), np.zeros_like(walker)], axis=-1)
        out.append(philox4x32(ctr, key).astype(np.uint64))
    to_unit = lambda hi, lo: (((hi << np.uint64(32)) | lo) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return to_unit(out[0][..., 0], out[0][..., 1]), to_unit(out[0][..., 2], out[0][..., 3]), to_unit(out[1][..., 0], out[1][..., 1])


def generate_markov_chain(num_steps, distribution, max_displacement, i, j, r, seed, want_chain=False):
    """mc.cpp:118-165 for all walkers at once (one batched distribution call per Monte-Carlo step); returns the last points
    and the acceptance ratio per walker (and, on request, the whole chains (num_steps + 1, n, 2)).  Uniform displacement in
    [-d, d) per dimension, accept when the new |rho| is larger or with probability new / old."""
    r = np.asarray(r, dtype=float).copy()
    walkers = np.arange(len(r))
    weight_old = np.abs(distribution(r, i, j))
    acc = np.zeros(len(r))
    chain = [r.copy()]
    for step in range(num_steps):
        u0, u1, u2 = philox_uniform(walkers, step, seed)
        r_new = r + np.stack([(2.0 * u0 - 1.0) * max_displacement, (2.0 * u1 - 1.0) * max_displacement], axis=1)
        weight_new = np.abs(distribution(r_new, i, j))
        with np.errstate(divide="ignore", invalid="ignore"):
            accept = (weight_new > weight_old) | (weight_new / weight_old > u2)
        r = np.where(accept[:, None], r_new, r)
        weight_old = np.where(accept, weight_new, weight_old)
        acc += accept
        if want_chain:
            chain.append(r.copy())
    if want_chain:
        return r, acc / max(1, num_steps), np.stack(chain)
    return r, acc / max(1, num_steps)
