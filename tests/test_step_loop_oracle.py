"""CPU: the numpy oracle of the step loop (oracle/evolve_oracle.py: Tully models, MQCLE back-propagation, Philox Metropolis)
against facts that do not depend on it: eigenvalues / Hellmann-Feynman forces of the diabatic matrix, the published
known-answer vectors of Philox4x32-10, and limits in which the back-propagation formula collapses."""
import numpy as np
import pytest

from oracle import evolve_oracle as E


@pytest.mark.parametrize("model", [E.SAC, E.DAC, E.ECR])
def test_adiabatic_representation(model):
    x = np.concatenate([np.linspace(-6, -0.05, 40), np.linspace(0.05, 6, 40)])
    v00, v01, v11 = E.diabatic_potential(x, model)
    e0, e1 = E.adiabatic_potential(x, model)
    for i in range(len(x)):
        w = np.linalg.eigvalsh(np.array([[v00[i], v01[i]], [v01[i], v11[i]]]))
        assert abs(w[0] - e0[i]) <= 1e-15 + 1e-13 * abs(w[0]) and abs(w[1] - e1[i]) <= 1e-15 + 1e-13 * abs(w[1])
    # diabatic force = -dV/dx; diagonal adiabatic force = -dE/dx (Hellmann-Feynman)
    h = 1e-6
    for k, (f, vp, vm) in enumerate(zip(E.diabatic_force(x, model), E.diabatic_potential(x + h, model), E.diabatic_potential(x - h, model))):
        assert np.abs(f + (vp - vm) / (2 * h)).max() <= 1e-8, k
    f00, f10, f11 = E.adiabatic_force(x, model)
    ep, em = E.adiabatic_potential(x + h, model), E.adiabatic_potential(x - h, model)
    assert np.abs(f00 + (ep[0] - em[0]) / (2 * h)).max() <= 1e-8 and np.abs(f11 + (ep[1] - em[1]) / (2 * h)).max() <= 1e-8
    assert np.allclose(np.abs(E.adiabatic_coupling_01(x, model)), np.abs(f10 / (e1 - e0)), rtol=1e-14)
    assert E.is_coupling(x, 10.0 + 0 * x, 2000.0, 1.0, model).all()  # CouplingCriterion = 0 with >=


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32 with 10 rounds"""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        assert tuple(int(v) for v in E.philox4x32(np.array(ctr), key)) == out
    u0, u1, u2 = E.philox_uniform(np.arange(4000), 3, 0x1234567890ABCDEF)
    for u in (u0, u1, u2):
        assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    assert abs(np.corrcoef(u0, u2)[0, 1]) < 0.05


def test_back_propagation_preserves_a_stationary_density():
    """If every element's distribution is constant in phase space and the exact densities equal those constants, the three
    branches recombine to the same constants wherever the coupling vanishes (SAC far from the crossing): the combination
    coefficients of evolve.cpp:339-367 sum to the identity."""
    rng = np.random.default_rng(1)
    r = np.stack([rng.uniform(9.0, 11.0, 50), rng.uniform(8.0, 12.0, 50)], axis=1)  # |x| ~ 10: V01 = 0.005 exp(-100) ~ 0
    const = {(0, 0): 0.7, (1, 0): 0.2 - 0.1j, (1, 1): 0.3}
    dist = lambda pts, i, j: np.full(len(pts), const[(i, j)], dtype=complex)
    dens = {e: (r.copy(), np.full(len(r), c, dtype=complex)) for e, c in const.items()}
    out = E.evolve(dens, 2000.0, 0.5, dist, E.SAC)
    for e, c in const.items():
        rn, rho = out[e]
        assert np.all(np.isfinite(rn)) and np.abs(rn - r).max() > 1e-4  # the points moved
        if e == (1, 0):  # the coherence picks up exp(-i (E1 - E0) dt) relative to a constant field: compare moduli
            assert np.abs(np.abs(rho) - abs(c)).max() <= 1e-9
        else:
            assert np.abs(rho - c).max() <= 1e-9


def test_metropolis_samples_the_target():
    """chains on a Gaussian target |rho| reproduce its mean and variance; the acceptance ratio falls with the step size"""
    target = lambda pts, i, j: np.exp(-0.5 * (((pts[:, 0] + 10.0) / 0.7) ** 2 + ((pts[:, 1] - 14.0) / 0.5) ** 2)) + 0j
    start = np.tile([-10.0, 14.0], (3000, 1))
    r, acc = E.generate_markov_chain(60, target, 0.8, 0, 0, start, seed=42)
    assert abs(r[:, 0].mean() + 10.0) < 0.05 and abs(r[:, 1].mean() - 14.0) < 0.05
    assert abs(r[:, 0].std() - 0.7) < 0.06 and abs(r[:, 1].std() - 0.5) < 0.05
    _, acc_big = E.generate_markov_chain(60, target, 3.0, 0, 0, start, seed=42)
    assert 0.15 < acc.mean() < 0.9 and acc_big.mean() < acc.mean()


def test_chain_autocorrelation_against_the_double_loop():
    """steploop.chain_autocorrelation (FFT) == the literal double loop of mc.cpp:205-226"""
    from gaussian_process_liouville_equation_amd import steploop
    rng = np.random.default_rng(2)
    whole = np.cumsum(rng.normal(size=(41, 5, 2)), axis=0)  # 5 random walks of 41 points
    got = steploop.chain_autocorrelation(whole)
    n = whole.shape[0]
    ref = np.zeros(n // 2)
    for w in range(whole.shape[1]):
        ave = whole[:, w].mean(axis=0)
        for j in range(n // 2):
            ref[j] += sum(np.dot(whole[i, w] - ave, whole[i + j, w] - ave) for i in range(n - j)) / (n - j)
    ref /= whole.shape[1]
    assert len(got) == n // 2 and np.abs(got - ref).max() <= 1e-11 * np.abs(ref).max()


def test_monte_carlo_tuning_on_an_analytic_density():
    """mc.cpp:167-372 on a Gaussian density (no GP involved: distribution = the exact density): the chosen displacement is the largest
    of the table whose acceptance lies inside the band, the chain length is a lag inside the measured window, and
    element_monte_carlo returns points with the density evaluated there."""
    from gaussian_process_liouville_equation_amd import steploop as S
    rng = np.random.default_rng(3)
    sig = np.array([0.7086, 0.7056])
    rho = lambda r: np.exp(-0.5 * (((r - [-10.0, 14.112]) / sig) ** 2).sum(axis=1)) / (2 * np.pi * sig.prod())
    dist = lambda r, i, j: rho(r).astype(complex)
    chain = lambda n, d, r, seed, want_chain=False: E.generate_markov_chain(n, dist, d, 0, 0, r, seed, want_chain)
    r0 = np.array([-10.0, 14.112]) + sig * rng.normal(size=(60, 2))
    mc, seeds = S.MCParameters(), S._Seeds(99)
    S.acceptance_optimize_displacement(mc, chain, r0, seeds, MaxNOMC=300)
    d = mc.get_max_displacement()
    ratio = lambda dd: float(np.mean(chain(300, dd, r0, 12345)[1]))
    assert d in S.PossibleDisplacement and S.MinAcceptRatio < ratio(d) < S.MaxAcceptRatio
    bigger = [x for x in S.PossibleDisplacement if x > d]
    assert all(not (S.MinAcceptRatio < ratio(x) < S.MaxAcceptRatio) or abs(ratio(x) - S.MinAcceptRatio) < 0.03 for x in bigger)  # (other seed: band edge)
    S.autocorrelation_optimize_steps(mc, chain, r0, seeds, MaxNOMC=400)
    assert 1 <= mc.get_num_MC_steps() < 200
    r1, rho1 = S.element_monte_carlo((r0, rho(r0).astype(complex)), S.MCParameters(), lambda n, dd, r, s, w=False: chain(min(n, 300), dd, r, s, w),
                                     lambda r: rho(r), S._Seeds(5))
    assert r1.shape == r0.shape and np.allclose(rho1, rho(r1)) and np.abs(r1 - r0).max() > 0


def test_extra_points_and_host_chain():
    """steploop.generate_element_extra_points (mc.cpp:59-98) and the host-driven Metropolis chain for elements without a fit"""
    from gaussian_process_liouville_equation_amd import steploop as S
    rng = np.random.default_rng(4)
    sig = np.array([0.7086, 0.7056])
    rho = lambda r: np.exp(-0.5 * (((np.asarray(r) - [-10.0, 14.112]) / sig) ** 2).sum(axis=1)) / (2 * np.pi * sig.prod())
    r0 = np.array([-10.0, 14.112]) + sig * rng.normal(size=(50, 2))
    new, val = S.generate_element_extra_points((r0, rho(r0)), 130, rho, np.random.default_rng(9))
    assert new.shape == (130, 2) and np.allclose(val, rho(new))
    std = r0.std(axis=0)
    dev = new - r0[np.arange(130) % 50]
    assert np.all(np.abs(dev.std(axis=0) / std - 1.0) < 0.35)
    chain = S.host_chain(rho, np.random.default_rng(10))
    last, acc, whole = chain(200, 0.5, r0, None, True)
    assert whole.shape == (201, 50, 2) and np.array_equal(whole[0], r0) and np.array_equal(whole[-1], last)
    assert 0.2 < acc.mean() < 0.95 and np.all(np.abs(np.diff(whole, axis=0)).max(axis=(1, 2)) <= 0.5)
    # the chain samples |rho|: after many steps the walkers' spread is that of the density
    last, _ = chain(1500, 0.8, np.tile([[-10.0, 14.112]], (400, 1)))
    assert np.all(np.abs(last.std(axis=0) / sig - 1.0) < 0.2)


# ---- N-level step loop (oracle/evolve_oracle_n.py): the derived generalisation of evolve.cpp:184-372 ------------------------------------------

def _smooth_distribution(pts, i, j):
    x, p = pts[:, 0], pts[:, 1]
    g = np.exp(-0.5 * (((x + 1.5) / 0.8) ** 2 + ((p - 14.0) / 0.9) ** 2))
    if i == j:
        return g * (0.6, 0.3, 0.1)[i] + 0j
    return g * 0.2 * np.exp(0.4j * (x + 1.5) + 0.1j * (i + j))


@pytest.mark.parametrize("model", [E.SAC, E.DAC, E.ECR])
def test_n_level_oracle_reduces_to_the_two_level_reference_code(model):
    """The projector form of the back-propagation (P_a rho P_b shifted by (lambda_a + lambda_b) / 2; rho <- O rho O^T with O = exp(-v D t))
    at N = 2 IS the reference's three-branch code (evolve.cpp:184-372, restated in oracle/evolve_oracle.py): same values to rounding for every
    element, with the exact density on the zero-shift branch and without it, and the same adiabatic quantities."""
    from oracle import evolve_oracle_n as EN
    rng = np.random.default_rng(model)
    r = rng.normal([-1.5, 14.0], [0.8, 0.9], size=(60, 2))
    En, _, Fn, NACn = EN.adiabatic(r[:, 0], model, 2)
    e0, e1 = E.adiabatic_potential(r[:, 0], model)
    f00, f10, f11 = E.adiabatic_force(r[:, 0], model)
    assert np.abs(En[:, 0] - e0).max() <= 1e-16 and np.abs(En[:, 1] - e1).max() <= 1e-16
    assert max(np.abs(Fn[:, 0, 0] - f00).max(), np.abs(Fn[:, 1, 0] - f10).max(), np.abs(Fn[:, 1, 1] - f11).max()) <= 1e-14
    assert np.abs(NACn[:, 0, 1] - E.adiabatic_coupling_01(r[:, 0], model)).max() <= 1e-12 * np.abs(NACn).max()
    for (i, j) in [(0, 0), (1, 0), (1, 1)]:
        rho = _smooth_distribution(r, i, j) * 1.01
        for dens in (rho, None):
            a = E.non_adiabatic_evolve_predict(r, dens, 2000.0, 1.0, _smooth_distribution, i, j, model)
            b = EN.non_adiabatic_evolve_predict(r, dens, 2000.0, 1.0, _smooth_distribution, i, j, model, 2)
            assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max(), (i, j)
    dens = {e: (r, _smooth_distribution(r, *e)) for e in [(0, 0), (1, 0), (1, 1)]}
    a, b = E.evolve(dens, 2000.0, 1.0, _smooth_distribution, model), EN.evolve(dens, 2000.0, 1.0, _smooth_distribution, model, 2)
    for e in dens:
        assert np.abs(a[e][0] - b[e][0]).max() <= 1e-13 * np.abs(a[e][0]).max() and np.abs(a[e][1] - b[e][1]).max() <= 1e-13 * np.abs(a[e][1]).max()


@pytest.mark.parametrize("model", [E.SAC, E.DAC, E.ECR])
def test_three_levels_with_a_spectator_reproduce_the_two_level_block(model):
    """pes.cpp compiled for NumPES = 3 leaves the third diabat uncoupled at V = 0: the two Tully states evolve among themselves exactly as in the
    two-level code, the spectator's population moves on its own flat surface, and its coherences only turn with the 0-1 rotation"""
    from oracle import evolve_oracle_n as EN
    rng = np.random.default_rng(10 + model)
    r = rng.normal([-1.5, 14.0], [0.8, 0.9], size=(40, 2))
    E3 = EN.adiabatic(r[:, 0], model, 3)[0]
    E2 = EN.adiabatic(r[:, 0], model, 2)[0]
    spect = int(np.argmin(np.abs(E3[0])))            # the adiabatic index the spectator (E = 0 exactly) has at these positions
    assert np.all(E3[:, spect] == 0.0)
    tully = [k for k in range(3) if k != spect]
    assert np.abs(E3[:, tully] - E2).max() <= 1e-16
    emb = {0: tully[0], 1: tully[1]}

    def dist3(pts, i, j):  # the two-level distribution on the Tully block, nothing on the spectator
        inv = {v: k for k, v in emb.items()}
        if i in inv and j in inv:
            a, b = inv[i], inv[j]
            v = _smooth_distribution(pts, max(a, b), min(a, b))
            return v if a >= b else np.conj(v)
        return np.zeros(len(pts), dtype=complex)

    for (i, j) in [(0, 0), (1, 0), (1, 1)]:
        a = E.non_adiabatic_evolve_predict(r, None, 2000.0, 1.0, _smooth_distribution, i, j, model)
        k, l = emb[i], emb[j]
        b = EN.non_adiabatic_evolve_predict(r, None, 2000.0, 1.0, dist3, max(k, l), min(k, l), model, 3)
        if k < l:
            b = np.conj(b)
        assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max(), (i, j)
    # nothing leaks onto the spectator
    assert np.abs(EN.non_adiabatic_evolve_predict(r, None, 2000.0, 1.0, dist3, spect, spect, model, 3)).max() <= 1e-18


def test_three_state_model_invariants():
    """TSAC (the library's own three-level model): avoided crossings of sqrt(2) C, Hellmann-Feynman forces, NAC from the force matrix,
    continuous eigenvector signs; the back-propagation maps a multiple of the identity onto itself (unitary mixing + translations), keeps
    Hermitian structure (a real symmetric, momentum-independent density stays real symmetric) and is exact for a flat density"""
    from oracle import evolve_oracle_n as EN
    x = np.linspace(-12, 12, 4801)
    En, Cn, Fn, NACn = EN.adiabatic(x, EN.TSAC, 3)
    assert abs((En[:, 1] - En[:, 0]).min() - np.sqrt(2) * EN.TSAC_C) < 1e-6 and abs((En[:, 2] - En[:, 1]).min() - np.sqrt(2) * EN.TSAC_C) < 1e-6
    h = 1e-6
    dE = (EN.adiabatic(x + h, EN.TSAC, 3)[0] - EN.adiabatic(x - h, EN.TSAC, 3)[0]) / (2 * h)
    assert np.abs(np.einsum("mkk->mk", Fn) + dE).max() <= 1e-8
    assert np.abs(np.diff(Cn, axis=0)).max() < 0.02              # no sign flips along x
    dC = (EN.adiabatic(x + h, EN.TSAC, 3)[1] - EN.adiabatic(x - h, EN.TSAC, 3)[1]) / (2 * h)
    d_num = np.einsum("mik,mil->mkl", Cn, dC)                      # <k | d/dx | l>
    assert np.abs(d_num - NACn).max() <= 1e-6 * np.abs(NACn).max()  # pes.cpp:137-155's F / (E_j - E_k) is the derivative coupling
    rng = np.random.default_rng(3)
    r = rng.normal([0.3, 14.0], [0.8, 0.9], size=(25, 2))
    ident = lambda pts, i, j: (np.ones(len(pts)) if i == j else np.zeros(len(pts))) + 0j
    for (i, j) in EN.elements(3):
        v = EN.non_adiabatic_evolve_predict(r, None, 2000.0, 1.0, ident, i, j, EN.TSAC, 3)
        assert np.abs(v - (1.0 if i == j else 0.0)).max() <= 1e-13
    sym = np.array([[0.5, 0.1, -0.05], [0.1, 0.3, 0.07], [-0.05, 0.07, 0.2]])
    flat = lambda pts, i, j: np.full(len(pts), sym[i, j], dtype=complex)
    # a constant real symmetric matrix: the result stays Hermitian (real diagonal); its trace moves only by the branch-dependent phases
    # (E_k - E_l) t / 2 of the off-diagonal entries, second order in the step — the same holds for the two-level code
    vals = {e: EN.non_adiabatic_evolve_predict(r, None, 2000.0, 1.0, flat, e[0], e[1], EN.TSAC, 3) for e in EN.elements(3)}
    assert np.abs(sum(vals[(k, k)] for k in range(3)) - np.trace(sym)).max() <= 1e-4
    assert max(np.abs(vals[(k, k)].imag).max() for k in range(3)) <= 1e-15
