"""GPU: the device step loop (gple_pes_adiabatic, gple_evolve, gple_markov_chain) against the numpy oracle
of evolve.cpp / pes.cpp / mc.cpp (oracle/evolve_oracle.py), whose distribution function is the C++ oracle's predictor."""
import numpy as np
import pytest

from gaussian_process_liouville_equation_amd import kernels as K
from oracle import evolve_oracle as E
from tests import parity

pytestmark = pytest.mark.gpu
MASS, DT = 2000.0, 1.0
TH, THC = [1.0, 0.7086, 0.7056, 1e-2], [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05]


@pytest.mark.parametrize("model", [E.SAC, E.DAC, E.ECR])
def test_pes_against_oracle(gpu, model):
    x = np.concatenate([np.linspace(-12, -0.01, 200), np.linspace(0.01, 12, 200), [0.37, -2.5]])
    got = gpu.pes_adiabatic(model, x)
    e0, e1 = E.adiabatic_potential(x, model)
    f00, f10, f11 = E.adiabatic_force(x, model)
    ref = np.stack([e0, e1, f00, f10, f11, E.adiabatic_coupling_01(x, model)], axis=1)
    scale = np.abs(ref).max(axis=0)
    assert np.all(np.abs(got - ref) <= 1e-12 * scale + 1e-300)


def _case(N, seed, x_centre=-1.5):
    """a wave packet approaching the crossing: samples of all three elements with their (exact-like) densities"""
    rng = np.random.Generator(np.random.PCG64(seed))
    dens = {}
    for e, (i, j) in enumerate(K.element_order(2)):
        r = rng.normal([x_centre, 14.0], [0.7086, 0.7056], size=(N + 7 * e, 2))
        g = np.exp(-0.5 * (((r[:, 0] - x_centre) / 0.7086) ** 2 + ((r[:, 1] - 14.0) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
        rho = g * (0.6, 0.3 * np.exp(0.4j * (r[:, 0] - x_centre)), 0.4)[e]
        dens[(i, j)] = (r, rho.astype(complex))
    return dens


def _fits(api, dens):
    return [api.real_fit(TH, *dens[(0, 0)], 0), api.complex_fit(THC, *dens[(1, 0)], 0), api.real_fit(TH, dens[(1, 1)][0], dens[(1, 1)][1], 0)]


def _oracle_distribution(oracle, fits):
    def distribution(pts, i, j):
        f = fits[i * (i + 1) // 2 + j]
        if f is None:
            return np.zeros(len(pts), dtype=complex)
        pred = oracle.complex_predict if i != j else oracle.real_predict
        return np.asarray(pred(f, pts, want=("cutoff",))["cutoff"], dtype=complex)
    return distribution


@pytest.mark.parametrize("model,N", [(E.DAC, 150), (E.SAC, 90), (E.ECR, 60)])
def test_evolve_tick_against_oracle(gpu, oracle, model, N):
    dens = _case(N, 20240607 + N)
    out = gpu.evolve(_fits(gpu, dens), model, MASS, DT, dens)
    ref = E.evolve(dens, MASS, DT, _oracle_distribution(oracle, _fits(oracle, dens)), model)
    for e in dens:
        assert out[e][0].shape == ref[e][0].shape
        assert np.abs(out[e][0] - ref[e][0]).max() <= 1e-12 * np.abs(ref[e][0]).max()   # coordinates: pure propagation
        scale = max(np.abs(ref[e][1]).max(), np.abs(dens[e][1]).max())
        assert np.abs(out[e][1] - ref[e][1]).max() <= 1e-8 * scale, e                     # densities: through two GP predicts
        assert np.abs(out[e][1] - dens[e][1]).max() > 1e-6 * scale                        # the tick did something
    # the diagonal elements of a density matrix stay real under the back-propagation of real diagonal predictions
    assert np.abs(out[(0, 0)][1].imag).max() <= 1e-9 * np.abs(out[(0, 0)][1]).max()


def test_evolve_with_unpopulated_elements(gpu, oracle):
    """only rho_00 populated (the reference's initial state, main.cpp:38): absent elements predict 0 (main.cpp:86-88) and
    empty point sets stay empty"""
    dens = _case(80, 5)
    dens[(1, 0)] = (np.zeros((0, 2)), np.zeros(0, complex))
    dens[(1, 1)] = (np.zeros((0, 2)), np.zeros(0, complex))
    fits_g = [gpu.real_fit(TH, *dens[(0, 0)], 0), None, None]
    fits_o = [oracle.real_fit(TH, *dens[(0, 0)], 0), None, None]
    out = gpu.evolve(fits_g, E.DAC, MASS, DT, dens)
    ref = E.evolve(dens, MASS, DT, _oracle_distribution(oracle, fits_o), E.DAC)
    assert len(out[(1, 0)][0]) == 0 and len(out[(1, 1)][1]) == 0
    assert np.abs(out[(0, 0)][0] - ref[(0, 0)][0]).max() <= 1e-12 * np.abs(ref[(0, 0)][0]).max()
    assert np.abs(out[(0, 0)][1] - ref[(0, 0)][1]).max() <= 1e-8 * np.abs(ref[(0, 0)][1]).max()


def test_metropolis_chains_against_oracle(gpu, oracle):
    """same Philox stream, same decisions: every walker ends where the oracle's walker ends"""
    dens = _case(200, 11, x_centre=-10.0)
    X, rho = dens[(0, 0)]
    fg, fo = gpu.real_fit(TH, X, rho, 0), oracle.real_fit(TH, X, rho, 0)
    start = X[:120]
    rg, ag = gpu.markov_chain(fg, 25, 0.3, 0xC0FFEE1234, start)
    ro, ao = E.generate_markov_chain(25, lambda pts, i, j: oracle.real_predict(fo, pts, want=("cutoff",))["cutoff"], 0.3, 0, 0, start, 0xC0FFEE1234)
    same = np.abs(rg - ro).max(axis=1) <= 1e-12
    assert same.mean() >= 0.98, same.mean()  # a decision can flip only where new/old sits within rounding of the random number
    assert np.abs(ag[same] - ao[same]).max() <= 1e-15 and 0.05 < ag.mean() < 0.95
    # complex element: weights are the modulus of the complex cut-off prediction
    Xc, rc = dens[(1, 0)]
    fgc, foc = gpu.complex_fit(THC, Xc, rc, 0), oracle.complex_fit(THC, Xc, rc, 0)
    rg, ag = gpu.markov_chain(fgc, 10, 0.2, 7, Xc[:64])
    ro, ao = E.generate_markov_chain(10, lambda pts, i, j: oracle.complex_predict(foc, pts, want=("cutoff",))["cutoff"], 0.2, 1, 0, Xc[:64], 7)
    assert (np.abs(rg - ro).max(axis=1) <= 1e-12).mean() >= 0.95
    # no steps: nothing moves
    r0, a0 = gpu.markov_chain(fg, 0, 0.3, 1, start)
    assert np.array_equal(r0, start) and np.all(a0 == 0)


def test_tick_mirror_and_phase_files(gpu, oracle):
    """steploop.tick (main.cpp:143-176: evolve density and extra points, refit) keeps the packet's population, and the phase /
    variance files written from the new kernels (output.cpp:180-232, row N4) match the oracle's line by line"""
    import io
    from gaussian_process_liouville_equation_amd import output, steploop
    dens = _case(120, 77)
    extra = _case(200, 78)
    params = {(0, 0): TH, (1, 0): THC, (1, 1): TH}
    k0 = K.TrainingKernels(params, K.construct_training_sets(dens), True, True, False, api=gpu)
    pop0 = k0.calculate_population()
    d1, x1, k1 = steploop.tick(dens, extra, params, MASS, DT, k0, steploop.DAC)
    assert all(len(d1[e][0]) == len(dens[e][0]) and len(x1[e][0]) == len(extra[e][0]) for e in dens)
    assert abs(k1.calculate_population() - pop0) <= 0.05 * abs(pop0)  # one tick of unitary dynamics on a fitted density
    # the same tick on the oracle side, then both sets of files
    fo = _fits(oracle, dens)
    ref = E.evolve(dens, MASS, DT, _oracle_distribution(oracle, fo), E.DAC)
    ko = K.TrainingKernels(params, K.construct_training_sets(ref), True, True, False, api=oracle)
    grid = np.stack(np.meshgrid(np.linspace(-3.5, 0.5, 9), np.linspace(12.5, 15.5, 7), indexing="ij"), axis=-1).reshape(-1, 2)
    files = {}
    for name, ks in (("gpu", k1), ("oracle", ko)):
        ph, va = io.StringIO(), io.StringIO()
        output.output_phase(ph, va, ks, grid)
        files[name] = (ph.getvalue(), va.getvalue())
    for a, b in zip(files["gpu"], files["oracle"]):
        la, lb = a.split("\n"), b.split("\n")
        assert len(la) == len(lb)
        for x, y in zip(la, lb):
            assert len(x.split()) == len(y.split())
            if x.strip():
                assert np.allclose(np.array(x.split(), float), np.array(y.split(), float), rtol=2e-4, atol=1e-7)  # %g keeps 6 digits


def test_tick_at_the_reference_initial_complex_parameters(gpu, oracle):
    """The reference STARTS its complex kernel at sR = sI = 1, lR = lI = sigma (opt.cpp:306-332).  There K~ = 2i K_C with K_C = K_R = K_I: the
    pseudo-covariance equals the covariance in modulus, the widely-linear model degenerates (it can only represent labels of one fixed phase),
    and one tick against that fit loses a good part of |rho_10| — on the reference's own algorithm just as here (round 3 moved
    test_tick_at_c5_size to distinct sub-kernels for that reason).  What happens at the reference's own starting point stays pinned by this
    test: one tick at exactly those parameters, HIP against the oracle (C++ restatement as predictor, numpy restatement of evolve.cpp).
    What was measured (N = 300, gpurun_out/r04/item46_tests.log): the back-propagated densities agree with the oracle at 1e-9 of their scale,
    like every other tick test — the PREDICTOR is fine at this point.  The purity SCALAR is not well conditioned there: with R = I = C the
    purity form (complex_kernel.cpp:357-377) collapses to 4 (Re v + Im v)^T K' (Re v + Im v), and the weights of a model that cannot tell
    the phase have Re v ~ -Im v, each ~ 1 / sn^2 large: the sum keeps three to four digits less than v.  HIP 0.61042, oracle 0.60997
    (7e-4 apart: two routes to v — real embedding there, complex LDLT + Schur complement here — round differently at cond eps), and
    0.54857 / 0.54826 after the tick: the same 10 % drop on both sides (ratios 0.89868 / 0.89883).  So: densities at 1e-7, purity at 2e-3,
    drift at 1e-3 — and the drop itself is asserted to be there (6-14 %): it is what the reference does at its starting point."""
    from gaussian_process_liouville_equation_amd import steploop
    TC0 = [1.0, 1.0, 0.7086, 0.7056, 1.0, 0.7086, 0.7056, 1e-2]  # InitialComplexParameter, opt.cpp:306-332
    dens, extra = _case(300, 501), _case(120, 502)
    params = {(0, 0): TH, (1, 0): TC0, (1, 1): TH}
    kg = K.TrainingKernels(params, K.construct_training_sets(dens), True, True, False, api=gpu)
    ko = K.TrainingKernels(params, K.construct_training_sets(dens), True, True, False, api=oracle)
    pur0_g, pur0_o = kg.calculate_purity(), ko.calculate_purity()
    assert abs(kg.calculate_population() - ko.calculate_population()) <= 1e-7  # the diagonal elements: well conditioned
    d1, _, k1 = steploop.tick(dens, extra, params, MASS, DT, kg, steploop.DAC)
    fo = [oracle.real_fit(TH, *dens[(0, 0)], 0), oracle.complex_fit(TC0, *dens[(1, 0)], 0), oracle.real_fit(TH, *dens[(1, 1)], 0)]
    ref = E.evolve(dens, MASS, DT, _oracle_distribution(oracle, fo), E.DAC)
    worst = {}
    for e in dens:
        assert np.abs(d1[e][0] - ref[e][0]).max() <= 1e-12 * np.abs(ref[e][0]).max()
        scale = max(np.abs(ref[e][1]).max(), np.abs(dens[e][1]).max())
        worst[e] = np.abs(d1[e][1] - ref[e][1]).max() / scale
    k1o = K.TrainingKernels(params, K.construct_training_sets(ref), True, True, False, api=oracle)
    pur1_g, pur1_o = k1.calculate_purity(), k1o.calculate_purity()
    print(f"initial complex parameters, N = 300: purity {pur0_g:.5f} (HIP) / {pur0_o:.5f} (oracle) -> after one tick {pur1_g:.5f} / {pur1_o:.5f}; "
          f"element-wise HIP vs oracle after the tick: " + ", ".join(f"{e}: {w:.1e}" for e, w in worst.items()))
    assert abs(pur0_g - pur0_o) <= 2e-3 * pur0_o
    assert all(w <= 1e-7 for w in worst.values()), worst
    drift_g, drift_o = pur1_g / pur0_g - 1.0, pur1_o / pur0_o - 1.0
    assert abs(drift_g - drift_o) <= 1e-3, (drift_g, drift_o)  # the same drift on both sides ...
    assert -0.14 <= drift_o <= -0.06, drift_o                    # ... and it is the reference algorithm's own loss at its initial parameters


def test_average_line_on_the_device_against_oracle(gpu, oracle):
    """ave.txt (output.cpp:24-118): the kernels' analytic averages from the HIP fits and the surface energies from
    gple_pes_adiabatic, against the same line computed with the oracle fits and the numpy Tully model"""
    import io
    from gaussian_process_liouville_equation_amd import output
    dens = _case(150, 909)
    params = {(0, 0): TH, (1, 0): THC, (1, 1): TH}
    sets = K.construct_training_sets(dens)
    lines = []
    for api, pot in ((gpu, output.tully_potential(gpu, E.DAC)), (oracle, lambda x, i: E.adiabatic_potential(x, E.DAC)[i])):
        f = io.StringIO()
        output.output_average(f, K.TrainingKernels(params, sets, True, True, False, api=api), dens, MASS, 1.0, potential=pot)
        lines.append(np.array(f.getvalue().split(), dtype=float))
    a, b = lines
    assert a.shape == b.shape == (34,)
    both = ~(np.isnan(a) | np.isnan(b))
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(a[both], b[both], rtol=2e-5, atol=1e-9)  # %g keeps 6 digits


def test_monte_carlo_selection_on_the_device(gpu, oracle):
    """steploop.monte_carlo_selection (mc.cpp:333-403: displacement and chain-length tuning, then the walk) on the device chains
    against the same host logic driven by the oracle's chains and predictor: same tuned parameters; the re-selected points agree
    wherever no accept decision sat within rounding of its random number; densities are the fit's cut-off prediction there."""
    from gaussian_process_liouville_equation_amd import steploop as S
    dens = _case(96, 31)
    dens = {(0, 0): dens[(0, 0)], (1, 0): (np.zeros((0, 2)), np.zeros(0, dtype=complex)), (1, 1): (np.zeros((0, 2)), np.zeros(0, dtype=complex))}
    params = {(0, 0): TH, (1, 0): THC, (1, 1): TH}
    kg = K.TrainingKernels(params, K.construct_training_sets(dens), True, True, False, api=gpu)
    mcg = {e: S.MCParameters() for e in dens}
    # shorter tuning runs than the reference's 1000 / 2000 steps keep the oracle side of this test at seconds
    orig = (S.acceptance_optimize_displacement.__defaults__, S.autocorrelation_optimize_steps.__defaults__)
    S.acceptance_optimize_displacement.__defaults__, S.autocorrelation_optimize_steps.__defaults__ = (120,), (160,)
    try:
        new = S.monte_carlo_selection(dens, mcg, kg, seed=2024)
        X, rho = dens[(0, 0)]
        fo = oracle.real_fit(TH, X, rho, 0)
        dist = lambda pts, i=0, j=0: oracle.real_predict(fo, pts, want=("cutoff",))["cutoff"]
        chain = lambda n, d, r, seed, want_chain=False: E.generate_markov_chain(n, dist, d, 0, 0, r, seed, want_chain)
        mco = S.MCParameters()
        ro, rhoo = S.element_monte_carlo((X, rho), mco, chain, lambda r: dist(r), S._Seeds(2024))
    finally:
        S.acceptance_optimize_displacement.__defaults__, S.autocorrelation_optimize_steps.__defaults__ = orig
    assert mcg[(0, 0)].get_max_displacement() == mco.get_max_displacement()
    assert abs(mcg[(0, 0)].get_num_MC_steps() - mco.get_num_MC_steps()) <= 2
    rg, rhog = new[(0, 0)]
    assert rg.shape == X.shape and len(new[(1, 0)][0]) == 0
    if mcg[(0, 0)].get_num_MC_steps() == mco.get_num_MC_steps():
        assert (np.abs(rg - ro).max(axis=1) <= 1e-10).mean() >= 0.9
    fg = kg(0)._fit
    assert np.abs(rhog.real - gpu.real_predict(fg, rg, want=("cutoff",))["cutoff"]).max() <= 1e-12 * np.abs(rhog).max() and np.all(rhog.imag == 0)


def test_new_point_predict_and_is_very_small(gpu, oracle):
    """gple_evolve with GPLE_EVOLVE_NEW_POINTS (evolve.cpp:425-443: back-propagated prediction at points that are not moved, no exact
    density, 0 where uncoupled) and is_very_small (:445-478) against the oracle, for every target element"""
    from gaussian_process_liouville_equation_amd import steploop as S
    dens = _case(100, 41)
    kg = K.TrainingKernels({(0, 0): TH, (1, 0): THC, (1, 1): TH}, K.construct_training_sets(dens), True, True, False, api=gpu)
    fo = _fits(oracle, dens)
    dist = _oracle_distribution(oracle, fo)
    rng = np.random.default_rng(5)
    pts = dens[(0, 0)][0][:60] + rng.normal(0, 0.1, (60, 2))
    pts[:5, 0] = 25.0  # far outside the coupling region of the model: uncoupled points -> exactly 0
    for e in [(0, 0), (1, 0), (1, 1)]:
        got = S.new_point_predict(pts, e[0], e[1], MASS, DT, kg, S.DAC)
        ref = E.new_point_predict(pts, MASS, DT, dist, e[0], e[1], E.DAC)
        scale = max(np.abs(ref).max(), 1e-30)
        assert np.abs(got - ref).max() <= 1e-9 * scale, (e, np.abs(got - ref).max(), scale)
        cpl = E.is_coupling(pts[:, 0], pts[:, 1], MASS, DT, E.DAC)
        assert np.all(got[~cpl] == 0)
    # an element without points: small or not as the oracle says; elements with points are never small
    only00 = {(0, 0): dens[(0, 0)], (1, 0): (np.zeros((0, 2)), np.zeros(0, dtype=complex)), (1, 1): (np.zeros((0, 2)), np.zeros(0, dtype=complex))}
    k00 = K.TrainingKernels({(0, 0): TH, (1, 0): THC, (1, 1): TH}, K.construct_training_sets(only00), True, True, False, api=gpu)
    fo00 = [oracle.real_fit(TH, *only00[(0, 0)], 0), None, None]
    small_g = S.is_very_small(only00, MASS, DT, k00, S.DAC)
    small_o = E.is_very_small(only00, MASS, DT, _oracle_distribution(oracle, fo00), E.DAC)
    assert small_g == small_o and small_g[(0, 0)] is False


def test_new_element_point_selection_on_the_device(gpu):
    """mc.cpp:405-537 on the device predictors: the newly populated element gets NumPoints points whose densities are the new-point
    prediction there, and NumExtraPoints extra points; an element that became small is emptied; nothing changes without a change"""
    from gaussian_process_liouville_equation_amd import steploop as S
    dens = _case(64, 43)
    extra = _case(96, 44)
    empty = (np.zeros((0, 2)), np.zeros(0, dtype=complex))
    d0 = {(0, 0): dens[(0, 0)], (1, 0): dens[(1, 0)], (1, 1): empty}
    x0 = {(0, 0): extra[(0, 0)], (1, 0): extra[(1, 0)], (1, 1): empty}
    params = {(0, 0): TH, (1, 0): THC, (1, 1): TH}
    kg = K.TrainingKernels(params, K.construct_training_sets(d0), True, True, False, api=gpu)
    mc = {e: S.MCParameters() for e in d0}
    old = {(0, 0): False, (1, 0): False, (1, 1): True}
    same_d, same_x = S.new_element_point_selection(d0, x0, old, dict(old), mc, kg, MASS, DT, np.random.default_rng(1))
    assert same_d is d0 and same_x is x0
    new = {(0, 0): False, (1, 0): True, (1, 1): False}  # (1,1) appears, (1,0) disappears
    orig = (S.acceptance_optimize_displacement.__defaults__, S.autocorrelation_optimize_steps.__defaults__)
    S.acceptance_optimize_displacement.__defaults__, S.autocorrelation_optimize_steps.__defaults__ = (12,), (24,)  # short tuning runs
    try:
        d1, x1 = S.new_element_point_selection(d0, x0, old, new, mc, kg, MASS, DT, np.random.default_rng(2))
    finally:
        S.acceptance_optimize_displacement.__defaults__, S.autocorrelation_optimize_steps.__defaults__ = orig
    assert len(d1[(1, 0)][0]) == 0 and len(x1[(1, 0)][0]) == 0
    r11, rho11 = d1[(1, 1)]
    assert r11.shape == (64, 2) and x1[(1, 1)][0].shape == (96, 2)
    ref = S.new_point_predict(r11, 1, 1, MASS, DT, kg, S.DAC)
    assert np.abs(rho11 - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-30)
    assert np.array_equal(d1[(0, 0)][0], d0[(0, 0)][0])


def test_main_tick_rules(gpu):
    """steploop.main_tick (main.cpp:136-186): without an element change and off the re-optimisation beat the kernels are only refitted
    on the evolved points; on the beat (iTick % ReoptFreq == 0) the optimiser runs and fresh extra points are drawn"""
    from gaussian_process_liouville_equation_amd import optimization as O, steploop as S
    rng = np.random.default_rng(3)
    sig, x0, p0 = np.array([0.7086, 0.7056]), -10.0, 14.112  # far from the crossing: nothing couples, no element appears
    wig = lambda r: np.exp(-0.5 * (((r - [x0, p0]) / sig) ** 2).sum(axis=1)) / (2 * np.pi * sig.prod())
    r, re = rng.normal(size=(80, 2)) * sig + [x0, p0], rng.normal(size=(160, 2)) * sig + [x0, p0]
    empty = (np.zeros((0, 2)), np.zeros(0, dtype=complex))
    dens = {(0, 0): (r, wig(r).astype(complex)), (1, 0): empty, (1, 1): empty}
    extra = {(0, 0): (re, wig(re).astype(complex)), (1, 0): empty, (1, 1): empty}
    e0 = O.calculate_total_energy_average_one_surface(dens[(0, 0)], MASS, 0)
    opt = O.Optimization(sig, (x0 - 5, p0 - 5), (x0 + 5, p0 + 5), MASS, e0, 1.0, api=gpu, num_pes=2, local_maxeval=60, searches="native")
    opt.optimize({(0, 0): dens[(0, 0)]}, {(0, 0): extra[(0, 0)]})
    k0 = K.TrainingKernels(opt.get_parameters(), K.construct_training_sets({(0, 0): dens[(0, 0)]}, 2), True, True, False, api=gpu, num_pes=2)
    small = {(0, 0): False, (1, 0): True, (1, 1): True}
    mc = {e: S.MCParameters() for e in dens}
    d1, x1, s1, k1, res1 = S.main_tick(1, dens, extra, small, mc, opt, k0, MASS, DT, 2, 160, 1.0, rng, S.DAC, gpu)
    assert res1 is None and s1 == small and len(d1[(1, 0)][0]) == 0
    assert np.abs(d1[(0, 0)][0][:, 0] - (r[:, 0] + DT * r[:, 1] / MASS)).max() < 1e-2  # free flight on a flat stretch of the surface
    assert abs(k1.calculate_population() - 1.0) < 0.1
    d2, x2, s2, k2, res2 = S.main_tick(2, d1, x1, s1, mc, opt, k1, MASS, DT, 2, 160, 1.0, rng, S.DAC, gpu)
    assert res2 is not None and len(x2[(0, 0)][0]) == 160 and not np.array_equal(x2[(0, 0)][0], x1[(0, 0)][0])
    assert abs(k2.calculate_population() - 1.0) < 0.1


def test_tick_at_c5_size(gpu):
    """configs[4]: one tick of main.cpp:143-176 at N = 8192 points per element (evolve the density and 5N extra points per element — 48 N
    back-propagated predicts per element in one batch each — then refit).  Population and purity of the refitted kernels stay within the drift
    the reference tolerates before it re-optimises (main.cpp:179-188: 10 %), coordinates move by the classical step, and the batched tick equals
    the same tick on 8 spot-checked points alone (the few-points predict path) to the accuracy two GP predicts allow."""
    from gaussian_process_liouville_equation_amd import steploop
    N = 8192
    rng = np.random.Generator(np.random.PCG64(20240607 + 8))
    dens, extra = {}, {}
    xc = -1.5
    for e, (i, j) in enumerate(K.element_order(2)):
        for store, n in ((dens, N), (extra, 5 * N)):
            r = rng.normal([xc, 14.112], [0.7086, 0.7056], size=(n, 2))
            g = np.exp(-0.5 * (((r[:, 0] - xc) / 0.7086) ** 2 + ((r[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
            store[(i, j)] = (r, (g * (0.6, 0.3 * np.exp(0.4j * (r[:, 0] - xc)), 0.4)[e]).astype(complex))
    # complex hyper-parameters with distinct sub-kernels: at the reference's INITIAL ones (opt.cpp:306-332: sR = sI, lR = lI) the pseudo-covariance
    # 2i K_C equals the covariance K_R + K_I in modulus, the widely-linear model degenerates to labels of fixed phase, and one tick of
    # back-propagation against that fit halves |rho_10| (purity 0.61 -> 0.55 on the oracle as well, N = 300) — the optimiser moves away from them
    params = {(0, 0): TH, (1, 0): THC, (1, 1): TH}
    k0 = K.TrainingKernels(params, K.construct_training_sets(dens), True, True, False, api=gpu)
    pop0, pur0 = k0.calculate_population(), k0.calculate_purity()
    assert abs(pop0 - 1.0) <= 0.02
    d1, x1, k1 = steploop.tick(dens, extra, params, MASS, DT, k0, steploop.DAC)
    for e in dens:
        assert d1[e][0].shape == dens[e][0].shape and x1[e][0].shape == extra[e][0].shape
        assert np.all(np.isfinite(d1[e][1])) and np.all(np.isfinite(x1[e][1]))
        # one classical step: dx = p dt / m up to the force's second-order term
        assert np.abs(d1[e][0][:, 0] - dens[e][0][:, 0] - dens[e][0][:, 1] * DT / MASS).max() <= 1e-3
    pop1, pur1 = k1.calculate_population(), k1.calculate_purity()
    assert abs(pop1 - pop0) <= 0.02 * abs(pop0), (pop0, pop1)
    assert abs(pur0 - 0.70) <= 0.02 and abs(pur1 - pur0) <= 0.03 * abs(pur0), (pur0, pur1)  # 2 pi int (0.36 + 0.16 + 2 * 0.09) g^2 = 0.70
    # spot check: the same tick for 8 points per element alone
    idx = rng.choice(N, 8, replace=False)
    sub = {e: (dens[e][0][idx], dens[e][1][idx]) for e in dens}
    s1 = steploop.evolve(sub, MASS, DT, k0, steploop.DAC)
    for e in dens:
        assert np.array_equal(s1[e][0], d1[e][0][idx])
        scale = np.abs(dens[e][1]).max()
        assert np.abs(s1[e][1] - d1[e][1][idx]).max() <= 1e-8 * scale, e


# ---- N-level step loop on the device (gple_evolve_n, gple_pes_adiabatic_n) ----------------------------------------------------------------------

@pytest.mark.parametrize("num_pes,model", [(2, 0), (2, 1), (2, 2), (3, 0), (3, 1), (3, 2), (3, 3)])
def test_n_level_pes_against_oracle(gpu, num_pes, model):
    """adiabatic energies, force matrix and non-adiabatic couplings of the N-level path (cyclic Jacobi on the device) against numpy's eigh with
    the same ordering and sign convention (oracle/evolve_oracle_n.py); models 0-2 at three levels = Tully + the uncoupled third diabat"""
    from oracle import evolve_oracle_n as EN
    x = np.concatenate([np.linspace(-12, -0.01, 150), np.linspace(0.01, 12, 150), [0.37, -2.5]])
    Eg, Fg, Ng = gpu.pes_adiabatic_n(num_pes, model, x)
    Eo, _, Fo, No = EN.adiabatic(x, model, num_pes)
    assert np.abs(Eg - Eo).max() <= 1e-14 * max(np.abs(Eo).max(), 1e-3)
    assert np.abs(Fg - Fo).max() <= 1e-11 * np.abs(Fo).max()
    assert np.abs(Ng - No).max() <= 1e-9 * np.abs(No).max()


def _case_n(num_pes, N, seed, x_centre=-1.5):
    rng = np.random.Generator(np.random.PCG64(seed))
    dens = {}
    for e, (i, j) in enumerate(K.element_order(num_pes)):
        r = rng.normal([x_centre, 14.0], [0.7086, 0.7056], size=(N + 5 * e, 2))
        g = np.exp(-0.5 * (((r[:, 0] - x_centre) / 0.7086) ** 2 + ((r[:, 1] - 14.0) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
        rho = g * ((0.5, 0.3, 0.2)[i] if i == j else 0.15 * np.exp(0.4j * (r[:, 0] - x_centre) + 0.3j * (i + j)))
        dens[(i, j)] = (r, rho.astype(complex))
    return dens


def _fits_n(api, dens, num_pes):
    return [api.real_fit(TH, *dens[e], 0) if e[0] == e[1] else api.complex_fit(THC, *dens[e], 0) for e in K.element_order(num_pes)]


@pytest.mark.parametrize("model", [E.SAC, E.DAC, E.ECR])
def test_n_level_tick_at_two_levels_is_the_two_level_tick(gpu, model):
    """gple_evolve_n with num_pes = 2 against gple_evolve (the reference's three-branch code on the device): same coordinates bit for bit, same
    densities to the rounding of two different arithmetic routes (Jacobi + projectors vs the closed two-level forms)"""
    dens = _case_n(2, 120, 40 + model)
    fits = _fits_n(gpu, dens, 2)
    a = gpu.evolve(fits, model, MASS, DT, dens)
    b = gpu.evolve_n(2, fits, model, MASS, DT, dens)
    for e in dens:
        assert np.abs(a[e][0] - b[e][0]).max() <= 1e-13 * np.abs(a[e][0]).max()
        scale = np.abs(a[e][1]).max()
        assert np.abs(a[e][1] - b[e][1]).max() <= 1e-9 * scale, e
    a = gpu.evolve(fits, model, MASS, DT, dens, new_points=True)
    b = gpu.evolve_n(2, fits, model, MASS, DT, dens, new_points=True)
    for e in dens:
        assert np.array_equal(b[e][0], dens[e][0])
        assert np.abs(a[e][1] - b[e][1]).max() <= 1e-9 * np.abs(a[e][1]).max(), e


@pytest.mark.parametrize("model", [3, E.DAC])
def test_three_level_tick_against_oracle(gpu, oracle, model):
    """one tick of a three-level system (6 elements, 36 back-propagated predicts per point) on the device against the numpy oracle of the derived
    N-level back-propagation with the C++ oracle's predictor as its DistributionFunction; model 3 = three coupled states, DAC = Tully II plus
    the uncoupled third diabat pes.cpp gives for NumPES = 3"""
    from oracle import evolve_oracle_n as EN
    dens = _case_n(3, 60, 70 + model, x_centre=-0.8)
    fg, fo = _fits_n(gpu, dens, 3), _fits_n(oracle, dens, 3)
    order = K.element_order(3)

    def distribution(pts, i, j):
        f = fo[order.index((i, j))]
        pred = oracle.complex_predict if i != j else oracle.real_predict
        return np.asarray(pred(f, pts, want=("cutoff",))["cutoff"], dtype=complex)

    out = gpu.evolve_n(3, fg, model, MASS, DT, dens)
    ref = EN.evolve(dens, MASS, DT, distribution, model, 3)
    for e in order:
        assert np.abs(out[e][0] - ref[e][0]).max() <= 1e-12 * np.abs(ref[e][0]).max()
        scale = max(np.abs(ref[e][1]).max(), np.abs(dens[e][1]).max())
        assert np.abs(out[e][1] - ref[e][1]).max() <= 1e-8 * scale, e
        if model == 3:  # (with the uncoupled third diabat the spectator's population is carried along unchanged: flat surface, zero-shift branch)
            assert np.abs(out[e][1] - dens[e][1]).max() > 1e-6 * scale
    for k in range(3):  # populations stay real
        assert np.abs(out[(k, k)][1].imag).max() <= 1e-9 * np.abs(out[(k, k)][1]).max()


def test_three_level_tick_through_the_step_loop(gpu):
    """steploop.tick with NumPES = 3 (configs[4]'s "3-state PES ... full step loop" at test size): evolve density and extra points of all six
    elements, refit six kernels; population and purity of the refitted kernels stay within the reference's re-optimisation thresholds"""
    from gaussian_process_liouville_equation_amd import steploop
    dens, extra = _case_n(3, 300, 91, x_centre=-0.8), _case_n(3, 600, 92, x_centre=-0.8)
    params = {e: (TH if e[0] == e[1] else THC) for e in K.element_order(3)}
    k0 = K.TrainingKernels(params, K.construct_training_sets(dens, 3), True, True, False, api=gpu, num_pes=3)
    pop0, pur0 = k0.calculate_population(), k0.calculate_purity()
    d1, x1, k1 = steploop.tick(dens, extra, params, MASS, DT, k0, steploop.TSAC)
    assert set(d1) == set(dens) and all(len(d1[e][0]) == len(dens[e][0]) and len(x1[e][0]) == len(extra[e][0]) for e in dens)
    assert abs(k1.calculate_population() - pop0) <= 0.05 * abs(pop0)
    assert abs(k1.calculate_purity() - pur0) <= 0.10 * abs(pur0)


def test_three_level_tick_at_c5_size(gpu):
    """configs[4] AS STATED: "3-state PES ... N = 8192 ... full step loop" — one tick of main.cpp:143-176 for a three-level system at N = 8192
    points per element: six elements (3 real + 3 complex GPs, n = 8192 / 16384), 36 back-propagated predicts per point (gple_evolve_n), the
    density and an extra set of N points per element evolved (bench.py --workload C5step3 times the tick with the full 5N extra points:
    profiles/r04_bench/C5step3.json), then six refits.  The reference asserts for NumPES > 2 (evolve.cpp:367-371), so what is checked is what the
    two-level test at this size checks: population and purity of the refitted kernels within the drift the reference tolerates before it
    re-optimises (main.cpp:179-188; population 2 %, purity 10 %), coordinates moved by the classical step, and the batched tick equal to the
    same tick on 8 spot-checked points alone (the few-points predict path) to the accuracy two routes through the GP predict allow."""
    from gaussian_process_liouville_equation_amd import steploop
    N, xc = 8192, -0.5
    rng = np.random.Generator(np.random.PCG64(20240607 + 9))
    weight = (0.5, 0.3, 0.2)
    order = K.element_order(3)
    dens, extra = {}, {}
    for (i, j) in order:
        for store in (dens, extra):
            r = rng.normal([xc, 14.112], [0.7086, 0.7056], size=(N, 2))
            g = np.exp(-0.5 * (((r[:, 0] - xc) / 0.7086) ** 2 + ((r[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
            amp = weight[i] if i == j else 0.6 * np.sqrt(weight[i] * weight[j]) * np.exp(0.4j * (r[:, 0] - xc) + 0.3j * (i + j))
            store[(i, j)] = (r, (g * amp).astype(complex))
    params = {e: (TH if e[0] == e[1] else THC) for e in order}
    k0 = K.TrainingKernels(params, K.construct_training_sets(dens, 3), True, True, False, api=gpu, num_pes=3)
    pop0, pur0 = k0.calculate_population(), k0.calculate_purity()
    assert abs(pop0 - 1.0) <= 0.02
    # purity = 2 pi int g^2 (sum_k w_k^2 + 2 sum_{k<l} 0.36 w_k w_l) with 2 pi int g^2 = 1 / (2 sx sp) = 1.00: 0.38 + 0.72 * 0.31 = 0.6032
    assert abs(pur0 - 0.6032) <= 0.03 * 0.6032, pur0
    d1, x1, k1 = steploop.tick(dens, extra, params, MASS, DT, k0, steploop.TSAC)
    for e in order:
        assert d1[e][0].shape == dens[e][0].shape and x1[e][0].shape == extra[e][0].shape
        assert np.all(np.isfinite(d1[e][1])) and np.all(np.isfinite(x1[e][1]))
        assert np.abs(d1[e][0][:, 0] - dens[e][0][:, 0] - dens[e][0][:, 1] * DT / MASS).max() <= 1e-3
    pop1, pur1 = k1.calculate_population(), k1.calculate_purity()
    assert abs(pop1 - pop0) <= 0.02 * abs(pop0), (pop0, pop1)
    assert abs(pur1 - pur0) <= 0.10 * abs(pur0), (pur0, pur1)
    for k in range(3):  # populations stay real
        assert np.abs(d1[(k, k)][1].imag).max() <= 1e-9 * np.abs(d1[(k, k)][1]).max()
    idx = rng.choice(N, 8, replace=False)
    sub = {e: (dens[e][0][idx], dens[e][1][idx]) for e in order}
    s1 = steploop.evolve(sub, MASS, DT, k0, steploop.TSAC)
    for e in order:
        assert np.array_equal(s1[e][0], d1[e][0][idx])
        scale = np.abs(dens[e][1]).max()
        assert np.abs(s1[e][1] - d1[e][1][idx]).max() <= 1e-8 * scale, e


def test_three_level_main_tick_with_elements_appearing(gpu):
    """steploop.main_tick for a three-level system (the reference's loop of main.cpp:136-186 compiled for NumPES = 3, where it asserts): all
    population on the lowest state just before the three-state crossing of the library's model; is_very_small asks the N-level new-point
    prediction for the five empty elements, elements that stopped being negligible get their points by the Metropolis re-selection, the
    optimiser (six-element parameter layout) runs because the element set changed, and the refitted kernels keep the population"""
    from gaussian_process_liouville_equation_amd import optimization as O, steploop as S
    rng = np.random.default_rng(5)
    sig, x0, p0 = np.array([0.7086, 0.7056]), -0.6, 14.112
    wig = lambda r: np.exp(-0.5 * (((r - [x0, p0]) / sig) ** 2).sum(axis=1)) / (2 * np.pi * sig.prod())
    r, re = rng.normal(size=(100, 2)) * sig + [x0, p0], rng.normal(size=(200, 2)) * sig + [x0, p0]
    empty = (np.zeros((0, 2)), np.zeros(0, dtype=complex))
    order = K.element_order(3)
    dens = {e: empty for e in order}
    extra = {e: empty for e in order}
    dens[(0, 0)], extra[(0, 0)] = (r, wig(r).astype(complex)), (re, wig(re).astype(complex))
    e0 = O.calculate_total_energy_average_one_surface(dens[(0, 0)], MASS, 0)
    opt = O.Optimization(sig, (x0 - 5, p0 - 5), (x0 + 5, p0 + 5), MASS, e0, 1.0, api=gpu, num_pes=3, local_maxeval=40, searches="native")
    opt.optimize({(0, 0): dens[(0, 0)]}, {(0, 0): extra[(0, 0)]})
    k0 = K.TrainingKernels(opt.get_parameters(), K.construct_training_sets({(0, 0): dens[(0, 0)]}, 3), True, True, False, api=gpu, num_pes=3)
    assert k0.num_pes == 3 and abs(k0.calculate_population() - 1.0) < 0.05
    small0 = {e: e != (0, 0) for e in order}
    # the new-point prediction of the coupled elements at the packet: population flows to the neighbouring state through the coherence
    np10 = S.new_point_predict(r[:20], 1, 0, MASS, DT, k0, S.TSAC, gpu)
    np22 = S.new_point_predict(r[:20], 2, 2, MASS, DT, k0, S.TSAC, gpu)
    assert np.abs(np10).max() > 1e-5 and np.abs(np10).max() > 10 * np.abs(np22).max()  # 0 -> 1 is first order in the coupling, 0 -> 2 second
    small = S.is_very_small(dens, MASS, DT, k0, S.TSAC, gpu)
    assert set(small) == set(order) and small[(0, 0)] is False and small[(1, 0)] is False
    mc = {e: S.MCParameters(20, 0.3) for e in order}
    d1, x1, s1, k1, res1 = S.main_tick(1, dens, extra, small0, mc, opt, k0, MASS, DT, 50, 200, 1.0, rng, S.TSAC, gpu)
    assert res1 is not None and s1 == small                      # the element set changed: re-selection + optimisation (main.cpp:148-163)
    assert len(d1[(1, 0)][0]) == 100 and len(x1[(1, 0)][0]) == 200  # the coherence with the neighbouring state now has its points
    for e in order:
        assert (len(d1[e][0]) == 0) == s1[e]
    assert abs(k1.calculate_population() - 1.0) < 0.1 and k1.num_pes == 3
