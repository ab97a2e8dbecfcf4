"""Host mirror of the per-tick loop around the GP, on the device entry points of include/gple.h.

reference (paths relative to /root/reference/gaussian_process_liouville_equation/):
    evolve                      evolve.h:19-24, evolve.cpp:377-423   -> evolve()
    generate_markov_chain       mc.cpp:118-165                       -> generate_markov_chain()
    MCParameters                mc.h:46-92                           -> MCParameters
    acceptance_optimize_displacement   mc.cpp:287-331                -> acceptance_optimize_displacement()
    autocorrelation_optimize_steps     mc.cpp:167-285                -> autocorrelation_optimize_steps()
    element_monte_carlo, monte_carlo_selection   mc.cpp:333-403      -> element_monte_carlo(), monte_carlo_selection()
    new_point_predict, is_very_small   evolve.cpp:425-478            -> new_point_predict(), is_very_small()
    generate_extra_points       mc.cpp:59-117                        -> generate_element_extra_points(), generate_extra_points()
    new_element_point_selection mc.cpp:405-537                       -> new_element_point_selection()
    one tick of main()          main.cpp:143-176                     -> tick(): evolve density and extra points, refit
    the loop body of main()     main.cpp:136-186                     -> main_tick(): tick + element changes + the three re-optimisation rules
The reference passes a per-point DistributionFunction (stdafx.h:155) into these loops; here `all_kernels` (a TrainingKernels)
plays that role and every tick costs one batched predict per element instead of 8 one-point predicts per sample.  NumPES = 2 is what
the reference instantiates (evolve.cpp:367-371 asserts beyond); with all_kernels.num_pes = 3 the same functions run on the N-level
back-propagation of gple_evolve_n and the six elements of a three-level density matrix.
"""
import numpy as np

from . import kernels as K

SAC, DAC, ECR = 0, 1, 2  # pes.h:27-32; TestModel defaults to DAC (pes.h:38-41)
TSAC = 3  # three-state avoided crossings: this library's three-level model (num_pes = 3 only; include/gple.h, gple_evolve_n)
_ORDER = [(0, 0), (1, 0), (1, 1)]


def element_order(num_pes):
    return [(i, j) for i in range(num_pes) for j in range(i + 1)]


def _fits(all_kernels):
    return [None if all_kernels(i, j) is None else all_kernels(i, j)._fit for (i, j) in element_order(all_kernels.num_pes)]


def _api(all_kernels, api):
    if api is not None:
        return api
    for (i, j) in element_order(getattr(all_kernels, "num_pes", 2)):
        if all_kernels(i, j) is not None:
            return all_kernels(i, j)._api
    return K.default_api()


def _points(density, num_pes=2):
    empty = (np.zeros((0, 2)), np.zeros(0, dtype=complex))
    return {e: (density.get(e) if density.get(e) is not None else empty) for e in element_order(num_pes)}


def evolve(density, mass, dt, all_kernels, model=DAC, api=None):
    """evolve.cpp:377-423: every selected point of every element one time step further, with its density rebuilt by
    back-propagation against the current fit.  density: {(iPES, jPES): (r (n, 2), rho (n,) complex)}; returns the same."""
    if all_kernels.num_pes != 2:  # beyond the reference (evolve.cpp:367-371 asserts): the N-level back-propagation of gple_evolve_n
        return _api(all_kernels, api).evolve_n(all_kernels.num_pes, _fits(all_kernels), model, float(np.ravel(mass)[0]), dt, density)
    return _api(all_kernels, api).evolve(_fits(all_kernels), model, float(np.ravel(mass)[0]), dt, _points(density))


def generate_markov_chain(NumSteps, all_kernels, MaxDisplacement, iPES, jPES, r, seed, api=None):
    """mc.cpp:118-165 for all start points r (n, 2) at once: (last points, acceptance ratio per chain)."""
    k = all_kernels(iPES, jPES) if iPES != jPES else all_kernels(iPES)
    return _api(all_kernels, api).markov_chain(None if k is None else k._fit, NumSteps, MaxDisplacement, seed, r)


MaxAcceptRatio, MinAcceptRatio = 0.5, 0.15  # mc.cpp:19-21
PossibleDisplacement = (1e-4, 2e-4, 5e-4, 1e-3, 2e-3, 5e-3, 0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0, 2.0, 5.0, 10.0)  # mc.cpp:301


class MCParameters:
    """mc.h:46-92"""
    AboveMinFactor = 1.1

    def __init__(self, InitialSteps=200, InitialDisplacement=1.0):
        self.NOMC, self.displacement = int(InitialSteps), float(InitialDisplacement)

    def set_num_MC_steps(self, n):
        self.NOMC = int(n)

    def set_displacement(self, d):
        self.displacement = float(d)

    def get_num_MC_steps(self):
        return self.NOMC

    def get_max_displacement(self):
        return self.displacement


class _Seeds:
    """The reference draws from one clock-seeded engine shared by all threads (mc.cpp:17); here every chain launch gets the next
    key of a counter so that a run is reproducible from its first seed."""

    def __init__(self, seed):
        self.next = int(seed)

    def __call__(self):
        self.next += 1
        return self.next - 1


def _device_chain(all_kernels, iPES, jPES, api):
    k = all_kernels(iPES, jPES) if iPES != jPES else all_kernels(iPES)
    fit = None if k is None else k._fit
    a = _api(all_kernels, api)
    return lambda NumSteps, d, r, seed, want_chain=False: a.markov_chain(fit, NumSteps, d, seed, r, want_chain=want_chain)


def acceptance_optimize_displacement(MCParams, chain, r, seeds, MaxNOMC=2 * 500):
    """mc.cpp:287-331: the largest displacement of the table whose mean acceptance ratio over all points' chains of MaxNOMC steps lies
    inside (MinAcceptRatio, MaxAcceptRatio).  chain(NumSteps, d, r, seed[, want_chain]) runs the chains of all points at once."""
    for d in reversed(PossibleDisplacement):
        ratio = float(np.mean(chain(MaxNOMC, d, r, seeds())[1]))
        if MinAcceptRatio < ratio < MaxAcceptRatio:
            MCParams.set_displacement(d)
            return


def chain_autocorrelation(whole):
    """mc.cpp:205-226 for all chains at once: AutoCors[j] = < sum_i (x_i - <x>).(x_{i+j} - <x>) / (NSteps - j) >_chains, j < NSteps / 2.
    whole: (NSteps, n, 2).  The lagged sums come from one FFT per chain and dimension instead of the O(NSteps^2) double loop."""
    NSteps = whole.shape[0]
    a = whole - whole.mean(axis=0, keepdims=True)
    f = np.fft.rfft(a, n=2 * NSteps, axis=0)
    c = np.fft.irfft(f * np.conj(f), n=2 * NSteps, axis=0)[:NSteps // 2].sum(axis=2)  # (NSteps / 2, n)
    return (c / (NSteps - np.arange(NSteps // 2))[:, None]).mean(axis=1)


def autocorrelation_optimize_steps(MCParams, chain, r, seeds, MaxNOMC=2 * 1000):
    """mc.cpp:167-285: the number of Monte-Carlo steps after which a chain has forgotten its start — the first lag whose
    |autocorrelation| is within AboveMinFactor of the minimum found from a lag on at which one chain's acceptance is acceptable."""
    d = MCParams.get_max_displacement()
    AutoCors = chain_autocorrelation(chain(MaxNOMC, d, r, seeds(), True)[2])
    n = len(AutoCors)
    min_start_step, min_autocor_step, min_auto_cor = 0, 0, 0.0
    while True:  # mc.cpp:252-267
        min_start_step = min_autocor_step + 1
        if min_start_step >= n:
            min_start_step = 1
            min_autocor_step = int(np.argmin(np.abs(AutoCors)))
            min_auto_cor = float(np.abs(AutoCors[min_autocor_step]))
            break
        rel = int(np.argmin(np.abs(AutoCors[min_start_step:])))
        min_auto_cor = float(np.abs(AutoCors[min_start_step + rel]))
        min_autocor_step = min_start_step + rel
        acc = float(chain(min_autocor_step, d, r[:1], seeds())[1][0])
        if MinAcceptRatio <= acc <= MaxAcceptRatio:
            break
    for iStep in range(min_start_step, min_autocor_step):  # mc.cpp:268-275
        if abs(AutoCors[iStep]) <= MCParameters.AboveMinFactor * min_auto_cor:
            min_autocor_step = iStep
            break
    MCParams.set_num_MC_steps(min_autocor_step)


def element_monte_carlo(points, MCParams, chain, distribution, seeds):
    """mc.cpp:333-372: tune displacement and chain length, walk every point NumSteps further, take the fitted density there.
    points: (r (n, 2), rho (n,)); distribution(r) -> rho at all points at once.  Returns the new (r, rho)."""
    r = np.asarray(points[0], dtype=float)
    acceptance_optimize_displacement(MCParams, chain, r, seeds)
    autocorrelation_optimize_steps(MCParams, chain, r, seeds)
    r_new = chain(MCParams.get_num_MC_steps(), MCParams.get_max_displacement(), r, seeds())[0]
    return r_new, np.asarray(distribution(r_new), dtype=complex)


def monte_carlo_selection(density, MCParams, all_kernels, seed, api=None):
    """mc.cpp:378-403: Metropolis re-selection of the points of every populated element against the current fit.
    density: {(iPES, jPES): (r, rho)}; MCParams: {(iPES, jPES): MCParameters}.  Returns the new density."""
    a = _api(all_kernels, api)
    seeds = _Seeds(seed)
    fits = _fits(all_kernels)
    out = {}
    for e, (iPES, jPES) in enumerate(element_order(all_kernels.num_pes)):
        pts = _points(density, all_kernels.num_pes)[(iPES, jPES)]
        if len(pts[0]) == 0:
            out[(iPES, jPES)] = pts
            continue
        which = np.full(len(pts[0]), e, dtype=np.int32)
        distribution = lambda r, which=which: a.predict_batch(fits, r, which[:len(r)])
        out[(iPES, jPES)] = element_monte_carlo(pts, MCParams[(iPES, jPES)], _device_chain(all_kernels, iPES, jPES, a), distribution, seeds)
    return out


def new_point_predict(r, iPES, jPES, mass, dt, all_kernels, model=DAC, api=None):
    """evolve.cpp:425-443 at all points r (n, 2) at once: what element (iPES, jPES) would be there after one more tick according to
    the current fits of all elements (three-branch back-propagation, no exact density); 0 where the point does not couple."""
    r = np.asarray(r, dtype=float).reshape(-1, 2)
    dens = _points({}, all_kernels.num_pes)
    dens[(iPES, jPES)] = (r, np.zeros(len(r), dtype=complex))
    a = _api(all_kernels, api)
    if all_kernels.num_pes != 2:  # the N-level back-propagation (gple_evolve_n)
        return a.evolve_n(all_kernels.num_pes, _fits(all_kernels), model, float(np.ravel(mass)[0]), dt, dens, new_points=True)[(iPES, jPES)][1]
    return a.evolve(_fits(all_kernels), model, float(np.ravel(mass)[0]), dt, dens, new_points=True)[(iPES, jPES)][1]


def is_very_small(density, mass, dt, all_kernels, model=DAC, api=None):
    """evolve.cpp:445-478: {(iPES, jPES): bool}.  An element that has points is not small; one without is small when the new-point
    prediction stays below 1e-5 in modulus at every point of element (0, 0)."""
    pts = _points(density, all_kernels.num_pes)
    test = pts[(0, 0)][0]
    out = {}
    for e in element_order(all_kernels.num_pes):
        out[e] = len(pts[e][0]) == 0 and bool(np.all(np.abs(new_point_predict(test, e[0], e[1], mass, dt, all_kernels, model, api)) ** 2 < 1e-10))
    return out


def generate_element_extra_points(points, NumExtraPoints, distribution, rng):
    """mc.cpp:59-98: NumExtraPoints points, each a selected point (cyclically) plus a normal deviate with the per-coordinate standard
    deviation of the selected points, with the distribution evaluated there.  rng: numpy Generator (the reference: its shared engine)."""
    r = np.asarray(points[0], dtype=float)
    std = np.sqrt(np.maximum((r ** 2).mean(axis=0) - r.mean(axis=0) ** 2, 0.0))  # calculate_standard_deviation_one_surface
    new = r[np.arange(NumExtraPoints) % len(r)] + rng.normal(size=(NumExtraPoints, 2)) * std
    return new, np.asarray(distribution(new), dtype=complex)


def generate_extra_points(density, NumExtraPoints, all_kernels, rng, api=None):
    """mc.cpp:100-117 with the fits as the distribution (main.cpp:67,160,169,185)"""
    a, fits, out = _api(all_kernels, api), _fits(all_kernels), {}
    for k, e in enumerate(element_order(all_kernels.num_pes)):
        pts = _points(density, all_kernels.num_pes)[e]
        if len(pts[0]) == 0:
            out[e] = pts
            continue
        out[e] = generate_element_extra_points(pts, NumExtraPoints, lambda r, k=k: a.predict_batch(fits, r, np.full(len(r), k, dtype=np.int32)), rng)
    return out


def host_chain(distribution, rng):
    """Metropolis chains (mc.cpp:118-165) against an arbitrary batched distribution — for an element that has no fit yet, whose weight
    is the new-point prediction; one distribution call (= one device batch) per step.  Same call shape as the device chains."""

    def run(NumSteps, d, r, seed=None, want_chain=False):
        r = np.asarray(r, dtype=float).copy()
        w_old = np.abs(distribution(r))
        acc, whole = np.zeros(len(r)), [r.copy()]
        for _ in range(int(NumSteps)):
            r_new = r + rng.uniform(-d, d, size=r.shape)
            w_new = np.abs(distribution(r_new))
            with np.errstate(divide="ignore", invalid="ignore"):
                ok = (w_new > w_old) | (w_new / w_old > rng.uniform(size=len(r)))
            r, w_old = np.where(ok[:, None], r_new, r), np.where(ok, w_new, w_old)
            acc += ok
            if want_chain:
                whole.append(r.copy())
        res = (r, acc / max(1, int(NumSteps)))
        return res + (np.stack(whole),) if want_chain else res

    return run


def new_element_point_selection(density, extra_points, IsSmallOld, IsSmall, MCParams, all_kernels, mass, dt, rng, model=DAC, api=None):
    """mc.cpp:405-537: an element that stopped being small gets the NumPoints candidates of largest |new-point prediction| among all
    current points of all elements (replicated up to NumPoints if fewer are non-zero), a Metropolis re-selection against that
    prediction, and fresh extra points; an element that became small loses its points.  Returns (density, extra_points)."""
    if IsSmallOld == IsSmall:
        return density, extra_points
    order = element_order(all_kernels.num_pes)
    dens, extra = dict(_points(density, all_kernels.num_pes)), dict(_points(extra_points, all_kernels.num_pes))
    NumPoints, NumExtraPoints = len(dens[(0, 0)][0]), len(extra[(0, 0)][0])
    candidates = np.concatenate([np.asarray(x[e][0], dtype=float).reshape(-1, 2) for e in order for x in (dens, extra)])
    seeds = _Seeds(int(rng.integers(1, 2 ** 62)))
    for e in order:
        if IsSmallOld[e] and not IsSmall[e]:
            distribution = lambda r, e=e: new_point_predict(r, e[0], e[1], mass, dt, all_kernels, model, api)
            rho = distribution(candidates)
            keep = min(NumPoints, int(np.count_nonzero(rho)))
            order = np.argsort(-np.abs(rho) ** 2, kind="stable")[:keep]  # the `keep` most important points (nth_element + erase)
            r_sel, rho_sel = candidates[order], rho[order]
            while NumPoints >= 2 * len(r_sel) > 0:
                r_sel, rho_sel = np.concatenate([r_sel, r_sel]), np.concatenate([rho_sel, rho_sel])
            if 0 < len(r_sel) < NumPoints:
                fill = NumPoints - len(r_sel)
                r_sel, rho_sel = np.concatenate([r_sel, r_sel[:fill]]), np.concatenate([rho_sel, rho_sel[:fill]])
            dens[e] = element_monte_carlo((r_sel, rho_sel), MCParams[e], host_chain(distribution, rng), distribution, seeds)
            extra[e] = generate_element_extra_points(dens[e], NumExtraPoints, distribution, rng)
        elif not IsSmallOld[e] and IsSmall[e]:
            dens[e] = extra[e] = (np.zeros((0, 2)), np.zeros(0, dtype=complex))
    return dens, extra


def tick(density, extra_points, ParameterVectors, mass, dt, all_kernels, model=DAC, api=None):
    """main.cpp:143-176 without the re-optimisation branches: evolve the density and the extra points against the current
    kernels, then refit the kernels on the evolved density (TrainingKernels(params, density), predict.cpp:390-393)."""
    api = _api(all_kernels, api)
    num_pes = all_kernels.num_pes
    density = evolve(density, mass, dt, all_kernels, model, api)
    extra_points = evolve(extra_points, mass, dt, all_kernels, model, api)
    sets = K.construct_training_sets({e: v for e, v in density.items() if len(v[0])}, num_pes)
    new_kernels = K.TrainingKernels(ParameterVectors, sets, True, True, False, api=api, num_pes=num_pes)
    return density, extra_points, new_kernels


def main_tick(iTick, density, extra_points, IsSmall, MCParams, optimizer, all_kernels, mass, dt, ReoptFreq, NumExtraPoints, Purity, rng,
              model=DAC, api=None):
    """main.cpp:136-186, one pass of the evolution loop with everything the reference does in it: evolve the density and the extra
    points against the current kernels; detect elements that became populated / negligible (is_very_small) and re-select their
    points (new_element_point_selection); re-optimise the hyper-parameters when an element changed, every ReoptFreq ticks, or when
    the refitted kernels miss the population or the purity by more than twice AverageTolerance; refit the kernels.
    optimizer: optimization.Optimization.  Returns (density, extra_points, IsSmall, all_kernels, opt_result | None)."""
    from . import optimization as O
    api = _api(all_kernels, api)
    IsSmallOld = dict(IsSmall)
    density = evolve(density, mass, dt, all_kernels, model, api)
    extra_points = evolve(extra_points, mass, dt, all_kernels, model, api)
    IsSmall = is_very_small(density, mass, dt, all_kernels, model, api)
    opt_result = None

    num_pes = all_kernels.num_pes

    def refit(params):
        sets = K.construct_training_sets({e: v for e, v in density.items() if len(v[0])}, num_pes)
        return K.TrainingKernels(params, sets, True, True, False, api=api, num_pes=num_pes)

    def reoptimise():
        nonlocal all_kernels, extra_points
        res = optimizer.optimize({e: v for e, v in density.items() if len(v[0])}, {e: v for e, v in extra_points.items() if len(v[0])})
        all_kernels = refit(optimizer.get_parameters())
        extra_points = generate_extra_points(density, NumExtraPoints, all_kernels, rng, api)
        return res

    if IsSmallOld != IsSmall:  # main.cpp:148-163
        density, extra_points = new_element_point_selection(density, extra_points, IsSmallOld, IsSmall, MCParams, all_kernels, mass, dt, rng, model, api)
        opt_result = reoptimise()
    elif iTick % ReoptFreq == 0:  # :165-172
        opt_result = reoptimise()
    else:  # :174-189
        all_kernels = refit(optimizer.get_parameters())
        pop, pur = all_kernels.calculate_population(), all_kernels.calculate_purity()
        tol = 2.0 * O.AverageTolerance
        if pur > (1.0 + tol) * Purity or pop > 1.0 + tol or pop < 1.0 - tol:
            opt_result = reoptimise()
    return density, extra_points, IsSmall, all_kernels, opt_result
