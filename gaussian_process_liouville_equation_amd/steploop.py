"""Host mirror of the per-tick loop around the GP, on the device entry points of include/gple.h.

reference (paths relative to /root/reference/gaussian_process_liouville_equation/):
    evolve                      evolve.h:19-24, evolve.cpp:377-423   -> evolve()
    generate_markov_chain       mc.cpp:118-165                       -> generate_markov_chain()
    MCParameters                mc.h:46-92                           -> MCParameters
    acceptance_optimize_displacement   mc.cpp:287-331                -> acceptance_optimize_displacement()
    autocorrelation_optimize_steps     mc.cpp:167-285                -> autocorrelation_optimize_steps()
    element_monte_carlo, monte_carlo_selection   mc.cpp:333-403      -> element_monte_carlo(), monte_carlo_selection()
    one tick of main()          main.cpp:143-176                     -> tick(): evolve density and extra points, refit
The reference passes a per-point DistributionFunction (stdafx.h:155) into these loops; here `all_kernels` (a TrainingKernels)
plays that role and every tick costs three batched predicts instead of 8 one-point predicts per sample (NumPES = 2: what the
reference instantiates, evolve.cpp:367-371).
"""
import numpy as np

from . import kernels as K

SAC, DAC, ECR = 0, 1, 2  # pes.h:27-32; TestModel defaults to DAC (pes.h:38-41)
_ORDER = [(0, 0), (1, 0), (1, 1)]


def _fits(all_kernels):
    return [None if all_kernels(i, j) is None else all_kernels(i, j)._fit for (i, j) in _ORDER]


def _api(all_kernels, api):
    if api is not None:
        return api
    for (i, j) in _ORDER:
        if all_kernels(i, j) is not None:
            return all_kernels(i, j)._api
    return K.default_api()


def _points(density):
    empty = (np.zeros((0, 2)), np.zeros(0, dtype=complex))
    return {e: (density.get(e) if density.get(e) is not None else empty) for e in _ORDER}


def evolve(density, mass, dt, all_kernels, model=DAC, api=None):
    """evolve.cpp:377-423: every selected point of every element one time step further, with its density rebuilt by
    back-propagation against the current fit.  density: {(iPES, jPES): (r (n, 2), rho (n,) complex)}; returns the same."""
    assert all_kernels.num_pes == 2, "the reference instantiates the two-level system only (evolve.cpp:367-371)"
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
    for e, (iPES, jPES) in enumerate(_ORDER):
        pts = _points(density)[(iPES, jPES)]
        if len(pts[0]) == 0:
            out[(iPES, jPES)] = pts
            continue
        which = np.full(len(pts[0]), e, dtype=np.int32)
        distribution = lambda r, which=which: a.predict_batch(fits, r, which[:len(r)])
        out[(iPES, jPES)] = element_monte_carlo(pts, MCParams[(iPES, jPES)], _device_chain(all_kernels, iPES, jPES, a), distribution, seeds)
    return out


def tick(density, extra_points, ParameterVectors, mass, dt, all_kernels, model=DAC, api=None):
    """main.cpp:143-176 without the re-optimisation branches: evolve the density and the extra points against the current
    kernels, then refit the kernels on the evolved density (TrainingKernels(params, density), predict.cpp:390-393)."""
    api = _api(all_kernels, api)
    density = evolve(density, mass, dt, all_kernels, model, api)
    extra_points = evolve(extra_points, mass, dt, all_kernels, model, api)
    sets = K.construct_training_sets({e: v for e, v in density.items() if len(v[0])}, 2)
    new_kernels = K.TrainingKernels(ParameterVectors, sets, True, True, False, api=api, num_pes=2)
    return density, extra_points, new_kernels
