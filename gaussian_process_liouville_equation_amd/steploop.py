"""Host mirror of the per-tick loop around the GP, on the device entry points of include/gple.h.

reference (paths relative to /root/reference/gaussian_process_liouville_equation/):
    evolve                      evolve.h:19-24, evolve.cpp:377-423   -> evolve()
    generate_markov_chain       mc.cpp:118-165                       -> generate_markov_chain()
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


def tick(density, extra_points, ParameterVectors, mass, dt, all_kernels, model=DAC, api=None):
    """main.cpp:143-176 without the re-optimisation branches: evolve the density and the extra points against the current
    kernels, then refit the kernels on the evolved density (TrainingKernels(params, density), predict.cpp:390-393)."""
    api = _api(all_kernels, api)
    density = evolve(density, mass, dt, all_kernels, model, api)
    extra_points = evolve(extra_points, mass, dt, all_kernels, model, api)
    sets = K.construct_training_sets({e: v for e, v in density.items() if len(v[0])}, 2)
    new_kernels = K.TrainingKernels(ParameterVectors, sets, True, True, False, api=api, num_pes=2)
    return density, extra_points, new_kernels
