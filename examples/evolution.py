"""A few passes of the reference's evolution loop (main.cpp:48-186) on the MI355X path, end to end on Tully's dual avoided
crossing: initial Gaussian on the lower surface -> optimise -> for every tick: evolve points and extra points by the MQCLE
back-propagation, detect newly populated elements and re-select their points by Metropolis, re-optimise when the reference
would, refit, write the ave.txt line.  Run on a GPU box: python examples/evolution.py [n_points] [n_ticks]"""
import math, os, sys, time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gaussian_process_liouville_equation_amd import kernels as K, optimization as O, output, steploop as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(11)
sigma, x0, p0, mass, dt = (0.7086, 0.7056), -4.0, 14.112, 2000.0, 10.0  # started close to the crossing so that a few ticks show it
wigner = lambda r: np.exp(-0.5 * (((r[:, 0] - x0) / sigma[0]) ** 2 + ((r[:, 1] - p0) / sigma[1]) ** 2)) / (2 * math.pi * sigma[0] * sigma[1])
draw = lambda m: rng.normal(size=(m, 2)) * sigma + (x0, p0)
empty = (np.zeros((0, 2)), np.zeros(0, dtype=complex))
r, re = draw(n), draw(5 * n)
density = {(0, 0): (r, wigner(r).astype(complex)), (1, 0): empty, (1, 1): empty}
extra = {(0, 0): (re, wigner(re).astype(complex)), (1, 0): empty, (1, 1): empty}
api = K.default_api()
potential = output.tully_potential(api, S.DAC)
e0 = O.calculate_total_energy_average_one_surface(density[(0, 0)], mass, 0, potential)
t0 = time.perf_counter()
opt = O.Optimization(sigma, (x0 - 6, p0 - 6), (x0 + 12, p0 + 6), mass, e0, 1.0, potential=potential, api=api, searches="native")
res = opt.optimize({(0, 0): density[(0, 0)]}, {(0, 0): extra[(0, 0)]})
kernels = K.TrainingKernels(opt.get_parameters(), K.construct_training_sets({(0, 0): density[(0, 0)]}, 2), True, True, False, api=api, num_pes=2)
print(f"t = 0: optimisation {res[2].name}, population {kernels.calculate_population():.5f}, purity {kernels.calculate_purity():.5f} ({time.perf_counter() - t0:.1f} s)")
small = {(0, 0): False, (1, 0): True, (1, 1): True}
mc = {e: S.MCParameters() for e in density}
S.acceptance_optimize_displacement.__defaults__, S.autocorrelation_optimize_steps.__defaults__ = (60,), (120,)  # short tuning runs for a demo
for it in range(1, ticks + 1):
    t1 = time.perf_counter()
    density, extra, small, kernels, res = S.main_tick(it, density, extra, small, mc, opt, kernels, mass, dt, 4, 5 * n, 1.0, rng, S.DAC, api)
    pops = [0.0 if kernels(i) is None else kernels(i).get_population() for i in range(2)]
    print(f"tick {it}: elements {[e for e in density if len(density[e][0])]}, populations {pops[0]:.4f} / {pops[1]:.4f}, purity {kernels.calculate_purity():.4f}, "
          f"{'re-optimised (' + res[2].name + ')' if res else 'refit only'}, {time.perf_counter() - t1:.1f} s")
