"""One time step of the reference's GP pipeline (main.cpp:160-185) on the MI355X path, end to end on synthetic samples:
Optimization (hyper-parameter search over the device objective) -> TrainingKernels -> grid prediction written as
phase.txt / var.txt / param.txt (the layout plot.py reads).  Run on a GPU box: python examples/one_step.py [outdir]"""
import math, os, sys, time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gaussian_process_liouville_equation_amd import kernels as K, optimization as O, output  # noqa: E402

outdir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/one_step"
os.makedirs(outdir, exist_ok=True)
rng = np.random.default_rng(7)
sigma, x0, p0, mass, n = (0.7086, 0.7056), -10.0, 14.112, 2000.0, 500  # the reference's test case: hbar = 1, sigma_x sigma_p = 1/2
wigner = lambda r: np.exp(-0.5 * (((r[:, 0] - x0) / sigma[0]) ** 2 + ((r[:, 1] - p0) / sigma[1]) ** 2)) / (2 * math.pi * sigma[0] * sigma[1])
draw = lambda m: rng.normal(size=(m, 2)) * sigma + (x0, p0)
r = draw(n)
density = {(0, 0): (r, wigner(r).astype(complex))}   # everything on surface 0, as at t = 0
re = draw(2 * n)
extra = {(0, 0): (re, wigner(re).astype(complex))}
e0 = O.calculate_total_energy_average_one_surface(density[(0, 0)], mass, 0)

pool = K.ApiPool(3)
t = time.perf_counter()
opt = O.Optimization(sigma, (x0 - 5, p0 - 5), (x0 + 5, p0 + 5), mass, e0, 1.0, api=pool)
err, steps, kind = opt.optimize(density, extra)
t_opt = time.perf_counter() - t
ks = K.TrainingKernels(opt.get_parameters(), K.construct_training_sets(density), False, True, False, api=pool)
g = 256
xs, ps = np.linspace(x0 - 5, x0 + 5, g), np.linspace(p0 - 5, p0 + 5, g)
grid = np.stack(np.meshgrid(xs, ps, indexing="ij"), axis=-1).reshape(-1, 2)
t = time.perf_counter()
with open(os.path.join(outdir, "phase.txt"), "w") as ph, open(os.path.join(outdir, "var.txt"), "w") as va:
    output.output_phase(ph, va, ks, grid)
t_out = time.perf_counter() - t
with open(os.path.join(outdir, "param.txt"), "w") as f:
    output.output_param(f, opt)
rho = np.loadtxt(os.path.join(outdir, "phase.txt"))[0].reshape(g, g)
dx, dp = xs[1] - xs[0], ps[1] - ps[0]
print(f"optimisation: {kind.name}, error {err:.3e}, evaluations per stage {steps}, {t_opt:.2f} s")
print(f"parameters rho[0][0]: {[round(float(v), 4) for v in opt.get_parameters()[(0, 0)]]}")
print(f"population analytic {ks.calculate_population():.6f}, grid quadrature of phase.txt {rho.sum() * dx * dp:.6f}, purity {ks.calculate_purity():.6f}")
print(f"grid predict + write of {g}x{g} points for 3 elements: {t_out:.2f} s -> {outdir}/phase.txt, var.txt, param.txt")
pool.close()
