"""Where a tick of the step loop spends its time (steploop.tick at N points per element, two levels): evolve(density), evolve(extra points, 5N),
the three refits — wall clock around each call, after a warm-up tick.  usage: python probes/tick_breakdown.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import kernels as K, steploop as S

api = pkg.open_api(0)
TH, THC = [1.0, 0.7086, 0.7056, 1e-2], [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05]
for N in [int(a) for a in sys.argv[1:]] or [256, 1024, 4096]:
    rng = np.random.default_rng(N)
    dens, extra = {}, {}
    for e, (i, j) in enumerate(K.element_order(2)):
        for store, n in ((dens, N), (extra, 5 * N)):
            r = rng.normal([-1.5, 14.112], [0.7086, 0.7056], size=(n, 2))
            g = np.exp(-0.5 * (((r[:, 0] + 1.5) / 0.7086) ** 2 + ((r[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
            store[(i, j)] = (r, (g * (0.6, 0.3 * np.exp(0.4j * (r[:, 0] + 1.5)), 0.4)[e]).astype(complex))
    params = {(0, 0): TH, (1, 0): THC, (1, 1): TH}
    k = K.TrainingKernels(params, K.construct_training_sets(dens), True, True, False, api=api)
    S.tick(dens, extra, params, 2000.0, 1.0, k, S.DAC, api)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        d1 = S.evolve(dens, 2000.0, 1.0, k, S.DAC, api)
        t1 = time.perf_counter()
        x1 = S.evolve(extra, 2000.0, 1.0, k, S.DAC, api)
        t2 = time.perf_counter()
        k1 = K.TrainingKernels(params, K.construct_training_sets(d1), True, True, False, api=api)
        pop = k1.calculate_population()
        t3 = time.perf_counter()
        t.append((t1 - t0, t2 - t1, t3 - t2))
    a = 1e3 * np.median(np.array(t), axis=0)
    print(f"N={N}: evolve(density) {a[0]:.2f} ms, evolve(extra 5N) {a[1]:.2f} ms, refit 3 elements {a[2]:.2f} ms, tick {a.sum():.2f} ms", flush=True)
api.close()
