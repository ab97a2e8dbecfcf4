"""First-light check of the HIP path against the golden fixtures and the oracle (diagnostic script)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from oracle import binding
o = binding.load()
g = pkg.open_api(0)
np.set_printoptions(linewidth=200, precision=6)

def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)

for name in ['real_a', 'real_b', 'real_c']:
    gd = dict(np.load(f'tests/golden/{name}.npz'))
    fit = g.real_fit(gd['theta'], gd['X'], gd['y'], 3)
    s = fit.scalars
    print('==', name, 'info', s['info'])
    for k, gk in [('rescale_factor', 'rescale'), ('magnitude', 'magnitude'), ('error', 'error'), ('population', 'population'), ('purity', 'purity')]:
        print(f'  {k:16s} {s[k]:.15g} {float(gd[gk]):.15g} rel {abs(s[k]-gd[gk])/abs(gd[gk]):.2e}')
    print('  first_order', rel(s['first_order_average'], gd['first_order']))
    print('  K', rel(fit.get(c.R_KERNEL), gd['K']), ' W', rel(fit.get(c.R_INVERSE), gd['W']), ' v', rel(fit.get(c.R_INVLBL), gd['v']),
          ' diagW', rel(fit.get(c.R_INVERSE_DIAG), np.diag(gd['W'])))
    p = g.real_predict(fit, gd['Xs'])
    print('  pred mean', np.abs(p['prediction'] - gd['t_mean']).max(), ' var', np.abs(p['variance'] - gd['t_var']).max(), ' cut', np.abs(p['cutoff'] - gd['t_cut']).max())
    p = g.real_predict(fit, gd['Xv'], labels=gd['tv'])
    print('  val error', p['error'], float(gd['v_error']))
    Kg, dKg = g.real_gram(gd['theta'], gd['X'], gd['X'], True, True)
    print('  gram', rel(Kg, gd['K']), rel(dKg, gd['dK']))

for name in ['complex_a', 'complex_c', 'complex_b']:
    gd = dict(np.load(f'tests/golden/{name}.npz'))
    fit = g.complex_fit(gd['theta'], gd['X'], gd['y'], 3)
    s = fit.scalars
    print('==', name, 'info', s['info'])
    for k, gk in [('rescale_factor', 'rescale'), ('magnitude', 'magnitude'), ('error', 'error'), ('purity', 'purity')]:
        print(f'  {k:16s} {s[k]:.15g} {float(gd[gk]):.15g} rel {abs(s[k]-gd[gk])/abs(gd[gk]):.2e}')
    print('  K', rel(fit.get(c.C_KERNEL), gd['K']), ' Kt', rel(fit.get(c.C_PSEUDO), gd['Kt']), ' P', rel(fit.get(c.C_UPPER_LEFT), gd['P']),
          ' Q', rel(fit.get(c.C_LOWER_LEFT), gd['Q']), ' v', rel(fit.get(c.C_INVLBL), gd['v']))
    p = g.complex_predict(fit, gd['Xs'])
    print('  pred mean', np.abs(p['prediction'] - gd['t_mean']).max(), ' var', np.abs(p['variance'] - gd['t_var']).max(), ' cut', np.abs(p['cutoff'] - gd['t_cut']).max())
    p = g.complex_predict(fit, gd['Xv'], labels=gd['tv'])
    print('  val error', p['error'], float(gd['v_error']))

# medium size vs oracle
rng = np.random.default_rng(1)
for N, M in [(300, 1000), (1024, 4096)]:
    X = rng.normal([-10, 14.112], [0.7086, 0.7056], size=(N, 2))
    y = np.exp(-0.5 * (((X[:, 0] + 10) / 0.7086) ** 2 + ((X[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
    Xs = rng.normal([-10, 14.112], [1.5, 1.5], size=(M, 2))
    th = [1.0, 0.7086, 0.7056, 1e-2]
    t0 = time.time(); fo = o.real_fit(th, X, y, 3); po = o.real_predict(fo, Xs); t1 = time.time()
    fg = g.real_fit(th, X, y, 3); pg = g.real_predict(fg, Xs); t2 = time.time()
    fg2 = g.real_fit(th, X, y, 3); pg = g.real_predict(fg2, Xs); t3 = time.time()
    print(f'== N={N} M={M}: oracle {t1-t0:.3f}s gpu first {t2-t1:.3f}s second {t3-t2:.4f}s info {fg.scalars["info"]}')
    for k in ['error', 'population', 'purity', 'magnitude']:
        print(f'  {k:12s} gpu {fg.scalars[k]:.12g} oracle {fo.scalars[k]:.12g} rel {abs(fg.scalars[k]-fo.scalars[k])/abs(fo.scalars[k]):.2e}')
    print('  v', rel(fg.get(c.R_INVLBL), fo.get(c.R_INVLBL)), ' mean', np.abs(pg['prediction'] - po['prediction']).max(),
          ' var', np.abs(pg['variance'] - po['variance']).max(), ' cut', np.abs(pg['cutoff'] - po['cutoff']).max())
g.close()
