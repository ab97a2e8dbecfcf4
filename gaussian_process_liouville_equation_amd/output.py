"""Row N4: the writers of output.cpp in the text layout plot.py reads — `output_phase`, `output_param`,
`output_average`, `output_point`, `output_logging` (output.cpp:24-302) — one line per row, values separated by one blank,
default stream precision (6 significant digits, `%g`), an empty line after each time step where the reference writes one.
The Monte-Carlo observables they print (predict.cpp:65-244) are O(N) host glue over the selected points and live here too."""
import numpy as np

from . import kernels as K


def _fmt(v):
    return " ".join("%g" % x for x in np.asarray(v, dtype=float).ravel())


def output_phase(phase, variance, AllKernels, PhaseGrids, api=None):
    """output.cpp:180-232.  phase / variance: text streams; PhaseGrids (M, 2).  Per element two lines (real, imaginary part of
    the cut-off prediction) in `phase` and one line in `variance`; zeros for an element without a kernel.  The elements are
    predicted by `PredictiveKernel` / `PredictiveComplexKernel` on the device — the grid predict of the north-star path."""
    grid = np.asarray(PhaseGrids, dtype=float)
    zero = _fmt(np.zeros(len(grid)))
    for (i, j) in K.element_order(AllKernels.num_pes):
        k = AllKernels(i, j)
        if k is None:
            phase.write(zero + "\n" + zero + "\n")
            variance.write(zero + "\n")
        elif i == j:
            p = K.PredictiveKernel(grid, k, False)
            phase.write(_fmt(p.get_cutoff_prediction()) + "\n" + zero + "\n")
            variance.write(_fmt(p.get_variance()) + "\n")
        else:
            p = K.PredictiveComplexKernel(grid, k, False)
            pred = np.asarray(p.get_cutoff_prediction())
            phase.write(_fmt(pred.real) + "\n" + _fmt(pred.imag) + "\n")
            variance.write(_fmt(p.get_variance()) + "\n")
    phase.write("\n")
    variance.write("\n")


def output_param(os, Optimizer):
    """output.cpp:120-133: lower bound, parameters, upper bound of every element, one line each."""
    lb, param, ub = Optimizer.get_lower_bounds(), Optimizer.get_parameters(), Optimizer.get_upper_bounds()
    for e in K.element_order(Optimizer.num_pes):
        os.write(_fmt(lb[e]) + "\n" + _fmt(param[e]) + "\n" + _fmt(ub[e]) + "\n")
    os.write("\n")


# ---- the Monte-Carlo observables the writers need (predict.cpp:65-244): O(N) host glue over the selected points -------------
_ORDER2 = [(0, 0), (1, 0), (1, 1)]


def _elem(density, e):
    v = density.get(e)
    return (np.zeros((0, 2)), np.zeros(0, dtype=complex)) if v is None else (np.asarray(v[0], dtype=float).reshape(-1, 2), np.asarray(v[1], dtype=complex))


def calculate_population_each_surface(density, num_pes=2):  # predict.cpp:65-86 (normalised)
    w = np.array([_elem(density, (i, i))[1].real.sum() for i in range(num_pes)])
    return w / w.sum()


def calculate_1st_order_average_one_surface(points):  # predict.cpp:88-107
    r, rho = points
    return (r * rho.real[:, None]).sum(axis=0) / rho.real.sum()


def calculate_1st_order_average_all_surface(density, num_pes=2):  # predict.cpp:129-156
    num, den = np.zeros(2), 0.0
    for i in range(num_pes):
        r, rho = _elem(density, (i, i))
        if len(r):
            num, den = num + (r * rho.real[:, None]).sum(axis=0), den + rho.real.sum()
    return num / den


def _energies(points, mass, iPES, potential):
    r, rho = points
    e = r[:, 1] ** 2 / float(mass) / 2.0
    return e + (0.0 if potential is None else np.asarray(potential(r[:, 0], iPES), dtype=float))


def calculate_total_energy_average_each_surface(density, mass, potential=None, num_pes=2):  # predict.cpp:158-192
    out = np.zeros(num_pes)
    for i in range(num_pes):
        pts = _elem(density, (i, i))
        if len(pts[0]):
            out[i] = (_energies(pts, mass, i, potential) * pts[1].real).sum() / pts[1].real.sum()
    return out


def calculate_total_energy_average_all_surface(density, mass, potential=None, num_pes=2):  # predict.cpp:194-224
    eng, ppl = 0.0, 0.0
    for i in range(num_pes):
        pts = _elem(density, (i, i))
        if len(pts[0]):
            eng, ppl = eng + (_energies(pts, mass, i, potential) * pts[1].real).sum(), ppl + pts[1].real.sum()
    return eng / ppl


def calculate_purity_each_element(density, num_pes=2):  # predict.cpp:226-244: sum |rho|^2, lower triangle mirrored
    m = np.zeros((num_pes, num_pes))
    for i in range(num_pes):
        for j in range(i + 1):
            m[i, j] = m[j, i] = (np.abs(_elem(density, (i, j))[1]) ** 2).sum()
    return m


def tully_potential(api, model):
    """adiabatic_potential(x)[iPES] of pes.cpp:98-120 through the device entry gple_pes_adiabatic: potential(x (n,), iPES) -> (n,)"""
    return lambda x, iPES: api.pes_adiabatic(model, x)[:, iPES]


def output_average(os, AllKernels, density, mass, PurityFactor, potential=None):
    """output.cpp:24-118: one line per output tick of ave.txt — per surface (population, <x>, <p>, NaN) from the kernels'
    analytic integrals and (population, <x>, <p>, <E>) from the Monte-Carlo points; the same summed over surfaces; the purity
    matrix and its sum from the kernels and from the points.  `potential` as in optimization.py (None: flat surfaces)."""
    n = AllKernels.num_pes
    vals = []
    ppl_mci_each = calculate_population_each_surface(density, n)
    e_mci_each = calculate_total_energy_average_each_surface(density, mass, potential, n)
    for i in range(n):
        k = AllKernels(i)
        if k is not None:
            vals += [k.get_population(), *(np.asarray(k.get_1st_order_average()) / k.get_population())]
        else:
            vals += [0.0, np.nan, np.nan]
        vals.append(np.nan)
        vals.append(ppl_mci_each[i])
        pts = _elem(density, (i, i))
        vals += list(calculate_1st_order_average_one_surface(pts)) if len(pts[0]) else [np.nan, np.nan]
        vals.append(e_mci_each[i])
    ppl_prm_all = AllKernels.calculate_population()
    vals += [ppl_prm_all, *(AllKernels.calculate_1st_order_average() / ppl_prm_all), AllKernels.calculate_total_energy_average(e_mci_each) / ppl_prm_all]
    ppl_mci_all = ppl_mci_each.sum()
    vals += [ppl_mci_all, *(calculate_1st_order_average_all_surface(density, n) / ppl_mci_all),
             calculate_total_energy_average_all_surface(density, mass, potential, n) / ppl_mci_all]
    prt = np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1):
            k = AllKernels(i, j) if i != j else AllKernels(i)
            prt[i, j] = prt[j, i] = 0.0 if k is None else k.get_purity()
    vals += [*prt.ravel(), AllKernels.calculate_purity()]
    prt_mci = calculate_purity_each_element(density, n) * PurityFactor
    vals += [*prt_mci.ravel(), prt_mci.sum()]
    os.write(" " + " ".join("%g" % v for v in vals) + "\n")


def output_point(coord, value, density, extra_points, num_pes=2):
    """output.cpp:135-178: per element the coordinates of the selected + extra points (one line of x, one of p) in coord.txt and
    their densities (real line, imaginary line) in value.txt; zeros for an unpopulated element."""
    n0, n1 = len(_elem(density, (0, 0))[0]), len(_elem(extra_points, (0, 0))[0])
    for e in K.element_order(num_pes):
        r, rho = np.zeros((n0 + n1, 2)), np.zeros(n0 + n1, dtype=complex)
        d, x = _elem(density, e), _elem(extra_points, e)
        if len(d[0]):
            r[:n0], rho[:n0] = d[0], d[1]
            r[n0:n0 + len(x[0])], rho[n0:n0 + len(x[0])] = x[0], x[1]
        coord.write(_fmt(r[:, 0]) + "\n" + _fmt(r[:, 1]) + "\n")  # MatrixFormatter: rows of the 2 x n matrix on separate lines
        value.write(_fmt(rho.real) + "\n" + _fmt(rho.imag) + "\n")
    coord.write("\n")
    value.write("\n")


def output_logging(os, time, OptResult, MCParams, CPUTime, AllKernels):
    """output.cpp:235-302: time, seconds since the last output, Metropolis steps and displacements per element, rescale factor per
    kernel, optimisation error, steps, kind, wall-clock stamp.  MCParams: {(i, j): (num_steps, max_displacement)}."""
    import time as _t
    error, steps, kind = OptResult
    order = K.element_order(AllKernels.num_pes)
    vals = [time, CPUTime] + [MCParams[e][0] for e in order] + [MCParams[e][1] for e in order]
    for (i, j) in order:
        k = AllKernels(i, j) if i != j else AllKernels(i)
        vals.append(np.nan if k is None else k.get_rescale_factor())
    os.write(" ".join("%g" % v for v in vals) + " %g " % error + " ".join(str(int(s)) for s in steps) + " %d " % int(kind) + _t.strftime("%F %T %Z") + "\n")
