"""Row N4, the two writers that consume this path's outputs directly: `output_phase` and `output_param`
(output.cpp:120-133, 180-232), in the text layout plot.py reads — one line per row, values separated by one blank, default
stream precision (6 significant digits, `%g`), an empty line after each time step.  The other writers of output.cpp
(averages over Monte-Carlo samples, points, logging) belong to callers that are out of scope."""
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
