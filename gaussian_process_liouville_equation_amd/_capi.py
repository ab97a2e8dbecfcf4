"""ctypes view of include/gple.h.

`Api` binds one shared library that exports the gple.h entry points under a given prefix.  The product binds
libgple_hip.so with prefix "gple_" (see __init__.py); the test-only CPU oracle exports the same interface under
"oracle_" (without contexts or device pointers) and is bound by oracle/binding.py with this same class, so the
parity tests drive both sides with identical code.
"""
import ctypes as C
import weakref

import numpy as np

GPLE_OK = 0
GPLE_ERR_TIMEOUT = 6  # a draining call noticed that the one-launch factorisation had given up: the fit is recovered, work enqueued before it is NaN (include/gple.h)
CALC_ERROR = 0x1
CALC_AVERAGE = 0x2
CALC_DERIVATIVE = 0x4
IO_DEVICE = 0x100
EVOLVE_NEW_POINTS = 0x400  # gple_evolve: new_point_predict instead of a tick
PREDICT_FULL = 0x200  # contract every test row (default: far rows whose contraction cannot move the variance are skipped)

# gple_real_array / gple_complex_array
R_KERNEL, R_INVERSE, R_INVLBL, R_INVLBL_DERIV, R_LABEL, R_INVERSE_DIAG = range(6)
C_KERNEL, C_PSEUDO, C_UPPER_LEFT, C_LOWER_LEFT, C_INVLBL, C_INVLBL_DERIV, C_LABEL = range(7)

_dp = C.POINTER(C.c_double)


class RealFitScalars(C.Structure):
    _fields_ = [
        ("rescale_factor", C.c_double),
        ("magnitude", C.c_double),
        ("error", C.c_double),
        ("population", C.c_double),
        ("first_order_average", C.c_double * 2),
        ("purity", C.c_double),
        ("error_derivative", C.c_double * 4),
        ("population_derivative", C.c_double * 4),
        ("purity_derivative", C.c_double * 4),
        ("info", C.c_int),
    ]


class ComplexFitScalars(C.Structure):
    _fields_ = [
        ("rescale_factor", C.c_double),
        ("magnitude", C.c_double),
        ("error", C.c_double),
        ("purity", C.c_double),
        ("error_derivative", C.c_double * 8),
        ("purity_derivative", C.c_double * 8),
        ("info", C.c_int),
    ]


class Element(C.Structure):
    """gple_element: the fit of one density-matrix element (exactly one of the two set, or neither)"""
    _fields_ = [("real", C.c_void_p), ("cplx", C.c_void_p)]


class OptOptions(C.Structure):
    _fields_ = [("xtol_rel", C.c_double), ("ftol_rel", C.c_double), ("xtol_abs", C.c_double), ("ftol_abs", C.c_double),
                ("initial_step", C.c_double), ("max_eval", C.c_int)]


OBJECTIVE_FN = C.CFUNCTYPE(C.c_double, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)           # nlopt_func
CONSTRAINT_FN = C.CFUNCTYPE(None, C.c_uint, C.POINTER(C.c_double), C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)  # nlopt_mfunc


def _opt_options(maxeval=0, initial_step=0.5):
    return OptOptions(1e-5, 1e-5, 1e-15, 1e-15, initial_step, int(maxeval or 0))


def minimize_neldermead(lib, fun, x0, lb, ub, maxeval=0):
    """the library's own Nelder-Mead (gple_minimize_neldermead) on a Python objective fun(x list) -> float; (x, f, n_eval)"""
    n = len(x0)
    cb = OBJECTIVE_FN(lambda nn, xp, gp, data: float(fun([xp[i] for i in range(nn)])))
    x, lbv, ubv = _f64(x0).copy(), _f64(lb), _f64(ub)
    f, ne = C.c_double(), C.c_int()
    lib.gple_minimize_neldermead.argtypes = [OBJECTIVE_FN, C.c_void_p, C.c_uint, _dp, _dp, C.POINTER(OptOptions), _dp, _dp, C.POINTER(C.c_int)]
    st = lib.gple_minimize_neldermead(cb, None, n, _ptr(lbv), _ptr(ubv), C.byref(_opt_options(maxeval)), _ptr(x), C.cast(C.byref(f), _dp), C.byref(ne))
    if st != GPLE_OK:
        raise GpleError(f"gple_minimize_neldermead: status {st}")
    return list(x), f.value, ne.value


def minimize_auglag_eq(lib, fun, constraint, m, x0, lb, ub, maxeval=0):
    """the library's augmented-Lagrangian search: fun(x, want_grad) -> (f, grad or None); constraint(x, want_grad) ->
    (h (m,), grad (m*n,) row-major or None); returns (x, f, n_eval)"""
    n = len(x0)

    def f_cb(nn, xp, gp, data):
        v, g = fun([xp[i] for i in range(nn)], bool(gp))
        if gp:
            for i in range(nn):
                gp[i] = g[i]
        return float(v)

    def h_cb(mm, rp, nn, xp, gp, data):
        h, g = constraint([xp[i] for i in range(nn)], bool(gp))
        for i in range(mm):
            rp[i] = h[i]
        if gp:
            for i in range(mm * nn):
                gp[i] = g[i]

    fc, hc = OBJECTIVE_FN(f_cb), CONSTRAINT_FN(h_cb)
    x, lbv, ubv = _f64(x0).copy(), _f64(lb), _f64(ub)
    f, ne = C.c_double(), C.c_int()
    lib.gple_minimize_auglag_eq.argtypes = [OBJECTIVE_FN, C.c_void_p, CONSTRAINT_FN, C.c_void_p, C.c_uint, C.c_uint, _dp, _dp, C.POINTER(OptOptions), _dp, _dp,
                                            C.POINTER(C.c_int)]
    st = lib.gple_minimize_auglag_eq(fc, None, hc, None, m, n, _ptr(lbv), _ptr(ubv), C.byref(_opt_options(maxeval)), _ptr(x), C.cast(C.byref(f), _dp), C.byref(ne))
    if st != GPLE_OK:
        raise GpleError(f"gple_minimize_auglag_eq: status {st}")
    return list(x), f.value, ne.value


def objective_minimize_neldermead(lib, objectives, x0, lb, ub, maxeval=0):
    """gple_objective_minimize_neldermead over resident objectives (same data, different contexts): vertices evaluated concurrently"""
    n = len(x0)
    arr = (C.c_void_p * len(objectives))(*[o.handle.value for o in objectives])
    x, lbv, ubv = _f64(x0).copy(), _f64(lb), _f64(ub)
    f, ne = C.c_double(), C.c_int()
    lib.gple_objective_minimize_neldermead.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, _dp, _dp, C.POINTER(OptOptions), _dp, _dp, C.POINTER(C.c_int)]
    st = lib.gple_objective_minimize_neldermead(arr, len(objectives), n, _ptr(lbv), _ptr(ubv), C.byref(_opt_options(maxeval)), _ptr(x), C.cast(C.byref(f), _dp), C.byref(ne))
    if st != GPLE_OK:
        raise GpleError(f"gple_objective_minimize_neldermead: status {st}")
    return list(x), f.value, ne.value


def minimize_direct_l(lib, fun, x0, lb, ub, maxeval=0):
    """the library's DIRECT-L (gple_minimize_direct_l, the GN_DIRECT_L stand-in) on a Python objective fun(x list) -> float; (x, f, n_eval)"""
    n = len(x0)
    cb = OBJECTIVE_FN(lambda nn, xp, gp, data: float(fun([xp[i] for i in range(nn)])))
    x, lbv, ubv = _f64(x0).copy(), _f64(lb), _f64(ub)
    f, ne = C.c_double(), C.c_int()
    lib.gple_minimize_direct_l.argtypes = [OBJECTIVE_FN, C.c_void_p, C.c_uint, _dp, _dp, C.POINTER(OptOptions), _dp, _dp, C.POINTER(C.c_int)]
    st = lib.gple_minimize_direct_l(cb, None, n, _ptr(lbv), _ptr(ubv), C.byref(_opt_options(maxeval)), _ptr(x), C.cast(C.byref(f), _dp), C.byref(ne))
    if st != GPLE_OK:
        raise GpleError(f"gple_minimize_direct_l: status {st}")
    return list(x), f.value, ne.value


def objective_minimize_direct_l(lib, objectives, x0, lb, ub, is_log=None, maxeval=0):
    """gple_objective_minimize_direct_l over resident objectives (same data, different contexts): the new rectangle centres of an iteration
    are evaluated concurrently; is_log flags the coordinates that are logarithms of their parameter (the global tier's reparametrisation)"""
    n = len(x0)
    arr = (C.c_void_p * len(objectives))(*[o.handle.value for o in objectives])
    x, lbv, ubv = _f64(x0).copy(), _f64(lb), _f64(ub)
    flags = (C.c_ubyte * n)(*[1 if (is_log is not None and is_log[i]) else 0 for i in range(n)])
    f, ne = C.c_double(), C.c_int()
    lib.gple_objective_minimize_direct_l.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, _dp, _dp, C.POINTER(C.c_ubyte), C.POINTER(OptOptions), _dp, _dp,
                                                     C.POINTER(C.c_int)]
    st = lib.gple_objective_minimize_direct_l(arr, len(objectives), n, _ptr(lbv), _ptr(ubv), flags, C.byref(_opt_options(maxeval)), _ptr(x), C.cast(C.byref(f), _dp),
                                              C.byref(ne))
    if st != GPLE_OK:
        raise GpleError(f"gple_objective_minimize_direct_l: status {st}")
    return list(x), f.value, ne.value


class Points(C.Structure):
    """gple_points: the selected phase-space points of one density-matrix element"""
    _fields_ = [("r", C.POINTER(C.c_double)), ("rho", C.POINTER(C.c_double)), ("n", C.c_size_t)]


class PredictScalars(C.Structure):
    _fields_ = [("error", C.c_double), ("error_derivative", C.c_double * 8)]


def scalars_to_dict(s):
    out = {}
    for name, _ in s._fields_:
        v = getattr(s, name)
        out[name] = np.array(list(v)) if hasattr(v, "__len__") else v
    return out


class GpleError(RuntimeError):
    pass


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _points(X):
    """(N,2) array of phase-space points -> contiguous interleaved [x0,p0,x1,p1,...] (stdafx.h:153)."""
    X = _f64(X)
    if X.ndim != 2 or X.shape[1] != 2:
        raise ValueError("phase-space points must have shape (N, 2)")
    return X


def _cplx(y):
    return np.ascontiguousarray(y, dtype=np.complex128)


# every symbol include/gple.h declares (checked by tests/test_capi_symbols.py)
GPLE_SYMBOLS = [
    "ctx_create", "ctx_destroy", "ctx_synchronize", "ctx_trim", "status_string", "ctx_last_error", "ctx_enable_timing", "ctx_get_timing", "ctx_get_prune_stats",
    "real_gram", "complex_gram", "cutoff_factor", "predict_batch", "shard_bounds", "set_allgather_function", "real_predict_sharded", "complex_predict_sharded", "real_predict_dealt", "complex_predict_dealt", "deal_share",
    "real_fit_create", "real_fit_get_scalars", "real_fit_retain", "real_fit_release", "real_fit_size", "real_fit_get", "real_predict",
    "complex_fit_create", "complex_fit_get_scalars", "complex_fit_retain", "complex_fit_release", "complex_fit_size", "complex_fit_get",
    "complex_predict", "loose_function", "objective_create", "objective_eval", "objective_eval_part", "objective_release", "minimize_neldermead", "objective_minimize_neldermead", "minimize_direct_l", "objective_minimize_direct_l", "minimize_auglag_eq", "pes_adiabatic", "evolve", "evolve_n", "pes_adiabatic_n", "markov_chain", "markov_chain_trace", "nlml", "nlml_predict", "nlml_cross", "nlml_cross_predict",
]


class _Fit:
    """Owns one fit handle of either backend (RAII, like the reference's value-semantic kernel objects)."""

    def __init__(self, api, handle, kind, N, scalars=None):
        self.api, self.handle, self.kind, self.N = api, handle, kind, N
        self._scalars = scalars  # None: deferred create, fetched (with a stream sync) when a getter first needs them
        api._fits.add(self)

    @property
    def scalars(self):
        if self._scalars is None:
            sc = RealFitScalars() if self.kind == "real" else ComplexFitScalars()
            self.api._check(self.api._fn(f"{self.kind}_fit_get_scalars")(self.handle, C.byref(sc)))
            self._scalars = scalars_to_dict(sc)
        return self._scalars

    def release(self):
        if self.handle:
            getattr(self.api.lib, f"{self.api.prefix}{self.kind}_fit_release")(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def get(self, which):
        N = self.N
        if self.kind == "real":
            shape = {R_KERNEL: (N, N), R_INVERSE: (N, N), R_INVLBL: (N,), R_INVLBL_DERIV: (4, N), R_LABEL: (N,),
                     R_INVERSE_DIAG: (N,)}[which]
            buf = np.empty(int(np.prod(shape)), dtype=np.float64)
            self.api._fit_get(self, which, buf)
            # N x N matrices arrive column-major: viewed as a C-ordered numpy array that is the transpose
            return buf.reshape(shape).T.copy() if which in (R_KERNEL, R_INVERSE) else buf.reshape(shape)
        shape, cplx = {C_KERNEL: ((N, N), False), C_PSEUDO: ((N, N), True), C_UPPER_LEFT: ((N, N), True),
                       C_LOWER_LEFT: ((N, N), True), C_INVLBL: ((N,), True), C_INVLBL_DERIV: ((8, N), True),
                       C_LABEL: ((N,), True)}[which]
        buf = np.empty(int(np.prod(shape)) * (2 if cplx else 1), dtype=np.float64)
        self.api._fit_get(self, which, buf)
        arr = buf.view(np.complex128) if cplx else buf
        arr = arr.reshape(shape)
        return arr.T.copy() if which in (C_KERNEL, C_PSEUDO, C_UPPER_LEFT, C_LOWER_LEFT) else arr


class _Objective:
    """Owns one gple_objective handle (training and extra set resident on the device)."""

    def __init__(self, api, X, y, Xe, ye):
        self.api, self.handle = api, C.c_void_p()
        api._check(api.lib.gple_objective_create(api.ctx, _ptr(X), _ptr(y.view(np.float64)), len(X), _ptr(Xe), _ptr(ye.view(np.float64)),
                                                 len(Xe), C.byref(self.handle)))
        api._fits.add(self)  # released with the context, like the fits

    def __call__(self, x, want_grad=True):
        x = _f64(x)
        val = C.c_double()
        grad = np.empty(len(x)) if want_grad else None
        self.api._check(self.api.lib.gple_objective_eval(self.handle, _ptr(x), len(x), C.cast(C.byref(val), _dp), _ptr(grad)))
        return val.value, grad

    def part(self, x, part, nparts, want_grad=True):
        """gple_objective_eval_part: this rank's share of the value and gradient (sum over the parts = the whole; make_normal after the sum)"""
        x = _f64(x)
        val = C.c_double()
        grad = np.empty(len(x)) if want_grad else None
        self.api.lib.gple_objective_eval_part.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_int, C.c_int, _dp, _dp]
        self.api._check(self.api.lib.gple_objective_eval_part(self.handle, _ptr(x), len(x), int(part), int(nparts), C.cast(C.byref(val), _dp), _ptr(grad)))
        return val.value, grad

    def release(self):
        if self.handle:
            self.api.lib.gple_objective_release(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Api:
    def __init__(self, lib, prefix, with_ctx, device=0, stream=None):
        self.lib, self.prefix, self.with_ctx = lib, prefix, with_ctx
        self.ctx = None
        self._fits = weakref.WeakSet()  # a context must outlive its fit handles: close() releases them first
        self._declare()
        if with_ctx:
            ctx = C.c_void_p()
            self._check(lib.gple_ctx_create(int(device), C.c_void_p(stream), C.byref(ctx)))
            self.ctx = ctx

    # ---- plumbing --------------------------------------------------------------------------------------------
    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def _declare(self):
        ctx = [C.c_void_p] if self.with_ctx else []
        fl = [C.c_uint] if self.with_ctx else []  # the oracle has no flags on gram/cutoff/get
        sz, vp, ci = C.c_size_t, C.c_void_p, C.c_int
        sig = {
            "real_gram": ctx + [_dp, _dp, sz, _dp, sz, ci] + fl + [_dp, _dp],
            "complex_gram": ctx + [_dp, _dp, sz, _dp, sz, ci] + fl + [_dp, _dp, _dp, _dp],
            "cutoff_factor": ctx + [_dp, ci, _dp, sz] + fl + [_dp],
            "real_fit_create": ctx + [_dp, _dp, _dp, ci, sz, C.c_uint, C.POINTER(RealFitScalars), C.POINTER(vp)],
            "real_fit_release": [vp],
            "real_fit_get": [vp, ci] + fl + [_dp],
            "real_predict": ctx + [vp, _dp, sz, C.c_uint, _dp, _dp, _dp, _dp, C.POINTER(PredictScalars)],
            "complex_fit_create": ctx + [_dp, _dp, _dp, sz, C.c_uint, C.POINTER(ComplexFitScalars), C.POINTER(vp)],
            "complex_fit_release": [vp],
            "complex_fit_get": [vp, ci] + fl + [_dp],
            "complex_predict": ctx + [vp, _dp, sz, C.c_uint, _dp, _dp, _dp, _dp, C.POINTER(PredictScalars)],
            "loose_function": ctx + [_dp, sz, _dp, _dp, sz, _dp, _dp, sz, _dp, _dp],
            "nlml": ctx + [_dp, _dp, _dp, sz, _dp, _dp],
            "nlml_predict": ctx + [_dp, _dp, _dp, sz, _dp, sz] + fl + [_dp],
            "nlml_cross": ctx + [_dp, _dp, _dp, sz, _dp, _dp],
            "nlml_cross_predict": ctx + [_dp, _dp, _dp, sz, _dp, sz] + fl + [_dp],
        }
        for name, argtypes in sig.items():
            f = self._fn(name)
            f.argtypes, f.restype = argtypes, C.c_int
        if self.with_ctx:
            for kind, st in (("real", RealFitScalars), ("complex", ComplexFitScalars)):
                f = self._fn(f"{kind}_fit_get_scalars")
                f.argtypes, f.restype = [vp, C.POINTER(st)], C.c_int
            self.lib.gple_predict_batch.argtypes = [vp, C.POINTER(Element), sz, _dp, C.POINTER(ci), sz, _dp]
            self.lib.gple_predict_batch.restype = C.c_int
            self.lib.gple_objective_create.argtypes = [vp, _dp, _dp, sz, _dp, _dp, sz, C.POINTER(vp)]
            self.lib.gple_objective_eval.argtypes = [vp, _dp, sz, _dp, _dp]
            self.lib.gple_objective_release.argtypes = [vp]
            self.lib.gple_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
            self.lib.gple_ctx_destroy.argtypes = [C.c_void_p]
            self.lib.gple_ctx_synchronize.argtypes = [C.c_void_p]
            self.lib.gple_ctx_trim.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
            self.lib.gple_status_string.argtypes, self.lib.gple_status_string.restype = [C.c_int], C.c_char_p
            self.lib.gple_ctx_last_error.argtypes, self.lib.gple_ctx_last_error.restype = [C.c_void_p], C.c_char_p

    def trim(self):
        """Free the pooled device buffers no live fit owns; returns the number of bytes released."""
        n = C.c_size_t(0)
        self._check(self.lib.gple_ctx_trim(self.ctx, C.byref(n)))
        return n.value

    def _check(self, status):
        if status != GPLE_OK:
            msg = f"status {status}"
            if self.with_ctx:
                msg = self.lib.gple_status_string(status).decode()
                if self.ctx:
                    msg += ": " + self.lib.gple_ctx_last_error(self.ctx).decode()
            raise GpleError(msg)

    def _c(self):
        return [self.ctx] if self.with_ctx else []

    def _fl(self, flags=0):
        return [flags] if self.with_ctx else []

    def close(self):
        for f in list(self._fits):
            f.release()
        if self.ctx:
            self.lib.gple_ctx_destroy(self.ctx)
            self.ctx = None

    def enable_timing(self, on=True):
        self.lib.gple_ctx_enable_timing.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.gple_ctx_enable_timing(self.ctx, int(on)))

    def timing(self, which):
        """(last_ms, total_ms, count) of timer 0 = fit, 1 = predict call, 2 = fused predict kernel."""
        last, total, count = C.c_double(), C.c_double(), C.c_long()
        self.lib.gple_ctx_get_timing.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                 C.POINTER(C.c_long)]
        self._check(self.lib.gple_ctx_get_timing(self.ctx, which, C.byref(last), C.byref(total), C.byref(count)))
        return last.value, total.value, count.value

    def last_contraction_kernel(self):
        """Name of the kernel this context's last predict ran its variance contraction on (csrc/gple_debug.h; the roofline label of bench.py)."""
        fn = self.lib.gple_debug_last_contraction_kernel
        fn.argtypes, fn.restype = [C.c_void_p], C.c_char_p
        return (fn(self.ctx) or b"").decode()

    def prune_stats(self, reset=False):
        """(contracted, seen) test rows of the pruned predicts in units of 128 rows since creation / the last reset."""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        self._check(self.lib.gple_ctx_get_prune_stats(self.ctx, C.byref(a), C.byref(b), int(bool(reset))))
        return a.value, b.value

    def synchronize(self):
        if self.ctx:
            self._check(self.lib.gple_ctx_synchronize(self.ctx))

    def _fit_get(self, fit, which, buf):
        self._check(self._fn(f"{fit.kind}_fit_get")(fit.handle, which, *self._fl(), _ptr(buf)))

    # ---- KernelBase ------------------------------------------------------------------------------------------
    def real_gram(self, theta, left, right, same_features=False, derivative=False):
        theta, left, right = _f64(theta), _points(left), _points(right)
        R, Cc = len(left), len(right)
        K = np.empty(R * Cc)
        dK = np.empty(4 * R * Cc) if derivative else None
        self._check(self._fn("real_gram")(*self._c(), _ptr(theta), _ptr(left), R, _ptr(right), Cc, int(same_features),
                                          *self._fl(), _ptr(K), _ptr(dK)))
        K = K.reshape(Cc, R).T  # column-major R x C
        if derivative:
            return K, dK.reshape(4, Cc, R).transpose(0, 2, 1)
        return K

    def complex_gram(self, theta, left, right, same_features=False, derivative=False):
        """ComplexKernelBase: (K, Kt) or (K, Kt, dK[8], dKt[8]), R x C arrays"""
        theta, left, right = _f64(theta), _points(left), _points(right)
        R, Cc = len(left), len(right)
        K, Kt = np.empty(R * Cc), np.empty(2 * R * Cc)
        dK = np.empty(8 * R * Cc) if derivative else None
        dKt = np.empty(16 * R * Cc) if derivative else None
        self._check(self._fn("complex_gram")(*self._c(), _ptr(theta), _ptr(left), R, _ptr(right), Cc, int(same_features),
                                             *self._fl(), _ptr(K), _ptr(Kt), _ptr(dK), _ptr(dKt)))
        K, Kt = K.reshape(Cc, R).T, Kt.view(np.complex128).reshape(Cc, R).T
        if derivative:
            return K, Kt, dK.reshape(8, Cc, R).transpose(0, 2, 1), dKt.view(np.complex128).reshape(8, Cc, R).transpose(0, 2, 1)
        return K, Kt

    def predict_batch(self, elements, points, element_of_request):
        """gple_predict_batch: elements = list of _Fit or None; points (n,2); element_of_request (n,) ints -> complex (n,)"""
        arr = (Element * max(1, len(elements)))()
        for i, f in enumerate(elements):
            if f is not None:
                setattr(arr[i], "real" if f.kind == "real" else "cplx", f.handle.value)
        pts = _points(points)
        idx = np.ascontiguousarray(element_of_request, dtype=np.int32)
        out = np.empty(2 * len(pts))
        self._check(self.lib.gple_predict_batch(self.ctx, arr, len(elements), _ptr(pts), idx.ctypes.data_as(C.POINTER(C.c_int)), len(pts), _ptr(out)))
        return out.view(np.complex128)

    def _elements(self, fits):
        arr = (Element * max(1, len(fits)))()
        for i, f in enumerate(fits):
            if f is not None:
                setattr(arr[i], "real" if f.kind == "real" else "cplx", f.handle.value)
        return arr

    def pes_adiabatic(self, model, x):
        """(M, 6): E0, E1, F00, F10, F11, NAC01 of Tully's model `model` (0 SAC, 1 DAC, 2 ECR) at positions x"""
        x = _f64(x)
        out = np.empty(6 * len(x))
        self.lib.gple_pes_adiabatic.argtypes = [C.c_void_p, C.c_int, _dp, C.c_size_t, C.c_uint, _dp]
        self._check(self.lib.gple_pes_adiabatic(self.ctx, int(model), _ptr(x), len(x), 0, _ptr(out)))
        return out.reshape(-1, 6)

    def evolve(self, fits, model, mass, dt, density, new_points=False):
        """one tick of evolve(): fits = [fit(0,0) | None, fit(1,0) | None, fit(1,1) | None]; density = {(i, j): (r (n,2), rho (n,))}
        -> the same structure one tick later.  new_points: new_point_predict (evolve.cpp:425-443) at the given points instead — they
        stay where they are and rho becomes what the back-propagation predicts there from the fits alone (0 where uncoupled)."""
        order = [(0, 0), (1, 0), (1, 1)]
        rs = [np.ascontiguousarray(np.asarray(density[e][0], dtype=np.float64).reshape(-1, 2)).copy() for e in order]
        rhos = [np.ascontiguousarray(np.asarray(density[e][1], dtype=np.complex128)).copy() for e in order]
        pts = (Points * 3)()
        for k in range(3):
            pts[k].r, pts[k].rho, pts[k].n = _ptr(rs[k]), _ptr(rhos[k].view(np.float64)), len(rs[k])
        self.lib.gple_evolve.argtypes = [C.c_void_p, C.POINTER(Element), C.c_int, C.c_double, C.c_double, C.POINTER(Points), C.c_uint]
        self._check(self.lib.gple_evolve(self.ctx, self._elements(fits), int(model), float(mass), float(dt), pts, EVOLVE_NEW_POINTS if new_points else 0))
        return {e: (rs[k], rhos[k]) for k, e in enumerate(order)}

    def pes_adiabatic_n(self, num_pes, model, x):
        """N-level adiabatic quantities at positions x: (E (M, N), F (M, N, N) symmetric, NAC (M, N, N) antisymmetric, NAC[j, k] = F[j, k] / (E_j - E_k))"""
        x = _f64(x)
        ne = num_pes * (num_pes + 1) // 2
        w = num_pes + 2 * ne
        out = np.empty(w * len(x))
        self.lib.gple_pes_adiabatic_n.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, C.c_size_t, C.c_uint, _dp]
        self._check(self.lib.gple_pes_adiabatic_n(self.ctx, int(num_pes), int(model), _ptr(x), len(x), 0, _ptr(out)))
        out = out.reshape(-1, w)
        E, F, NAC = out[:, :num_pes].copy(), np.zeros((len(x), num_pes, num_pes)), np.zeros((len(x), num_pes, num_pes))
        e = 0
        for k in range(num_pes):
            for l in range(k + 1):
                F[:, k, l] = F[:, l, k] = out[:, num_pes + e]
                NAC[:, k, l], NAC[:, l, k] = out[:, num_pes + ne + e], -out[:, num_pes + ne + e]
                e += 1
        return E, F, NAC

    def evolve_n(self, num_pes, fits, model, mass, dt, density, new_points=False):
        """gple_evolve_n: one tick for an N-level system; fits and density in the packing order (0,0), (1,0), (1,1), (2,0), ...;
        density = {(i, j): (r (n, 2), rho (n,))} -> the same structure one tick later"""
        order = [(i, j) for i in range(num_pes) for j in range(i + 1)]
        empty = (np.zeros((0, 2)), np.zeros(0, dtype=complex))
        rs = [np.ascontiguousarray(np.asarray(density.get(e, empty)[0], dtype=np.float64).reshape(-1, 2)).copy() for e in order]
        rhos = [np.ascontiguousarray(np.asarray(density.get(e, empty)[1], dtype=np.complex128)).copy() for e in order]
        pts = (Points * len(order))()
        for k in range(len(order)):
            pts[k].r, pts[k].rho, pts[k].n = _ptr(rs[k]), _ptr(rhos[k].view(np.float64)), len(rs[k])
        self.lib.gple_evolve_n.argtypes = [C.c_void_p, C.c_int, C.POINTER(Element), C.c_int, C.c_double, C.c_double, C.POINTER(Points), C.c_uint]
        self._check(self.lib.gple_evolve_n(self.ctx, int(num_pes), self._elements(fits), int(model), float(mass), float(dt), pts,
                                           EVOLVE_NEW_POINTS if new_points else 0))
        return {e: (rs[k], rhos[k]) for k, e in enumerate(order)}

    def markov_chain(self, fit, num_steps, max_displacement, seed, r, want_chain=False):
        """Metropolis chains of all walkers on |cut-off prediction| of `fit`: (last points (n,2), acceptance ratio (n,)); with
        want_chain also the whole chains (num_steps + 1, n, 2)"""
        r = np.ascontiguousarray(np.asarray(r, dtype=np.float64).reshape(-1, 2)).copy()
        acc = np.empty(len(r))
        if want_chain:
            chain = np.empty((int(num_steps) + 1, len(r), 2))
            self.lib.gple_markov_chain_trace.argtypes = [C.c_void_p, C.POINTER(Element), C.c_size_t, C.c_double, C.c_ulonglong, _dp, C.c_size_t, _dp, _dp]
            self._check(self.lib.gple_markov_chain_trace(self.ctx, self._elements([fit]), int(num_steps), float(max_displacement), int(seed), _ptr(r), len(r),
                                                         _ptr(acc), _ptr(chain)))
            return r, acc, chain
        self.lib.gple_markov_chain.argtypes = [C.c_void_p, C.POINTER(Element), C.c_size_t, C.c_double, C.c_ulonglong, _dp, C.c_size_t, _dp]
        self._check(self.lib.gple_markov_chain(self.ctx, self._elements([fit]), int(num_steps), float(max_displacement), int(seed), _ptr(r), len(r), _ptr(acc)))
        return r, acc

    def cutoff_factor(self, prediction, variance):
        is_c = np.iscomplexobj(prediction)
        p = _cplx(prediction).view(np.float64) if is_c else _f64(prediction)
        var = _f64(variance)
        out = np.empty(len(var))
        self._check(self._fn("cutoff_factor")(*self._c(), _ptr(p), int(is_c), _ptr(var), len(var), *self._fl(), _ptr(out)))
        return out

    # ---- TrainingKernel / PredictiveKernel -------------------------------------------------------------------
    def real_fit(self, theta, X, y, flags, defer_scalars=False):
        defer_scalars = defer_scalars and self.with_ctx  # the oracle computes everything at once
        theta, X = _f64(theta), _points(X)
        is_c = np.iscomplexobj(y)
        yy = _cplx(y).view(np.float64) if is_c else _f64(y)
        sc, h = RealFitScalars(), C.c_void_p()
        self._check(self._fn("real_fit_create")(*self._c(), _ptr(theta), _ptr(X), _ptr(yy), int(is_c), len(X), flags,
                                                None if defer_scalars else C.byref(sc), C.byref(h)))
        return _Fit(self, h, "real", len(X), None if defer_scalars else scalars_to_dict(sc))

    def real_predict(self, fit, Xs, flags=0, labels=None, want=("prediction", "variance", "cutoff")):
        Xs = _points(Xs)
        M = len(Xs)
        lab = None if labels is None else _f64(labels)
        pred = np.empty(M) if "prediction" in want else None
        var = np.empty(M) if "variance" in want else None
        cut = np.empty(M) if "cutoff" in want else None
        ps = PredictScalars()
        self._check(self._fn("real_predict")(*self._c(), fit.handle, _ptr(Xs), M, flags, _ptr(lab), _ptr(pred), _ptr(var),
                                             _ptr(cut), C.byref(ps)))
        d = scalars_to_dict(ps)
        d["error_derivative"] = d["error_derivative"][:4]
        d.update(prediction=pred, variance=var, cutoff=cut)
        return d

    # ---- TrainingComplexKernel / PredictiveComplexKernel ------------------------------------------------------
    def complex_fit(self, theta, X, y, flags, defer_scalars=False):
        defer_scalars = defer_scalars and self.with_ctx
        theta, X, yy = _f64(theta), _points(X), _cplx(y).view(np.float64)
        sc, h = ComplexFitScalars(), C.c_void_p()
        self._check(self._fn("complex_fit_create")(*self._c(), _ptr(theta), _ptr(X), _ptr(yy), len(X), flags,
                                                   None if defer_scalars else C.byref(sc), C.byref(h)))
        return _Fit(self, h, "complex", len(X), None if defer_scalars else scalars_to_dict(sc))

    def complex_predict(self, fit, Xs, flags=0, labels=None, want=("prediction", "variance", "cutoff")):
        Xs = _points(Xs)
        M = len(Xs)
        lab = None if labels is None else _cplx(labels).view(np.float64)
        pred = np.empty(2 * M) if "prediction" in want else None
        var = np.empty(M) if "variance" in want else None
        cut = np.empty(2 * M) if "cutoff" in want else None
        ps = PredictScalars()
        self._check(self._fn("complex_predict")(*self._c(), fit.handle, _ptr(Xs), M, flags, _ptr(lab), _ptr(pred),
                                                _ptr(var), _ptr(cut), C.byref(ps)))
        d = scalars_to_dict(ps)
        d.update(prediction=None if pred is None else pred.view(np.complex128), variance=var,
                 cutoff=None if cut is None else cut.view(np.complex128))
        return d

    # ---- objective / NLML ------------------------------------------------------------------------------------
    def loose_function(self, x, X, y, X_extra, y_extra, want_grad=True):
        x, X, Xe = _f64(x), _points(X), _points(X_extra)
        yy, ye = _cplx(y).view(np.float64), _cplx(y_extra).view(np.float64)
        val = C.c_double()
        grad = np.empty(len(x)) if want_grad else None
        self._check(self._fn("loose_function")(*self._c(), _ptr(x), len(x), _ptr(X), _ptr(yy), len(X), _ptr(Xe), _ptr(ye),
                                               len(Xe), C.cast(C.byref(val), _dp), _ptr(grad)))
        return val.value, grad

    def objective(self, X, y, X_extra, y_extra):
        """loose_function with its data resident on the device: returns f(x, want_grad=True) -> (value, grad or None).  The
        oracle has no device: there the closure simply keeps the arrays."""
        X, Xe = _points(X), _points(X_extra)
        yy, ye = _cplx(y), _cplx(y_extra)
        if not self.with_ctx:
            return lambda x, want_grad=True: self.loose_function(x, X, yy, Xe, ye, want_grad=want_grad)
        return _Objective(self, X, yy, Xe, ye)

    def nlml(self, x, X, y, want_grad=True):
        """len(x) == 4: (w_d, w_g, a_x, a_p), the NOCROSS build; len(x) == 5: (w_d, w_g, a, c, b), the lower-triangular ARD weight
        matrix of the default build of test/gpr.cpp"""
        x, X, y = _f64(x), _points(X), _f64(y)
        assert len(x) in (4, 5)
        val = C.c_double()
        grad = np.empty(len(x)) if want_grad else None
        self._check(self._fn("nlml" if len(x) == 4 else "nlml_cross")(*self._c(), _ptr(x), _ptr(X), _ptr(y), len(X), C.cast(C.byref(val), _dp), _ptr(grad)))
        return val.value, grad

    def nlml_predict(self, x, X, y, Xs):
        x, X, y, Xs = _f64(x), _points(X), _f64(y), _points(Xs)
        assert len(x) in (4, 5)
        out = np.empty(len(Xs))
        self._check(self._fn("nlml_predict" if len(x) == 4 else "nlml_cross_predict")(*self._c(), _ptr(x), _ptr(X), _ptr(y), len(X), _ptr(Xs), len(Xs),
                                                                                       *self._fl(), _ptr(out)))
        return out
