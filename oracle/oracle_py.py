"""ctypes binding of the CPU oracle (oracle/liboracle.so) and, when present, of the real reference
build (oracle/_ref/libref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under optimization_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_LIB_OMP = os.path.join(_HERE, "liboracle_omp.so")  # same sources, OpenMP loops: bench.py's all-cores CPU leg only
_REF = os.path.join(_HERE, "_ref", "libref.so")

c_double_p = C.POINTER(C.c_double)
c_size_p = C.POINTER(C.c_size_t)
c_int_p = C.POINTER(C.c_int)

APPLY_FN = C.CFUNCTYPE(None, C.c_void_p, c_double_p, c_double_p)
INNER_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, c_double_p, c_double_p)
MATOP_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, c_double_p, c_double_p)


def build(force=False):
    """Compile liboracle.so (and _ref/libref.so when /root/reference is mounted)."""
    if force or not os.path.exists(_LIB) or not os.path.exists(_LIB_OMP) or _stale():
        subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)
    return _LIB


def _stale():
    try:
        t = min(os.path.getmtime(_LIB), os.path.getmtime(_LIB_OMP))
        return any(os.path.getmtime(os.path.join(_HERE, f)) > t
                   for f in ("oracle.c", "problems.c", "oracle.h"))
    except OSError:
        return True


class Problem(C.Structure):
    _fields_ = [
        ("nvar", C.c_size_t), ("ntan", C.c_size_t), ("user", C.c_void_p),
        ("f", C.CFUNCTYPE(C.c_double, C.c_void_p, c_double_p)),
        ("grad", C.CFUNCTYPE(None, C.c_void_p, c_double_p, c_double_p)),
        ("hess", C.CFUNCTYPE(None, C.c_void_p, c_double_p, c_double_p, c_double_p)),
        ("metric", C.CFUNCTYPE(C.c_double, C.c_void_p, c_double_p, c_double_p, c_double_p)),
        ("retract", C.CFUNCTYPE(None, C.c_void_p, c_double_p, c_double_p, c_double_p)),
        ("precon", C.CFUNCTYPE(None, C.c_void_p, c_double_p, c_double_p, c_double_p)),
        ("destroy", C.CFUNCTYPE(None, C.c_void_p)),
        ("n_f", C.c_size_t), ("n_grad", C.c_size_t), ("n_hess", C.c_size_t),
        ("n_metric", C.c_size_t), ("n_retract", C.c_size_t), ("n_precon", C.c_size_t),
    ]


class StpcgTrace(C.Structure):
    _fields_ = [("cap", C.c_size_t), ("len", C.c_size_t), ("alpha", c_double_p),
                ("beta", c_double_p), ("kappa", c_double_p), ("rv", c_double_p)]


class TntParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_size_t), ("max_computation_time", C.c_double),
        ("gradient_tolerance", C.c_double), ("relative_decrease_tolerance", C.c_double),
        ("stepsize_tolerance", C.c_double), ("Delta0", C.c_double), ("eta1", C.c_double),
        ("eta2", C.c_double), ("alpha1", C.c_double), ("alpha2", C.c_double),
        ("max_TPCG_iterations", C.c_size_t), ("kappa_fgr", C.c_double), ("theta", C.c_double),
        ("preconditioned_gradient_tolerance", C.c_double), ("Delta_tolerance", C.c_double),
    ]


class TntResult(C.Structure):
    _fields_ = [
        ("x", c_double_p), ("f", C.c_double), ("gradfx_norm", C.c_double),
        ("preconditioned_gradfx_norm", C.c_double), ("status", C.c_int),
        ("outer_iterations", C.c_size_t), ("n_trace", C.c_size_t),
        ("objective_values", c_double_p), ("gradient_norms", c_double_p),
        ("preconditioned_gradient_norms", c_double_p), ("trust_region_radius", c_double_p),
        ("inner_iterations", c_size_p), ("update_step_norms", c_double_p),
        ("update_step_M_norms", c_double_p), ("gain_ratios", c_double_p), ("accepted", C.c_size_t),
    ]


TNT_STATUS = ["Gradient", "PreconditionedGradient", "RelativeDecrease", "Stepsize", "TrustRegion",
              "IterationLimit", "ElapsedTime", "UserFunction"]


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


class _Lib:
    def __init__(self, path, prefix, with_problems):
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        self.stpcg_fn = getattr(L, prefix + "_stpcg")
        self.stpcg_fn.restype = C.c_int
        self.stpcg_fn.argtypes = [C.c_size_t, c_double_p, APPLY_FN, C.c_void_p, INNER_FN, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_double, C.c_size_t, C.c_double,
                                  C.c_double, C.c_double, c_double_p, c_double_p, c_size_p, c_int_p,
                                  C.POINTER(StpcgTrace)]
        self.tnt_fn = getattr(L, prefix + "_tnt")
        self.tnt_fn.restype = C.c_int
        self.tnt_fn.argtypes = [C.POINTER(Problem), c_double_p, C.POINTER(TntParams),
                                C.POINTER(TntResult)]
        if hasattr(L, prefix + "_gd"):  # template drivers (oracle/template_driver.inc)
            self.gd_fn = getattr(L, prefix + "_gd")
            self.gd_fn.restype = C.c_int
            self.gd_fn.argtypes = [C.POINTER(Problem), c_double_p, C.c_size_t, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_double, C.c_double, C.c_size_t, c_double_p,
                                   c_double_p, c_double_p, c_int_p, c_size_p, C.c_size_t, c_double_p,
                                   c_size_p]
            self.lsqr_fn = getattr(L, prefix + "_lsqr_dense")
            self.lsqr_fn.restype = C.c_int
            self.lsqr_fn.argtypes = [C.c_size_t, C.c_size_t, c_double_p, c_double_p, C.c_size_t, C.c_double,
                                     C.c_double, C.c_double, C.c_double, C.c_double, c_double_p, c_double_p,
                                     c_size_p]
            self.tnls_fn = getattr(L, prefix + "_tnls_sinfit")
            self.tnls_affine_fn = getattr(L, prefix + "_tnls_affine")
            self.tnls_affine_fn.restype = C.c_int
            self.tnls_affine_fn.argtypes = [C.c_size_t, C.c_size_t, c_double_p, c_double_p, c_double_p, C.c_double,
                                            C.c_double, C.c_size_t, C.c_size_t, c_double_p, c_double_p,
                                            c_double_p, C.POINTER(C.c_int), c_size_p, c_size_p]
            self.tnls_fn.restype = C.c_int
            self.tnls_fn.argtypes = [C.c_size_t, c_double_p, c_double_p, c_double_p, C.c_int, C.c_double,
                                     C.c_double, C.c_double, C.c_size_t, c_double_p, c_double_p, c_double_p,
                                     c_int_p, c_size_p, c_size_p]
        if with_problems:
            L.orc_tnt_default_params.argtypes = [C.POINTER(TntParams)]
            L.orc_problem_free.argtypes = [C.POINTER(Problem)]
            for name, args in {
                "orc_problem_sphere": [c_double_p, C.c_int],
                "orc_problem_rosenbrock": [C.c_size_t, C.c_int],
                "orc_problem_diag_quadratic": [C.c_size_t, c_double_p, c_double_p, c_double_p],
                "orc_problem_stiefel_rq": [C.c_size_t, C.c_size_t, c_int_p, c_int_p, c_double_p,
                                           c_double_p],
                "orc_problem_so3n": [C.c_size_t, C.c_size_t, c_int_p, c_int_p, c_double_p,
                                     c_double_p, C.c_int],
            }.items():
                fn = getattr(L, name)
                fn.restype = C.POINTER(Problem)
                fn.argtypes = args
            L.orc_rayleigh_ritz.restype = C.c_int
            L.orc_rayleigh_ritz.argtypes = [C.c_size_t, c_double_p, c_double_p, c_double_p,
                                            c_double_p]
            L.orc_lobpcg.restype = C.c_int
            L.orc_lobpcg.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_double_p,
                                     c_double_p, C.c_size_t, C.c_double, c_double_p, c_double_p,
                                     c_size_p, c_size_p, c_double_p]
            L.orc_csr_spmm.argtypes = [C.c_size_t, C.c_size_t, c_int_p, c_int_p, c_double_p,
                                       c_double_p, c_double_p]
            L.orc_sym3_invsqrt.argtypes = [c_double_p, c_double_p]
            L.orc_so3_exp.argtypes = [c_double_p, c_double_p]

    # ---- STPCG with numpy callables -------------------------------------------------------
    def stpcg(self, g, H, P=None, inner=None, Delta=1.0, max_iterations=1000, kappa_fgr=.1,
              theta=.5, epsilon=1e-8, trace_cap=0):
        """H, P: callables numpy(n)->numpy(n).  inner: callable (a,b)->float (default: sequential dot
        done in C order via np.dot is NOT used -- we sum sequentially to match the C oracle)."""
        g = np.ascontiguousarray(g, dtype=np.float64)
        n = g.size

        def wrap_apply(fn):
            def cb(_u, pin, pout):
                a = np.ctypeslib.as_array(pin, shape=(n,))
                o = np.ctypeslib.as_array(pout, shape=(n,))
                o[:] = fn(a)
            return APPLY_FN(cb)

        if inner is None:
            def inner(a, b):
                return _seq_dot(a, b)

        def ipcb(_u, pa, pb):
            a = np.ctypeslib.as_array(pa, shape=(n,))
            b = np.ctypeslib.as_array(pb, shape=(n,))
            return float(inner(a, b))

        Hc = wrap_apply(H)
        Pc = wrap_apply(P) if P is not None else None
        ipc = INNER_FN(ipcb)
        s = np.zeros(n)
        mnorm = C.c_double(0)
        iters = C.c_size_t(0)
        reason = C.c_int(-1)
        tr = None
        arrs = {}
        if trace_cap:
            arrs = {k: np.zeros(trace_cap) for k in ("alpha", "beta", "kappa", "rv")}
            tr = StpcgTrace(trace_cap, 0, _dp(arrs["alpha"]), _dp(arrs["beta"]), _dp(arrs["kappa"]),
                            _dp(arrs["rv"]))
        rc = self.stpcg_fn(n, _dp(g), Hc, None, ipc, None,
                           C.cast(Pc, C.c_void_p) if Pc is not None else None, None, Delta,
                           max_iterations, kappa_fgr, theta, epsilon, _dp(s), C.byref(mnorm),
                           C.byref(iters), C.byref(reason), C.byref(tr) if tr else None)
        out = dict(rc=rc, s=s, M_norm=mnorm.value, iterations=iters.value, exit_reason=reason.value)
        if tr:
            out["trace"] = {k: v[:tr.len].copy() for k, v in arrs.items()}
        return out

    # ---- TNT on a C problem handle ---------------------------------------------------------
    def tnt(self, prob, x0, params):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        p = prob.contents
        cap = params.max_iterations + 2
        bufs = dict(x=np.zeros(p.nvar), objective_values=np.zeros(cap), gradient_norms=np.zeros(cap),
                    preconditioned_gradient_norms=np.zeros(cap), trust_region_radius=np.zeros(cap),
                    inner_iterations=np.zeros(cap, dtype=np.uint64), update_step_norms=np.zeros(cap),
                    update_step_M_norms=np.zeros(cap), gain_ratios=np.zeros(cap))
        res = TntResult()
        res.x = _dp(bufs["x"])
        for k in ("objective_values", "gradient_norms", "preconditioned_gradient_norms",
                  "trust_region_radius", "update_step_norms", "update_step_M_norms", "gain_ratios"):
            setattr(res, k, _dp(bufs[k]))
        res.inner_iterations = bufs["inner_iterations"].ctypes.data_as(c_size_p)
        rc = self.tnt_fn(prob, _dp(x0), C.byref(params), C.byref(res))
        if rc:
            return dict(rc=rc)
        nt, no = res.n_trace, res.outer_iterations
        return dict(
            rc=0, x=bufs["x"], f=res.f, gradfx_norm=res.gradfx_norm,
            preconditioned_gradfx_norm=res.preconditioned_gradfx_norm, status=res.status,
            status_name=TNT_STATUS[res.status], outer_iterations=no, accepted=res.accepted,
            objective_values=bufs["objective_values"][:nt].copy(),
            gradient_norms=bufs["gradient_norms"][:nt].copy(),
            preconditioned_gradient_norms=bufs["preconditioned_gradient_norms"][:nt].copy(),
            trust_region_radius=bufs["trust_region_radius"][:nt].copy(),
            inner_iterations=bufs["inner_iterations"][:no].astype(np.int64),
            update_step_norms=bufs["update_step_norms"][:no].copy(),
            update_step_M_norms=bufs["update_step_M_norms"][:no].copy(),
            gain_ratios=bufs["gain_ratios"][:no].copy(),
            calls=dict(f=p.n_f, grad=p.n_grad, hess=p.n_hess, metric=p.n_metric,
                       retract=p.n_retract, precon=p.n_precon))


    # ---- template drivers: GradientDescent / LSQR / TNLS -------------------------------------
    def gd(self, prob, x0, max_iterations=1000, gradient_tolerance=1e-6, relative_decrease_tolerance=1e-6,
           stepsize_tolerance=1e-6, alpha=1.0, beta=.5, sigma=.5, max_ls_iterations=100, cap=4096):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        p = prob.contents
        x = np.zeros(p.nvar)
        f, gn = C.c_double(0), C.c_double(0)
        st, it = C.c_int(-1), C.c_size_t(0)
        fv = np.zeros(cap)
        ls = np.zeros(cap, dtype=np.uint64)
        rc = self.gd_fn(prob, _dp(x0), max_iterations, gradient_tolerance, relative_decrease_tolerance,
                        stepsize_tolerance, alpha, beta, sigma, max_ls_iterations, _dp(x), C.byref(f),
                        C.byref(gn), C.byref(st), C.byref(it), cap, _dp(fv), ls.ctypes.data_as(c_size_p))
        n = min(it.value, cap)
        return dict(rc=rc, x=x, f=f.value, gradfx_norm=gn.value, status=st.value, iterations=it.value,
                    objective_values=fv[:n].copy(), linesearch_iterations=ls[:n].astype(np.int64))

    def lsqr_dense(self, A, b, max_iterations=1000, lam=0.0, btol=1e-6, Atol=1e-6, Acond_limit=1e8,
                   Delta=None):
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        m, n = A.shape
        if Delta is None:
            Delta = float(np.sqrt(np.finfo(np.float64).max))
        x = np.zeros(n)
        xn, it = C.c_double(0), C.c_size_t(0)
        rc = self.lsqr_fn(m, n, _dp(A), _dp(b), max_iterations, lam, btol, Atol, Acond_limit, Delta, _dp(x),
                          C.byref(xn), C.byref(it))
        return dict(rc=rc, x=x, xnorm=xn.value, iterations=it.value)

    def tnls_sinfit(self, t, y, beta0, with_precon=False, root_tolerance=1e-6, gradient_tolerance=0.0,
                    Delta_tolerance=0.0, max_iterations=100):
        t = np.ascontiguousarray(t, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        b0 = np.ascontiguousarray(beta0, dtype=np.float64)
        beta = np.zeros(2)
        f, gn = C.c_double(0), C.c_double(0)
        st, outer, inner = C.c_int(-1), C.c_size_t(0), C.c_size_t(0)
        rc = self.tnls_fn(t.size, _dp(t), _dp(y), _dp(b0), int(with_precon), root_tolerance,
                          gradient_tolerance, Delta_tolerance, max_iterations, _dp(beta), C.byref(f),
                          C.byref(gn), C.byref(st), C.byref(outer), C.byref(inner))
        return dict(rc=rc, beta=beta, f=f.value, gradfx_norm=gn.value, status=st.value, outer=outer.value,
                    inner_total=inner.value)


def _tnls_affine(self, A, b, x0, root_tolerance=1e-9, gradient_tolerance=0.0, max_iterations=20,
                 max_LSQR_iterations=1000):
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    b = np.ascontiguousarray(b, dtype=np.float64)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    x = np.zeros(n)
    f, gn = np.zeros(1), np.zeros(1)
    st, outer, inner = C.c_int(-1), C.c_size_t(0), C.c_size_t(0)
    rc = self.tnls_affine_fn(m, n, _dp(A), _dp(b), _dp(x0), root_tolerance, gradient_tolerance, max_iterations,
                             max_LSQR_iterations, _dp(x), _dp(f), _dp(gn), C.byref(st), C.byref(outer),
                             C.byref(inner))
    return dict(rc=rc, x=x, f=float(f[0]), gradfx_norm=float(gn[0]), status=st.value, outer=outer.value,
                inner_total=inner.value)


_Lib.tnls_affine = _tnls_affine


def _seq_dot(a, b):
    # strictly sequential summation, like the C oracle (np.dot uses pairwise/BLAS order)
    s = 0.0
    for x, y in zip(a.tolist(), b.tolist()):
        s += x * y
    return s


def projected_stpcg_problem(case, n=1000, m=100):
    """Seeded inputs of the reference's two equality-constrained STPCG tests
    (tests/IterativeSolvers_unit_test.cpp:316-496; Eigen's Random -> numpy PCG64): g in [-1,1]^n, diagonal
    Hessian and diagonal M in [1000,3000], constraint matrix 1000 * U(-1,1)^(m x n).
    case "exact": kappa 1e-8; "truncated": kappa .1 (theta .7, Delta DBL_MAX, max 5 n iterations both)."""
    rng = np.random.Generator(np.random.PCG64({"exact": 316, "truncated": 413}[case]))
    g = rng.uniform(-1, 1, n)
    P = 2000 + 1000 * rng.uniform(-1, 1, n)
    M = 2000 + 1000 * rng.uniform(-1, 1, n)
    A = np.ascontiguousarray(1000 * rng.uniform(-1, 1, (m, n)))
    return dict(n=n, m=m, g=g, P=P, M=M, A=A, Delta=float(np.finfo(np.float64).max), max_iterations=5 * n,
                kappa={"exact": 1e-8, "truncated": .1}[case], theta=.7)


def projected_stpcg_sparse_problem(n=2000, m=150, per_row=8, seed=904):
    """A SPARSE equality-constrained STPCG case (r04; the reference only tests dense constraints): every constraint
    couples `per_row` random unknowns plus one of its own (unknown 13 a, so the rows are independent), diagonal Hessian
    and diagonal M as in projected_stpcg_problem.  A is returned dense (the reference driver takes it that way)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = rng.uniform(-1, 1, n)
    P = 2000 + 1000 * rng.uniform(-1, 1, n)
    M = 2000 + 1000 * rng.uniform(-1, 1, n)
    A = np.zeros((m, n))
    for a in range(m):
        cols = rng.choice(n, size=per_row, replace=False)
        A[a, cols] = 1000 * rng.uniform(-1, 1, per_row)
        A[a, (13 * a) % n] += 3000.0
    return dict(n=n, m=m, g=g, P=P, M=M, A=np.ascontiguousarray(A), Delta=float(np.finfo(np.float64).max),
                max_iterations=5 * n, kappa=1e-9, theta=.7)


def stpcg_projected(lib, prefix, pr):
    """<prefix>_stpcg_projected of a template-driver library (oracle/template_driver.inc)"""
    fn = getattr(lib, prefix + "_stpcg_projected")
    fn.restype = C.c_int
    fn.argtypes = [C.c_size_t, C.c_size_t, c_double_p, c_double_p, c_double_p, c_double_p, C.c_double,
                   C.c_size_t, C.c_double, C.c_double, c_double_p, c_double_p, c_size_p]
    s = np.zeros(pr["n"])
    mn, it = C.c_double(0), C.c_size_t(0)
    rc = fn(pr["n"], pr["m"], _dp(pr["g"]), _dp(pr["P"]), _dp(pr["M"]), _dp(pr["A"]), pr["Delta"],
            pr["max_iterations"], pr["kappa"], pr["theta"], _dp(s), C.byref(mn), C.byref(it))
    return dict(rc=rc, s=s, M_norm=mn.value, iterations=it.value)


def stpcg_stop_problem(n=200, seed=365):
    """Seeded inputs of the user-function stop test: diagonal SPD Hessian, diagonal preconditioner."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return dict(n=n, g=rng.normal(size=n), D=rng.uniform(1.0, 100.0, n), Minv=1.0 / rng.uniform(1.0, 50.0, n))


def stpcg_diag_stop(lib, prefix, g, D, Minv, stop_at, Delta=1e6, max_iterations=100, kappa=1e-10, theta=1.0):
    """<prefix>_stpcg_diag_stop (template drivers: prefix ref / hz; device harness: prefix hd)"""
    fn = getattr(lib, prefix + "_stpcg_diag_stop")
    fn.restype = C.c_int
    fn.argtypes = [C.c_size_t, c_double_p, c_double_p, c_double_p, C.c_double, C.c_size_t, C.c_double, C.c_double,
                   C.c_size_t, c_double_p, c_double_p, c_size_p, c_size_p]
    n = g.size
    s = np.zeros(n)
    mn, it, calls = C.c_double(0), C.c_size_t(0), C.c_size_t(0)
    rc = fn(n, _dp(g), _dp(D), _dp(Minv) if Minv is not None else None, Delta, max_iterations, kappa, theta,
            stop_at, _dp(s), C.byref(mn), C.byref(it), C.byref(calls))
    return dict(rc=rc, s=s, M_norm=mn.value, iterations=it.value, calls=calls.value)


class Oracle(_Lib):
    def __init__(self, omp=False):
        """omp=True: the OpenMP build (vector loops and row loops in parallel; sums re-associated, so NOT the
        bit-for-bit restatement -- used only as the all-cores CPU baseline)."""
        build()
        super().__init__(_LIB_OMP if omp else _LIB, "orc", True)
        self.lib.orc_max_threads.restype = C.c_int

    def set_threads(self, n):
        self.lib.orc_set_threads(int(n))
        return self.lib.orc_max_threads()

    def default_params(self, **kw):
        p = TntParams()
        self.lib.orc_tnt_default_params(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p

    def free(self, prob):
        self.lib.orc_problem_free(prob)

    def sphere(self, P=(0.0, 0.0, 1.0), with_precon=False):
        P = np.asarray(P, dtype=np.float64)
        return self.lib.orc_problem_sphere(_dp(P), int(with_precon))

    def rosenbrock(self, n, precon_kind=0):
        return self.lib.orc_problem_rosenbrock(n, precon_kind)

    def diag_quadratic(self, D, g, Minv=None):
        D = np.ascontiguousarray(D, dtype=np.float64)
        g = np.ascontiguousarray(g, dtype=np.float64)
        Mi = np.ascontiguousarray(Minv, dtype=np.float64) if Minv is not None else None
        return self.lib.orc_problem_diag_quadratic(D.size, _dp(D), _dp(g),
                                                   _dp(Mi) if Mi is not None else None)

    def stiefel_rq(self, n, p, rowptr, col, val, dinv=None):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        di = np.ascontiguousarray(dinv, dtype=np.float64) if dinv is not None else None
        return self.lib.orc_problem_stiefel_rq(n, p, _ip(rowptr), _ip(col), _dp(val),
                                               _dp(di) if di is not None else None)

    def so3n(self, N, ei, ej, Rt, w, precon_kind=0):
        ei = np.ascontiguousarray(ei, dtype=np.int32)
        ej = np.ascontiguousarray(ej, dtype=np.int32)
        Rt = np.ascontiguousarray(Rt, dtype=np.float64)
        w = np.ascontiguousarray(w, dtype=np.float64)
        return self.lib.orc_problem_so3n(N, ei.size, _ip(ei), _ip(ej), _dp(Rt), _dp(w), precon_kind)

    # direct evaluation of a problem's callables (for kernel-level parity tests)
    def eval_f(self, prob, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        p = prob.contents
        return p.f(p.user, _dp(x))

    def eval_grad(self, prob, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        p = prob.contents
        g = np.zeros(p.ntan)
        p.grad(p.user, _dp(x), _dp(g))
        return g

    def eval_hess(self, prob, x, v):
        x = np.ascontiguousarray(x, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        p = prob.contents
        h = np.zeros(p.ntan)
        p.hess(p.user, _dp(x), _dp(v), _dp(h))
        return h

    def eval_retract(self, prob, x, v):
        x = np.ascontiguousarray(x, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        p = prob.contents
        y = np.zeros(p.nvar)
        p.retract(p.user, _dp(x), _dp(v), _dp(y))
        return y

    def eval_precon(self, prob, x, v):
        x = np.ascontiguousarray(x, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        p = prob.contents
        y = np.zeros(p.ntan)
        p.precon(p.user, _dp(x), _dp(v), _dp(y))
        return y

    def stpcg_problem(self, prob, x, g, Delta, max_iterations=1000, kappa_fgr=.1, theta=.5,
                      epsilon=1e-8, trace_cap=0, lib=None):
        """STPCG on the Hessian/metric/precon of `prob` at x (x must be the point of the last
        grad() call so that cached model state is valid), entirely in C (no Python callbacks)."""
        lib = lib or self
        return _stpcg_on_problem(self, lib, prob, x, g, Delta, max_iterations, kappa_fgr, theta,
                                 epsilon, trace_cap)

    def rayleigh_ritz(self, A, B):
        A = np.asfortranarray(A, dtype=np.float64)
        B = np.asfortranarray(B, dtype=np.float64)
        n = A.shape[0]
        Th = np.zeros(n)
        Cm = np.zeros((n, n), order="F")
        rc = self.lib.orc_rayleigh_ritz(n, _dp(A), _dp(B), _dp(Th), _dp(Cm))
        return rc, Th, Cm

    def lobpcg(self, A, B, T, X0, Omega, nev, max_iters, tau=1e-6):
        """A, B, T: callables on (m x k) Fortran-ordered numpy panels (B, T may be None)."""
        X0 = np.asfortranarray(X0, dtype=np.float64)
        Omega = np.asfortranarray(Omega, dtype=np.float64)
        m, nx = X0.shape

        def wrap(fn):
            if fn is None:
                return None

            def cb(_u, mm, k, pin, pout):
                Xi = np.ctypeslib.as_array(pin, shape=(k, mm)).T
                Yo = np.ctypeslib.as_array(pout, shape=(k, mm)).T
                Yo[:, :] = fn(Xi)
            return MATOP_FN(cb)

        Ac, Bc, Tc = wrap(A), wrap(B), wrap(T)
        Th = np.zeros(nev)
        X = np.zeros((m, nev), order="F")
        it = C.c_size_t(0)
        nc = C.c_size_t(0)
        res = np.zeros(nx)
        rc = self.lib.orc_lobpcg(m, nx, nev, C.cast(Ac, C.c_void_p), None,
                                 C.cast(Bc, C.c_void_p) if Bc else None, None,
                                 C.cast(Tc, C.c_void_p) if Tc else None, None, _dp(X0), _dp(Omega),
                                 max_iters, tau, _dp(Th), _dp(X), C.byref(it), C.byref(nc), _dp(res))
        return dict(rc=rc, Theta=Th, X=X, num_iters=it.value, nc=nc.value, residuals=res)


def _stpcg_on_problem(orc, lib, prob, x, g, Delta, max_iterations, kappa_fgr, theta, epsilon,
                      trace_cap):
    x = np.ascontiguousarray(x, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    p = prob.contents
    n = p.ntan
    xp = _dp(x)

    def Hcb(_u, pin, pout):
        p.hess(p.user, xp, pin, pout)

    def Pcb(_u, pin, pout):
        p.precon(p.user, xp, pin, pout)

    def ipcb(_u, pa, pb):
        return p.metric(p.user, xp, pa, pb)

    Hc, ipc = APPLY_FN(Hcb), INNER_FN(ipcb)
    Pc = APPLY_FN(Pcb) if bool(p.precon) else None
    s = np.zeros(n)
    mnorm, iters, reason = C.c_double(0), C.c_size_t(0), C.c_int(-1)
    tr, arrs = None, {}
    if trace_cap:
        arrs = {k: np.zeros(trace_cap) for k in ("alpha", "beta", "kappa", "rv")}
        tr = StpcgTrace(trace_cap, 0, _dp(arrs["alpha"]), _dp(arrs["beta"]), _dp(arrs["kappa"]),
                        _dp(arrs["rv"]))
    rc = lib.stpcg_fn(n, _dp(g), Hc, None, ipc, None,
                      C.cast(Pc, C.c_void_p) if Pc is not None else None, None, Delta,
                      max_iterations, kappa_fgr, theta, epsilon, _dp(s), C.byref(mnorm),
                      C.byref(iters), C.byref(reason), C.byref(tr) if tr else None)
    out = dict(rc=rc, s=s, M_norm=mnorm.value, iterations=iters.value, exit_reason=reason.value)
    if tr:
        out["trace"] = {k: v[:tr.len].copy() for k, v in arrs.items()}
    return out


class Reference(_Lib):
    """The real reference templates (oracle/_ref/libref.so).  Available only if prebuilt."""

    def __init__(self):
        if not os.path.exists(_REF):
            raise FileNotFoundError(_REF)
        super().__init__(_REF, "ref", False)


def lobpcg_dense_template(lib, m, nx, nev, Adiag=None, csr=None, Bdiag=None, Tdiag=None, X0=None, max_iters=1000,
                          tau=1e-6, trace_cap=0):
    """hz_lobpcg_dense: the template layer's GENERIC LOBPCG path on a plain dense host matrix
    (tests/cpp/harness_host.cpp); trace_cap > 0 also returns the per-iteration Ritz values / residual norms."""
    ip32 = C.POINTER(C.c_int32)
    fn = lib.hz_lobpcg_dense
    fn.restype = C.c_int
    fn.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, c_double_p, ip32, ip32, c_double_p, c_double_p, c_double_p,
                   c_double_p, C.c_size_t, C.c_double, c_double_p, c_double_p, c_size_p, c_size_p, c_double_p,
                   c_double_p, C.c_size_t]

    def arr(a):
        return None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    Ad, Bd, Td = arr(Adiag), arr(Bdiag), arr(Tdiag)
    X0f = np.asfortranarray(X0, dtype=np.float64)
    rp = cl = vl = None
    if csr is not None:
        rp = np.ascontiguousarray(csr[0], dtype=np.int32)
        cl = np.ascontiguousarray(csr[1], dtype=np.int32)
        vl = np.ascontiguousarray(csr[2], dtype=np.float64)
    th = np.zeros(nev)
    X = np.zeros((m, nev), order="F")
    it, nc = C.c_size_t(0), C.c_size_t(0)
    tt = np.zeros((max(trace_cap, 1), nx))
    rt = np.zeros((max(trace_cap, 1), nx))
    rc = fn(m, nx, nev, _dp(Ad) if Ad is not None else None, rp.ctypes.data_as(ip32) if rp is not None else None,
            cl.ctypes.data_as(ip32) if cl is not None else None, _dp(vl) if vl is not None else None,
            _dp(Bd) if Bd is not None else None, _dp(Td) if Td is not None else None, _dp(X0f), max_iters, tau,
            _dp(th), _dp(X), C.byref(it), C.byref(nc), _dp(tt) if trace_cap else None,
            _dp(rt) if trace_cap else None, trace_cap)
    k = min(it.value, trace_cap)
    return dict(rc=rc, Theta=th, X=X, num_iters=it.value, nc=nc.value, theta_trace=tt[:k], r_trace=rt[:k])


def have_reference():
    return os.path.exists(_REF)


class TemplateHarness(_Lib):
    """The MI355X build's own template layer instantiated on a host vector through the same driver
    code as Reference (tests/cpp/libharness_host.so, entry points hz_*)."""

    PATH = os.path.join(os.path.dirname(_HERE), "tests", "cpp", "libharness_host.so")

    def __init__(self):
        if not os.path.exists(self.PATH):
            raise FileNotFoundError(self.PATH + " (run __graft_entry__.build())")
        super().__init__(self.PATH, "hz", False)
