"""ctypes access to tests/cpp/libharness_device.so: the drop-in C++ template API exercised with
MI355::DeviceVector (built by optimization_amd.build.build_harness())."""
import ctypes as C
import os

import numpy as np

import oracle_py as op

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "cpp", "libharness_device.so")

dp = C.POINTER(C.c_double)
sp = C.POINTER(C.c_size_t)
ip32 = C.POINTER(C.c_int32)


def _dp(a):
    return a.ctypes.data_as(dp)


class DeviceHarness:
    def __init__(self):
        if not os.path.exists(PATH):
            raise FileNotFoundError(PATH + " (run __graft_entry__.build())")
        L = self.L = C.CDLL(PATH)
        L.hd_last_error.restype = C.c_char_p
        L.hd_last_tnt_syncs.restype = C.c_size_t
        L.hd_last_tnt_seconds.restype = C.c_double
        L.hd_stpcg_diag.restype = C.c_int
        L.hd_stpcg_diag.argtypes = [C.c_size_t, dp, dp, dp, C.c_double, C.c_size_t, C.c_double, C.c_double,
                                    C.c_int, dp, dp, sp]
        L.hd_tnt_stiefel.restype = C.c_int
        L.hd_tnt_stiefel.argtypes = [C.c_size_t, C.c_int, ip32, ip32, dp, dp, C.POINTER(op.TntParams), C.c_int,
                                     C.POINTER(op.TntResult)]
        L.hd_tnt_sphere.restype = C.c_int
        L.hd_tnt_sphere.argtypes = [C.c_int, dp, C.POINTER(op.TntParams), C.POINTER(op.TntResult)]
        L.hd_gd_sphere.restype = C.c_int
        L.hd_gd_sphere.argtypes = [dp, dp, dp, dp, C.POINTER(C.c_int), sp, C.c_size_t, dp, sp]
        L.hd_gd_stiefel.restype = C.c_int
        L.hd_gd_stiefel.argtypes = [C.c_size_t, C.c_int, ip32, ip32, dp, dp, C.c_size_t, C.c_double, C.c_double,
                                    C.c_double, C.c_double, C.c_size_t, C.c_int, dp, dp, dp, C.POINTER(C.c_int), sp,
                                    C.c_size_t, dp, sp, sp]

    def err(self):
        return self.L.hd_last_error().decode()

    def fusion_counters(self):
        """mi_ctx_fusion_counters of the context of the last harness call (hd_last_fusion_counters)"""
        from optimization_amd import capi
        fc = capi.FusionCounters()
        self.L.hd_last_fusion_counters.restype = None
        self.L.hd_last_fusion_counters.argtypes = [C.POINTER(capi.FusionCounters)]
        self.L.hd_last_fusion_counters(C.byref(fc))
        return {k: int(getattr(fc, k)) for k, _ in capi.FusionCounters._fields_}

    def stpcg_diag(self, g, D, Minv, Delta, max_iterations, kappa, theta, mode):
        g = np.ascontiguousarray(g, dtype=np.float64)
        D = np.ascontiguousarray(D, dtype=np.float64)
        Mi = np.ascontiguousarray(Minv, dtype=np.float64) if Minv is not None else None
        s = np.zeros(g.size)
        mn, it = C.c_double(0), C.c_size_t(0)
        rc = self.L.hd_stpcg_diag(g.size, _dp(g), _dp(D), _dp(Mi) if Mi is not None else None, Delta,
                                  max_iterations, kappa, theta, mode, _dp(s), C.byref(mn), C.byref(it))
        return dict(rc=rc, s=s, M_norm=mn.value, iterations=it.value, err=self.err() if rc else "")

    @staticmethod
    def _result_buffers(nvar, params):
        cap = params.max_iterations + 2
        bufs = dict(x=np.zeros(nvar), objective_values=np.zeros(cap), gradient_norms=np.zeros(cap),
                    preconditioned_gradient_norms=np.zeros(cap), trust_region_radius=np.zeros(cap),
                    inner_iterations=np.zeros(cap, dtype=np.uint64), update_step_norms=np.zeros(cap),
                    update_step_M_norms=np.zeros(cap), gain_ratios=np.zeros(cap))
        res = op.TntResult()
        res.x = _dp(bufs["x"])
        for k in ("objective_values", "gradient_norms", "preconditioned_gradient_norms", "trust_region_radius",
                  "update_step_norms", "update_step_M_norms", "gain_ratios"):
            setattr(res, k, _dp(bufs[k]))
        res.inner_iterations = bufs["inner_iterations"].ctypes.data_as(sp)
        return bufs, res

    @staticmethod
    def _unpack(rc, bufs, res, err):
        if rc:
            return dict(rc=rc, err=err)
        nt, no = res.n_trace, res.outer_iterations
        return dict(rc=0, x=bufs["x"], f=res.f, gradfx_norm=res.gradfx_norm, status=res.status,
                    outer_iterations=no, accepted=res.accepted,
                    objective_values=bufs["objective_values"][:nt].copy(),
                    gradient_norms=bufs["gradient_norms"][:nt].copy(),
                    preconditioned_gradient_norms=bufs["preconditioned_gradient_norms"][:nt].copy(),
                    trust_region_radius=bufs["trust_region_radius"][:nt].copy(),
                    inner_iterations=bufs["inner_iterations"][:no].astype(np.int64),
                    update_step_norms=bufs["update_step_norms"][:no].copy(),
                    update_step_M_norms=bufs["update_step_M_norms"][:no].copy(),
                    gain_ratios=bufs["gain_ratios"][:no].copy())

    def tnt_stiefel(self, n, p, rowptr, col, val, X0, params, mode=0):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        X0 = np.ascontiguousarray(X0, dtype=np.float64).ravel()
        bufs, res = self._result_buffers(n * p, params)
        rc = self.L.hd_tnt_stiefel(n, p, rowptr.ctypes.data_as(ip32), col.ctypes.data_as(ip32), _dp(val), _dp(X0),
                                   C.byref(params), mode, C.byref(res))
        return self._unpack(rc, bufs, res, self.err() if rc else "")

    def stpcg_projected(self, pr, mode=0):
        """hd_stpcg_projected on the inputs of oracle_py.projected_stpcg_problem"""
        self.L.hd_stpcg_projected.restype = C.c_int
        self.L.hd_stpcg_projected.argtypes = [C.c_size_t, C.c_size_t, dp, dp, dp, dp, C.c_double, C.c_size_t,
                                              C.c_double, C.c_double, C.c_int, dp, dp, C.POINTER(C.c_size_t)]
        s = np.zeros(pr["n"])
        mn, it = C.c_double(0), C.c_size_t(0)
        rc = self.L.hd_stpcg_projected(pr["n"], pr["m"], _dp(pr["g"]), _dp(pr["P"]), _dp(pr["M"]),
                                       _dp(np.ascontiguousarray(pr["A"])), pr["Delta"], pr["max_iterations"],
                                       pr["kappa"], pr["theta"], mode, _dp(s), C.byref(mn), C.byref(it))
        self.L.hd_last_kkt_inner.restype = C.c_size_t
        self.L.hd_last_kkt_worst_residual.restype = C.c_double
        return dict(rc=rc, err=self.err() if rc else "", s=s, M_norm=mn.value, iterations=it.value,
                    syncs=self.L.hd_last_tnt_syncs(), kkt_inner=self.L.hd_last_kkt_inner(),
                    kkt_worst_residual=self.L.hd_last_kkt_worst_residual())

    def tnt_rosenbrock(self, n, precon_kind, x0, params, mode=0):
        """BASELINE cfg1 through EuclideanTNT<DeviceVector> (hd_tnt_rosenbrock)"""
        x0 = np.ascontiguousarray(x0, dtype=np.float64).ravel()
        bufs, res = self._result_buffers(n, params)
        self.L.hd_tnt_rosenbrock.restype = C.c_int
        self.L.hd_tnt_rosenbrock.argtypes = [C.c_size_t, C.c_int, dp, C.POINTER(op.TntParams), C.c_int,
                                             C.POINTER(op.TntResult)]
        rc = self.L.hd_tnt_rosenbrock(n, precon_kind, _dp(x0), C.byref(params), mode, C.byref(res))
        return self._unpack(rc, bufs, res, self.err() if rc else "")

    def tnt_so3n(self, N, ei, ej, Rt, w, R0, params, with_precon):
        ei = np.ascontiguousarray(ei, dtype=np.int32)
        ej = np.ascontiguousarray(ej, dtype=np.int32)
        Rt = np.ascontiguousarray(Rt, dtype=np.float64)
        w = np.ascontiguousarray(w, dtype=np.float64)
        R0 = np.ascontiguousarray(R0, dtype=np.float64).ravel()
        self.L.hd_tnt_so3n.restype = C.c_int
        self.L.hd_tnt_so3n.argtypes = [C.c_size_t, C.c_size_t, ip32, ip32, dp, dp, dp, C.POINTER(op.TntParams),
                                       C.c_int, C.POINTER(op.TntResult)]
        bufs, res = self._result_buffers(9 * N, params)
        rc = self.L.hd_tnt_so3n(N, ei.size, ei.ctypes.data_as(ip32), ej.ctypes.data_as(ip32), _dp(Rt), _dp(w),
                                _dp(R0), C.byref(params), int(with_precon), C.byref(res))
        return self._unpack(rc, bufs, res, self.err() if rc else "")

    def gaussian_probe(self, m, nx):
        out = np.zeros((m, nx), order="F")
        self.L.hd_gaussian_probe.restype = C.c_int
        self.L.hd_gaussian_probe.argtypes = [C.c_size_t, C.c_size_t, dp]
        rc = self.L.hd_gaussian_probe(m, nx, _dp(out))
        assert rc == 0, self.err()
        return out

    def lobpcg(self, m, nx, nev, Adiag=None, csr=None, Bdiag=None, Tdiag=None, X0=None, max_iters=1000,
               tau=1e-6):
        self.L.hd_lobpcg.restype = C.c_int
        self.L.hd_lobpcg.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, dp, ip32, ip32, dp, dp, dp, dp,
                                     C.c_size_t, C.c_double, dp, dp, sp, sp, dp]

        def arr(a):
            return None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        Ad, Bd, Td = arr(Adiag), arr(Bdiag), arr(Tdiag)
        X0f = None if X0 is None else np.asfortranarray(X0, dtype=np.float64)
        rp = cl = vl = None
        if csr is not None:
            rp = np.ascontiguousarray(csr[0], dtype=np.int32)
            cl = np.ascontiguousarray(csr[1], dtype=np.int32)
            vl = np.ascontiguousarray(csr[2], dtype=np.float64)
        th = np.zeros(nev)
        X = np.zeros((m, nev), order="F")
        it, nc = C.c_size_t(0), C.c_size_t(0)
        res = np.zeros(nx)
        rc = self.L.hd_lobpcg(m, nx, nev, _dp(Ad) if Ad is not None else None,
                              rp.ctypes.data_as(ip32) if rp is not None else None,
                              cl.ctypes.data_as(ip32) if cl is not None else None,
                              _dp(vl) if vl is not None else None, _dp(Bd) if Bd is not None else None,
                              _dp(Td) if Td is not None else None, _dp(X0f) if X0f is not None else None,
                              max_iters, tau, _dp(th), _dp(X), C.byref(it), C.byref(nc), _dp(res))
        out = dict(rc=rc, err=self.err() if rc else "", Theta=th, X=X, num_iters=it.value, nc=nc.value,
                   residuals=res)
        # per-iteration Ritz values / residual norms as the user function saw them (hd_lobpcg_trace)
        self.L.hd_lobpcg_trace.restype = C.c_size_t
        self.L.hd_lobpcg_trace.argtypes = [dp, dp, C.c_size_t]
        n = self.L.hd_lobpcg_trace(None, None, 0)
        tt, rt = np.zeros(max(n, 1)), np.zeros(max(n, 1))
        self.L.hd_lobpcg_trace(_dp(tt), _dp(rt), n)
        out["theta_trace"] = tt[:n].reshape(-1, nx)
        out["r_trace"] = rt[:n].reshape(-1, nx)
        return out

    def tnt_sphere(self, with_precon, x0, params):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        bufs, res = self._result_buffers(3, params)
        rc = self.L.hd_tnt_sphere(int(with_precon), _dp(x0), C.byref(params), C.byref(res))
        return self._unpack(rc, bufs, res, self.err() if rc else "")

    @staticmethod
    def _csr_pair(A):
        """scipy CSR A -> (rowptr, col, val) of A and of A' as int32/float64 arrays"""
        import scipy.sparse as sps
        A = sps.csr_matrix(A)
        At = sps.csr_matrix(A.T)
        out = []
        for M in (A, At):
            M.sort_indices()
            out += [np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32),
                    np.ascontiguousarray(M.data, dtype=np.float64)]
        return out

    def lsqr_csr(self, A, b, max_iterations=1000, lam=0.0, btol=1e-6, Atol=1e-6, Acond_limit=1e8, Delta=None,
                 mode=0):
        """LinearAlgebra::LSQR on DeviceVector, A (square, scipy sparse) and A' as CSR operators"""
        n = A.shape[0]
        rp, cl, vl, rpt, clt, vlt = self._csr_pair(A)
        b = np.ascontiguousarray(b, dtype=np.float64)
        if Delta is None:
            Delta = float(np.sqrt(np.finfo(np.float64).max))
        x = np.zeros(n)
        xn, it = C.c_double(0), C.c_size_t(0)
        self.L.hd_lsqr_csr.restype = C.c_int
        self.L.hd_lsqr_csr.argtypes = [C.c_size_t, ip32, ip32, dp, ip32, ip32, dp, dp, C.c_size_t, C.c_double, C.c_double,
                                       C.c_double, C.c_double, C.c_double, C.c_int, dp, dp, sp]
        rc = self.L.hd_lsqr_csr(n, rp.ctypes.data_as(ip32), cl.ctypes.data_as(ip32), _dp(vl), rpt.ctypes.data_as(ip32),
                                clt.ctypes.data_as(ip32), _dp(vlt), _dp(b), max_iterations, lam, btol, Atol,
                                Acond_limit, Delta, mode, _dp(x), C.byref(xn), C.byref(it))
        return dict(rc=rc, err=self.err() if rc else "", x=x, xnorm=xn.value, iterations=it.value)

    def tnls_affine(self, A, b, x0, root_tolerance=1e-9, gradient_tolerance=0.0, max_iterations=20,
                    max_LSQR_iterations=1000, mode=0):
        """Riemannian::TNLS on DeviceVector for F(x) = A x - b (mode 0: tagged callables => fused mi_lsqr)"""
        n = A.shape[0]
        rp, cl, vl, rpt, clt, vlt = self._csr_pair(A)
        b = np.ascontiguousarray(b, dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        x = np.zeros(n)
        f, gn = C.c_double(0), C.c_double(0)
        st, outer, inner = C.c_int(-1), C.c_size_t(0), C.c_size_t(0)
        self.L.hd_tnls_affine.restype = C.c_int
        self.L.hd_tnls_affine.argtypes = [C.c_size_t, ip32, ip32, dp, ip32, ip32, dp, dp, dp, C.c_double, C.c_double,
                                          C.c_size_t, C.c_size_t, C.c_int, dp, dp, dp, C.POINTER(C.c_int), sp, sp]
        rc = self.L.hd_tnls_affine(n, rp.ctypes.data_as(ip32), cl.ctypes.data_as(ip32), _dp(vl),
                                   rpt.ctypes.data_as(ip32), clt.ctypes.data_as(ip32), _dp(vlt), _dp(b), _dp(x0),
                                   root_tolerance, gradient_tolerance, max_iterations, max_LSQR_iterations, mode, _dp(x),
                                   C.byref(f), C.byref(gn), C.byref(st), C.byref(outer), C.byref(inner))
        return dict(rc=rc, err=self.err() if rc else "", x=x, f=f.value, gradfx_norm=gn.value, status=st.value,
                    outer=outer.value, inner_total=inner.value)

    def lobpcg_on(self, ctx, csr, m, nx, nev, X0, max_iters=500, tau=1e-8):
        """hd_lobpcg_on: LOBPCG on the caller's capi.Context / capi.Csr (a rank's shard when the context carries a
        communicator); X0 is the local m x nx block, column-major"""
        self.L.hd_lobpcg_on.restype = C.c_int
        self.L.hd_lobpcg_on.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, dp, C.c_size_t,
                                        C.c_double, dp, dp, sp, sp, dp]
        X0 = np.asfortranarray(X0, dtype=np.float64)
        th = np.zeros(nev)
        X = np.zeros((m, nev), order="F")
        it, nc = C.c_size_t(0), C.c_size_t(0)
        res = np.zeros(nx)
        rc = self.L.hd_lobpcg_on(ctx.h, csr.h, m, nx, nev, _dp(X0.ravel(order="F")), max_iters, tau, _dp(th),
                                 X.ctypes.data_as(dp), C.byref(it), C.byref(nc), _dp(res))
        return dict(rc=rc, err=self.err() if rc else "", theta=th, X=X, iterations=it.value, nconv=nc.value,
                    residuals=res)

    def gd_sphere(self, x0, cap=4096):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        x = np.zeros(3)
        f, gn = C.c_double(0), C.c_double(0)
        st, it = C.c_int(-1), C.c_size_t(0)
        fv = np.zeros(cap)
        ls = np.zeros(cap, dtype=np.uint64)
        rc = self.L.hd_gd_sphere(_dp(x0), _dp(x), C.byref(f), C.byref(gn), C.byref(st), C.byref(it), cap, _dp(fv),
                                 ls.ctypes.data_as(sp))
        k = min(it.value, cap)
        return dict(rc=rc, err=self.err() if rc else "", x=x, f=f.value, gradfx_norm=gn.value, status=st.value,
                    iterations=it.value, objective_values=fv[:k].copy(),
                    linesearch_iterations=ls[:k].astype(np.int64))

    def gd_stiefel(self, n, p, rowptr, col, val, X0, max_iterations, gradient_tolerance, alpha=1.0, beta=.5,
                   sigma=.5, max_ls_iterations=100, mode=0, cap=4096):
        """hd_gd_stiefel: GradientDescent<DeviceVector> on the Stiefel Rayleigh quotient (mode 0 fused Armijo
        trials, mode 1 the reference's statement sequence)"""
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        X0 = np.ascontiguousarray(X0, dtype=np.float64).ravel()
        x = np.zeros(n * p)
        f, gn = C.c_double(0), C.c_double(0)
        st, it, sy = C.c_int(-1), C.c_size_t(0), C.c_size_t(0)
        fv = np.zeros(cap)
        ls = np.zeros(cap, dtype=np.uint64)
        rc = self.L.hd_gd_stiefel(n, p, rowptr.ctypes.data_as(ip32), col.ctypes.data_as(ip32), _dp(val), _dp(X0),
                                  max_iterations, gradient_tolerance, alpha, beta, sigma, max_ls_iterations, mode,
                                  _dp(x), C.byref(f), C.byref(gn), C.byref(st), C.byref(it), cap, _dp(fv),
                                  ls.ctypes.data_as(sp), C.byref(sy))
        k = min(it.value, cap)
        return dict(rc=rc, err=self.err() if rc else "", x=x.reshape(n, p), f=f.value, gradfx_norm=gn.value,
                    status=st.value, iterations=it.value, objective_values=fv[:k].copy(),
                    linesearch_iterations=ls[:k].astype(np.int64), syncs=sy.value)


class SinfitHarness:
    """tests/cpp/libharness_sinfit.so (hipcc): the reference's TNLS sin-fit problem with HIP kernels of its own; the
    Jacobian pair handed back by J(x) is a pair of FRESH device operators at every linearisation."""
    PATH = os.path.join(HERE, "cpp", "libharness_sinfit.so")

    def __init__(self):
        from optimization_amd import capi
        if not os.path.exists(self.PATH):
            raise FileNotFoundError(self.PATH + " (run __graft_entry__.build())")
        self.L = C.CDLL(self.PATH)
        self.L.hs_last_error.restype = C.c_char_p
        self.FC = capi.FusionCounters
        self.L.hs_tnls_sinfit.restype = C.c_int
        self.L.hs_tnls_sinfit.argtypes = [C.c_size_t, dp, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                          C.c_size_t, dp, dp, dp, C.POINTER(C.c_int), sp, sp, sp, C.POINTER(self.FC)]

    def tnls_sinfit(self, t, y, beta0, with_precon=False, mode=0, root_tolerance=1e-6, gradient_tolerance=0.0,
                    Delta_tolerance=0.0, max_iterations=100):
        t = np.ascontiguousarray(t, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        b0 = np.ascontiguousarray(beta0, dtype=np.float64)
        beta = np.zeros(2)
        f, gn = C.c_double(0), C.c_double(0)
        st, outer, inner, jac = C.c_int(-1), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        fc = self.FC()
        rc = self.L.hs_tnls_sinfit(t.size, _dp(t), _dp(y), _dp(b0), int(with_precon), mode, root_tolerance,
                                   gradient_tolerance, Delta_tolerance, max_iterations, _dp(beta), C.byref(f),
                                   C.byref(gn), C.byref(st), C.byref(outer), C.byref(inner), C.byref(jac), C.byref(fc))
        return dict(rc=rc, err=self.L.hs_last_error().decode() if rc else "", beta=beta, f=f.value,
                    gradfx_norm=gn.value, status=st.value, outer=outer.value, inner_total=inner.value,
                    jacobians=jac.value, counters={k: int(getattr(fc, k)) for k, _ in self.FC._fields_})
