"""ctypes binding of the C ABI in include/mi355opt.h (libmi355opt.so).

This is harness plumbing for tests/ and bench.py: the product is the shared library (HIP kernels
behind a C ABI) and the C++ template layer in optimization_amd/include/.  There is no CPU fallback:
if the library is missing or no GPU is present the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355OPT_LIB") or os.path.join(_HERE, "libmi355opt.so")  # MI355OPT_LIB: experiment builds

MI_OK = 0
STATUS = {0: "MI_OK", 1: "MI_ERR_INVALID_ARGUMENT", 2: "MI_ERR_HIP", 3: "MI_ERR_OOM",
          4: "MI_ERR_NO_DEVICE", 5: "MI_ERR_COMM", 6: "MI_ERR_INTERNAL"}
KERNELS = ["none", "cg_init", "cg_dot3", "cg_scalar_a", "cg_update", "cg_scalar_b", "cg_pupdate",
           "csr_spmm", "stiefel_spmm_gram", "stiefel_gram_reduce", "stiefel_finish_dots",
           "stiefel_retract", "bsr3_spmv_dots", "blas1", "lobpcg_gram", "lobpcg_update",
           "lobpcg_residual", "stiefel_hess_fused", "comm_allreduce", "comm_halo"]
KID = {k: i for i, k in enumerate(KERNELS)}
STPCG_EXIT = ["RESIDUAL", "MAXIT", "KERNEL", "BOUNDARY"]

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_size_p = C.POINTER(C.c_size_t)
vp = C.c_void_p

APPLY_FN = C.CFUNCTYPE(C.c_int, vp, vp, vp)


class PanelBlocks(C.Structure):
    """mi_panel_blocks (include/mi355opt.h): a panel as 1..3 column blocks that need not be adjacent"""
    _fields_ = [("nblocks", C.c_int), ("block", C.c_void_p * 3), ("cols", C.c_int * 3)]

    @classmethod
    def of(cls, blocks):
        pb = cls()
        pb.nblocks = len(blocks)
        for i, (v, c) in enumerate(blocks):
            pb.block[i] = v.h
            pb.cols[i] = c
        pb._keep = [v for v, _ in blocks]
        return pb


class FusedArgs(C.Structure):
    _fields_ = [("partials", C.POINTER(C.c_double)), ("partial_stride", C.c_size_t), ("max_rows", C.c_int),
                ("required_rows", C.c_int), ("stream", vp)]


APPLY_FUSED_FN = C.CFUNCTYPE(C.c_int, vp, vp, vp, C.POINTER(FusedArgs), C.POINTER(C.c_int))


class MiError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS.get(status, status)}: {msg}")
        self.status = status


class StpcgParams(C.Structure):
    _fields_ = [("Delta", C.c_double), ("max_iterations", C.c_size_t), ("kappa_fgr", C.c_double),
                ("theta", C.c_double), ("epsilon", C.c_double), ("run_ahead", C.c_int),
                ("constraint_At", C.c_int), ("defer_result", C.c_int)]


class StpcgResult(C.Structure):
    _fields_ = [("update_step_M_norm", C.c_double), ("num_iterations", C.c_size_t),
                ("exit_reason", C.c_int), ("hvp_calls", C.c_size_t), ("rv_final", C.c_double),
                ("precon_status", C.c_int)]


class LsqrParams(C.Structure):
    _fields_ = [("max_iterations", C.c_size_t), ("lam", C.c_double), ("btol", C.c_double), ("Atol", C.c_double),
                ("Acond_limit", C.c_double), ("Delta", C.c_double), ("run_ahead", C.c_int)]


class LsqrResult(C.Structure):
    _fields_ = [("xnorm", C.c_double), ("num_iterations", C.c_size_t), ("exit_reason", C.c_int),
                ("rbar_norm", C.c_double), ("Arnorm", C.c_double), ("Anorm", C.c_double), ("Acond", C.c_double),
                ("operator_applications", C.c_size_t)]


LSQR_EXIT = ["MAXIT", "S1", "S2", "S3", "S4", "TRIVIAL"]


class StpcgTrace(C.Structure):
    _fields_ = [("cap", C.c_size_t), ("len", C.c_size_t), ("alpha", c_double_p), ("beta", c_double_p),
                ("kappa", c_double_p), ("rv", c_double_p)]


_lib = None


def load():
    """Load libmi355opt.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not found: run `python -m optimization_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.mi_version.restype = C.c_char_p
    L.mi_last_error.restype = C.c_char_p
    L.mi_status_string.restype = C.c_char_p
    L.mi_status_string.argtypes = [C.c_int]
    L.mi_kernel_name.restype = C.c_char_p
    L.mi_kernel_name.argtypes = [C.c_int]
    sigs = {
        "mi_device_count": [C.POINTER(C.c_int)],
        "mi_ctx_create": [C.c_int, C.POINTER(vp)],
        "mi_ctx_destroy": [vp],
        "mi_ctx_sync": [vp],
        "mi_ctx_set_option": [vp, C.c_char_p, C.c_long],
        "mi_ctx_sync_count": [vp, c_size_p],
        "mi_ctx_fusion_counters": [vp, C.POINTER(FusionCounters)],
        "mi_ctx_fusion_counters_reset": [vp],
        "mi_ctx_note_generic": [vp, C.c_int, C.c_char_p],
        "mi_op_create_compose": [vp, vp, vp, C.POINTER(vp)],
        "mi_ctx_stream": [vp, C.POINTER(vp)],
        "mi_ctx_device_name": [vp, C.c_char_p, C.c_size_t],
        "mi_ctx_pool_bytes": [vp, c_size_p],
        "mi_ktime_enable": [vp, C.c_int, C.c_int],
        "mi_ktime_reset": [vp],
        "mi_ktime_read": [vp, C.c_int, c_size_p, c_double_p],
        "mi_range_push": [C.c_char_p],
        "mi_range_pop": [],
        "mi_timer_start": [vp],
        "mi_timer_stop": [vp, c_double_p],
        "mi_vec_create": [vp, C.c_size_t, C.POINTER(vp)],
        "mi_vec_destroy": [vp],
        "mi_vec_len": [vp, c_size_p],
        "mi_vec_data": [vp, C.POINTER(vp)],
        "mi_vec_touch": [vp],
        "mi_vec_upload": [vp, c_double_p, C.c_size_t],
        "mi_vec_download": [vp, c_double_p, C.c_size_t],
        "mi_vec_copy": [vp, vp],
        "mi_vec_fill": [vp, C.c_double],
        "mi_vec_scale": [vp, C.c_double],
        "mi_vec_div": [vp, C.c_double],
        "mi_vec_scale_to": [vp, C.c_double, vp],
        "mi_vec_axpy": [vp, C.c_double, vp],
        "mi_vec_axpby": [vp, C.c_double, vp, C.c_double, vp],
        "mi_vec_dot": [vp, vp, c_double_p],
        "mi_vec_dot_batch": [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), c_double_p],
        "mi_csr_create": [vp, C.c_size_t, C.c_size_t, c_int32_p, c_int32_p, c_double_p, C.POINTER(vp)],
        "mi_csr_destroy": [vp],
        "mi_csr_spmm": [vp, C.c_int, vp, vp],
        "mi_op_create_callback": [vp, C.c_size_t, APPLY_FN, vp, C.POINTER(vp)],
        "mi_op_create_callback_rect": [vp, C.c_size_t, C.c_size_t, APPLY_FN, vp, C.POINTER(vp)],
        "mi_op_create_callback_fused": [vp, C.c_size_t, APPLY_FN, APPLY_FUSED_FN, vp, C.POINTER(vp)],
        "mi_lsqr_default_params": [C.POINTER(LsqrParams)],
        "mi_lsqr": [vp, vp, vp, vp, C.POINTER(LsqrParams), vp, C.POINTER(LsqrResult)],
        "mi_op_create_diag": [vp, vp, C.POINTER(vp)],
        "mi_op_create_csr": [vp, vp, C.c_int, C.POINTER(vp)],
        "mi_op_apply": [vp, vp, vp],
        "mi_op_dims": [vp, c_size_p, c_size_p],
        "mi_op_destroy": [vp],
        "mi_precon_create_callback": [vp, C.c_size_t, APPLY_FN, vp, C.POINTER(vp)],
        "mi_precon_create_diag": [vp, vp, C.POINTER(vp)],
        "mi_precon_create_block3": [vp, vp, C.POINTER(vp)],
        "mi_precon_apply": [vp, vp, vp],
        "mi_precon_create_constraint": [vp, C.c_size_t, C.c_size_t, vp, vp, C.POINTER(vp)],
        "mi_precon_create_constraint_csr": [vp, C.c_size_t, C.c_size_t, c_int32_p, c_int32_p, c_double_p, vp, C.c_double,
                                            C.c_size_t, C.POINTER(vp)],
        "mi_precon_constraint_info": [vp, c_size_p, c_double_p, c_double_p, C.POINTER(C.c_int)],
        "mi_precon_constraint_solve": [vp, vp, vp, vp],
        "mi_precon_constraint_At": [vp, vp, vp],
        "mi_precon_destroy": [vp],
        "mi_stpcg": [vp, vp, vp, vp, C.POINTER(StpcgParams), vp, C.POINTER(StpcgResult),
                     C.POINTER(StpcgTrace)],
        "mi_stiefel_gram": [vp, C.c_size_t, C.c_int, vp, vp, c_double_p],
        "mi_stiefel_project": [vp, C.c_size_t, C.c_int, vp, vp, vp],
        "mi_stiefel_retract": [vp, C.c_size_t, C.c_int, vp, vp, vp],
        "mi_stiefel_rq_create": [vp, vp, C.c_size_t, C.c_int, C.POINTER(vp)],
        "mi_stiefel_rq_destroy": [vp],
        "mi_stiefel_rq_objective": [vp, vp, c_double_p],
        "mi_stiefel_rq_model": [vp, vp, vp, C.POINTER(vp)],
        "mi_stiefel_rq_trial": [vp, vp, vp, vp, vp, c_double_p],
        "mi_stiefel_rq_armijo_trial": [vp, vp, vp, C.c_double, vp, vp, c_double_p],
        "mi_stiefel_rq_precon": [vp, vp, vp, C.POINTER(vp)],
        "mi_so3n_create": [vp, C.c_size_t, C.c_size_t, c_int32_p, c_int32_p, c_double_p, c_double_p,
                           C.POINTER(vp)],
        "mi_so3n_destroy": [vp],
        "mi_so3n_objective": [vp, vp, c_double_p],
        "mi_so3n_model": [vp, vp, vp, C.POINTER(vp), C.POINTER(vp)],
        "mi_so3n_retract": [vp, vp, vp, vp],
        "mi_so3n_trial": [vp, vp, vp, vp, C.c_int, vp, c_double_p],
        "mi_stpcg_collect": [vp, C.POINTER(StpcgResult)],
        "mi_lobpcg_gram": [vp, C.c_size_t, C.c_int, C.c_int, vp, vp, c_double_p],
        "mi_lobpcg_update": [vp, C.c_size_t, C.c_int, C.c_int, vp, c_double_p, C.c_int, vp],
        "mi_lobpcg_update2": [vp, C.c_size_t, C.c_int, C.c_int, vp, c_double_p, C.c_int, vp, C.c_int, vp],
        "mi_lobpcg_residual": [vp, C.c_size_t, C.c_int, vp, vp, vp, c_double_p, vp, c_double_p,
                               c_double_p],
        "mi_rayleigh_ritz": [C.c_int, c_double_p, c_double_p, c_double_p, c_double_p],
        "mi_rayleigh_ritz_lowest": [C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p],
        "mi_csr_spmm_colmajor": [vp, C.c_int, vp, vp],
        "mi_csr_spmm_colmajor_residual": [vp, C.c_int, vp, c_double_p, vp, vp, c_double_p, c_double_p],
        "mi_lobpcg_gram_split": [vp, C.c_size_t, C.c_int, vp, C.c_int, vp, vp, c_double_p],
        "mi_debug_window_runs": [C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)],
        "mi_lobpcg_gram_pair": [vp, C.c_size_t, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, vp, c_double_p, c_double_p],
        "mi_lobpcg_gram_pair_sym": [vp, C.c_size_t, C.c_int, vp, C.c_int, vp, vp, c_double_p, c_double_p],
        "mi_lobpcg_gram_pair_sym_blocks": [vp, C.c_size_t, vp, C.c_int, vp, vp, c_double_p, c_double_p],
        "mi_lobpcg_gram_pair_gen_blocks": [vp, C.c_size_t, vp, vp, vp, c_double_p, c_double_p],
        "mi_lobpcg_gram_pair_sym_tblocks": [vp, C.c_size_t, vp, vp, c_double_p, c_double_p],
        "mi_lobpcg_update2_blocks": [vp, C.c_size_t, vp, C.c_int, c_double_p, C.c_int, vp, C.c_int, vp],
        "mi_csr_spmm_colmajor_blocks": [vp, vp, vp],
        "mi_panel_rowscale": [vp, C.c_size_t, C.c_int, vp, vp, vp],
        "mi_vec_view": [vp, C.c_size_t, C.c_size_t, C.POINTER(vp)],
        "mi_comm_unique_id": [C.POINTER(C.c_ubyte)],
        "mi_comm_init": [vp, C.c_int, C.c_int, C.POINTER(C.c_ubyte)],
        "mi_comm_finalize": [vp],
        "mi_comm_info": [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "mi_comm_rccl_count": [vp, C.POINTER(C.c_int)],
        "mi_comm_ipc_export": [vp, C.POINTER(C.c_ubyte)],
        "mi_comm_ipc_attach": [vp, C.c_int, C.c_int, C.POINTER(C.c_ubyte)],
        "mi_comm_ipc_selftest": [vp, C.POINTER(C.c_int)],
        "mi_comm_ipc_enable": [vp, C.c_int],
        "mi_comm_ipc_error": [vp, C.POINTER(C.c_int)],
        "mi_comm_kernel_launches": [vp, C.POINTER(C.c_ulonglong)],
        "mi_comm_ipc_fold": [vp, C.c_int],
        "mi_debug_csr_window_info": [vp, c_size_p],
        "mi_debug_time_fused_apply": [vp, vp, vp, C.c_int, c_double_p],
        "mi_debug_set_rank": [vp, C.c_int, C.c_int],
        "mi_debug_csr_set_halo": [vp, C.c_int, c_double_p],
        "mi_csr_create_sharded": [vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, c_int32_p,
                                  c_int64_p, c_double_p, c_size_p, C.POINTER(vp)],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = args
    L.mi_stpcg_default_params.restype = None
    L.mi_lsqr_default_params.restype = None
    L.mi_stpcg_default_params.argtypes = [C.POINTER(StpcgParams)]
    _lib = L
    return L


class FusionCounters(C.Structure):
    _fields_ = [(k, C.c_ulonglong) for k in ("fused_stpcg_solves", "generic_stpcg_solves", "fused_lsqr_solves",
                                             "generic_lsqr_solves", "fused_trial_steps", "generic_trial_steps",
                                             "generic_inner_products")]


def check(status):
    if status != MI_OK:
        raise MiError(status, load().mi_last_error().decode())


def window_runs(ntiles, max_wgs, num_cu=256, far_stride=0):
    """run plan of the LDS-window kernels (host-only, mi_debug_window_runs): array of run starts + the end"""
    L = load()
    cap = max(ntiles, max_wgs, 1) + 2
    buf = (C.c_int * cap)()
    nb = C.c_int(0)
    check(L.mi_debug_window_runs(ntiles, max_wgs, num_cu, far_stride, buf, cap, C.byref(nb)))
    return np.array(buf[:nb.value + 1], dtype=np.int64)


def csr_shard_plan(n_global, world_size, rank, row_starts, col_global):
    """Host-only (no GPU): (col_local int32, need_lo, need_hi) for this rank's slab of a row-sharded matrix."""
    L = load()
    L.mi_csr_shard_plan.restype = C.c_int
    L.mi_csr_shard_plan.argtypes = [C.c_size_t, C.c_int, C.c_int, c_size_p, C.c_size_t, c_int64_p, c_int32_p,
                                    c_size_p, c_size_p]
    col_global = np.ascontiguousarray(col_global, dtype=np.int64)
    starts = (C.c_size_t * len(row_starts))(*[int(x) for x in row_starts])
    col_local = np.zeros(col_global.size, dtype=np.int32)
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    check(L.mi_csr_shard_plan(n_global, world_size, rank, starts, col_global.size,
                              col_global.ctypes.data_as(c_int64_p), col_local.ctypes.data_as(c_int32_p),
                              C.byref(lo), C.byref(hi)))
    return col_local, lo.value, hi.value


def device_count():
    n = C.c_int(0)
    check(load().mi_device_count(C.byref(n)))
    return n.value


def _dp(a):
    return a.ctypes.data_as(c_double_p)


class Context:
    def __init__(self, device=0):
        self.L = load()
        self.h = vp()
        check(self.L.mi_ctx_create(device, C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.mi_ctx_destroy(self.h)
            self.h = vp()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        check(self.L.mi_ctx_sync(self.h))

    def set_option(self, name, value):
        """mi_ctx_set_option: one of the library's MI355OPT_<NAME> switches on this live context"""
        check(self.L.mi_ctx_set_option(self.h, name.encode(), int(value)))
        return self

    def sync_count(self):
        n = C.c_size_t(0)
        check(self.L.mi_ctx_sync_count(self.h, C.byref(n)))
        return n.value

    def fusion_counters(self, reset=False):
        """mi_ctx_fusion_counters: which side of the fusion boundary the work of this context ran on"""
        fc = FusionCounters()
        check(self.L.mi_ctx_fusion_counters(self.h, C.byref(fc)))
        if reset:
            check(self.L.mi_ctx_fusion_counters_reset(self.h))
        return {k: int(getattr(fc, k)) for k, _ in FusionCounters._fields_}

    def device_name(self):
        buf = C.create_string_buffer(256)
        check(self.L.mi_ctx_device_name(self.h, buf, 256))
        return buf.value.decode()

    # vectors ------------------------------------------------------------------------------
    def vec(self, n):
        return Vec(self, n)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        v = Vec(self, arr.size)
        check(self.L.mi_vec_upload(v.h, _dp(arr), arr.size))
        return v

    def dot_batch(self, xs, ys):
        k = len(xs)
        X = (vp * k)(*[x.h for x in xs])
        Y = (vp * k)(*[y.h for y in ys])
        out = np.zeros(k)
        check(self.L.mi_vec_dot_batch(self.h, k, X, Y, _dp(out)))
        return out

    # timing ---------------------------------------------------------------------------------
    def timer_start(self):
        check(self.L.mi_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_double(0)
        check(self.L.mi_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def range_push(self, name):
        check(self.L.mi_range_push(name.encode()))

    def range_pop(self):
        check(self.L.mi_range_pop())

    def ktime_enable(self, name, on=True):
        check(self.L.mi_ktime_enable(self.h, KID[name], int(on)))

    def ktime_reset(self):
        check(self.L.mi_ktime_reset(self.h))

    def ktime_read(self, name):
        n = C.c_size_t(0)
        ms = C.c_double(0)
        check(self.L.mi_ktime_read(self.h, KID[name], C.byref(n), C.byref(ms)))
        return n.value, ms.value

    # operators ------------------------------------------------------------------------------
    def csr(self, n, rowptr, col, val):
        return Csr(self, n, rowptr, col, val)

    def csr_sharded(self, n_global, row_begin, row_end, rowptr, col_global, val, row_starts):
        A = Csr.__new__(Csr)
        A.ctx, A.L = self, self.L
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col_global = np.ascontiguousarray(col_global, dtype=np.int64)
        val = np.ascontiguousarray(val, dtype=np.float64)
        starts = (C.c_size_t * len(row_starts))(*[int(x) for x in row_starts])
        A.n, A.nnz = row_end - row_begin, int(rowptr[-1])
        A.h = vp()
        check(self.L.mi_csr_create_sharded(self.h, n_global, row_begin, row_end, A.nnz,
                                           rowptr.ctypes.data_as(c_int32_p),
                                           col_global.ctypes.data_as(c_int64_p), _dp(val), starts,
                                           C.byref(A.h)))
        return A

    def op_diag(self, d):
        h = vp()
        check(self.L.mi_op_create_diag(self.h, d.h, C.byref(h)))
        return Op(self, h, keep=[d])

    def op_csr(self, A, p):
        h = vp()
        check(self.L.mi_op_create_csr(self.h, A.h, p, C.byref(h)))
        op = Op(self, h, keep=[A])
        op.n_in = A.n * p
        return op

    def op_callback(self, n, fn):
        """fn(in_vec: Vec, out_vec: Vec) enqueues out = Op(in)."""
        def cb(_u, pin, pout):
            try:
                fn(Vec(self, 0, handle=vp(pin)), Vec(self, 0, handle=vp(pout)))
                return 0
            except Exception:  # noqa
                import traceback
                traceback.print_exc()
                return 6
        cfn = APPLY_FN(cb)
        h = vp()
        check(self.L.mi_op_create_callback(self.h, n, cfn, None, C.byref(h)))
        return Op(self, h, keep=[cfn])

    def op_callback_fused(self, n, fn, fused):
        """mi_op_create_callback_fused.  fn(in_vec, out_vec) enqueues out = Op(in); fused(in_vec, out_vec, args) does the
        same AND leaves the three curvature partial rows (args: FusedArgs), returning the number of rows written.  (A
        real client does both in a HIP kernel of its own: examples/stpcg_user_stencil.hip; from Python this serves the
        contract tests.)"""
        def cb(_u, pin, pout):
            try:
                fn(Vec(self, 0, handle=vp(pin)), Vec(self, 0, handle=vp(pout)))
                return 0
            except Exception:  # noqa
                import traceback
                traceback.print_exc()
                return 6

        def cbf(_u, pin, pout, args, rows):
            try:
                rows[0] = int(fused(Vec(self, 0, handle=vp(pin)), Vec(self, 0, handle=vp(pout)), args.contents))
                return 0
            except Exception:  # noqa
                import traceback
                traceback.print_exc()
                return 6
        c1, c2 = APPLY_FN(cb), APPLY_FUSED_FN(cbf)
        h = vp()
        check(self.L.mi_op_create_callback_fused(self.h, n, c1, c2, None, C.byref(h)))
        return Op(self, h, keep=[c1, c2])

    def precon_diag(self, dinv):
        h = vp()
        check(self.L.mi_precon_create_diag(self.h, dinv.h, C.byref(h)))
        return Precon(self, h, keep=[dinv])

    def precon_block3(self, inv_blocks):
        h = vp()
        check(self.L.mi_precon_create_block3(self.h, inv_blocks.h, C.byref(h)))
        return Precon(self, h, keep=[inv_blocks])

    def precon_constraint(self, A, Minv):
        """mi_precon_create_constraint: A (m x n numpy, dense constraints), Minv (n, inverse of the diagonal M)"""
        A = np.ascontiguousarray(A, dtype=np.float64)
        m, n = A.shape
        Ad, Md = self.upload(A), self.upload(Minv)
        h = vp()
        check(self.L.mi_precon_create_constraint(self.h, n, m, Ad.h, Md.h, C.byref(h)))
        P = ConstraintPrecon(self, h, keep=[Ad, Md])
        P.m, P.n = m, n
        return P

    def precon_constraint_csr(self, A, Minv, inner_tol=0.0, inner_max_iterations=0):
        """mi_precon_create_constraint_csr: A scipy.sparse (m x n) or dense numpy (its non-zeros are taken)"""
        import scipy.sparse as sps
        A = sps.csr_matrix(A)
        A.sort_indices()
        m, n = A.shape
        Md = self.upload(Minv)
        rp = np.ascontiguousarray(A.indptr, dtype=np.int32)
        cl = np.ascontiguousarray(A.indices, dtype=np.int32)
        vl = np.ascontiguousarray(A.data, dtype=np.float64)
        h = vp()
        check(self.L.mi_precon_create_constraint_csr(self.h, n, m, rp.ctypes.data_as(c_int32_p),
                                                     cl.ctypes.data_as(c_int32_p), _dp(vl), Md.h, inner_tol,
                                                     inner_max_iterations, C.byref(h)))
        P = ConstraintPrecon(self, h, keep=[Md])
        P.m, P.n = m, n
        return P

    def precon_callback(self, n, fn):
        def cb(_u, pin, pout):
            try:
                fn(Vec(self, 0, handle=vp(pin)), Vec(self, 0, handle=vp(pout)))
                return 0
            except Exception:  # noqa
                import traceback
                traceback.print_exc()
                return 6
        cfn = APPLY_FN(cb)
        h = vp()
        check(self.L.mi_precon_create_callback(self.h, n, cfn, None, C.byref(h)))
        return Precon(self, h, keep=[cfn])

    # fused STPCG ------------------------------------------------------------------------------
    def stpcg(self, g, H, P=None, Delta=1.0, max_iterations=1000, kappa_fgr=.1, theta=.5,
              epsilon=1e-8, run_ahead=0, trace_cap=0, s_out=None, defer=False, constraint_At=False):
        """defer=True: mi_stpcg returns without waiting for the device (s is valid in stream order); the scalar
        results come from stpcg_collect()."""
        prm = StpcgParams(Delta, max_iterations, kappa_fgr, theta, epsilon, run_ahead, int(constraint_At),
                          int(defer))
        res = StpcgResult()
        s = s_out if s_out is not None else Vec(self, g.n)
        tr = None
        arrs = {}
        if trace_cap:
            arrs = {k: np.zeros(trace_cap) for k in ("alpha", "beta", "kappa", "rv")}
            tr = StpcgTrace(trace_cap, 0, _dp(arrs["alpha"]), _dp(arrs["beta"]), _dp(arrs["kappa"]),
                            _dp(arrs["rv"]))
        check(self.L.mi_stpcg(self.h, g.h, H.h, P.h if P is not None else None, C.byref(prm), s.h,
                              C.byref(res), C.byref(tr) if tr else None))
        out = dict(s=s, M_norm=res.update_step_M_norm, iterations=res.num_iterations,
                   exit_reason=res.exit_reason, hvp_calls=res.hvp_calls, rv_final=res.rv_final,
                   precon_status=res.precon_status)
        if tr:
            out["trace"] = {k: v[:tr.len].copy() for k, v in arrs.items()}
        return out

    def stpcg_collect(self):
        res = StpcgResult()
        check(self.L.mi_stpcg_collect(self.h, C.byref(res)))
        return dict(M_norm=res.update_step_M_norm, iterations=res.num_iterations, exit_reason=res.exit_reason,
                    hvp_calls=res.hvp_calls, rv_final=res.rv_final, precon_status=res.precon_status)

    # fused LSQR ---------------------------------------------------------------------------------
    def lsqr(self, A, At, b, x_out=None, **kw):
        """mi_lsqr: A, At are Op handles (A: n_x -> n_y); kw: max_iterations, lam, btol, Atol, Acond_limit, Delta"""
        prm = LsqrParams()
        self.L.mi_lsqr_default_params(C.byref(prm))
        for k, v in kw.items():
            if not hasattr(prm, k):
                raise AttributeError(k)
            setattr(prm, k, v)
        res = LsqrResult()
        x = x_out if x_out is not None else Vec(self, kw_nx if (kw_nx := getattr(A, "n_in", None)) else b.n)
        check(self.L.mi_lsqr(self.h, A.h, At.h, b.h, C.byref(prm), x.h, C.byref(res)))
        return dict(x=x, xnorm=res.xnorm, iterations=res.num_iterations, exit_reason=res.exit_reason,
                    rbar_norm=res.rbar_norm, Arnorm=res.Arnorm, Anorm=res.Anorm, Acond=res.Acond,
                    operator_applications=res.operator_applications)

    # Stiefel ----------------------------------------------------------------------------------
    def stiefel_gram(self, n, p, X, Z):
        G = np.zeros(p * p)
        check(self.L.mi_stiefel_gram(self.h, n, p, X.h, Z.h, _dp(G)))
        return G.reshape(p, p)

    def stiefel_project(self, n, p, X, Z):
        out = Vec(self, n * p)
        check(self.L.mi_stiefel_project(self.h, n, p, X.h, Z.h, out.h))
        return out

    def stiefel_retract(self, n, p, X, V):
        Y = Vec(self, n * p)
        check(self.L.mi_stiefel_retract(self.h, n, p, X.h, V.h, Y.h))
        return Y

    def stiefel_rq(self, A, n, p):
        return StiefelRQ(self, A, n, p)

    def so3n(self, N, ei, ej, Rt, w):
        return So3N(self, N, ei, ej, Rt, w)

    # LOBPCG panels (column-major m x k) ---------------------------------------------------------
    def lobpcg_gram(self, m, S, ka, T, kb):
        G = np.zeros((ka, kb), order="F")
        check(self.L.mi_lobpcg_gram(self.h, m, ka, kb, S.h, T.h, _dp(G)))
        return G

    def lobpcg_gram_split(self, m, S, k, T1, k1, T2):
        """S' [T1 | T2] with T in two panels (mi_lobpcg_gram_split)"""
        G = np.zeros((k, k), order="F")
        check(self.L.mi_lobpcg_gram_split(self.h, m, k, S.h, k1, T1.h, T2.h, _dp(G)))
        return G

    def lobpcg_gram_pair(self, m, S, k, Ta1, k1a, Ta2, Tb1, k1b, Tb2):
        """(S'[Ta1|Ta2], S'[Tb1|Tb2]) with one synchronisation; T?2 may be None (one panel of k columns)"""
        Ga, Gb = np.zeros((k, k), order="F"), np.zeros((k, k), order="F")
        check(self.L.mi_lobpcg_gram_pair(self.h, m, k, S.h, k1a, Ta1.h, Ta2.h if Ta2 is not None else None, k1b,
                                         Tb1.h, Tb2.h if Tb2 is not None else None, _dp(Ga), _dp(Gb)))
        return Ga, Gb

    def lobpcg_gram_pair_sym(self, m, S, k, Ta1, k1a, Ta2):
        """(S'[Ta1|Ta2] for a symmetric product, S'S) from one pass (mi_lobpcg_gram_pair_sym); Ta2 may be None"""
        Ga, Gb = np.zeros((k, k), order="F"), np.zeros((k, k), order="F")
        check(self.L.mi_lobpcg_gram_pair_sym(self.h, m, k, S.h, k1a, Ta1.h, Ta2.h if Ta2 is not None else None,
                                             _dp(Ga), _dp(Gb)))
        return Ga, Gb

    def lobpcg_gram_pair_sym_blocks(self, m, blocks, Ta1, k1a, Ta2):
        """the same on a basis held as column blocks [(panel or view, columns), ...] (mi_lobpcg_gram_pair_sym_blocks)"""
        k = sum(c for _, c in blocks)
        Ga, Gb = np.zeros((k, k), order="F"), np.zeros((k, k), order="F")
        pb = PanelBlocks.of(blocks)
        check(self.L.mi_lobpcg_gram_pair_sym_blocks(self.h, m, C.byref(pb), k1a, Ta1.h,
                                                    Ta2.h if Ta2 is not None else None, _dp(Ga), _dp(Gb)))
        return Ga, Gb

    def lobpcg_gram_pair_sym_tblocks(self, m, S_blocks, T_blocks):
        """(S'A(S), S'S), S and A(S) as column blocks (mi_lobpcg_gram_pair_sym_tblocks)"""
        k = sum(c for _, c in S_blocks)
        Ga, Gb = np.zeros((k, k), order="F"), np.zeros((k, k), order="F")
        ps, pt = PanelBlocks.of(S_blocks), PanelBlocks.of(T_blocks)
        check(self.L.mi_lobpcg_gram_pair_sym_tblocks(self.h, m, C.byref(ps), C.byref(pt), _dp(Ga), _dp(Gb)))
        return Ga, Gb

    def lobpcg_gram_pair_gen_blocks(self, m, S_blocks, AS_blocks, BS_blocks):
        """(S'A(S), S'B(S)) of the generalized problem, all three panels as column blocks (mi_lobpcg_gram_pair_gen_blocks)"""
        k = sum(c for _, c in S_blocks)
        Ga, Gb = np.zeros((k, k), order="F"), np.zeros((k, k), order="F")
        ps, pa, pb = PanelBlocks.of(S_blocks), PanelBlocks.of(AS_blocks), PanelBlocks.of(BS_blocks)
        check(self.L.mi_lobpcg_gram_pair_gen_blocks(self.h, m, C.byref(ps), C.byref(pa), C.byref(pb), _dp(Ga), _dp(Gb)))
        return Ga, Gb

    def lobpcg_update2_blocks(self, m, blocks, Cmat, k1):
        """mi_lobpcg_update2_blocks: S C for a basis held as column blocks, columns [0, k1) and the rest"""
        Cmat = np.asfortranarray(Cmat, dtype=np.float64)
        kc = Cmat.shape[1]
        Y1, Y2 = Vec(self, m * k1), Vec(self, m * max(kc - k1, 1))
        pb = PanelBlocks.of(blocks)
        check(self.L.mi_lobpcg_update2_blocks(self.h, m, C.byref(pb), kc, _dp(Cmat), Cmat.shape[0], Y1.h, k1,
                                              Y2.h if k1 < kc else None))
        return Y1, Y2

    def lobpcg_update(self, m, S, ks, Cmat):
        Cmat = np.asfortranarray(Cmat, dtype=np.float64)
        kc = Cmat.shape[1]
        Y = Vec(self, m * kc)
        check(self.L.mi_lobpcg_update(self.h, m, ks, kc, S.h, _dp(Cmat), Cmat.shape[0], Y.h))
        return Y

    def lobpcg_update2(self, m, S, ks, Cmat, k1):
        """columns [0, k1) of S C into one panel, the rest into another (mi_lobpcg_update2)"""
        Cmat = np.asfortranarray(Cmat, dtype=np.float64)
        kc = Cmat.shape[1]
        Y1, Y2 = Vec(self, m * k1), Vec(self, m * max(kc - k1, 1))
        check(self.L.mi_lobpcg_update2(self.h, m, ks, kc, S.h, _dp(Cmat), Cmat.shape[0], Y1.h, k1,
                                       Y2.h if k1 < kc else None))
        return Y1, Y2

    def lobpcg_residual(self, m, nx, AX, BX, X, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        R = Vec(self, m * nx)
        rn, xn = np.zeros(nx), np.zeros(nx)
        check(self.L.mi_lobpcg_residual(self.h, m, nx, AX.h, BX.h, X.h, _dp(theta), R.h, _dp(rn), _dp(xn)))
        return R, rn, xn

    def rayleigh_ritz(self, A, B):
        A = np.asfortranarray(A, dtype=np.float64)
        B = np.asfortranarray(B, dtype=np.float64)
        n = A.shape[0]
        th = np.zeros(n)
        Cm = np.zeros((n, n), order="F")
        check(self.L.mi_rayleigh_ritz(n, _dp(A), _dp(B), _dp(th), _dp(Cm)))
        return th, Cm

    def rayleigh_ritz_lowest(self, A, B, k):
        A = np.asfortranarray(A, dtype=np.float64)
        B = np.asfortranarray(B, dtype=np.float64)
        n = A.shape[0]
        th = np.zeros(k)
        Cm = np.zeros((n, k), order="F")
        check(self.L.mi_rayleigh_ritz_lowest(n, k, _dp(A), _dp(B), _dp(th), _dp(Cm)))
        return th, Cm

    def panel_rowscale(self, m, k, d, X):
        Y = Vec(self, m * k)
        check(self.L.mi_panel_rowscale(self.h, m, k, d.h, X.h, Y.h))
        return Y

    # comm -------------------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = (C.c_ubyte * 128)()
        check(self.L.mi_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, world_size, rank, uid):
        buf = (C.c_ubyte * 128)(*uid)
        check(self.L.mi_comm_init(self.h, world_size, rank, buf))

    # peer-memory exchange layer (include/mi355opt.h: mi_comm_ipc_*) -----------------------------
    IPC_HANDLE_BYTES = 64

    def comm_ipc_export(self):
        buf = (C.c_ubyte * self.IPC_HANDLE_BYTES)()
        check(self.L.mi_comm_ipc_export(self.h, buf))
        return bytes(buf)

    def comm_ipc_attach(self, world_size, rank, handles):
        assert len(handles) == world_size * self.IPC_HANDLE_BYTES
        buf = (C.c_ubyte * len(handles)).from_buffer_copy(handles)
        check(self.L.mi_comm_ipc_attach(self.h, world_size, rank, buf))

    def comm_ipc_selftest(self):
        ok = C.c_int(0)
        check(self.L.mi_comm_ipc_selftest(self.h, C.byref(ok)))
        return bool(ok.value)

    def comm_ipc_enable(self, on=True):
        check(self.L.mi_comm_ipc_enable(self.h, int(on)))

    def comm_ipc_error(self):
        e = C.c_int(0)
        check(self.L.mi_comm_ipc_error(self.h, C.byref(e)))
        return e.value

    def comm_ipc_fold(self, on):
        check(self.L.mi_comm_ipc_fold(self.h, int(on)))

    def comm_kernel_launches(self):
        """(scalar-exchange kernels, halo-push kernels, halo pushes folded into the producer kernel, of those: in the
        early form) so far"""
        out = (C.c_ulonglong * 4)()
        check(self.L.mi_comm_kernel_launches(self.h, out))
        return tuple(int(v) for v in out)

    def enable_peer_memory(self, world_size, rank, dist, force=False):
        """Collective bring-up of the peer-memory layer over a torch.distributed (gloo) control plane: export,
        gather handles, map, self-test; enabled only if EVERY rank succeeded at every step, else it stays off
        (RCCL then does the small exchanges).  Returns True when enabled.
        DEFAULT since r03: the layer is brought up and used whenever its collective self-test passes on every rank
        (bench.py additionally verifies the sharded data path through it and falls back to RCCL on any failure);
        MI355OPT_COMM=rccl keeps RCCL in charge.  force=True: the one-GPU multi-process tests, where RCCL cannot run."""
        import os

        def all_ok(flag, payload=None):
            lst = [None] * world_size
            dist.all_gather_object(lst, (bool(flag), payload))
            return all(f for f, _ in lst), [pl for _, pl in lst]

        if not force and os.environ.get("MI355OPT_COMM", "peer") == "rccl":
            return False
        try:
            handle, ok = self.comm_ipc_export(), True
        except MiError:
            handle, ok = b"\0" * self.IPC_HANDLE_BYTES, False
        ok, handles = all_ok(ok, handle)
        if not ok:
            return False
        try:
            self.comm_ipc_attach(world_size, rank, b"".join(handles))
            ok = True
        except MiError:
            ok = False
        ok, _ = all_ok(ok)
        if not ok:
            return False
        try:
            ok = self.comm_ipc_selftest()
        except MiError:
            ok = False
        ok, _ = all_ok(ok)
        self.comm_ipc_enable(ok)
        return ok

    def debug_set_rank(self, world_size, rank):
        """verification hook: act as `rank` of `world_size` without a communicator (tests only)"""
        check(self.L.mi_debug_set_rank(self.h, world_size, rank))

    def comm_finalize(self):
        check(self.L.mi_comm_finalize(self.h))

    def comm_rccl_count(self):
        """ranks of the RCCL communicator as RCCL reports them (ncclCommCount); 0 without one"""
        n = C.c_int(0)
        check(self.L.mi_comm_rccl_count(self.h, C.byref(n)))
        return n.value


class Vec:
    def __init__(self, ctx, n, handle=None):
        self.ctx = ctx
        self.L = ctx.L
        self.owned = handle is None
        if handle is None:
            self.h = vp()
            check(self.L.mi_vec_create(ctx.h, n, C.byref(self.h)))
            self.n = n
        else:
            self.h = handle
            m = C.c_size_t(0)
            check(self.L.mi_vec_len(self.h, C.byref(m)))
            self.n = m.value

    def __del__(self):
        try:
            if self.owned and self.h and self.ctx.h:
                self.L.mi_vec_destroy(self.h)
        except Exception:  # noqa
            pass

    def numpy(self):
        out = np.zeros(self.n)
        check(self.L.mi_vec_download(self.h, _dp(out), self.n))
        return out

    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        check(self.L.mi_vec_upload(self.h, _dp(arr), arr.size))
        return self

    def copy(self):
        v = Vec(self.ctx, self.n)
        check(self.L.mi_vec_copy(v.h, self.h))
        return v

    def fill(self, a):
        check(self.L.mi_vec_fill(self.h, a))
        return self

    def view(self, offset, n):
        """non-owning window [offset, offset + n) (mi_vec_view); this vector must outlive it"""
        h = vp()
        check(self.L.mi_vec_view(self.h, offset, n, C.byref(h)))
        v = Vec(self.ctx, 0, handle=h)
        v.owned, v.base = True, self      # the VIEW object is ours to destroy (not the storage)
        return v

    def data_ptr(self):
        p = vp()
        check(self.L.mi_vec_data(self.h, C.byref(p)))
        return p.value

    def touch(self):
        """announce a write made through data_ptr() outside the library (mi_vec_touch)"""
        check(self.L.mi_vec_touch(self.h))
        return self

    def scale(self, a):
        check(self.L.mi_vec_scale(self.h, a))
        return self

    def div(self, a):
        check(self.L.mi_vec_div(self.h, a))
        return self

    def scaled(self, a):
        z = Vec(self.ctx, self.n)
        check(self.L.mi_vec_scale_to(z.h, a, self.h))
        return z

    def axpy(self, a, x):
        check(self.L.mi_vec_axpy(self.h, a, x.h))
        return self

    def axpby(self, a, x, b, y):
        check(self.L.mi_vec_axpby(self.h, a, x.h, b, y.h))
        return self

    def dot(self, other):
        out = C.c_double(0)
        check(self.L.mi_vec_dot(self.h, other.h, C.byref(out)))
        return out.value


class Csr:
    def __init__(self, ctx, n, rowptr, col, val):
        self.ctx, self.L = ctx, ctx.L
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        self.n, self.nnz = n, int(rowptr[-1])
        self.h = vp()
        check(self.L.mi_csr_create(ctx.h, n, self.nnz, rowptr.ctypes.data_as(c_int32_p),
                                   col.ctypes.data_as(c_int32_p), _dp(val), C.byref(self.h)))

    def spmm(self, p, V, W=None):
        W = W if W is not None else Vec(self.ctx, self.n * p)
        check(self.L.mi_csr_spmm(self.h, p, V.h, W.h))
        return W

    def window_info(self):
        """(window half-width in chunks, widest slice, far stride D if the far columns are computed else 0, halo rows)"""
        out = (C.c_size_t * 4)()
        check(self.L.mi_debug_csr_window_info(self.h, out))
        return tuple(int(v) for v in out)

    def debug_set_halo(self, p, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        check(self.L.mi_debug_csr_set_halo(self.h, p, _dp(rows)))

    def spmm_colmajor(self, k, X, Y=None):
        Y = Y if Y is not None else Vec(self.ctx, self.n * k)
        check(self.L.mi_csr_spmm_colmajor(self.h, k, X.h, Y.h))
        return Y

    def spmm_colmajor_blocks(self, blocks):
        """mi_csr_spmm_colmajor_blocks: A [block 0 | block 1 | ...] for column blocks [(panel or view, columns), ...]"""
        k = sum(c for _, c in blocks)
        Y = Vec(self.ctx, self.n * k)
        pb = PanelBlocks.of(blocks)
        check(self.L.mi_csr_spmm_colmajor_blocks(self.h, C.byref(pb), Y.h))
        return Y

    def spmm_colmajor_residual(self, nx, X, theta):
        """mi_csr_spmm_colmajor_residual: (AX, R, rnorm, xnorm)"""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        AX, R = Vec(self.ctx, self.n * nx), Vec(self.ctx, self.n * nx)
        rn, xn = np.zeros(nx), np.zeros(nx)
        check(self.L.mi_csr_spmm_colmajor_residual(self.h, nx, X.h, _dp(theta), AX.h, R.h, _dp(rn), _dp(xn)))
        return AX, R, rn, xn

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.L.mi_csr_destroy(self.h)
        except Exception:  # noqa
            pass


class Op:
    def __init__(self, ctx, h, keep=(), borrowed=False):
        self.ctx, self.L, self.h, self.keep, self.borrowed = ctx, ctx.L, h, list(keep), borrowed

    def apply(self, x, out=None):
        out = out if out is not None else Vec(self.ctx, x.n)
        check(self.L.mi_op_apply(self.h, x.h, out.h))
        return out

    def time_fused_apply(self, x, out, reps=200):
        """average us per application in the fused form mi_stpcg uses (mi_debug_time_fused_apply)"""
        us = C.c_double(0)
        check(self.L.mi_debug_time_fused_apply(self.h, x.h, out.h, reps, C.byref(us)))
        return us.value

    def __del__(self):
        try:
            if not self.borrowed and self.h and self.ctx.h:
                self.L.mi_op_destroy(self.h)
        except Exception:  # noqa
            pass


class Precon:
    def __init__(self, ctx, h, keep=()):
        self.ctx, self.L, self.h, self.keep = ctx, ctx.L, h, list(keep)

    def apply(self, r, out=None):
        out = out if out is not None else Vec(self.ctx, r.n)
        check(self.L.mi_precon_apply(self.h, r.h, out.h))
        return out

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.L.mi_precon_destroy(self.h)
        except Exception:  # noqa
            pass


class ConstraintPrecon(Precon):
    """The constraint preconditioner of the projected STPCG (mi_precon_create_constraint)."""

    def solve(self, r):
        """(v, lambda) = P(r)"""
        v, lam = Vec(self.ctx, self.n), Vec(self.ctx, self.m)
        check(self.L.mi_precon_constraint_solve(self.h, r.h, v.h, lam.h))
        return v, lam

    def At(self, lam):
        out = Vec(self.ctx, self.n)
        check(self.L.mi_precon_constraint_At(self.h, lam.h, out.h))
        return out

    def info(self):
        """(inner iterations of the last application, its relative residual, the worst one so far, status code: 0 ok,
        1 the inner iteration broke down, 2 it stopped at its iteration limit short of its tolerance): sparse form only.
        Never raises on a valid handle -- the code is an output."""
        it, last, worst, code = C.c_size_t(0), C.c_double(0), C.c_double(0), C.c_int(0)
        check(self.L.mi_precon_constraint_info(self.h, C.byref(it), C.byref(last), C.byref(worst), C.byref(code)))
        return it.value, last.value, worst.value, code.value


class _BorrowedPrecon(Precon):
    """A preconditioner owned by a problem object (never destroyed from Python)."""

    def __del__(self):
        pass


class So3N:
    """Chordal rotation averaging on SO(3)^N (mi_so3n_*)."""

    def __init__(self, ctx, N, ei, ej, Rt, w):
        self.ctx, self.L, self.N = ctx, ctx.L, N
        ei = np.ascontiguousarray(ei, dtype=np.int32)
        ej = np.ascontiguousarray(ej, dtype=np.int32)
        Rt = np.ascontiguousarray(Rt, dtype=np.float64)
        w = np.ascontiguousarray(w, dtype=np.float64)
        self.E = ei.size
        self.h = vp()
        check(self.L.mi_so3n_create(ctx.h, N, ei.size, ei.ctypes.data_as(c_int32_p),
                                    ej.ctypes.data_as(c_int32_p), _dp(Rt), _dp(w), C.byref(self.h)))

    def objective(self, R):
        f = C.c_double(0)
        check(self.L.mi_so3n_objective(self.h, R.h, C.byref(f)))
        return f.value

    def model(self, R, with_precon=True):
        """returns (grad Vec, Hessian Op, block-Jacobi Precon or None), all bound to R"""
        g = Vec(self.ctx, 3 * self.N)
        hop, pc = vp(), vp()
        check(self.L.mi_so3n_model(self.h, R.h, g.h, C.byref(hop), C.byref(pc) if with_precon else None))
        P = None
        if with_precon:
            P = Precon.__new__(Precon)
            P.ctx, P.L, P.h, P.keep = self.ctx, self.L, pc, [self, R]
            P.__class__ = _BorrowedPrecon
        return g, Op(self.ctx, hop, keep=[self, R], borrowed=True), P

    def retract(self, R, xi):
        Y = Vec(self.ctx, 9 * self.N)
        check(self.L.mi_so3n_retract(self.h, R.h, xi.h, Y.h))
        return Y

    def trial(self, R, h, g, with_precon=True):
        """mi_so3n_trial: (R_trial Vec, dict(f, hh, gh, hHh, grad_sqnorm, precon_grad_sqnorm))"""
        Rt = Vec(self.ctx, 9 * self.N)
        out = np.zeros(6)
        check(self.L.mi_so3n_trial(self.h, R.h, h.h, g.h, int(with_precon), Rt.h, _dp(out)))
        return Rt, dict(f=out[0], hh=out[1], gh=out[2], hHh=out[3], grad_sqnorm=out[4], precon_grad_sqnorm=out[5])

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.L.mi_so3n_destroy(self.h)
        except Exception:  # noqa
            pass


class StiefelRQ:
    def __init__(self, ctx, A, n, p):
        self.ctx, self.L, self.A, self.n, self.p = ctx, ctx.L, A, n, p
        self.h = vp()
        check(self.L.mi_stiefel_rq_create(ctx.h, A.h, n, p, C.byref(self.h)))

    def objective(self, X):
        f = C.c_double(0)
        check(self.L.mi_stiefel_rq_objective(self.h, X.h, C.byref(f)))
        return f.value

    def model(self, X):
        """returns (grad Vec, Hessian Op bound to X)"""
        g = Vec(self.ctx, self.n * self.p)
        hop = vp()
        check(self.L.mi_stiefel_rq_model(self.h, X.h, g.h, C.byref(hop)))
        return g, Op(self.ctx, hop, keep=[self, X], borrowed=True)

    def trial(self, X, h, g):
        """mi_stiefel_rq_trial: (X_trial Vec, dict(f, hh, gh, hHh, grad_sqnorm))"""
        Xt = Vec(self.ctx, self.n * self.p)
        out = np.zeros(5)
        check(self.L.mi_stiefel_rq_trial(self.h, X.h, h.h, g.h, Xt.h, _dp(out)))
        return Xt, dict(f=out[0], hh=out[1], gh=out[2], hHh=out[3], grad_sqnorm=out[4])

    def precon(self, X, dinv_rows):
        h = vp()
        check(self.L.mi_stiefel_rq_precon(self.h, X.h, dinv_rows.h, C.byref(h)))
        return Precon(self.ctx, h, keep=[self, X, dinv_rows])

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.L.mi_stiefel_rq_destroy(self.h)
        except Exception:  # noqa
            pass
