"""Seeded synthetic inputs for the BASELINE.json configurations (host side, numpy).

The reference has no Stiefel / SO(3)^N / sparse code: these generators are the
harness's problem definitions.  The same arrays are handed to the GPU path and to the CPU oracle.
"""
import numpy as np


def laplacian_3d(nx, ny, nz, shift=0.1, z_range=None):
    """7-point Dirichlet Laplacian on an nx x ny x nz grid + shift*I, CSR (int32 columns), rows
    ordered x fastest.  BASELINE cfg2: 100^3, shift 0.1 -> n=1e6, nnz=6 940 000, kappa ~ 120.
    z_range=(z0,z1): only rows of those z-planes (row-sharded slab); columns stay GLOBAL (int64)."""
    z0, z1 = (0, nz) if z_range is None else z_range
    n_loc = nx * ny * (z1 - z0)
    idx = np.arange(nx * ny * z0, nx * ny * z1, dtype=np.int64)
    ix = idx % nx
    iy = (idx // nx) % ny
    iz = idx // (nx * ny)
    cols = [idx]
    vals = [np.full(n_loc, 6.0 + shift)]
    mask = [np.ones(n_loc, dtype=bool)]
    for cond, off in ((iz > 0, -nx * ny), (iy > 0, -nx), (ix > 0, -1), (ix < nx - 1, 1),
                      (iy < ny - 1, nx), (iz < nz - 1, nx * ny)):
        cols.append(idx + off)
        vals.append(np.full(n_loc, -1.0))
        mask.append(cond)
    # order entries by column within each row: -nxny, -nx, -1, diag, +1, +nx, +nxny
    order = [1, 2, 3, 0, 4, 5, 6]
    colm = np.stack([cols[o] for o in order], axis=1)
    valm = np.stack([vals[o] for o in order], axis=1)
    mskm = np.stack([mask[o] for o in order], axis=1)
    counts = mskm.sum(axis=1)
    rowptr = np.zeros(n_loc + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    col = colm[mskm]
    val = valm[mskm]
    if z_range is None:
        col = col.astype(np.int32)
    return rowptr.astype(np.int32), col, val.astype(np.float64)


def laplacian_3d_eigvec(nx, ny, nz, kx, ky, kz):
    """Exact eigenvector (unit norm) of laplacian_3d for mode (kx,ky,kz) >= 1 and its eigenvalue
    (without shift): 4 sum sin^2(pi k / (2 (n+1)))."""
    sx = np.sin(np.pi * kx * np.arange(1, nx + 1) / (nx + 1))
    sy = np.sin(np.pi * ky * np.arange(1, ny + 1) / (ny + 1))
    sz = np.sin(np.pi * kz * np.arange(1, nz + 1) / (nz + 1))
    v = (sz[:, None, None] * sy[None, :, None] * sx[None, None, :]).ravel()
    v /= np.linalg.norm(v)
    lam = 4 * (np.sin(np.pi * kx / (2 * (nx + 1))) ** 2 + np.sin(np.pi * ky / (2 * (ny + 1))) ** 2 +
               np.sin(np.pi * kz / (2 * (nz + 1))) ** 2)
    return v, lam


_WL = None


def _wl():
    """optimization_amd/libmi355wl.so (wlgen.c): mt19937_64, an own sin and an own thin QR -- integer arithmetic and
    IEEE +,-,*,/,sqrt in a fixed order, so that the arrays below are the same BYTES on every host (SURVEY.md 8(d):
    "mt19937_64(seed) U(-1,1), host-generated"; r01-r05 used numpy's PCG64, libm's sin and LAPACK's QR, whose last bits
    follow the CPU model).  Built on demand with gcc when it is not there."""
    global _WL
    if _WL is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmi355wl.so")
        src = os.path.join(os.path.dirname(path), "wlgen.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            from optimization_amd import build as _b
            _b.build_wlgen()
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        L.wl_uniform_pm1.argtypes = [C.c_uint64, C.c_size_t, dp]
        L.wl_uniform_pm1.restype = None
        L.wl_add_uniform_pm1.argtypes = [C.c_uint64, C.c_size_t, C.c_double, dp]
        L.wl_add_uniform_pm1.restype = None
        L.wl_mt19937_64_raw.argtypes = [C.c_uint64, C.c_size_t, C.POINTER(C.c_uint64)]
        L.wl_mt19937_64_raw.restype = None
        L.wl_thin_qr.argtypes = [C.c_size_t, C.c_size_t, dp]
        L.wl_thin_qr.restype = C.c_int
        L.wl_grid_mode.argtypes = [C.c_int64] * 6 + [dp, C.c_size_t]
        L.wl_grid_mode.restype = C.c_int
        L.wl_sin_pi_ratio.argtypes = [C.c_int64, C.c_int64]
        L.wl_sin_pi_ratio.restype = C.c_double
        _WL = (L, dp)
    return _WL


def mt19937_64_raw(seed, count):
    L, _ = _wl()
    import ctypes as C
    out = np.zeros(count, dtype=np.uint64)
    L.wl_mt19937_64_raw(seed, count, out.ctypes.data_as(C.POINTER(C.c_uint64)))
    return out


def uniform_pm1(seed, count):
    """count doubles U(-1,1) from mt19937_64(seed): 2 ((x >> 11) 2^-53) - 1"""
    L, dp = _wl()
    out = np.zeros(count)
    L.wl_uniform_pm1(seed, count, out.ctypes.data_as(dp))
    return out


def thin_qr(M):
    """Q factor (positive diagonal of R) of the n x p matrix M, own arithmetic (Gram-Schmidt twice, wlgen.c)"""
    L, dp = _wl()
    Q = np.array(M, dtype=np.float64, order="C")
    if L.wl_thin_qr(Q.shape[0], Q.shape[1], Q.ctypes.data_as(dp)):
        raise ValueError("thin_qr: dependent columns")
    return Q


def random_stiefel(n, p, seed=20260928):
    """X0 = Q factor of an n x p matrix of mt19937_64(seed) U(-1,1) entries (row-major n x p, X0' X0 = I): SURVEY.md
    8(d)'s cfg2 start, the same bytes on every host."""
    return thin_qr(uniform_pm1(seed, n * p).reshape(n, p))


def polar(Y):
    """polar factor Y (Y'Y)^-1/2 (host helper)."""
    G = Y.T @ Y
    w, Q = np.linalg.eigh(G)
    return Y @ (Q @ np.diag(w ** -0.5) @ Q.T)


def lowest_modes(nx, ny, nz, p):
    """the p lowest modes (kx, ky, kz) of the grid Laplacian; eigenvalues that agree to 1e-12 (the degenerate triples
    of a cubic grid) are ordered by the index tuple, so the choice does not hang on libm's last bits"""
    def lam(m):
        return 4 * sum(np.sin(np.pi * k / (2 * (n + 1))) ** 2 for k, n in zip(m, (nx, ny, nz)))
    ks = range(1, max(3, p) + 2)   # (on a long thin grid the p lowest modes all lie along one axis)
    return sorted(((kx, ky, kz) for kx in ks for ky in ks for kz in ks), key=lambda m: (round(lam(m), 12), m))[:p]


def stiefel_bench_iterate(nx, ny, nz, p=3, eps=1e-3, seed=7):
    """A point near the minimiser of f(X) = .5 tr(X'AX) on the grid Laplacian: the p lowest exact
    eigenvectors, perturbed by eps * U(-1,1) / sqrt(n) (mt19937_64(seed)) and re-orthonormalised (thin QR).  There the
    Riemannian Hessian is (numerically) positive semi-definite, so STPCG runs its full iteration budget -- the regime in
    which a TNT solve spends nearly all of its inner iterations.  Machine-independent bytes (wlgen.c)."""
    L, dp = _wl()
    modes = lowest_modes(nx, ny, nz, p)
    n = nx * ny * nz
    X = np.zeros((n, p))
    for j, m in enumerate(modes):
        if L.wl_grid_mode(nx, ny, nz, *m, X[:, j:].ctypes.data_as(dp), p):
            raise MemoryError
    L.wl_add_uniform_pm1(seed, n * p, eps / np.sqrt(float(n)), X.ctypes.data_as(dp))
    return thin_qr(X), modes


def hat(x):
    return np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]])


def so3_exp(xi):
    """batched Rodrigues: xi (N,3) -> (N,3,3)"""
    xi = np.atleast_2d(xi)
    th = np.linalg.norm(xi, axis=1)
    K = np.zeros((xi.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -xi[:, 2], xi[:, 1]
    K[:, 1, 0], K[:, 1, 2] = xi[:, 2], -xi[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -xi[:, 1], xi[:, 0]
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0, np.sin(ths) / ths)
    b = np.where(small, 0.5, (1 - np.cos(ths)) / (ths * ths))
    return np.eye(3)[None] + a[:, None, None] * K + b[:, None, None] * (K @ K)


def pose_graph(N, chords_per_node=2, sigma=0.05, seed=7, init_sigma=0.2):
    """Synthetic rotation-averaging problem (BASELINE cfg3): ring + `chords_per_node` random chords
    per node; measurements Rt_e = R_i' R_j exp(noise); returns (ei, ej, Rt (E,9), w (E,), R_true
    (N,9), R_init (N,9))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    R_true = so3_exp(rng.normal(size=(N, 3)) * 1.5)
    ei = [np.arange(N, dtype=np.int64)]
    ej = [(np.arange(N, dtype=np.int64) + 1) % N]
    for _ in range(chords_per_node):
        a = np.arange(N, dtype=np.int64)
        b = rng.integers(0, N, size=N)
        b = np.where(b == a, (b + 2) % N, b)
        ei.append(a)
        ej.append(b)
    ei = np.concatenate(ei)
    ej = np.concatenate(ej)
    E = ei.size
    noise = so3_exp(rng.normal(size=(E, 3)) * sigma)
    Rt = np.einsum("eba,ebc->eac", R_true[ei], R_true[ej]) @ noise  # R_i' R_j * noise
    w = np.ones(E)
    R_init = R_true @ so3_exp(rng.normal(size=(N, 3)) * init_sigma)
    return (ei.astype(np.int32), ej.astype(np.int32), np.ascontiguousarray(Rt.reshape(E, 9)), w,
            np.ascontiguousarray(R_true.reshape(N, 9)), np.ascontiguousarray(R_init.reshape(N, 9)))


def shard_rows(n_planes, world_size):
    """z-plane ranges of a slab partition: list of (z0, z1) per rank, as even as possible."""
    base, rem = divmod(n_planes, world_size)
    out, z = [], 0
    for r in range(world_size):
        k = base + (1 if r < rem else 0)
        out.append((z, z + k))
        z += k
    return out


def cfg2_grid(world_size=1):
    """Grid for the weak-scaled Stiefel(1e6 * world_size, 3) workload (BASELINE cfg2 / cfg4):
    1 -> 100^3, 2 -> 100x100x200, 4 -> 100x200x200, 8 -> 200^3; z-slab sharded."""
    return {1: (100, 100, 100), 2: (100, 100, 200), 4: (100, 200, 200), 8: (200, 200, 200)}.get(
        world_size, (100, 100, 100 * world_size))


# Algorithmic bytes -------------------------------------------------------------
def cg_bytes_per_iter(N, precon="none"):
    return {"none": 88, "diag": 104, "block3": 120}[precon] * N


def spmm_bytes(n, nnz, p):
    return 12 * nnz + 4 * (n + 1) + 16 * n * p


def stiefel_hvp_bytes(n, nnz, p):
    """CSR SpMM + tangent-space finish (+56 N)"""
    return spmm_bytes(n, nnz, p) + 56 * n * p
