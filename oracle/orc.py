"""ctypes binding of oracle/liborc.so -- the plain-C CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY (see hif_oracle.h): may be imported from tests/,
__graft_entry__.smoke() and the cpu_baseline leg of bench.py -- never from `hifir_amd`.

Hierarchy format ("levels"): a list of dicts, one per hif::Prec (alg/Prec.hpp:309-323), exactly as
`oracle.ref.RefHIF.levels()` returns them and as the golden fixtures store them:
  m, n, {L,U,E,F}_{colptr,rowind,vals} (CCS), d, s, t, p, p_inv, q, q_inv,
  and on the last level optionally dense_n + dense (unfactored, column-major).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liborc.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
        vp = C.c_void_p
        for k in "dz":
            g = lambda name: getattr(_lib, f"orc_{k}_{name}")
            g("create").restype = vp
            g("create").argtypes = [C.c_int]
            g("destroy").argtypes = [vp]
            g("set_level").argtypes = [vp, C.c_int, C.c_int64, C.c_int64] + [vp] * 9 + [C.c_int64] + [vp] * 10
            g("set_dense").argtypes = [vp, C.c_int64, vp, C.c_double]
            g("set_dense_symm").argtypes = [vp, C.c_int64, vp, C.c_int]
            g("set_dense_lup").argtypes = [vp, C.c_int64, vp]
            g("lup").argtypes = [C.c_int64, vp, C.c_int, vp, vp]
            g("syeig").argtypes = [C.c_int64, vp, C.c_int, C.c_int, vp, C.c_int64, vp, vp, vp]
            g("dense_rank").argtypes = [vp]
            g("dense_rank").restype = C.c_int64
            g("work_size").argtypes = [vp]
            g("work_size").restype = C.c_int64
            g("solve").argtypes = [vp, vp, vp, C.c_int64]
            g("solve_batch").argtypes = [vp, vp, vp, C.c_int64, C.c_int64, C.c_int]
            g("solve_tran").argtypes = [vp, vp, vp, C.c_int64]
            g("solve_tran_batch").argtypes = [vp, vp, vp, C.c_int64, C.c_int64, C.c_int]
            g("mmultiply").argtypes = [vp, vp, vp, C.c_int64]
            g("mmultiply_tran").argtypes = [vp, vp, vp, C.c_int64]
            g("hifir").argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.c_int, vp, C.c_int64, vp, vp]
            g("crs_mv").argtypes = [C.c_int64, vp, vp, vp, vp, vp]
            g("ccs_kernel").argtypes = [C.c_int, C.c_int64, C.c_int64, vp, vp, vp, vp, vp]
            g("ccs_kernel_mrhs").argtypes = [C.c_int, C.c_int64, C.c_int64, vp, vp, vp, C.c_int64, vp, vp]
            g("qrcp").argtypes = [C.c_int64, vp, C.c_double, C.c_int, vp, C.c_int64, vp, vp]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _kind(*arrs):
    return "z" if any(np.iscomplexobj(a) for a in arrs if a is not None) else "d"


class Oracle:
    """CPU restatement of hif::HIF<>::solve / mmultiply / hifir on a given hierarchy."""

    def __init__(self, levels, rrqr_cond=0.0, dtype=None):
        L = lib()
        if dtype is None:
            dtype = np.complex128 if any(np.iscomplexobj(lv["L_vals"]) or np.iscomplexobj(lv["d"]) for lv in levels) else np.float64
        self.dtype = np.dtype(dtype)
        self.k = "z" if self.dtype == np.complex128 else "d"
        self.n = int(levels[0]["n"])
        self.h = self._f("create")(len(levels))
        keep = []
        for l, lv in enumerate(levels):
            m, n = int(lv["m"]), int(lv["n"])
            a = {}
            for nm in "LUEF":
                a[nm] = (np.ascontiguousarray(lv[nm + "_colptr"], dtype=np.int64),
                         np.ascontiguousarray(lv[nm + "_rowind"], dtype=np.int32),
                         np.ascontiguousarray(lv[nm + "_vals"], dtype=self.dtype))
            f_ncols = len(a["F"][0]) - 1
            if a["F"][0][-1] == 0 and n - m == 0:
                f_ncols = 0
            vec = [np.ascontiguousarray(lv["d"], dtype=self.dtype),
                   np.ascontiguousarray(lv["s"], dtype=np.float64), np.ascontiguousarray(lv["t"], dtype=np.float64)]
            perms = [np.ascontiguousarray(lv[k], dtype=np.int32) for k in ("p", "p_inv", "q", "q_inv")]
            keep += [a, vec, perms]
            rc = self._f("set_level")(self.h, l, m, n, *[_p(x) for x in a["L"]], *[_p(x) for x in a["U"]],
                                      *[_p(x) for x in a["E"]], f_ncols, *[_p(x) for x in a["F"]],
                                      *[_p(x) for x in vec], *[_p(x) for x in perms])
            assert rc == 0
        last = levels[-1]
        if int(last.get("dense_n", 0)) > 0:
            mat = np.ascontiguousarray(last["dense"], dtype=self.dtype).ravel()
            if int(last.get("dense_lup", 0)):  # reference built with HIF_DENSE_MODE=0: LUP last level
                self._f("set_dense_lup")(self.h, int(last["dense_n"]), _p(mat))
            elif int(last.get("dense_symm", 0)):  # is_symm hierarchy: Prec::symm_dense_solver (SYEIG)
                self._f("set_dense_symm")(self.h, int(last["dense_n"]), _p(mat), int(last.get("spd", 0)))
            else:
                self._f("set_dense")(self.h, int(last["dense_n"]), _p(mat), float(rrqr_cond))

    def _f(self, name):
        return getattr(lib(), f"orc_{self.k}_{name}")

    def close(self):
        if getattr(self, "h", None):
            self._f("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def dense_rank(self):
        return self._f("dense_rank")(self.h)

    def solve(self, b, rank=0, trans=False):
        b = np.ascontiguousarray(b, dtype=self.dtype)
        x = np.zeros_like(b)
        assert self._f("solve_tran" if trans else "solve")(self.h, _p(b), _p(x), rank) == 0
        return x

    def solve_batch(self, B, rank=0, threads=1, trans=False):
        """B: (n, nrhs) row-interleaved (C-contiguous). Column-by-column solve."""
        B = np.ascontiguousarray(B, dtype=self.dtype)
        X = np.zeros_like(B)
        assert self._f("solve_tran_batch" if trans else "solve_batch")(self.h, _p(B), _p(X), B.shape[1], rank, threads) == 0
        return X

    def mmultiply(self, x, rank=0, trans=False):
        x = np.ascontiguousarray(x, dtype=self.dtype)
        y = np.zeros_like(x)
        assert self._f("mmultiply_tran" if trans else "mmultiply")(self.h, _p(x), _p(y), rank) == 0
        return y

    def hifir(self, indptr, indices, vals, b, nirs, betas=None, rank=-1):
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=self.dtype)
        b = np.ascontiguousarray(b, dtype=self.dtype)
        x = np.zeros_like(b)
        st = np.zeros(2, dtype=np.int32)
        bt = None if betas is None else np.ascontiguousarray(betas, dtype=np.float64)
        assert self._f("hifir")(self.h, len(b), _p(indptr), _p(indices), _p(vals), _p(b), nirs, _p(bt), rank, _p(x), _p(st)) == 0
        return x, (int(st[0]), int(st[1]))


def crs_mv(indptr, indices, vals, x):
    k = _kind(vals, x)
    dt = np.complex128 if k == "z" else np.float64
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=dt)
    x = np.ascontiguousarray(x, dtype=dt)
    y = np.zeros(len(indptr) - 1, dtype=dt)
    getattr(lib(), f"orc_{k}_crs_mv")(len(indptr) - 1, _p(indptr), _p(indices), _p(vals), _p(x), _p(y))
    return y


def ccs_kernel(op, nrows, ncols, colptr, rowind, vals, x, nrhs=None):
    """op 0 strict-lower solve, 1 strict-upper solve, 2 y = A x, 3 / 4 the solves with the conjugate
    transpose, 5 y = A^H x (single RHS only for 3-5); nrhs: x is (n, nrhs) interleaved."""
    k = _kind(vals, x)
    dt = np.complex128 if k == "z" else np.float64
    colptr = np.ascontiguousarray(colptr, dtype=np.int64)
    rowind = np.ascontiguousarray(rowind, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=dt)
    x = np.ascontiguousarray(x, dtype=dt)
    if nrhs is None:
        y = np.zeros(nrows, dtype=dt) if op == 2 else (np.zeros(ncols, dtype=dt) if op == 5 else x.copy())
        getattr(lib(), f"orc_{k}_ccs_kernel")(op, nrows, ncols, _p(colptr), _p(rowind), _p(vals), _p(x), _p(y))
    else:
        y = np.zeros((nrows, nrhs), dtype=dt) if op == 2 else x.copy()
        getattr(lib(), f"orc_{k}_ccs_kernel_mrhs")(op, nrows, ncols, _p(colptr), _p(rowind), _p(vals), nrhs, _p(x), _p(y))
    return y


def qrcp(mat_colmajor, b, op=0, rank=0, rrqr_cond=0.0):
    k = _kind(mat_colmajor, b)
    dt = np.complex128 if k == "z" else np.float64
    mat = np.ascontiguousarray(mat_colmajor, dtype=dt).ravel()
    b = np.ascontiguousarray(b, dtype=dt)
    x = np.zeros_like(b)
    rk = np.zeros(1, dtype=np.int64)
    getattr(lib(), f"orc_{k}_qrcp")(len(b), _p(mat), rrqr_cond, op, _p(b), rank, _p(x), _p(rk))
    return x, int(rk[0])


def lup(mat_colmajor, b, op=0):
    """LUP (small_scale/LUP.hpp) on one block: op 0 solve, 1 multiply, 2 solve 'T', 3 multiply 'C'. Returns (x, info)."""
    k = _kind(mat_colmajor, b)
    dt = np.complex128 if k == "z" else np.float64
    mat = np.ascontiguousarray(mat_colmajor, dtype=dt).ravel()
    b = np.ascontiguousarray(b, dtype=dt)
    x = np.zeros_like(b)
    info = getattr(lib(), f"orc_{k}_lup")(len(b), _p(mat), op, _p(b), _p(x))
    return x, int(info)


def syeig(mat_colmajor, b, op=0, rank=0, spd=0):
    """SYEIG (small_scale/SYEIG.hpp) on one symmetric / Hermitian block: op 0 solve, 1 multiply.
    Returns (x, numerical rank, eigenvalues)."""
    k = _kind(mat_colmajor, b)
    dt = np.complex128 if k == "z" else np.float64
    mat = np.ascontiguousarray(mat_colmajor, dtype=dt).ravel()
    b = np.ascontiguousarray(b, dtype=dt)
    x = np.zeros_like(b)
    rk = np.zeros(1, dtype=np.int64)
    w = np.zeros(len(b))
    getattr(lib(), f"orc_{k}_syeig")(len(b), _p(mat), int(spd), op, _p(b), rank, _p(x), _p(rk), _p(w))
    return x, int(rk[0]), w


def fgmres(O, indptr, indices, vals, b, restart=30, rtol=1e-6, maxit=500, full_rank=False):
    """fgmres_hifir (examples/advanced/gmres.hpp:127-231): the same loop as gmres_hif, but the preconditioner
    of outer cycle k is HIF::hifir with 2^k refinement sweeps (:160-162), the preconditioned vectors are
    kept (Z, :164) and x is updated with Z y directly (:214-218).  Returns (x, flag, iterations, sweeps)."""
    return gmres(O, indptr, indices, vals, b, restart, rtol, maxit, full_rank, flexible=True)


def gmres(O, indptr, indices, vals, b, restart=30, rtol=1e-6, maxit=500, full_rank=False, flexible=False):
    """numpy restatement of the reference's right-preconditioned GMRES driver gmres_hif
    (examples/advanced/gmres.hpp:19-123) around the oracle's apply `O.solve`:
    x0 = 0, modified Gram-Schmidt, Givens rotations, relative residual |y_{j+1}| / ||b||.
    Returns (x, flag, iterations); flag 0 converged / 1 stagnated / 2 reached maxit.
    Real data: the driver line by line (pinned to the compiled driver in tests/test_oracle_vs_ref.py).
    Complex data: the same lines, rotations with the conjugates of :75-83, but the Hessenberg entry is the
    Hermitian product h = sum conj(q_i) v_i; the example's hif::inner(v, q) = sum conj(v_i) q_i
    (utils/math.hpp:83-90) is its conjugate and does not orthogonalize v against q, so there is no
    reference behaviour to pin for complex systems (checked by the true residual instead)."""
    import scipy.sparse as sp

    cplx = np.iscomplexobj(vals) or np.iscomplexobj(b)
    dt = np.complex128 if cplx else np.float64
    b = np.ascontiguousarray(b, dtype=dt)
    n = len(b)
    A = sp.csr_matrix((np.asarray(vals, dtype=dt), indices, indptr), shape=(n, n))
    rr = -1 if full_rank else 0                                     # :26
    it, flag, num_mv = 0, 0, 0
    beta0 = np.linalg.norm(b)                                        # :30
    x = np.zeros(n, dtype=dt)
    if beta0 == 0.0:                                                 # :36
        return (x, 0, 0, 0) if flexible else (x, 0, 0)
    Q = np.zeros((n, restart), dtype=dt)
    Z = np.zeros((n, restart), dtype=dt) if flexible else None
    R = np.zeros((restart, restart), dtype=dt)
    J = np.zeros((restart, 2), dtype=dt)
    y = np.zeros(restart + 1, dtype=dt)
    w2 = np.zeros(restart, dtype=dt)
    resid = 1.0
    for outer in range(int(np.ceil(maxit / restart))):               # :43-45
        v = b - A @ x if it else b.copy()                            # :48-52
        beta = np.linalg.norm(v)
        y[0] = beta
        Q[:, 0] = v / beta
        j = 0
        nirs = 1 << outer                                            # :157
        while True:
            if flexible:
                w, _ = O.hifir(indptr, indices, vals, Q[:, j].copy(), nirs, None, rr)   # :160
                num_mv += nirs
                Z[:, j] = w
            else:
                w = O.solve(Q[:, j].copy(), rank=rr)                 # :59
            v = A @ w                                                # :60
            for k in range(j + 1):                                   # :63-66
                w2[k] = np.vdot(Q[:, k], v) if cplx else v @ Q[:, k]
                v = v - w2[k] * Q[:, k]
            v_norm2 = float(np.vdot(v, v).real)
            v_norm = np.sqrt(v_norm2)
            if j + 1 < restart:
                Q[:, j + 1] = v / v_norm
            for cj in range(j):                                      # :73-78
                t0 = w2[cj]
                w2[cj] = np.conj(J[cj, 0]) * t0 + np.conj(J[cj, 1]) * w2[cj + 1]
                w2[cj + 1] = -J[cj, 1] * t0 + J[cj, 0] * w2[cj + 1]
            rho = np.sqrt((np.conj(w2[j]) * w2[j]).real + v_norm2)   # :79
            J[j, 0] = w2[j] / rho
            J[j, 1] = v_norm / rho
            y[j + 1] = -J[j, 1] * y[j]
            y[j] = np.conj(J[j, 0]) * y[j]
            w2[j] = rho
            R[:j + 1, j] = w2[:j + 1]
            resid_prev = resid
            resid = abs(y[j + 1]) / beta0                            # :89
            if resid >= resid_prev * (1.0 - 1e-8):                   # :90-93
                flag = 1
                break
            elif it >= maxit:                                        # :94-97
                flag = 2
                break
            it += 1
            if resid <= rtol or j + 1 >= restart:                    # :102
                break
            j += 1
        for k in range(j, -1, -1):                                   # :106-110
            y[k] /= R[k, k]
            y[:k] -= y[k] * R[:k, k]
        if flexible:
            x = x + Z[:, :j + 1] @ y[:j + 1]                         # :214-218
        else:
            v = Q[:, :j + 1] @ y[:j + 1]                             # :112-116
            x = x + O.solve(v, rank=rr)                              # :118-119
        if resid <= rtol or flag != 0:                               # :120
            break
    return (x, flag, it, num_mv) if flexible else (x, flag, it)
