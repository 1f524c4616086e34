"""ctypes binding of oracle/_ref/libhifref.so (the REAL reference, prebuilt by oracle/Makefile).

TEST INFRASTRUCTURE ONLY: may be imported from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from the product package `hifir_amd`.
The library wraps hif::HIF<> of /root/reference (src/hif/builder.hpp:109); see ref_shim.cpp.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libhifref.so")
_PATH_LUP = os.path.join(_HERE, "_ref", "libhifref_lup.so")  # the same reference built with HIF_DENSE_MODE=0


def available():
    return os.path.exists(_PATH)


def available_lup():
    return os.path.exists(_PATH_LUP)


_lib = None
_lib_lup = None


def lib(lup=False):
    """lup=True: the reference compiled with -DHIF_DENSE_MODE=0, whose last level is LU with partial pivoting
    (small_scale/LUP.hpp) instead of QRCP."""
    global _lib, _lib_lup
    if lup:
        if _lib_lup is None:
            keep, _lib = _lib, None
            globals()["_PATH"], path0 = _PATH_LUP, _PATH
            try:
                _lib_lup = lib()
            finally:
                globals()["_PATH"] = path0
                _lib = keep
        return _lib_lup
    if _lib is None:
        _lib = C.CDLL(_PATH)
        vp, i64p, i32p, dp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
        _lib.hifref_error.restype = C.c_char_p
        for k in "dz":
            f = getattr(_lib, f"hifref_{k}_factorize")
            f.restype = C.c_void_p
            f.argtypes = [C.c_size_t, i64p, i32p, vp, dp]
            getattr(_lib, f"hifref_{k}_destroy").argtypes = [C.c_void_p]
            getattr(_lib, f"hifref_{k}_nlevels").argtypes = [C.c_void_p]
            getattr(_lib, f"hifref_{k}_nnz").argtypes = [C.c_void_p]
            getattr(_lib, f"hifref_{k}_nnz").restype = C.c_int64
            getattr(_lib, f"hifref_{k}_level_sizes").argtypes = [C.c_void_p, C.c_int, i64p]
            getattr(_lib, f"hifref_{k}_level_ccs").argtypes = [C.c_void_p, C.c_int, C.c_int, i64p, i32p, vp]
            getattr(_lib, f"hifref_{k}_level_vectors").argtypes = [C.c_void_p, C.c_int] + [vp] * 7
            getattr(_lib, f"hifref_{k}_level_dense").argtypes = [C.c_void_p, C.c_int, vp]
            getattr(_lib, f"hifref_{k}_solve").argtypes = [C.c_void_p, vp, vp, C.c_int64]
            getattr(_lib, f"hifref_{k}_solve_tran").argtypes = [C.c_void_p, vp, vp, C.c_int64]
            getattr(_lib, f"hifref_{k}_mmultiply").argtypes = [C.c_void_p, vp, vp, C.c_int64]
            getattr(_lib, f"hifref_{k}_mmultiply_tran").argtypes = [C.c_void_p, vp, vp, C.c_int64]
            getattr(_lib, f"hifref_{k}_hifir").argtypes = [C.c_void_p, vp, C.c_int, dp, vp, i32p]
            if k == "d":
                _lib.hifref_d_gmres.argtypes = [C.c_void_p, vp, C.c_int, C.c_double, C.c_int, C.c_int, vp, i32p]
            getattr(_lib, f"hifref_{k}_spmv").argtypes = [C.c_size_t, i64p, i32p, vp, vp, vp]
            getattr(_lib, f"hifref_{k}_qrcp").argtypes = [C.c_size_t, vp, C.c_double, C.c_int, vp, C.c_int64, vp, i64p]
            getattr(_lib, f"hifref_{k}_ccs_kernel").argtypes = [C.c_int, C.c_size_t, C.c_size_t, i64p, i32p, vp, vp, vp]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_params(tau=0.0, kappa=0.0, alpha=0.0, dense_thres=0, rrqr_cond=0.0, is_symm=0, spd=0):
    """0 keeps the reference default (Options.h:135-164). PDE-tuned set: tau=1e-2, kappa=5, alpha=3.
    is_symm: symmetric factorization (symm_factor.hpp; last level = SYEIG); spd: Options::spd (+1 / 0 / -1)."""
    return np.array([tau, kappa, alpha, dense_thres, rrqr_cond, is_symm, spd], dtype=np.float64)


class RefHIF:
    """The reference's hif::HIF<double|complex<double>, int> behind ctypes."""

    def __init__(self, indptr, indices, vals, params=None, lup=False):
        self.lup = bool(lup)
        vals = np.ascontiguousarray(vals)
        self.k = "z" if np.iscomplexobj(vals) else "d"
        self.dtype = np.complex128 if self.k == "z" else np.float64
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.vals = vals.astype(self.dtype)
        self.n = len(self.indptr) - 1
        self.params = None if params is None else np.ascontiguousarray(params, dtype=np.float64)
        if self.params is not None and len(self.params) < 7:  # (older 5-entry parameter arrays)
            self.params = np.concatenate([self.params, np.zeros(7 - len(self.params))])
        L = lib(self.lup)
        self.h = getattr(L, f"hifref_{self.k}_factorize")(self.n, _p(self.indptr), _p(self.indices), _p(self.vals), _p(self.params))
        if not self.h:
            raise RuntimeError("reference factorize failed: " + L.hifref_error().decode())

    def _f(self, name):
        return getattr(lib(self.lup), f"hifref_{self.k}_{name}")

    def close(self):
        if self.h:
            self._f("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def nlevels(self):
        return self._f("nlevels")(self.h)

    @property
    def nnz(self):
        return self._f("nnz")(self.h)

    def level(self, l):
        """All arrays of one hif::Prec (Prec.hpp:309-323), CCS exactly as stored."""
        sz = np.zeros(11, dtype=np.int64)
        self._f("level_sizes")(self.h, l, _p(sz))
        m, n, nl, nu, ne, nf, nd, rk, enc, fnc, symm = [int(v) for v in sz]
        out = dict(m=m, n=n, dense_n=nd, dense_rank=rk)
        if symm == 1:  # Prec::symm_dense_solver: the last level of an is_symm factorization (SYEIG)
            out.update(dense_symm=1, spd=0 if self.params is None else int(self.params[6]))
        elif symm == 2:  # built with HIF_DENSE_MODE=0: the last level is LUP
            out.update(dense_lup=1)
        for which, (name, nz, nc) in enumerate([("L", nl, m), ("U", nu, m), ("E", ne, enc), ("F", nf, fnc)]):
            cp = np.zeros(nc + 1, dtype=np.int64)
            ri = np.zeros(nz, dtype=np.int32)
            v = np.zeros(nz, dtype=self.dtype)
            self._f("level_ccs")(self.h, l, which, _p(cp), _p(ri), _p(v))
            out[name + "_colptr"], out[name + "_rowind"], out[name + "_vals"] = cp, ri, v
        d = np.zeros(m, dtype=self.dtype)
        s = np.zeros(n)
        t = np.zeros(n)
        perms = [np.zeros(n, dtype=np.int32) for _ in range(4)]
        self._f("level_vectors")(self.h, l, _p(d), _p(s), _p(t), *[_p(a) for a in perms])
        out.update(d=d, s=s, t=t, p=perms[0], p_inv=perms[1], q=perms[2], q_inv=perms[3])
        if nd:
            mat = np.zeros(nd * nd, dtype=self.dtype)
            self._f("level_dense")(self.h, l, _p(mat))
            out["dense"] = mat  # column-major nd x nd, UNFACTORED (Prec.hpp:286-287)
        return out

    def levels(self):
        return [self.level(l) for l in range(self.nlevels)]

    def solve(self, b, rank=0, trans=False):
        """HIF::solve (builder.hpp:409-423); trans=True: x = M^{-H} b (prec_solve_tran)."""
        b = np.ascontiguousarray(b, dtype=self.dtype)
        x = np.zeros_like(b)
        if self._f("solve_tran" if trans else "solve")(self.h, _p(b), _p(x), rank):
            raise RuntimeError(lib().hifref_error().decode())
        return x

    def mmultiply(self, x, rank=0, trans=False):
        """HIF::mmultiply (builder.hpp:503-513): y = M x, or M^H x with trans=True."""
        x = np.ascontiguousarray(x, dtype=self.dtype)
        y = np.zeros_like(x)
        if self._f("mmultiply_tran" if trans else "mmultiply")(self.h, _p(x), _p(y), rank):
            raise RuntimeError(lib().hifref_error().decode())
        return y

    def hifir(self, b, nirs, betas=None):
        b = np.ascontiguousarray(b, dtype=self.dtype)
        x = np.zeros_like(b)
        st = np.zeros(2, dtype=np.int32)
        bt = None if betas is None else np.ascontiguousarray(betas, dtype=np.float64)
        if self._f("hifir")(self.h, _p(b), nirs, _p(bt), _p(x), _p(st)):
            raise RuntimeError(lib().hifref_error().decode())
        return x, (int(st[0]), int(st[1]))


def _gmres(self, b, restart=30, rtol=1e-6, maxit=500, full_rank=False):
    """gmres_hif of the reference's examples/advanced/gmres.hpp:19-123 (real only) on the factorized
    matrix: returns (x, flag, iterations); flag 0 converged / 1 stagnated / 2 reached maxit."""
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b)
    out = np.zeros(2, dtype=np.int32)
    if lib().hifref_d_gmres(self.h, _p(b), restart, rtol, maxit, int(full_rank), _p(x), _p(out)):
        raise RuntimeError(lib().hifref_error().decode())
    return x, int(out[0]), int(out[1])


RefHIF.gmres = _gmres


def _fgmres(self, b, restart=30, rtol=1e-6, maxit=500, full_rank=False):
    """fgmres_hifir of examples/advanced/gmres.hpp:127-231: returns (x, flag, iterations, sweeps)."""
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b)
    out = np.zeros(3, dtype=np.int32)
    lib().hifref_d_fgmres.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    if lib().hifref_d_fgmres(self.h, _p(b), restart, rtol, maxit, int(full_rank), _p(x), _p(out)):
        raise RuntimeError(lib().hifref_error().decode())
    return x, int(out[0]), int(out[1]), int(out[2])


RefHIF.fgmres = _fgmres


def _set_nsp_const(self, start=0, end=-1, trans=False):
    """HIF::nsp (trans: nsp_tran) = constant-mode filter on rows [start, end) (end < 0: to the end);
    start > end >= 0 removes it.  Real handles only."""
    lib().hifref_d_set_nsp_const.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64]
    lib().hifref_d_set_nsp_const(self.h, int(trans), start, end)


RefHIF.set_nsp_const = _set_nsp_const


def spmv(indptr, indices, vals, x):
    vals = np.ascontiguousarray(vals)
    k = "z" if np.iscomplexobj(vals) else "d"
    dt = np.complex128 if k == "z" else np.float64
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    x = np.ascontiguousarray(x, dtype=dt)
    y = np.zeros(len(indptr) - 1, dtype=dt)
    getattr(lib(), f"hifref_{k}_spmv")(len(indptr) - 1, _p(indptr), _p(indices), _p(vals.astype(dt)), _p(x), _p(y))
    return y


def ccs_kernel(op, nrows, ncols, colptr, rowind, vals, x):
    """op 0: y=x; solve_as_strict_lower(y) | 1: strict_upper | 2: y = A x (CCS multiply_nt_low) |
    3: solve_as_strict_lower_tran | 4: solve_as_strict_upper_tran | 5: y = A^H x (multiply_t_low)."""
    vals = np.ascontiguousarray(vals)
    k = "z" if (np.iscomplexobj(vals) or np.iscomplexobj(x)) else "d"
    dt = np.complex128 if k == "z" else np.float64
    colptr = np.ascontiguousarray(colptr, dtype=np.int64)
    rowind = np.ascontiguousarray(rowind, dtype=np.int32)
    vals = vals.astype(dt)
    x = np.ascontiguousarray(x, dtype=dt)
    if op == 2:
        y = np.zeros(nrows, dtype=dt)
    elif op == 5:
        y = np.zeros(ncols, dtype=dt)
    else:
        y = x.copy()
    getattr(lib(), f"hifref_{k}_ccs_kernel")(op, nrows, ncols, _p(colptr), _p(rowind), _p(vals), _p(x), _p(y))
    return y


def qrcp(mat_colmajor, b, op=0, rank=0, rrqr_cond=0.0):
    """hif::QRCP on a dense n x n block (column-major flat array): op 0 solve, 1 multiply, 2 solve with A^H, 3 multiply with A^H.
    Returns (x, numerical_rank)."""
    mat = np.ascontiguousarray(mat_colmajor)
    k = "z" if (np.iscomplexobj(mat) or np.iscomplexobj(b)) else "d"
    dt = np.complex128 if k == "z" else np.float64
    mat = mat.astype(dt).ravel()
    b = np.ascontiguousarray(b, dtype=dt)
    n = len(b)
    x = np.zeros(n, dtype=dt)
    rk = np.zeros(1, dtype=np.int64)
    if getattr(lib(), f"hifref_{k}_qrcp")(n, _p(mat), rrqr_cond, op, _p(b), rank, _p(x), _p(rk)):
        raise RuntimeError(lib().hifref_error().decode())
    return x, int(rk[0])
