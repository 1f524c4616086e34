"""ctypes binding of shim/libhifir.so exactly as a C user of the reference's libhifir would bind it
(libhifir/include/libhifir.h), plus the additive entry points of include/libhifir_amd_ext.h.  Test helper."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_PATH = os.path.join(ROOT, "shim", "libhifir.so")

LHF_S, LHF_SH, LHF_M, LHF_MH = 0, 1, 2, 3
LHF_SUCCESS, LHF_NULL_OBJ, LHF_MISMATCHED_SIZES, LHF_BAD_PREC, LHF_HIFIR_ERROR = range(5)
LHF_DEFAULT_RANK = -2
LHF_VERBOSE, LHF_NUMBER_PARAMS = 6, 17  # libhifir.h:94-117
ROW_MAJOR = 1

# the REAL libhifir: the reference's own libhifir/src/libhifir.cpp compiled where it lies (oracle/Makefile -> oracle/_ref/,
# git-ignored, travels to the GPU box as a binary).  Test oracle for the entry points the golden fixtures do not cover.
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libhifir_ref.so")

_vp, _sz, _int, _dp = C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)
_lib = None
_ref = None


def available():
    return os.path.exists(SHIM_PATH)


def ref_available():
    return os.path.exists(REF_PATH)


def ref_lib():
    """ctypes binding of the reference's own libhifir (91 symbols, no additive ones): a CHECKER."""
    global _ref
    if _ref is None:
        _ref = _bind(C.CDLL(REF_PATH), additive=False)
    return _ref


def lib():
    global _lib
    if _lib is None:
        import hifir_amd  # noqa: F401  (torch first, then libhifir_amd.so: one HIP runtime per process)

        hifir_amd.lib()
        _lib = _bind(C.CDLL(SHIM_PATH), additive=True)
    return _lib


def _bind(L, additive):
    if True:  # (one block: the binding is a flat list of signatures)
        L.lhfGetErrorMsg.restype = C.c_char_p
        L.lhfGetErrorMsg.argtypes = []
        L.lhfGetVersions.argtypes = [C.POINTER(_int)]
        for f in ("lhfSetDefaultParams",):
            getattr(L, f).argtypes = [_dp]
        for f in ("lhfSetDroptol", "lhfSetAlpha", "lhfSetKappa"):
            getattr(L, f).argtypes = [C.c_double, _dp]
        if additive:
            L.lhfSetDevices.argtypes = [C.POINTER(_int), _int]
            L.lhfGetDeviceCount.restype = _int
        for t in "dszc":
            g = lambda name: getattr(L, f"lhf{t}{name}")
            g("CreateMatrix").restype = _vp
            g("CreateMatrix").argtypes = [_int, _sz, _vp, _vp, _vp]
            g("DestroyMatrix").argtypes = [_vp]
            g("WrapMatrix").argtypes = [_vp, _sz, _vp, _vp, _vp]
            for q in ("GetMatrixSize", "GetMatrixNnz"):
                g(q).restype = _sz
                g(q).argtypes = [_vp]
            g("Create").restype = _vp
            g("Create").argtypes = [_vp, _vp, _dp]
            g("Destroy").argtypes = [_vp]
            g("Setup").argtypes = [_vp, _vp, _vp, _dp]
            g("Update").argtypes = [_vp, _vp]
            g("Refactorize").argtypes = [_vp, _vp, _dp]
            g("Apply").argtypes = [_vp, _int, _vp, _int, _dp, _int, _vp, C.POINTER(_int)]
            g("Solve").argtypes = [_vp, _vp, _vp]
            g("GetStats").argtypes = [_vp, C.POINTER(_sz)]
            for q in ("GetNnz", "GetLevels", "GetSchurSize", "GetSchurRank"):
                g(q).restype = _sz
                g(q).argtypes = [_vp]
        for t in ("dz" if additive else ""):
            g = lambda name: getattr(L, f"lhf{t}{name}")
            g("ApplyBatch").argtypes = [_vp, _int, _vp, _sz, _sz, _int, _dp, _int, _vp, _sz, C.POINTER(_int)]
            g("ApplyBatchDev").argtypes = [_vp, _int, _int, _vp, C.POINTER(_sz), C.POINTER(_sz), _vp, C.POINTER(_sz)]
            g("GatherBatchDev").argtypes = [_vp, _int, _vp, C.POINTER(_sz), C.POINTER(_sz), _vp, _sz]
            g("SyncDevices").argtypes = [_vp]
            g("SaveHierarchy").argtypes = [_vp, C.c_char_p]
            g("LoadHierarchy").restype = _vp
            g("LoadHierarchy").argtypes = [C.c_char_p]
            g("GetResidentBytes").argtypes = [_vp, C.POINTER(_sz)]
        for f in ("lhfsdUpdate", "lhfczUpdate"):
            getattr(L, f).argtypes = [_vp, _vp]
        for f in ("lhfsdApply", "lhfczApply"):
            getattr(L, f).argtypes = [_vp, _int, _vp, _int, _dp, _int, _vp, C.POINTER(_int)]
        for f in ("lhfsdSolve", "lhfczSolve"):
            getattr(L, f).argtypes = [_vp, _vp, _vp]
    return L


def errmsg():
    m = lib().lhfGetErrorMsg()
    return m.decode() if m else None


def default_params(verbose=0):
    p = (C.c_double * LHF_NUMBER_PARAMS)()
    assert lib().lhfSetDefaultParams(p) == LHF_SUCCESS
    p[LHF_VERBOSE] = verbose
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Matrix:
    """lhf?CreateMatrix: the handle ALIASES the arrays (libhifir.cpp:316-321), so they are kept alive here."""

    def __init__(self, t, indptr, indices, vals, rowmajor=True, L=None):
        self.t = t
        self.L = L or lib()
        dt = {"d": np.float64, "z": np.complex128, "s": np.float32, "c": np.complex64}[t]
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.vals = np.ascontiguousarray(vals, dtype=dt)
        self.n = len(self.indptr) - 1
        self.h = getattr(self.L, f"lhf{t}CreateMatrix")(int(rowmajor), self.n, _ptr(self.indptr), _ptr(self.indices),
                                                        _ptr(self.vals))
        assert self.h

    def close(self):
        if self.h:
            getattr(self.L, f"lhf{self.t}DestroyMatrix")(self.h)
            self.h = None


class Hif:
    def __init__(self, t, A=None, S=None, params=None, handle=None, L=None):
        self.t = t
        self.L = L or lib()
        self.dt = {"d": np.float64, "z": np.complex128, "s": np.float32, "c": np.complex64}[t]
        self.A = A
        self.h = handle if handle is not None else getattr(self.L, f"lhf{t}Create")(
            A.h if A else None, S.h if S else None, params)

    def _f(self, name):
        return getattr(self.L, f"lhf{self.t}{name}")

    # the mixed entry points of a single-precision handle (libhifir.cpp:1185-1284): double vectors, double matrix
    def mixed(self, name):
        return getattr(self.L, "lhf%s%s" % ({"s": "sd", "c": "cz"}[self.t], name))

    def mixed_solve(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64 if self.t == "s" else np.complex128)
        x = np.empty_like(b)
        return self.mixed("Solve")(self.h, _ptr(b), _ptr(x)), x

    def mixed_apply(self, op, b, nirs=1, betas=None, rank=LHF_DEFAULT_RANK):
        b = np.ascontiguousarray(b, dtype=np.float64 if self.t == "s" else np.complex128)
        x = np.empty_like(b)
        bt = None if betas is None else (C.c_double * 2)(*betas)
        irs = (C.c_int * 2)(-7, -7)
        return self.mixed("Apply")(self.h, op, _ptr(b), nirs, bt, rank, _ptr(x), irs), x

    def solve(self, b):
        b = np.ascontiguousarray(b, dtype=self.dt)
        x = np.empty_like(b)
        st = self._f("Solve")(self.h, _ptr(b), _ptr(x))
        return st, x

    def apply(self, op, b, nirs=1, betas=None, rank=LHF_DEFAULT_RANK, want_status=False):
        b = np.ascontiguousarray(b, dtype=self.dt)
        x = np.empty_like(b)
        bt = None if betas is None else (C.c_double * 2)(*betas)
        irs = (C.c_int * 2)(-7, -7)
        st = self._f("Apply")(self.h, op, _ptr(b), nirs, bt, rank, _ptr(x), irs)
        return (st, x, (irs[0], irs[1])) if want_status else (st, x)

    def apply_batch(self, op, B, nirs=1, betas=None, rank=LHF_DEFAULT_RANK):
        B = np.ascontiguousarray(B, dtype=self.dt)
        X = np.empty_like(B)
        bt = None if betas is None else (C.c_double * 2)(*betas)
        irs = (C.c_int * (2 * B.shape[1]))()
        st = self._f("ApplyBatch")(self.h, op, _ptr(B), B.shape[1], B.shape[1], nirs, bt, rank, _ptr(X), X.shape[1], irs)
        return st, X, np.array(list(irs)).reshape(-1, 2)

    def stats(self):
        s = (C.c_size_t * 9)()
        self._f("GetStats")(self.h, s)
        return list(s)

    def close(self):
        if self.h:
            self._f("Destroy")(self.h)
            self.h = None
