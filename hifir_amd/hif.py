"""class HIF -- Python mirror of hif::HIF<ValueType,int> (reference src/hif/builder.hpp:109) for the
apply path, over the C ABI.  The names and argument meaning follow the reference:

  reference                                   here
  HIF::solve(b, x, trans=false, r=0)   :410   HIF.solve(b, rank=0) -> x
  HIF::solve_mrhs<Nrhs>(b, x, r)       :434   HIF.solve_mrhs(B, rank=0) -> X      (B is [n][nrhs])
  HIF::hifir(A, b, N, x, trans, r)     :459   HIF.hifir(b, N, rank=-1) -> x
  HIF::hifir(A, b, N, betas, x, ...)   :482   HIF.hifir(b, N, betas=(lo, hi)) -> (x, iters, flag)
  HIF::levels()/nnz()/rank()/schur_*   :141-190  same names

numpy arrays use the host-pointer entry points; torch CUDA tensors (device memory is the only thing
torch is used for) use the device-pointer entry points and are not synchronized.
Errors follow the reference convention (status code + message, libhifir.cpp:34-53): a non-zero
HifAmdStatus raises HifAmdError carrying the code and the library's message.
"""
import ctypes as C
import os

import numpy as np

from ._lib import lib

STATUS = {0: "HIFAMD_SUCCESS", 1: "HIFAMD_NULL_OBJ", 2: "HIFAMD_MISMATCHED_SIZES", 3: "HIFAMD_BAD_PREC",
          4: "HIFAMD_HIFIR_ERROR"}


OP_S, OP_SH, OP_M, OP_MH = 0, 1, 2, 3  # HifAmdOp == LhfOperationType (libhifir.h:159-164)


class HifAmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code
        self.msg = msg


def _check(code):
    if code != 0:
        m = lib().hifamd_last_error()
        raise HifAmdError(code, m.decode() if m else "")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _np_dtype_of(t):
    import torch

    return {torch.float64: np.dtype(np.float64), torch.complex128: np.dtype(np.complex128)}.get(t.dtype)


class HIF:
    """A multilevel preconditioner resident on one MI355X."""

    def __init__(self, dtype=np.float64, device=-1):
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.complex128)):
            raise HifAmdError(3, "only float64 and complex128 hierarchies are supported")
        self._h = C.c_void_p()
        _check(lib().hifamd_create(0 if self.dtype == np.float64 else 1, device, C.byref(self._h)))
        self._A = None

    # ---- construction --------------------------------------------------------------------------
    @classmethod
    def from_levels(cls, levels, max_nrhs=64, rrqr_cond=0.0, device=-1, dtype=None):
        """levels: list of dicts with the fields of hif::Prec (alg/Prec.hpp:309-323), CCS matrices:
        m, n, {L,U,E,F}_{colptr,rowind,vals}, d, s, t, p, q_inv (+ p_inv, q), and on the last one
        optionally dense_n, dense (unfactored column-major Schur complement); dense_symm (+ spd) marks the block of
        a hierarchy factorized with is_symm, whose last level is the reference's SYEIG solver."""
        if dtype is None:
            cplx = any(np.iscomplexobj(lv["L_vals"]) or np.iscomplexobj(lv["d"]) or np.iscomplexobj(lv["E_vals"])
                       for lv in levels)
            dtype = np.complex128 if cplx else np.float64
        self = cls(dtype, device)
        for lv in levels:
            self.add_level(lv)
        last = levels[-1]
        if int(last.get("dense_n", 0)) > 0:
            if int(last.get("dense_lup", 0)):
                self.set_dense_lup(last["dense"])
            elif int(last.get("dense_symm", 0)):
                self.set_dense_symm(last["dense"], int(last.get("spd", 0)))
            else:
                self.set_dense(last["dense"], rrqr_cond)
        self.finalize(max_nrhs)
        return self

    def save(self, path, analysis=False):
        """Write the imported hierarchy (the add_level / set_dense arguments) to a file (hifamd_save).  analysis=True
        appends the host analysis of every level (hifamd_save_ex, HIFAMD_SAVE_ANALYSIS): a load under the same planner
        options adopts it instead of analyzing again (stats_ext()["analysis_cached_levels"])."""
        _check(lib().hifamd_save_ex(self._h, os.fsencode(path), 1 if analysis else 0))

    @classmethod
    def load(cls, path, max_nrhs=64, device=-1):
        """Read a hierarchy written by save() and ship it to the device (hifamd_load + finalize)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self._A = None
        _check(lib().hifamd_load(os.fsencode(path), device, C.byref(self._h)))
        with open(path, "rb") as f:
            f.seek(8)
            self.dtype = np.dtype(np.complex128 if int.from_bytes(f.read(8), "little") == 1 else np.float64)
        if max_nrhs:
            self.finalize(max_nrhs)
        return self

    def add_level(self, lv):
        dt = self.dtype
        m, n = int(lv["m"]), int(lv["n"])
        mats = []
        for k in "LUEF":
            mats += [np.ascontiguousarray(lv[k + "_colptr"], dtype=np.int64),
                     np.ascontiguousarray(lv[k + "_rowind"], dtype=np.int32),
                     np.ascontiguousarray(lv[k + "_vals"], dtype=dt)]
        f_ncols = len(mats[9]) - 1 if (n - m) else 0
        if f_ncols and mats[9][-1] == 0 and len(mats[9]) - 1 != n - m:
            f_ncols = 0
        d = np.ascontiguousarray(lv["d"], dtype=dt)
        s = np.ascontiguousarray(lv["s"], dtype=np.float64)
        t = np.ascontiguousarray(lv["t"], dtype=np.float64)
        perms = [None if lv.get(k) is None else np.ascontiguousarray(lv[k], dtype=np.int32)
                 for k in ("p", "p_inv", "q", "q_inv")]
        _check(lib().hifamd_add_level(self._h, m, n, *[_p(a) for a in mats[:9]], f_ncols, *[_p(a) for a in mats[9:]],
                                      _p(d), _p(s), _p(t), *[_p(a) for a in perms]))

    def set_dense(self, mat_colmajor, rrqr_cond=0.0):
        mat = np.ascontiguousarray(mat_colmajor, dtype=self.dtype).ravel()
        nd = int(round(np.sqrt(mat.size)))
        _check(lib().hifamd_set_dense(self._h, nd, _p(mat), float(rrqr_cond)))

    def set_dense_symm(self, mat_colmajor, spd=0):
        """Last level of a symmetric factorization (Prec::symm_dense_solver, SYEIG): eigendecomposition on the host."""
        mat = np.ascontiguousarray(mat_colmajor, dtype=self.dtype).ravel()
        nd = int(round(np.sqrt(mat.size)))
        _check(lib().hifamd_set_dense_symm(self._h, nd, _p(mat), int(spd)))

    def set_dense_lup(self, mat_colmajor):
        """Last level of a reference built with HIF_DENSE_MODE=0 (LU with partial pivoting, small_scale/LUP.hpp)."""
        mat = np.ascontiguousarray(mat_colmajor, dtype=self.dtype).ravel()
        nd = int(round(np.sqrt(mat.size)))
        _check(lib().hifamd_set_dense_lup(self._h, nd, _p(mat)))

    def finalize(self, max_nrhs=64):
        _check(lib().hifamd_finalize(self._h, int(max_nrhs)))
        # (the host copy is sealed at import and verified here, DESIGN 8: a copy that changed behind the library's back and
        #  was rebuilt is worth a Python warning as well -- pytest lists those even for tests that pass)
        rep = self.stats_ext().get("host_copy_repairs", 0.0)
        if rep:
            import warnings

            warnings.warn("hifir_amd: %d array(s) of the host copy had changed between add_level and finalize and were "
                          "rebuilt from the imported arrays (details on stderr)" % int(rep), RuntimeWarning)

    def set_matrix(self, indptr, indices, vals):
        """Attach the user's CRS matrix for iterative refinement (what lhf?Setup keeps, libhifir.cpp:413)."""
        ip = np.ascontiguousarray(indptr, dtype=np.int64)
        ix = np.ascontiguousarray(indices, dtype=np.int32)
        v = np.ascontiguousarray(vals, dtype=self.dtype)
        _check(lib().hifamd_set_matrix(self._h, len(ip) - 1, _p(ip), _p(ix), _p(v)))
        self._A = True

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().hifamd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- queries (builder.hpp:141-190) ------------------------------------------------------------
    def nrows(self):
        return lib().hifamd_nrows(self._h)

    ncols = nrows

    def levels(self):
        return lib().hifamd_levels(self._h)

    def nnz(self):
        return lib().hifamd_nnz(self._h)

    def schur_size(self):
        return lib().hifamd_schur_size(self._h)

    def schur_rank(self):
        return lib().hifamd_schur_rank(self._h)

    def rank(self):
        return self.nrows() - (self.schur_size() - self.schur_rank()) if self.schur_size() else self.nrows()

    def stats(self):
        s = np.zeros(16)
        _check(lib().hifamd_stats(self._h, _p(s)))
        keys = ["sum_n", "sum_m", "nnz_LU", "nnz_EF", "dense_n", "B_mat", "B_vec", "wavefronts_L", "wavefronts_U",
                "launches", "sparse_levels", "bands", "band_workgroups", "finalize_s", "operator_bytes", "graph_capture_ms"]
        return {k: float(s[i]) for i, k in enumerate(keys)}

    def stats_ext(self):
        """Set-up cost and resident explicit operators (hifamd_stats_ext)."""
        s = np.zeros(32)
        k = lib().hifamd_stats_ext(self._h, _p(s), 32)
        keys = ["finalize_s", "graph_capture_ms", "bytes_inverses", "bytes_top", "bytes_tail", "tail_rows", "tail_level",
                "tail_probe_relerr", "tail_max_abs", "tail_rejected", "tail_probe_tol", "tail_max_growth",
                "analysis_cached_levels", "analysis_s", "arena_bytes", "arena_cols", "tile_bytes", "factor_bytes", "max_nrhs",
                "rows_not_stored_L", "rows_not_stored_U", "host_copy_repairs"]
        return {key: float(s[i]) for i, key in enumerate(keys[:max(0, k)])}

    def level_stats(self, level):
        s = np.zeros(16)
        k = lib().hifamd_level_stats(self._h, int(level), _p(s), 16)
        if k < 0:
            raise HifAmdError(2, "no such level")
        keys = ["m", "n", "nnz_L", "nnz_U", "nnz_E", "nnz_F", "wavefronts_L", "wavefronts_U", "bands_L", "bands_U", "top_rows"]
        return {key: float(s[i]) for i, key in enumerate(keys[:k])}

    def launch_map(self):
        """(level, stage) of every kernel launch of the last batched apply, in launch order (hifamd_launch_map)."""
        k = lib().hifamd_launch_map(self._h, None, 0)
        o = np.zeros(max(k, 1), dtype=np.int32)
        lib().hifamd_launch_map(self._h, _p(o), int(k))
        return [(int(v) // 16, int(v) % 16) for v in o[:k]]

    def level_bytes(self, nrhs):
        """B_alg(nrhs) of SURVEY 8(d) level by level and stage by stage: {level: {stage: bytes}} with the stage codes of
        launch_map(); the dense block's bytes sit at stage 4 of the level that owns it."""
        sv = np.dtype(self.dtype).itemsize
        si, sp, ss = 4, 8, 8
        st = self.stats()
        out = {}
        nlev = int(st["sparse_levels"])
        for l in range(nlev):
            q = self.level_stats(l)
            m, n = q["m"], q["n"]
            nm = n - m
            ldu = 2 * m * sv * nrhs + (q["nnz_L"] + q["nnz_U"]) * (sv + si) + m * sv + 2 * (m + 1) * sp  # one LDU solve
            out[l] = {
                1: 2 * n * sv * nrhs + n * (si + ss),  # S1: read + write n rows, p and s
                2: ldu,
                3: (m + 2 * nm) * sv * nrhs + q["nnz_E"] * (sv + si) + (nm + 1) * sp,
                4: (st["dense_n"] ** 2 * sv) if l == nlev - 1 else 0.0,
                5: (nm + 2 * m) * sv * nrhs + q["nnz_F"] * (sv + si) + (m + 1) * sp,
                6: ldu,
                7: 2 * n * sv * nrhs + n * (si + ss),  # S7: read + write n rows, q_inv and t
            }
        tot = sum(sum(v.values()) for v in out.values())
        ref = self.algorithmic_bytes(nrhs)
        assert abs(tot - ref) <= 1e-6 * ref, (tot, ref)
        return out

    def algorithmic_bytes(self, nrhs):
        """B_alg(nrhs) = B_mat + nrhs * B_vec of SURVEY 8(d), for the hierarchy actually resident."""
        st = self.stats()
        return st["B_mat"] + nrhs * st["B_vec"]

    def stage_bytes(self, nrhs):
        """The same B_alg split by stage group (SURVEY 8(d) terms): which kernels stream which bytes.
        permute = S1 + S7 (k_gather_scale / k_scatter_scale), ldu = S2 + S6 (k_trsv_* / k_thin_update /
        k_tri_gemm_d), schur = S3 + S5 (k_spmm_epi), dense = last-level operators."""
        st = self.stats()
        sv = np.dtype(self.dtype).itemsize
        si, sp, ss = 4, 8, 8
        n, m, L = st["sum_n"], st["sum_m"], st["sparse_levels"]
        nm = n - m
        out = {
            "permute": 4 * n * sv * nrhs + n * (2 * si + 2 * ss),
            "ldu": 4 * m * sv * nrhs + 2 * st["nnz_LU"] * (sv + si) + 2 * m * sv + 4 * (m + L) * sp,
            "schur": (3 * m + 3 * nm) * sv * nrhs + st["nnz_EF"] * (sv + si) + (n + 2 * L) * sp,
            "dense": st["dense_n"] ** 2 * sv,
        }
        assert abs(sum(out.values()) - self.algorithmic_bytes(nrhs)) <= 1e-9 * self.algorithmic_bytes(nrhs)
        return out

    def level_schedule(self, level, which):
        """(order, wf_ptr) of the L (which=0) or U (which=1) factor of a level."""
        nwf = C.c_int64()
        _check(lib().hifamd_level_schedule(self._h, level, which, C.byref(nwf), None, None))
        stt = np.zeros(16)
        lib().hifamd_stats(self._h, _p(stt))
        # m is not exported separately: take it from wf_ptr's last entry
        wf = np.zeros(nwf.value + 1, dtype=np.int64)
        _check(lib().hifamd_level_schedule(self._h, level, which, None, None, _p(wf)))
        order = np.zeros(int(wf[-1]), dtype=np.int32)
        _check(lib().hifamd_level_schedule(self._h, level, which, None, _p(order), None))
        return order, wf

    # ---- argument checks of the device-pointer entry points ------------------------------------------
    def _dev_block(self, T, what, shape=None):
        """A CUDA tensor handed to a *_dev entry point must be what the kernels assume: on the handle's device, of
        the handle's value type, [nrows][nrhs] with unit column stride.  Anything else would be an out-of-bounds
        device access (a GPU memory fault aborts the process), so it is refused here with MISMATCHED_SIZES."""
        if not _is_torch(T) or not T.is_cuda:
            raise HifAmdError(2, f"{what}: expected a CUDA tensor")
        dev = lib().hifamd_device(self._h)
        if dev >= 0 and T.device.index != dev:
            raise HifAmdError(2, f"{what}: tensor lives on cuda:{T.device.index}, the hierarchy on cuda:{dev}")
        if _np_dtype_of(T) != self.dtype:
            raise HifAmdError(2, f"{what}: dtype {T.dtype} does not match the hierarchy's {self.dtype}")
        if T.dim() != 2 or T.shape[0] != self.nrows() or T.shape[1] < 1:
            raise HifAmdError(2, f"{what}: expected an [nrows][nrhs] block, got {tuple(T.shape)}")
        if T.stride(1) != 1 or T.stride(0) < T.shape[1]:
            raise HifAmdError(2, f"{what}: blocks must be row-interleaved (unit column stride)")
        if shape is not None and tuple(T.shape) != tuple(shape):
            raise HifAmdError(2, f"{what}: shape {tuple(T.shape)} differs from the input's {tuple(shape)}")
        return T

    def _host_out(self, X, like, what="x"):
        """A caller-supplied host output is written through its pointer: it must be exactly that buffer."""
        if X is None:
            return np.empty_like(like)
        if not isinstance(X, np.ndarray) or X.dtype != self.dtype or X.shape != like.shape or not X.flags.c_contiguous:
            raise HifAmdError(2, f"{what}: output must be a C-contiguous {self.dtype} array of shape {like.shape}")
        return X

    # ---- apply -----------------------------------------------------------------------------------
    def solve(self, b, x=None, trans=False, rank=0):
        """x = M^{-1} b, or x = M^{-H} b with trans=True (HIF::solve, builder.hpp:409-423)."""
        if _is_torch(b):
            return self.solve_mrhs(b.reshape(-1, 1), None if x is None else x.reshape(-1, 1), rank, trans=trans).reshape(-1)
        b = np.ascontiguousarray(b, dtype=self.dtype)
        if b.ndim != 1 or b.shape[0] != self.nrows():
            raise HifAmdError(2, "unmatched sizes")
        x = self._host_out(x, b)
        if trans:
            _check(lib().hifamd_apply_batch(self._h, OP_SH, _p(b), 1, _p(x), 1, 1, 1, None, int(rank), None))
        else:
            _check(lib().hifamd_solve(self._h, _p(b), _p(x), int(rank)))
        return x

    def solve_mrhs(self, B, X=None, rank=0, stream=None, trans=False):
        """X = M^{-1} B (trans: M^{-H} B) for an [n][nrhs] row-interleaved block (the layout of
        Array<std::array<T,Nrhs>>, CompressedStorage.hpp:2127); nrhs is a run-time value here."""
        op = OP_SH if trans else OP_S
        if _is_torch(B):
            import torch

            self._dev_block(B, "B")
            if X is None:
                X = torch.empty_like(B)
            self._dev_block(X, "X", B.shape)
            _check(lib().hifamd_apply_batch_dev(self._h, op, B.data_ptr(), B.stride(0), X.data_ptr(), X.stride(0),
                                               B.shape[1], int(rank), stream))
            return X
        B = np.ascontiguousarray(B, dtype=self.dtype)
        if B.ndim != 2 or B.shape[0] != self.nrows():
            raise HifAmdError(2, "unmatched sizes")
        X = self._host_out(X, B, "X")
        if trans:
            _check(lib().hifamd_apply_batch(self._h, op, _p(B), B.shape[1], _p(X), X.shape[1], B.shape[1], 1, None,
                                           int(rank), None))
        else:
            _check(lib().hifamd_solve_batch(self._h, _p(B), B.shape[1], _p(X), X.shape[1], B.shape[1], int(rank)))
        return X

    def set_nsp_const(self, start=0, end=-1, trans=False):
        """Constant-mode null-space filter of solve() (HIF::nsp; trans: HIF::nsp_tran): rows [start, end) of
        every solution lose their mean; start > end >= 0 removes the filter."""
        _check(lib().hifamd_set_nsp_const(self._h, OP_SH if trans else OP_S, int(start), int(end)))

    def mmultiply(self, x, trans=False, rank=-1):
        """y = M x (trans: M^H x), the multilevel product HIF::mmultiply (builder.hpp:503-513) -- the inverse
        direction of solve().  x: [n] or [n][nrhs], host array or CUDA tensor."""
        op = OP_MH if trans else OP_M
        vec = (x.ndim == 1)
        if _is_torch(x):
            import torch

            X = self._dev_block(x.reshape(x.shape[0], -1), "x")
            Y = torch.empty_like(X)
            _check(lib().hifamd_apply_batch_dev(self._h, op, X.data_ptr(), X.stride(0), Y.data_ptr(), Y.stride(0),
                                               X.shape[1], int(rank), None))
        else:
            X = np.ascontiguousarray(x, dtype=self.dtype).reshape(x.shape[0], -1)
            Y = np.empty_like(X)
            _check(lib().hifamd_apply_batch(self._h, op, _p(X), X.shape[1], _p(Y), Y.shape[1], X.shape[1], 1, None,
                                           int(rank), None))
        return Y.reshape(-1) if vec else Y

    def spmv(self, X, Y=None, stream=None):
        """Y = A X on the device (torch CUDA tensors, [n][nrhs] or [n])."""
        import torch

        X2 = self._dev_block(X.reshape(X.shape[0], -1), "X")
        if Y is None:
            Y = torch.empty_like(X)
        Y2 = self._dev_block(Y.reshape(Y.shape[0], -1), "Y", X2.shape)
        _check(lib().hifamd_spmv_batch_dev(self._h, X2.data_ptr(), X2.stride(0), Y2.data_ptr(), Y2.stride(0),
                                          X2.shape[1], stream))
        return Y

    def hifir(self, b, N, betas=None, rank=-1, trans=False):
        """Iterative refinement (HIF::hifir, builder.hpp:459-489).  b: [n] or [n][nrhs].
        Without betas returns x; with betas returns (x, iters, flags) per column (ints for a vector).
        trans=True refines A^H x = b with M^{-H} (host arrays only)."""
        if trans:
            if _is_torch(b):
                raise HifAmdError(3, "hifir(trans=True) takes host arrays")
            vec = (b.ndim == 1)
            bt = None if betas is None else np.ascontiguousarray(betas, dtype=np.float64)
            B = np.ascontiguousarray(b, dtype=self.dtype).reshape(b.shape[0], -1)
            X = np.empty_like(B)
            st = np.zeros(2 * B.shape[1], dtype=np.int32)
            _check(lib().hifamd_apply_batch(self._h, OP_SH, _p(B), B.shape[1], _p(X), X.shape[1], B.shape[1], int(N),
                                           _p(bt), int(rank), _p(st)))
            x = X.reshape(-1) if vec else X
            if betas is None:
                return x
            it, fl = st[0::2].copy(), st[1::2].copy()
            return (x, int(it[0]), int(fl[0])) if vec else (x, it, fl)
        vec = (b.ndim == 1)
        bt = None if betas is None else np.ascontiguousarray(betas, dtype=np.float64)
        if _is_torch(b):
            import torch

            B = self._dev_block(b.reshape(b.shape[0], -1), "b")
            X = torch.empty_like(B)
            st = np.zeros(2 * B.shape[1], dtype=np.int32)
            _check(lib().hifamd_hifir_batch_dev(self._h, B.data_ptr(), B.stride(0), X.data_ptr(), X.stride(0),
                                               B.shape[1], int(N), _p(bt), int(rank), _p(st)))
        else:
            B = np.ascontiguousarray(b, dtype=self.dtype).reshape(b.shape[0], -1)
            X = np.empty_like(B)
            st = np.zeros(2 * B.shape[1], dtype=np.int32)
            _check(lib().hifamd_hifir_batch(self._h, _p(B), B.shape[1], _p(X), X.shape[1], B.shape[1], int(N), _p(bt),
                                           int(rank), _p(st)))
        x = X.reshape(-1) if vec else X
        if betas is None:
            return x
        it, fl = st[0::2].copy(), st[1::2].copy()
        return (x, int(it[0]), int(fl[0])) if vec else (x, it, fl)

    def gmres(self, b, restart=30, rtol=1e-6, maxit=500, full_rank=False):
        """Right-preconditioned restarted GMRES (the reference's examples/advanced/gmres.hpp:19-123),
        all columns of b ([n] or [n][nrhs], host array or CUDA tensor) in lock step on the device.
        Returns (x, flags, iters); ints for a vector."""
        vec = (b.ndim == 1)
        rank = -1 if full_rank else 0
        if _is_torch(b):
            import torch

            B = self._dev_block(b.reshape(b.shape[0], -1), "b")
            X = torch.empty_like(B)
            fl = np.zeros(B.shape[1], dtype=np.int32)
            it = np.zeros(B.shape[1], dtype=np.int32)
            _check(lib().hifamd_gmres_batch_dev(self._h, B.data_ptr(), B.stride(0), X.data_ptr(), X.stride(0), B.shape[1],
                                               int(restart), float(rtol), int(maxit), rank, _p(fl), _p(it)))
        else:
            B = np.ascontiguousarray(b, dtype=self.dtype).reshape(b.shape[0], -1)
            X = np.empty_like(B)
            fl = np.zeros(B.shape[1], dtype=np.int32)
            it = np.zeros(B.shape[1], dtype=np.int32)
            _check(lib().hifamd_gmres_batch(self._h, _p(B), B.shape[1], _p(X), X.shape[1], B.shape[1], int(restart),
                                           float(rtol), int(maxit), rank, _p(fl), _p(it)))
        if vec:
            return X.reshape(-1), int(fl[0]), int(it[0])
        return X, fl, it

    def fgmres(self, b, restart=30, rtol=1e-6, maxit=500, full_rank=False):
        """Flexible GMRES with 2^k refinement sweeps as the preconditioner of outer cycle k (the reference's
        fgmres_hifir, examples/advanced/gmres.hpp:127-231); host arrays.  Returns (x, flags, iters, sweeps)."""
        vec = (b.ndim == 1)
        B = np.ascontiguousarray(b, dtype=self.dtype).reshape(b.shape[0], -1)
        X = np.empty_like(B)
        fl, it, mv = (np.zeros(B.shape[1], dtype=np.int32) for _ in range(3))
        _check(lib().hifamd_fgmres_batch(self._h, _p(B), B.shape[1], _p(X), X.shape[1], B.shape[1], int(restart),
                                        float(rtol), int(maxit), -1 if full_rank else 0, _p(fl), _p(it), _p(mv)))
        if vec:
            return X.reshape(-1), int(fl[0]), int(it[0]), int(mv[0])
        return X, fl, it, mv

    def time_apply(self, B, X, rank=0, warmup=2, reps=10):
        """Average device milliseconds of one batched apply, HIP events on the handle's stream."""
        self._dev_block(B, "B")
        self._dev_block(X, "X", B.shape)
        ms = C.c_double()
        _check(lib().hifamd_time_apply(self._h, B.data_ptr(), B.stride(0), X.data_ptr(), X.stride(0), B.shape[1],
                                      int(rank), int(warmup), int(reps), C.byref(ms)))
        return ms.value

    def sync(self):
        _check(lib().hifamd_sync(self._h))
