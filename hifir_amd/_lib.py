"""Loader of the native library (hifir_amd/libhifir_amd.so, built in-tree by csrc/Makefile).

There is deliberately no fallback: if the HIP library is missing the import fails loudly, and
if no GPU is visible every compute entry point returns HIFAMD_HIFIR_ERROR.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIFIR_AMD_LIB") or os.path.join(_HERE, "libhifir_amd.so")  # override: development only

# name -> (restype, argtypes); mirrors include/hifir_amd.h one to one
_vp, _i64, _int, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
SIGNATURES = {
    "hifamd_version": (C.c_char_p, []),
    "hifamd_last_error": (C.c_char_p, []),
    "hifamd_device_count": (_int, []),
    "hifamd_create": (_int, [_int, _int, C.POINTER(_vp)]),
    "hifamd_destroy": (_int, [_vp]),
    "hifamd_add_level": (_int, [_vp, _i64, _i64] + [_vp] * 9 + [_i64] + [_vp] * 10),
    "hifamd_set_nsp_const": (_int, [_vp, _int, _i64, _i64]),
    "hifamd_save": (_int, [_vp, C.c_char_p]),
    "hifamd_save_ex": (_int, [_vp, C.c_char_p, _int]),
    "hifamd_load": (_int, [C.c_char_p, _int, C.POINTER(_vp)]),
    "hifamd_set_dense": (_int, [_vp, _i64, _vp, _dbl]),
    "hifamd_set_dense_symm": (_int, [_vp, _i64, _vp, _int]),
    "hifamd_set_dense_lup": (_int, [_vp, _i64, _vp]),
    "hifamd_finalize": (_int, [_vp, _i64]),
    "hifamd_value_type": (_int, [_vp]),
    "hifamd_device": (_int, [_vp]),
    "hifamd_nrows": (_i64, [_vp]),
    "hifamd_levels": (_i64, [_vp]),
    "hifamd_nnz": (_i64, [_vp]),
    "hifamd_schur_size": (_i64, [_vp]),
    "hifamd_schur_rank": (_i64, [_vp]),
    "hifamd_stats": (_int, [_vp, _vp]),
    "hifamd_stats_ext": (_int, [_vp, _vp, _int]),
    "hifamd_level_stats": (_int, [_vp, _int, _vp, _int]),
    "hifamd_launch_map": (_int, [_vp, _vp, _int]),
    "hifamd_level_schedule": (_int, [_vp, _int, _int, _vp, _vp, _vp]),
    "hifamd_solve": (_int, [_vp, _vp, _vp, _i64]),
    "hifamd_solve_batch": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64]),
    "hifamd_solve_batch_dev": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    "hifamd_set_matrix": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "hifamd_spmv_batch_dev": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "hifamd_hifir_batch": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _int, _vp, _i64, _vp]),
    "hifamd_hifir_batch_dev": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _int, _vp, _i64, _vp]),
    "hifamd_apply_batch": (_int, [_vp, _int, _vp, _i64, _vp, _i64, _i64, _int, _vp, _i64, _vp]),
    "hifamd_apply_batch_dev": (_int, [_vp, _int, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    "hifamd_gmres_batch": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _int, _dbl, _int, _i64, _vp, _vp]),
    "hifamd_gmres_batch_dev": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _int, _dbl, _int, _i64, _vp, _vp]),
    "hifamd_fgmres_batch": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _int, _dbl, _int, _i64, _vp, _vp, _vp]),
    "hifamd_time_apply": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _int, _int, _vp]),
    "hifamd_sync": (_int, [_vp]),
    "hifamd_copy_columns_dev": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _i64]),
    "hifamd_debug_checksums": (_int, [_vp, _vp, _int]),
}

_lib = None


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.check_call(cmd + ["clean"])
    subprocess.check_call(cmd)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C hifir_amd/csrc` "
                "(or __graft_entry__.build()); hifir_amd has no Python/CPU fallback")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as
        # /opt/rocm's).  If torch is going to be used for device memory / torch.distributed it must
        # be loaded FIRST so that this library binds to the same runtime; loading ours first and
        # torch later leaves torch with "No HIP GPUs are available".
        if os.environ.get("HIFIR_AMD_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
