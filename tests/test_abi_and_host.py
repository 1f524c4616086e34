"""CPU: the C-ABI library loads and exports every symbol include/hifir_amd.h declares (no compute
calls), and the host-side analysis (import checks, level schedules) is right.  No GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import hifir_amd
from hifir_amd import _lib
from util import HIER_NAMES, load_hier

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "hifir_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(hifamd_\w+)\s*\(", hdr)))
    assert len(declared) >= 20
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    # the ctypes table mirrors the header one to one
    assert sorted(_lib.SIGNATURES) == declared


def test_no_silent_fallback_without_gpu():
    if hifir_amd.lib().hifamd_device_count() > 0:
        pytest.skip("a GPU is present")
    levels, d = load_hier("p2d_5")
    M = hifir_amd.HIF()
    M.add_level(levels[0])
    M.set_dense(levels[0]["dense"])
    with pytest.raises(hifir_amd.HifAmdError) as e:
        M.finalize(1)
    assert e.value.code == 4 and "no CPU fallback" in e.value.msg
    with pytest.raises(hifir_amd.HifAmdError):
        M.solve(d["b"])  # not finalized -> HIFAMD_BAD_PREC, never a CPU result


def test_import_validation():
    levels, _ = load_hier("p2d_30")
    lv = dict(levels[0])
    M = hifir_amd.HIF()
    bad = dict(lv)
    bad["p"] = lv["p"].copy()
    bad["p"][0] = lv["n"] + 5
    with pytest.raises(hifir_amd.HifAmdError) as e:
        M.add_level(bad)
    assert e.value.code == 2
    M.add_level(lv)
    with pytest.raises(hifir_amd.HifAmdError):  # child size must be n-m of the parent
        M.add_level(lv)
    with pytest.raises(hifir_amd.HifAmdError):  # wrong dense size
        M.set_dense(np.eye(3).ravel())
    # error message is returned once, then cleared (libhifir.cpp:224-229)
    assert hifir_amd.lib().hifamd_last_error() is None


@pytest.mark.parametrize("name", HIER_NAMES)
def test_level_schedule_is_valid(name):
    levels, _ = load_hier(name)
    M = hifir_amd.HIF(dtype=np.complex128 if name == "young1c" else np.float64)
    for lv in levels:
        M.add_level(lv)
    if int(levels[-1].get("dense_n", 0)):
        M.set_dense(levels[-1]["dense"])
        assert M.schur_rank() == levels[-1]["dense_rank"]
    for l, lv in enumerate(levels):
        m = lv["m"]
        for which, key in enumerate("LU"):
            order, wf = M.level_schedule(l, which)
            assert sorted(order.tolist()) == list(range(m))
            A = sp.csc_matrix((np.ones(len(lv[key + "_rowind"])), lv[key + "_rowind"], lv[key + "_colptr"]), shape=(m, m)).tocsr()
            depth = np.empty(m, dtype=np.int64)
            for w in range(len(wf) - 1):
                depth[order[wf[w]:wf[w + 1]]] = w
            # every dependency sits in a strictly earlier wavefront, and the depth is minimal
            for i in range(m):
                deps = A.indices[A.indptr[i]:A.indptr[i + 1]]
                if len(deps):
                    assert depth[deps].max() + 1 == depth[i]
                else:
                    assert depth[i] == 0
    assert M.nnz() == sum((len(lv["L_vals"]) + len(lv["U_vals"]) + lv["m"] if lv["m"] else 0) +
                          (len(lv["E_vals"]) + len(lv["F_vals"]) if lv["n"] > lv["m"] else 0) for lv in levels) + \
        int(levels[-1].get("dense_n", 0)) ** 2
