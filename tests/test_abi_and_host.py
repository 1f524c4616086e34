"""CPU: the C-ABI library loads and exports every symbol include/hifir_amd.h declares (no compute
calls), and the host-side analysis (import checks, level schedules) is right.  No GPU needed."""
import ctypes
import os
import re
import subprocess
import sys

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
    M = hifir_amd.HIF(dtype=np.complex128 if np.iscomplexobj(levels[0]["d"]) else np.float64)
    for lv in levels:
        M.add_level(lv)
    if int(levels[-1].get("dense_n", 0)):
        if int(levels[-1].get("dense_lup", 0)):
            M.set_dense_lup(levels[-1]["dense"])
        elif int(levels[-1].get("dense_symm", 0)):  # host-side SYEIG: rank as the reference's symm_dense_solver found it
            M.set_dense_symm(levels[-1]["dense"], int(levels[-1].get("spd", 0)))
        else:
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


def test_cpp_facade_header_compiles_standalone(tmp_path):
    # include/hifir_amd.hpp must be usable with nothing but the C ABI header and the standard library
    # (std::vector stands in for hif::Array; a minimal mock stands in for the reference's hierarchy)
    import subprocess

    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include <vector>
#include <array>
#include "hifir_amd.hpp"
struct MockCcs { std::vector<long> cs; std::vector<int> ri; std::vector<double> v; size_t nc = 0;
  const std::vector<long>& col_start() const { return cs; } const std::vector<int>& row_ind() const { return ri; }
  const std::vector<double>& vals() const { return v; } size_t ncols() const { return nc; } };
struct MockDense { std::vector<double> a; size_t n = 0; bool empty() const { return true; }
  static const char* method() { return "QRCP"; }
  const MockDense& mat_backup() const { return *this; } size_t nrows() const { return n; } const double* data() const { return a.data(); } };
struct MockPrec { size_t m = 0, n = 0; MockCcs L_B, U_B, E, F; std::vector<double> d_B, s, t; std::vector<int> p, p_inv, q, q_inv;
  MockDense dense_solver, symm_dense_solver; };
struct MockHif { std::vector<MockPrec> ps; const std::vector<MockPrec>& precs() const { return ps; } };
int main() {
  hifamd::HIF<double> G;
  MockHif M;
  std::vector<double> b(4), x(4);
  std::vector<std::array<double, 2>> B(4), X(4);
  if (false) { G.attach(M); G.solve(b, x); G.solve(b, x, true, 3); G.solve_mrhs(B, X); G.mmultiply(b, x, true);
               G.set_matrix(4, nullptr, nullptr, nullptr); }
  return (int)G.levels() + (int)G.empty();
}
''')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-I", inc, str(src)])


@pytest.mark.parametrize("name", ["p2d_64_deep", "young1c", "p2d_100_tuned", "p2d_32_symm", "herm_24_symm", "p2d_30_lup"])
def test_hierarchy_file_roundtrip(name, tmp_path):
    # hifamd_save / hifamd_load (host side only: no GPU needed before finalize): the reloaded handle holds
    # the same hierarchy -- same counts, same level schedules -- and a corrupt file is refused
    levels, d = load_hier(name)
    M = hifir_amd.HIF(dtype=d["b"].dtype)
    for lv in levels:
        M.add_level(lv)
    if int(levels[-1].get("dense_n", 0)) > 0:
        M.set_dense(levels[-1]["dense"])
    path = str(tmp_path / "h.hifamd")
    M.save(path)
    M2 = hifir_amd.HIF.load(path, max_nrhs=0)  # 0: do not finalize (no device here)
    assert M2.dtype == M.dtype
    assert (M2.nnz(), M2.levels(), M2.nrows(), M2.schur_size(), M2.schur_rank()) == \
           (M.nnz(), M.levels(), M.nrows(), M.schur_size(), M.schur_rank())
    for l in range(len(levels)):
        for which in (0, 1):
            o1, w1 = M.level_schedule(l, which)
            o2, w2 = M2.level_schedule(l, which)
            assert np.array_equal(o1, o2) and np.array_equal(w1, w2)
    raw = open(path, "rb").read()
    bad = str(tmp_path / "bad.hifamd")
    open(bad, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(hifir_amd.HifAmdError):
        hifir_amd.HIF.load(bad, max_nrhs=0)
    open(bad, "wb").write(b"garbage!" + raw[8:])
    with pytest.raises(hifir_amd.HifAmdError):
        hifir_amd.HIF.load(bad, max_nrhs=0)


@pytest.mark.parametrize("name", ["p2d_64_deep", "young1c", "cd2d_48"])
def test_analysis_trailer_of_a_hierarchy_file(name, tmp_path):
    """hifamd_save_ex(HIFAMD_SAVE_ANALYSIS) / hifamd_load (host side only): the trailer is adopted when it fits, ignored
    -- with the same hierarchy as result -- when it is damaged, truncated, made under other planner options or
    switched off, and invisible to the records in front of it."""
    import subprocess

    levels, d = load_hier(name)
    plain, ana = str(tmp_path / "plain.hifamd"), str(tmp_path / "ana.hifamd")
    M = _save_fixture(name, plain)
    M.save(ana, analysis=True)
    raw_plain, raw = open(plain, "rb").read(), open(ana, "rb").read()
    assert raw[:len(raw_plain)] == raw_plain and len(raw) > len(raw_plain) + 24 and raw[-24:-16] == b"HIFAMDAF"

    def load(data, env=None):
        p = str(tmp_path / "t.hifamd")
        open(p, "wb").write(data)
        if env:  # (the planner options are read when a handle is created: another process)
            code = ("import sys; sys.path.insert(0, %r); import hifir_amd; M = hifir_amd.HIF.load(%r, max_nrhs=0); "
                    "print(int(M.stats_ext()['analysis_cached_levels']), M.nnz(), M.levels())" % (ROOT, p))
            out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **env)).decode().split()
            return int(out[0]), (int(out[1]), int(out[2]))
        M2 = hifir_amd.HIF.load(p, max_nrhs=0)
        for l in range(len(levels)):
            for which in (0, 1):
                o1, w1 = M.level_schedule(l, which)
                o2, w2 = M2.level_schedule(l, which)
                assert np.array_equal(o1, o2) and np.array_equal(w1, w2)
        # what a loaded handle saves is what was loaded (plain records; with the analysis: the same trailer)
        M2.save(p + ".again", analysis=True)
        assert open(p + ".again", "rb").read() == raw
        return int(M2.stats_ext()["analysis_cached_levels"]), (M2.nnz(), M2.levels())

    same = (M.nnz(), M.levels())
    assert load(raw) == (len(levels), same)
    assert load(raw_plain) == (0, same)
    assert load(raw, {"HIFIR_AMD_LOAD_ANALYSIS": "0"}) == (0, same)
    assert load(raw, {"HIFIR_AMD_BAND_DEPTH": "24"})[0] == 0    # made under other planner options: ignored
    assert load(raw, {"HIFIR_AMD_BAND_DEPTH": "32"})[0] == len(levels)
    nt = len(raw) - len(raw_plain)
    rng = np.random.default_rng(5)
    for off in [len(raw_plain) + 3, len(raw_plain) + nt // 2, len(raw) - 30, len(raw) - 5] + \
            [len(raw_plain) + int(k) for k in rng.integers(8, nt - 24, size=6)]:
        bad = bytearray(raw)
        bad[off] ^= 0x40
        assert load(bytes(bad)) == (0, same), off                # one flipped bit anywhere in the trailer: ignored
    for cut in (1, 8, 24, 25, nt // 2, nt - 1):
        assert load(raw[:len(raw) - cut]) == (0, same), cut      # truncated trailer: ignored
    # the trailer holds no matrix values: behind OTHER factors of the same pattern it is as good; of another pattern: ignored
    if name == "p2d_64_deep":
        lv2 = [dict(lv) for lv in levels]
        lv2[0]["L_vals"] = np.asarray(lv2[0]["L_vals"]) * 1.5
        M4 = hifir_amd.HIF(dtype=M.dtype)
        for lv in lv2:
            M4.add_level(lv)
        if int(levels[-1].get("dense_n", 0)) > 0:
            M4.set_dense(levels[-1]["dense"])
        other = str(tmp_path / "other.hifamd")
        M4.save(other)
        raw_other = open(other, "rb").read()
        assert len(raw_other) == len(raw_plain) and raw_other != raw_plain
        p = str(tmp_path / "o.hifamd")
        open(p, "wb").write(raw_other + raw[len(raw_plain):])
        M5 = hifir_amd.HIF.load(p, max_nrhs=0)
        assert int(M5.stats_ext()["analysis_cached_levels"]) == len(levels)
        lv3 = [dict(lv) for lv in levels]
        L0 = sp.csc_matrix((np.asarray(lv3[0]["L_vals"]), np.asarray(lv3[0]["L_rowind"]), np.asarray(lv3[0]["L_colptr"])),
                           shape=(int(lv3[0]["m"]),) * 2).tolil()
        i, j = [(int(a), int(b)) for a, b in zip(*L0.nonzero())][7]
        L0[i, j] = 0.0          # one entry fewer ...
        k = next(r for r in range(int(lv3[0]["m"]) - 1, j, -1) if L0[r, j] == 0 and r != i)
        L0[k, j] = 0.25         # ... one more elsewhere: same counts, another pattern
        L0 = sp.csc_matrix(L0)
        L0.eliminate_zeros()
        L0.sort_indices()
        lv3[0]["L_colptr"], lv3[0]["L_rowind"], lv3[0]["L_vals"] = L0.indptr.astype(np.int64), L0.indices.astype(np.int32), L0.data
        M6 = hifir_amd.HIF(dtype=M.dtype)
        for lv in lv3:
            M6.add_level(lv)
        if int(levels[-1].get("dense_n", 0)) > 0:
            M6.set_dense(levels[-1]["dense"])
        M6.save(other)
        raw3 = open(other, "rb").read()
        assert len(raw3) == len(raw_plain)
        open(p, "wb").write(raw3 + raw[len(raw_plain):])
        M7 = hifir_amd.HIF.load(p, max_nrhs=0)
        assert int(M7.stats_ext()["analysis_cached_levels"]) == len(levels) - 1  # (level 0 is analyzed afresh)
        for which in (0, 1):
            o1, w1 = M6.level_schedule(0, which)
            o2, w2 = M7.level_schedule(0, which)
            assert np.array_equal(o1, o2) and np.array_equal(w1, w2)


def test_host_eigensolver(tmp_path):
    """herm_eig / dense_factorize_symm of host.hpp (the SYEIG last level's host side) as a plain C++ program."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "eig_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(root, "hifir_amd", "csrc"),
                           os.path.join(root, "tests", "cpp", "eig_test.cpp"), "-o", exe])
    out = subprocess.check_output([exe]).decode()
    assert out.strip().endswith("OK"), out


def test_host_band_plan_program(tmp_path):
    """The host analysis (CCS->CSR, schedule, band plan incl. the nonzero reordering of block-dense bands, block
    inverses) on synthetic triangles as a plain C++ program (tests/cpp/plan_test.cpp; clean under ASan/UBSan)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "plan_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(root, "hifir_amd", "csrc"),
                           os.path.join(root, "tests", "cpp", "plan_test.cpp"), "-o", exe])
    out = subprocess.check_output([exe]).decode()
    assert out.strip().endswith("OK"), out


def _save_fixture(name, path):
    """hifamd_add_level + hifamd_set_dense* + hifamd_save need no GPU."""
    levels, d = load_hier(name)
    M = hifir_amd.HIF(dtype=np.complex128 if np.iscomplexobj(levels[0]["d"]) else np.float64)
    for lv in levels:
        M.add_level(lv)
    last = levels[-1]
    if int(last.get("dense_n", 0)) > 0:
        if int(last.get("dense_lup", 0)):
            M.set_dense_lup(last["dense"])
        elif int(last.get("dense_symm", 0)):
            M.set_dense_symm(last["dense"], int(last.get("spd", 0)))
        else:
            M.set_dense(last["dense"])
    M.save(path)
    return M


def test_load_refuses_truncated_and_inconsistent_files(tmp_path):
    """hifamd_load validates every array length against the level header and the column pointers BEFORE anything is
    handed to add_level (a short rowind / p_inv used to be over-read).  The same inputs run under
    AddressSanitizer in test_import_path_under_sanitizers."""
    path = str(tmp_path / "h.hifamd")
    _save_fixture("p2d_30", path)
    raw = open(path, "rb").read()
    L = hifir_amd.lib()
    h = ctypes.c_void_p()

    def try_load(data):
        p = str(tmp_path / "bad.hifamd")
        open(p, "wb").write(data)
        st = L.hifamd_load(p.encode(), -1, ctypes.byref(h))
        if st == 0:
            L.hifamd_destroy(h)
        else:
            assert L.hifamd_last_error()
        return st

    assert try_load(raw) == 0
    for cut in (17, 40, 100, len(raw) // 3, len(raw) // 2, len(raw) - 8, len(raw) - 1):
        assert try_load(raw[:cut]) == 3, cut  # HIFAMD_BAD_PREC
    # the int64 words right behind the header: #levels, has_dense, m, n, F_ncols, then L's shape and colptr count
    words = np.frombuffer(raw[16:16 + 8 * 8], dtype=np.int64)
    assert words[0] == 1 and words[2] < words[3]
    for k in range(8):
        for v in (-1, 1 << 40, int(words[k]) + 1):
            if k == 1 and v == 2:
                continue  # (has_dense 1 -> 2 is another VALID file: the same block factorized as a symmetric one)
            bad = bytearray(raw)
            bad[16 + 8 * k:24 + 8 * k] = int(v).to_bytes(8, "little", signed=True)
            assert try_load(bytes(bad)) in (2, 3), (k, v)
    # a permutation entry out of range / repeated (p_inv and q are gather indices on the device, too)
    levels, _ = load_hier("p2d_30")
    for key in ("p", "q_inv", "p_inv", "q"):
        for mut in (lambda a: a.__setitem__(3, len(a) + 2), lambda a: a.__setitem__(3, a[4])):
            lv = dict(levels[0])
            lv[key] = lv[key].copy()
            mut(lv[key])
            with pytest.raises(hifir_amd.HifAmdError) as e:
                hifir_amd.HIF().add_level(lv)
            assert e.value.code == 2
    # q / p_inv that are permutations but not the inverses of q_inv / p (the solve writes its output through q)
    for key in ("q", "p_inv"):
        lv = dict(levels[0])
        lv[key] = lv[key].copy()
        lv[key][[0, 1]] = lv[key][[1, 0]]
        with pytest.raises(hifir_amd.HifAmdError) as e:
            hifir_amd.HIF().add_level(lv)
        assert e.value.code == 2


def test_tensor_arguments_are_validated():
    """The device-pointer entry points get raw pointers: a CPU / wrong-dtype / wrongly shaped tensor must be refused
    in the wrapper (HifAmdError), never turned into a device access."""
    torch = pytest.importorskip("torch")
    levels, d = load_hier("p2d_5")
    M = hifir_amd.HIF()
    M.add_level(levels[0])
    M.set_dense(levels[0]["dense"])
    n = len(d["b"])
    for bad in (torch.zeros(n, 2, dtype=torch.float64),):  # CPU tensor
        for call in (lambda t: M.solve_mrhs(t), lambda t: M.mmultiply(t), lambda t: M.spmv(t), lambda t: M.hifir(t, 2),
                     lambda t: M.gmres(t), lambda t: M.time_apply(t, t)):
            with pytest.raises(hifir_amd.HifAmdError) as e:
                call(bad)
            assert e.value.code == 2
    with pytest.raises(hifir_amd.HifAmdError):
        M.solve(d["b"], x=np.zeros(n, dtype=np.float32))  # wrong dtype of a caller-supplied output
    with pytest.raises(hifir_amd.HifAmdError):
        M.solve_mrhs(np.zeros((n, 2)), X=np.zeros((n, 3)))
    with pytest.raises(hifir_amd.HifAmdError):
        M.solve_mrhs(np.zeros((n, 2)), X=np.zeros((2, n)).T)  # not C-contiguous


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_import_path_under_sanitizers(tmp_path, san):
    """The host half of hifamd_load / add_level / set_dense / finalize (import.hpp + host.hpp, the code engine.hip
    compiles) under ASan+UBSan and under TSan: several handles alive, two loads on two threads, the complex QRCP
    228^2 of young1c with the thread pool active, every block inverse, the adjoint hierarchy, ~1,500 hostile files
    (tests/cpp/import_san_test.cpp).  Round 1's intermittent corruption of E's row pointer happened on this path."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "import_san_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", f"-fsanitize={san}", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(root, "hifir_amd", "csrc"),
                           os.path.join(root, "tests", "cpp", "import_san_test.cpp"), "-o", exe])
    files = []
    for name in (["young1c", "p2d_5"] if san == "thread" else ["young1c", "p2d_32_symm", "p2d_30_lup", "p2d_5", "p2d_100_tuned"]):
        files.append(str(tmp_path / f"{name}.hifamd"))
        _save_fixture(name, files[-1])
    env = dict(os.environ, HIFIR_AMD_THREADS="4", ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe] + files, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    assert r.stderr.count("-> ok") == 2 * len(files)


def test_integration_doc_matches_headers():
    """INTEGRATION.md names real entry points with their real signatures: the additive prototypes quoted in section 3 are
    the ones of include/libhifir_amd_ext.h, token for token, and every `lhf...` / `hifamd_...` name the document mentions
    is declared by the reference's libhifir.h (tests/golden/libhifir_symbols.txt), the extension header or
    include/hifir_amd.h (round 2's document had drifted: argument order of lhfSetDevices, three names that do not exist)."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    ext = open(os.path.join(root, "include", "libhifir_amd_ext.h")).read()
    amd = open(os.path.join(root, "include", "hifir_amd.h")).read()
    norm = lambda t: re.sub(r"\s+", " ", t).strip()
    proto_re = r"^(?:LhfStatus|int|Lhf[dz]HifHdl)\s+lhf\w+\([^;]*\);"
    hdr_protos = [norm(p) for p in re.findall(proto_re, ext, flags=re.M)]
    assert len(hdr_protos) == 16
    blocks = re.findall(r"```c\n(.*?)```", doc, flags=re.S)
    doc_protos = [norm(p) for b in blocks for p in re.findall(proto_re, b, flags=re.M)]
    assert doc_protos == hdr_protos  # same prototypes, same order, same argument lists
    ref_syms = set(open(os.path.join(root, "tests", "golden", "libhifir_symbols.txt")).read().split())
    ext_syms = set(re.findall(r"\b(lhf\w+)\s*\(", ext))
    amd_syms = set(re.findall(r"\b(hifamd_\w+)\s*\(", amd))
    known = ref_syms | ext_syms
    # names written with a type placeholder (lhf?Apply, lhf{d,z}Create, `lhf{d,s,z,c}CreateMatrix / DestroyMatrix / ...`)
    # are expanded; plain names must exist as they stand
    for name in set(re.findall(r"\blhf[A-Za-z]+\b", doc)):
        assert name in known or any(name == s for s in known), name
    for fam, rest in re.findall(r"lhf(?:\?|\{([a-z,]+)\})([A-Z][A-Za-z]+)", doc):
        letters = fam.split(",") if fam else ["d", "z"]
        for t in letters:
            assert f"lhf{t}{rest}" in known, f"lhf{t}{rest}"
    for name in set(re.findall(r"\bhifamd_[a-z_0-9]+\b", doc)):
        if name.endswith("_"):
            continue  # (a prefix such as hifamd_set_dense*)
        assert name in amd_syms or any(s.startswith(name) for s in amd_syms), name
