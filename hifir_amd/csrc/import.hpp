// import.hpp -- host-only half of the hierarchy import: argument validation of hifamd_add_level, CCS -> CSR,
// the band analysis of one level, the adjoint level, the on-disk format (hifamd_save / hifamd_load) and the
// invariant check that hifamd_finalize runs before anything is shipped to HBM.
//
// Pure C++17, no HIP: engine.hip uses it for the product, and tests/cpp/import_san_test.cpp compiles the very same
// code with g++ / clang++ under -fsanitize=address,undefined and -fsanitize=thread (the GPU pool has no sanitizers).
// Reference data contract: hif::Prec, src/hif/alg/Prec.hpp:82-334.
#pragma once
#include "host.hpp"

namespace hifamd {

// status codes of include/hifir_amd.h (== LhfStatus, libhifir/include/libhifir.h:148-154), usable without that header
enum : int { kNullObj = 1, kMismatchedSizes = 2, kBadPrec = 3, kHifirError = 4 };

template <class T>
Ccs<T> make_ccs(int64_t nrows, int64_t ncols, const int64_t *cp, const int32_t *ri, const T *v) {
  Ccs<T> A;
  A.nrows = nrows;
  A.ncols = ncols;
  A.colptr.assign((size_t)ncols + 1, 0);
  if (ncols > 0 && cp) {
    if (cp[0] != 0) throw Error(kMismatchedSizes, "CCS column pointer must start at 0");
    for (int64_t j = 0; j < ncols; ++j)
      if (cp[j + 1] < cp[j]) throw Error(kMismatchedSizes, "CCS column pointer not monotone");
    A.colptr.assign(cp, cp + ncols + 1);
    const int64_t nz = cp[ncols];
    if (nz > 0 && (!ri || !v)) throw Error(kNullObj, "NULL index/value array with nnz > 0");
    A.rowind.assign(ri, ri + nz);
    A.vals.assign(v, v + nz);
  }
  return A;
}

// every entry in [0, n) exactly once
inline bool is_permutation(const std::vector<int32_t> &p, int64_t n) {
  if ((int64_t)p.size() != n) return false;
  std::vector<uint8_t> seen((size_t)n, 0);
  for (int32_t v : p) {
    if (v < 0 || v >= n || seen[(size_t)v]) return false;
    seen[(size_t)v] = 1;
  }
  return true;
}

// The arguments of hifamd_add_level -> one HostLevel with its four matrices in row-gather form (no analysis yet).
// parent_nm: n - m of the previous level, or -1 for the first one.
template <class T>
HostLevel<T> import_level(int64_t parent_nm, int64_t m, int64_t n, const int64_t *Lcp, const int32_t *Lri, const T *Lv,
                          const int64_t *Ucp, const int32_t *Uri, const T *Uv, const int64_t *Ecp, const int32_t *Eri,
                          const T *Ev, int64_t F_ncols, const int64_t *Fcp, const int32_t *Fri, const T *Fv, const T *d,
                          const double *s, const double *t, const int32_t *p, const int32_t *p_inv, const int32_t *q,
                          const int32_t *q_inv) {
  if (m < 0 || n < m || n <= 0) throw Error(kMismatchedSizes, "need 0 <= m <= n, n > 0");
  if (n > (int64_t)std::numeric_limits<int32_t>::max()) throw Error(kMismatchedSizes, "level size exceeds int32 indices");
  if (parent_nm >= 0 && parent_nm != n)
    throw Error(kMismatchedSizes, "level size must equal the parent's Schur complement size n-m");
  if (!s || !t || !p || !q_inv || (m > 0 && !d)) throw Error(kNullObj, "NULL level vector");
  const int64_t nm = n - m;
  if (F_ncols != 0 && F_ncols != nm) throw Error(kMismatchedSizes, "F must have n-m columns (or 0)");
  HostLevel<T> H;
  H.m = m;
  H.n = n;
  H.F_ncols = F_ncols;
  H.L = make_ccs(m, m, Lcp, Lri, Lv);
  H.U = make_ccs(m, m, Ucp, Uri, Uv);
  H.E = make_ccs(nm, nm ? m : 0, nm ? Ecp : nullptr, Eri, Ev);
  H.F = make_ccs(m, F_ncols, F_ncols ? Fcp : nullptr, Fri, Fv);
  H.d.assign(d, d + m);
  H.s.assign(s, s + n);
  H.t.assign(t, t + n);
  H.p.assign(p, p + n);
  H.q_inv.assign(q_inv, q_inv + n);
  if (p_inv) H.p_inv.assign(p_inv, p_inv + n);
  if (q) H.q.assign(q, q + n);
  // all four are gather indices on the device (p, q_inv: solve; q, p_inv: transpose solve and products)
  if (!is_permutation(H.p, n) || !is_permutation(H.q_inv, n) || (p_inv && !is_permutation(H.p_inv, n)) ||
      (q && !is_permutation(H.q, n)))
    throw Error(kMismatchedSizes, "p, q_inv (and p_inv, q when given) must be permutations of [0, n)");
  // ... and each other's inverses (Prec.hpp:317-321 builds them that way): the solve uses q to write its output from
  // inside the last triangular kernel, the transposed solve and the products use both
  for (int64_t i = 0; i < n; ++i)
    if ((p_inv && H.p_inv[(size_t)H.p[(size_t)i]] != i) || (q && H.q[(size_t)H.q_inv[(size_t)i]] != i))
      throw Error(kMismatchedSizes, "p_inv / q must be the inverse permutations of p / q_inv");
  H.Lr = ccs_to_csr(H.L, false);  // (row indices are range-checked there)
  H.Ur = ccs_to_csr(H.U, true);
  H.Er = ccs_to_csr(H.E, false);
  H.Fr = ccs_to_csr(H.F, false);
  return H;
}

// two independent pieces of host work side by side (an exception of either is rethrown after both have finished)
template <class FA, class FB>
void par2(FA &&fa, FB &&fb) {
  {  // HIFIR_AMD_THREADS=1 (or a single hardware thread): strictly one after the other, on the calling thread
    unsigned hw = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("HIFIR_AMD_THREADS")) hw = (unsigned)std::max(1, std::atoi(e));
    if (hw == 1) {
      fa();
      fb();
      return;
    }
  }
  std::exception_ptr ea, eb;
  std::thread ta([&] {
    try {
      fa();
    } catch (...) {
      ea = std::current_exception();
    }
  });
  try {
    fb();
  } catch (...) {
    eb = std::current_exception();
  }
  ta.join();
  if (ea) std::rethrow_exception(ea);
  if (eb) std::rethrow_exception(eb);
}

// schedules, band plans, slot-ordered matrices and the block cutting of one level (H.Lr .. H.Fr given)
template <class T>
void analyze_level(HostLevel<T> &H, const BandOptions &band_opt, bool dump = false, size_t level_no = 0) {
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  par2([&] { H.Ls = level_schedule(H.Lr, true); }, [&] { H.Us = level_schedule(H.Ur, false); });
  double t1 = now();
  // component-dense plan (host.hpp plan_bands_cd) for triangles with real rows to gather; the depth-cut bands for the
  // nearly diagonal ones (level 0 of a PDE hierarchy: ~2 nonzeros per row, shallow, bandwidth-bound) and in exact mode
  auto use_cd = [&](const Csr<T> &A) {
    return band_opt.cd_rows > 0 && band_opt.dense_block > 0 && A.nrows > 0 &&
           (double)A.col.size() >= band_opt.cd_min_row_nnz * (double)A.nrows;
  };
  H.top.clear();
  H.top_n = 0;
  H.top_bandL = H.top_bandU = -1;
  if (use_cd(H.Lr) && use_cd(H.Ur)) {
    H.Lp = plan_bands_cd(H.Lr, H.Ls, true, band_opt);
    // the narrow top of the level as ONE dense operator (host.hpp choose_top): replan both triangles around it
    if (band_opt.top_max > 0) H.top = choose_top(H.Lr, H.Ur, H.Lp, band_opt.top_max, band_opt.top_few_wgs);
    if (!H.top.empty()) {
      par2([&] { H.Lp = plan_bands_cd(H.Lr, H.Ls, true, band_opt, &H.top); },
           [&] { H.Up = plan_bands_cd(H.Ur, H.Us, false, band_opt, &H.top); });
      for (uint8_t t : H.top) H.top_n += t;
      H.top_bandL = (int32_t)H.Lp.nbands() - 1;  // lower: the rest band comes last; upper: first
      H.top_bandU = 0;
      if (H.top_n > band_opt.max_wg_rows) {  // (the rest would have been cut into several bands: no combined operator)
        H.top.clear();
        H.top_n = 0;
        H.top_bandL = H.top_bandU = -1;
        par2([&] { H.Lp = plan_bands_cd(H.Lr, H.Ls, true, band_opt); }, [&] { H.Up = plan_bands_cd(H.Ur, H.Us, false, band_opt); });
      }
    } else {
      H.Up = plan_bands_cd(H.Ur, H.Us, false, band_opt);
    }
  } else {
    // thin triangles (level 0 of a PDE hierarchy): the same subtree plan with the components' own nonzeros solved
    // sparsely in LDS (BandPlan::cd_sparse); exact mode and complex data keep the depth-cut flag bands
    auto plan_one = [&](const Csr<T> &A, const Schedule &S, bool lower) {
      if (use_cd(A)) return plan_bands_cd(A, S, lower, band_opt);
      // (shallow triangles only: inside a component the sparse substitution pays a barrier per depth level)
      if (band_opt.cd_rows > 0 && band_opt.cd_sparse_rows > 0 && band_opt.dense_block > 0 && A.nrows >= band_opt.cd_sparse_min_rows &&
          S.nwf() <= band_opt.cd_sparse_max_depth)
        return plan_bands_cd(A, S, lower, band_opt, nullptr, true);
      return plan_bands(A, S, lower, band_opt);
    };
    par2([&] { H.Lp = plan_one(H.Lr, H.Ls, true); }, [&] { H.Up = plan_one(H.Ur, H.Us, false); });
  }
  double t2 = now();
  // (the two triangles of a level are planned, permuted and cut independently: one host thread each)
  par2(
      [&] {
        H.Lr = permute_rows(H.Lr, H.Lp.order);
        finish_band_plan(H.Lp, H.Lr, band_opt);
      },
      [&] {
        H.Ur = permute_rows(H.Ur, H.Up.order);
        finish_band_plan(H.Up, H.Ur, band_opt);
      });
  double t3 = now();
  par2(
      [&] {
        H.Ltinv_elems = plan_dense_blocks<T>(H.Lp, band_opt);
        build_cd_streams(H.Lp, H.Lr.ptr);
      },
      [&] {
        H.Utinv_elems = plan_dense_blocks<T>(H.Up, band_opt);
        build_cd_streams(H.Up, H.Ur.ptr);
      });
  if (!dump) return;
  std::fprintf(stderr, "ANALYZE m=%ld: schedule %.2f s, band plan %.2f s, permute+finish %.2f s, block cutting %.2f s\n",
               (long)H.m, t1 - t0, t2 - t1, t3 - t2, now() - t3);
  for (int tri = 0; tri < 2; ++tri) {  // development aid: one line per band
    const BandPlan &P = tri ? H.Up : H.Lp;
    const Csr<T> &A = tri ? H.Ur : H.Lr;
    for (int64_t b = 0; b < P.nbands(); ++b) {
      const int32_t g0 = P.band_wg_ptr[(size_t)b], g1 = P.band_wg_ptr[(size_t)b + 1];
      const int32_t s0 = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g0]], s1 = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g1]];
      int64_t maxnnz = 0, maxdepth = 0, maxrows = 0;
      for (int32_t g = g0; g < g1; ++g) {
        const int32_t a = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g]], e = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g + 1]];
        maxnnz = std::max<int64_t>(maxnnz, A.ptr[(size_t)e] - A.ptr[(size_t)a]);
        maxrows = std::max<int64_t>(maxrows, e - a);
        maxdepth = std::max<int64_t>(maxdepth, P.wg_grp_ptr[(size_t)g + 1] - P.wg_grp_ptr[(size_t)g]);
      }
      int64_t distinct = 0;  // component-dense bands: distinct source rows per component, summed (reuse potential)
      if (P.band_cd[(size_t)b]) {
        std::vector<int32_t> u;
        for (int32_t c = P.wg_grp_ptr[(size_t)g0]; c < P.wg_grp_ptr[(size_t)g1]; ++c) {
          u.clear();
          for (int32_t q = P.grp_slot_ptr[(size_t)c]; q < P.grp_slot_ptr[(size_t)c + 1]; ++q)
            for (int32_t k = P.split[(size_t)q]; k < P.csplit[(size_t)q]; ++k) u.push_back(P.srcslot[(size_t)k]);
          std::sort(u.begin(), u.end());
          distinct += std::unique(u.begin(), u.end()) - u.begin();
        }
      }
      int32_t ck16 = 0, rowmax = 0, ne_max = 0;  // most entries of one wave chunk, of one row, of one component
      if (P.band_cd[(size_t)b] && !P.cd_desc.empty()) {
        for (int32_t c = P.wg_grp_ptr[(size_t)g0]; c < P.wg_grp_ptr[(size_t)g1]; ++c) {
          const uint16_t *wm = reinterpret_cast<const uint16_t *>(&P.cd_desc[(size_t)c * kCdDescWords + 11]);
          for (int q = 0; q < 16; ++q) ck16 = std::max<int32_t>(ck16, wm[q + 1] - wm[q]);
          ne_max = std::max(ne_max, P.cd_desc[(size_t)c * kCdDescWords + 3]);
        }
        for (int32_t q = s0; q < s1; ++q) rowmax = std::max(rowmax, P.csplit[(size_t)q] - P.split[(size_t)q]);
      }
      int64_t runs = 0, span_sum = 0;  // address locality of the band's rows: runs of consecutive row ids, id span per component
      for (int32_t q = s0; q < s1; ++q) runs += (q == s0 || A.rowid[(size_t)q] != A.rowid[(size_t)q - 1] + 1);
      for (int32_t c = P.wg_grp_ptr[(size_t)g0]; c < P.wg_grp_ptr[(size_t)g1]; ++c) {
        int32_t lo = std::numeric_limits<int32_t>::max(), hi = -1;
        for (int32_t q = P.grp_slot_ptr[(size_t)c]; q < P.grp_slot_ptr[(size_t)c + 1]; ++q)
          lo = std::min(lo, A.rowid[(size_t)q]), hi = std::max(hi, A.rowid[(size_t)q]);
        if (hi >= lo) span_sum += hi - lo + 1;
      }
      int64_t lvl_sum = 0, lvl_max = 0, ncomp_sp = 0;  // sparse-own components: depth levels of the LDS substitution
      if (P.band_cd[(size_t)b] && P.cd_sparse && !P.cd_desc.empty())
        for (int32_t c = P.wg_grp_ptr[(size_t)g0]; c < P.wg_grp_ptr[(size_t)g1]; ++c) {
          const int32_t nl = P.cd_desc[(size_t)c * kCdDescWords + 24];
          lvl_sum += nl, lvl_max = std::max<int64_t>(lvl_max, nl), ++ncomp_sp;
        }
      std::fprintf(stderr, "PLAN2 level=%zu tri=%c band=%ld comp_entries_max=%d chunk_entries_max=%d row_entries_max=%d id_runs=%ld id_span_sum=%ld own_levels_mean=%.1f own_levels_max=%ld\n", level_no,
                   tri ? 'U' : 'L', (long)b, ne_max, ck16, rowmax, (long)runs, (long)span_sum, ncomp_sp ? (double)lvl_sum / (double)ncomp_sp : 0.0, (long)lvl_max);
      int64_t own = 0, prevb = 0;  // nonzeros inside the rows' own component / gathered by the band kernel itself
      for (int32_t q = s0; q < s1; ++q) {
        own += A.ptr[(size_t)q + 1] - P.csplit[(size_t)q];
        prevb += P.csplit[(size_t)q] - P.split[(size_t)q];
      }
      std::fprintf(stderr, "PLAN level=%zu tri=%c band=%ld rows=%d nnz=%d wgs=%d prefix=%d dense=%d fused=%d cd=%d comps=%d own=%ld inband=%ld distinct=%ld maxwg_nnz=%ld maxwg_rows=%ld maxdepth=%ld\n",
                   level_no, tri ? 'U' : 'L', (long)b, s1 - s0, A.ptr[(size_t)s1] - A.ptr[(size_t)s0], g1 - g0,
                   (int)P.band_prefix[(size_t)b], (int)P.band_dense[(size_t)b], (int)P.band_fused[(size_t)b],
                   (int)P.band_cd[(size_t)b], P.wg_grp_ptr[(size_t)g1] - P.wg_grp_ptr[(size_t)g0], (long)own, (long)prevb, (long)distinct,
                   (long)maxnnz, (long)maxrows, (long)maxdepth);
    }
  }
}

// x = M^{-H} b (prec_solve_tran, alg/prec_solve.hpp:542-612) is the SAME machinery on the adjoint hierarchy:
// L' = U^H, U' = L^H, E' = F^H, F' = E^H, d' = conj(d), (s', p') = (t, q), (t', q_inv') = (s, p_inv).
template <class T>
HostLevel<T> adjoint_level(const HostLevel<T> &P) {
  if (P.q.empty() || P.p_inv.empty())
    throw Error(kBadPrec, "the transpose apply needs the q and p_inv permutations (hifamd_add_level)");
  HostLevel<T> H;
  H.m = P.m;
  H.n = P.n;
  const int64_t nm = P.n - P.m;
  H.F_ncols = nm;  // E^H prolongs whenever there is a Schur complement (prec_solve.hpp:602)
  H.Lr = adjoint_rows(P.U);  // U^H: strict lower
  H.Ur = adjoint_rows(P.L);  // L^H: strict upper
  if (P.F_ncols) {
    H.Er = adjoint_rows(P.F);  // F^H: nm x m
  } else {                     // no F: y[m:n] = t[q] b[q] (:574) -- an empty restriction
    H.E_void = true;
    H.Er.nrows = nm;
    H.Er.ncols = nm ? P.m : 0;
    H.Er.ptr.assign((size_t)nm + 1, 0);
    H.Er.rowid.resize((size_t)nm);
    for (int64_t i = 0; i < nm; ++i) H.Er.rowid[(size_t)i] = (int32_t)i;
  }
  H.Fr = adjoint_rows(P.E);  // E^H: m x nm
  H.d.resize(P.d.size());
  for (size_t i = 0; i < P.d.size(); ++i) H.d[i] = conj_(P.d[i]);
  H.s = P.t;
  H.t = P.s;
  H.p = P.q;
  H.q_inv = P.p_inv;
  H.q = P.p;  // product on the adjoint hierarchy = prec_prod_tran: (s, p) in, (t, q_inv) out
  H.p_inv = P.q_inv;
  return H;
}

// conjugate transpose of the user's CRS matrix (for iterative refinement with A^H, IterRefine.hpp:96)
template <class T>
Csr<T> adjoint_of_csr(const Csr<T> &A) {
  Csr<T> B;
  B.nrows = A.ncols;
  B.ncols = A.nrows;
  B.ptr.assign((size_t)A.ncols + 1, 0);
  for (size_t k = 0; k < A.col.size(); ++k) ++B.ptr[(size_t)A.col[k] + 1];
  for (int64_t j = 0; j < A.ncols; ++j) B.ptr[(size_t)j + 1] += B.ptr[(size_t)j];
  B.col.resize(A.col.size());
  B.val.resize(A.val.size());
  std::vector<int32_t> fill(B.ptr.begin(), B.ptr.end() - 1);
  for (int64_t i = 0; i < A.nrows; ++i)
    for (int32_t k = A.ptr[(size_t)i]; k < A.ptr[(size_t)i + 1]; ++k) {
      const int32_t pos = fill[(size_t)A.col[(size_t)k]]++;
      B.col[(size_t)pos] = (int32_t)i;
      B.val[(size_t)pos] = conj_(A.val[(size_t)k]);
    }
  return B;
}

// ---------------------------------------------------------------------------------------------
// Invariants of everything the kernels index with.  hifamd_finalize runs this right before the upload: a hierarchy
// whose converted arrays are not what the conversion must have produced (whatever the cause) is refused with
// HIFAMD_HIFIR_ERROR instead of being applied -- a wrong row pointer on the device is a hang or a wrong answer.
// ---------------------------------------------------------------------------------------------
template <class T>
void check_csr(const Csr<T> &A, int64_t nnz_expected, const char *what, size_t level_no) {
  auto fail = [&](const char *why, int64_t at) {
    throw Error(kHifirError, std::string("internal error: ") + what + " of level " + std::to_string(level_no) + ": " + why +
                                 " at " + std::to_string(at) + " (host copy of the hierarchy is corrupt)");
  };
  if ((int64_t)A.ptr.size() != A.nrows + 1) fail("row pointer length", (int64_t)A.ptr.size());
  if (A.ptr[0] != 0) fail("row pointer does not start at 0", 0);
  for (int64_t i = 0; i < A.nrows; ++i)
    if (A.ptr[(size_t)i + 1] < A.ptr[(size_t)i]) fail("row pointer not monotone", i);
  const int64_t nz = A.ptr[(size_t)A.nrows];
  if (nz != nnz_expected || (int64_t)A.col.size() != nz || (int64_t)A.val.size() != nz) fail("nonzero count", nz);
  for (int64_t k = 0; k < nz; ++k)
    if (A.col[(size_t)k] < 0 || A.col[(size_t)k] >= A.ncols) fail("column index out of range", k);
  if ((int64_t)A.rowid.size() != A.nrows) fail("row id length", (int64_t)A.rowid.size());
  if (!is_permutation(A.rowid, A.nrows)) fail("row ids are not a permutation", 0);
}

template <class T>
void check_band_plan(const BandPlan &P, const Csr<T> &A, const char *what, size_t level_no, const BandOptions *opt = nullptr) {
  // (components must fit the LDS block the kernels size from the planner options)
  const int64_t cd_rows_limit = opt ? (P.cd_sparse ? opt->cd_sparse_rows : opt->cd_rows) : (int64_t)255;
  auto fail = [&](const char *why, int64_t at) {
    throw Error(kHifirError, std::string("internal error: band plan of ") + what + " of level " + std::to_string(level_no) +
                                 ": " + why + " at " + std::to_string(at));
  };
  const int64_t m = A.nrows, nz = (int64_t)A.col.size();
  if ((int64_t)P.split.size() != m || (int64_t)P.srcslot.size() != nz) fail("array length", m);
  for (int64_t s = 0; s < m; ++s)
    if (P.split[(size_t)s] < A.ptr[(size_t)s] || P.split[(size_t)s] > A.ptr[(size_t)s + 1]) fail("split outside its row", s);
  for (int64_t k = 0; k < nz; ++k)
    if (P.srcslot[(size_t)k] < 0 || P.srcslot[(size_t)k] >= m) fail("source slot out of range", k);
  auto monotone = [&](const std::vector<int32_t> &v, int64_t last, const char *name) {
    if (v.empty() || v[0] != 0 || v.back() != last) fail(name, (int64_t)v.size());
    for (size_t i = 0; i + 1 < v.size(); ++i)
      if (v[i + 1] < v[i]) fail(name, (int64_t)i);
  };
  monotone(P.grp_slot_ptr, (int32_t)m, "group pointer");
  monotone(P.wg_grp_ptr, (int32_t)P.grp_slot_ptr.size() - 1, "workgroup pointer");
  monotone(P.band_wg_ptr, (int32_t)P.wg_grp_ptr.size() - 1, "band pointer");
  const int64_t nbands = (int64_t)P.band_wg_ptr.size() - 1;
  if ((int64_t)P.band_cd.size() != nbands || (int64_t)P.band_dense.size() != nbands || (int64_t)P.band_prefix.size() != nbands ||
      (int64_t)P.band_fused.size() != nbands || (int64_t)P.csplit.size() != m ||
      (int64_t)P.grp_inv_off.size() != (int64_t)P.grp_slot_ptr.size() - 1)
    fail("per-band array length", nbands);
  for (int64_t s = 0; s < m; ++s)
    if (P.csplit[(size_t)s] < P.split[(size_t)s] || P.csplit[(size_t)s] > A.ptr[(size_t)s + 1]) fail("component split outside its row", s);
  for (int64_t b = 0; b < nbands; ++b)
    for (int32_t g = P.band_wg_ptr[(size_t)b]; g < P.band_wg_ptr[(size_t)b + 1]; ++g) {
      if (P.band_cd[(size_t)b]) {  // components must fit the LDS block of the kernel and own an inverse
        for (int32_t c = P.wg_grp_ptr[(size_t)g]; c < P.wg_grp_ptr[(size_t)g + 1]; ++c) {
          const int32_t nb = P.grp_slot_ptr[(size_t)c + 1] - P.grp_slot_ptr[(size_t)c];
          if (nb < 1 || nb > cd_rows_limit) fail("component size", c);
          if (!P.cd_sparse && P.grp_inv_off[(size_t)c] < 0) fail("component without an inverse", c);
        }
      } else if (P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g + 1]] - P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g]] > 16384 &&
                 !P.band_dense[(size_t)b])
        fail("workgroup owns more rows than it has LDS flags", g);
    }
  for (size_t q = 0; q < P.blk_slot0.size(); ++q)
    if (P.blk_slot0[q] < 0 || P.blk_slot1[q] <= P.blk_slot0[q] || P.blk_slot1[q] > m) fail("block range", (int64_t)q);
}

template <class T>
void check_level_invariants(const HostLevel<T> &H, size_t level_no, bool adjoint = false, const BandOptions *opt = nullptr) {
  const int64_t m = H.m, n = H.n, nm = n - m;
  auto fail = [&](const char *why) {
    throw Error(kHifirError, std::string("internal error: level ") + std::to_string(level_no) + ": " + why +
                                 " (host copy of the hierarchy is corrupt)");
  };
  if (H.Lr.nrows != m || H.Ur.nrows != m || H.Er.nrows != nm) fail("matrix shapes");
  if (H.Lr.ncols != m || H.Ur.ncols != m || (nm && H.Er.ncols != m)) fail("matrix shapes");
  // (a level without a Schur complement has no E / F at all: their stored shapes are then whatever the import left)
  if (H.F_ncols && (H.Fr.nrows != m || H.Fr.ncols != nm)) fail("matrix shapes");
  // (the adjoint level keeps no CCS copies: its row forms ARE the CCS arrays of the primary level)
  check_csr(H.Lr, adjoint ? (int64_t)H.Lr.col.size() : H.L.nnz(), "L", level_no);
  check_csr(H.Ur, adjoint ? (int64_t)H.Ur.col.size() : H.U.nnz(), "U", level_no);
  check_csr(H.Er, adjoint ? (int64_t)H.Er.col.size() : H.E.nnz(), "E", level_no);
  check_csr(H.Fr, adjoint ? (int64_t)H.Fr.col.size() : H.F.nnz(), "F", level_no);
  check_band_plan(H.Lp, H.Lr, "L", level_no, opt);
  check_band_plan(H.Up, H.Ur, "U", level_no, opt);
  if ((int64_t)H.d.size() != m || (int64_t)H.s.size() != n || (int64_t)H.t.size() != n) fail("vector lengths");
  if (!is_permutation(H.p, n) || !is_permutation(H.q_inv, n)) fail("p / q_inv are not permutations");
  if (!H.p_inv.empty() && !is_permutation(H.p_inv, n)) fail("p_inv is not a permutation");
  if (!H.q.empty() && !is_permutation(H.q, n)) fail("q is not a permutation");
}

// ---------------------------------------------------------------------------------------------
// Integrity of the host copy between hifamd_add_level and hifamd_finalize.  seal_level checksums every array of a level
// once it is imported and analyzed; verify_level (first thing in finalize) recomputes them.  A host copy that changed in
// between is NOT what this library wrote: in rounds 1 and 4 two row pointers of E were found overwritten inside a pytest
// process (torch + numpy + the OpenMP oracle loaded; cause never shown, sanitizer runs clean).  What happens then:
//   * the row forms of E / F (identity row order: pure functions of the imported CCS arrays) are rebuilt from the imported
//     arrays when THOSE still carry their checksums; the first difference is reported on stderr (index, value found, value
//     expected -- evidence for whoever meets it next) and counted (hifamd_stats_ext slot 21);
//   * anything else that changed is refused with the array's name (HIFAMD_HIFIR_ERROR), as check_level_invariants would
//     refuse a structure that no longer holds.
// ---------------------------------------------------------------------------------------------
template <class V>
uint64_t array_sum(const std::vector<V> &a) {
  const size_t bytes = a.size() * sizeof(V);
  const unsigned char *b = reinterpret_cast<const unsigned char *>(a.data());
  const int64_t kChunk = 1 << 20;  // bytes per chunk: chunks hashed side by side, combined in order
  const int64_t nch = (int64_t)((bytes + kChunk - 1) / kChunk);
  std::vector<uint64_t> part((size_t)std::max<int64_t>(nch, 1), 0);
  parallel_for(nch, 1, [&](int64_t c0, int64_t c1) {
    for (int64_t c = c0; c < c1; ++c) {
      const size_t lo = (size_t)c * kChunk, hi = std::min(bytes, lo + (size_t)kChunk);
      uint64_t h = 1469598103934665603ull;
      size_t i = lo;
      for (; i + 8 <= hi; i += 8) {
        uint64_t w;
        std::memcpy(&w, b + i, 8);
        h = (h ^ w) * 1099511628211ull;
      }
      for (; i < hi; ++i) h = (h ^ b[i]) * 1099511628211ull;
      part[(size_t)c] = h;
    }
  });
  uint64_t h = 1469598103934665603ull ^ (uint64_t)bytes;
  for (int64_t c = 0; c < nch; ++c) h = (h ^ part[(size_t)c]) * 1099511628211ull;
  return h;
}

static const char *const kSumNames[] = {
    "L.colptr", "L.rowind", "L.vals", "U.colptr", "U.rowind", "U.vals", "E.colptr", "E.rowind", "E.vals", "F.colptr", "F.rowind",
    "F.vals", "d", "s", "t", "p", "p_inv", "q", "q_inv", "Lr.ptr", "Lr.col", "Lr.val", "Lr.rowid", "Ur.ptr", "Ur.col", "Ur.val",
    "Ur.rowid", "Er.ptr", "Er.col", "Er.val", "Er.rowid", "Fr.ptr", "Fr.col", "Fr.val", "Fr.rowid"};
constexpr int kSumE = 6, kSumF = 9, kSumEr = 27, kSumFr = 31, kSumCount = 35;

template <class T>
std::vector<uint64_t> level_sums(const HostLevel<T> &H) {
  std::vector<uint64_t> v;
  v.reserve(kSumCount);
  for (const Ccs<T> *A : {&H.L, &H.U, &H.E, &H.F}) {
    v.push_back(array_sum(A->colptr));
    v.push_back(array_sum(A->rowind));
    v.push_back(array_sum(A->vals));
  }
  v.push_back(array_sum(H.d));
  v.push_back(array_sum(H.s));
  v.push_back(array_sum(H.t));
  v.push_back(array_sum(H.p));
  v.push_back(array_sum(H.p_inv));
  v.push_back(array_sum(H.q));
  v.push_back(array_sum(H.q_inv));
  for (const Csr<T> *A : {&H.Lr, &H.Ur, &H.Er, &H.Fr}) {
    v.push_back(array_sum(A->ptr));
    v.push_back(array_sum(A->col));
    v.push_back(array_sum(A->val));
    v.push_back(array_sum(A->rowid));
  }
  return v;
}
template <class T>
void seal_level(HostLevel<T> &H) {
  H.sums = level_sums(H);
}

template <class V>
static std::string first_difference(const char *name, const std::vector<V> &found, const std::vector<V> &expect) {
  if (found.size() != expect.size())
    return std::string(name) + ": " + std::to_string(found.size()) + " entries, " + std::to_string(expect.size()) + " expected";
  for (size_t i = 0; i < found.size(); ++i)
    if (std::memcmp(&found[i], &expect[i], sizeof(V)) != 0) {
      size_t cnt = 0;
      for (size_t k = i; k < found.size(); ++k) cnt += std::memcmp(&found[k], &expect[k], sizeof(V)) != 0;
      char buf[256];
      unsigned long long fb = 0, eb = 0;
      std::memcpy(&fb, &found[i], std::min(sizeof(V), sizeof(fb)));
      std::memcpy(&eb, &expect[i], std::min(sizeof(V), sizeof(eb)));
      std::snprintf(buf, sizeof(buf), "%s[%zu] (of %zu, %zu-byte entries, at %p): found 0x%llx, expected 0x%llx; %zu entries differ", name, i,
                    found.size(), sizeof(V), (const void *)&found[i], fb, eb, cnt);
      std::string out = buf;
      // the 64 bytes around the first difference as they are now (whose data is it? doubles, pointers, text ...)
      const unsigned char *base = reinterpret_cast<const unsigned char *>(found.data());
      const size_t nbytes = found.size() * sizeof(V), at = i * sizeof(V);
      const size_t lo = at >= 24 ? (at - 24) & ~(size_t)7 : 0, hi = std::min(nbytes, lo + 64);
      out += "; bytes [" + std::to_string(lo) + ", " + std::to_string(hi) + "):";
      for (size_t k = lo; k < hi; ++k) {
        std::snprintf(buf, sizeof(buf), "%s%02x", (k % 8 == 0) ? " " : "", base[k]);
        out += buf;
      }
      return out;
    }
  return std::string(name) + ": equal";
}

// -> number of arrays repaired (0: the host copy is what seal_level saw); throws when it cannot vouch for the level
template <class T>
int verify_level(HostLevel<T> &H, size_t level_no, bool adjoint) {
  if (H.sums.empty()) return 0;  // (a level that was never sealed: nothing to compare with)
  const std::vector<uint64_t> now = level_sums(H);
  if (now == H.sums) return 0;
  int repaired = 0;
  std::string other;
  bool e_bad = false, f_bad = false;
  for (int k = 0; k < kSumCount; ++k) {
    if (now[(size_t)k] == H.sums[(size_t)k]) continue;
    if (k >= kSumEr && k < kSumEr + 4)
      e_bad = true;
    else if (k >= kSumFr && k < kSumFr + 4)
      f_bad = true;
    else
      other += std::string(other.empty() ? "" : ", ") + kSumNames[k];
  }
  auto rebuild = [&](const char *what, const Ccs<T> &src, int ksrc, Csr<T> &dst, int kdst) {
    for (int k = 0; k < 3; ++k)
      if (now[(size_t)(ksrc + k)] != H.sums[(size_t)(ksrc + k)]) return false;  // (the imported arrays changed as well)
    if (adjoint) return false;                                                    // (adjoint levels keep no CCS copies)
    Csr<T> R = ccs_to_csr(src, false);
    std::string d = first_difference((std::string(what) + "r.ptr").c_str(), dst.ptr, R.ptr);
    if (d.find(": equal") != std::string::npos) d = first_difference((std::string(what) + "r.col").c_str(), dst.col, R.col);
    if (d.find(": equal") != std::string::npos) d = first_difference((std::string(what) + "r.val").c_str(), dst.val, R.val);
    if (d.find(": equal") != std::string::npos) d = first_difference((std::string(what) + "r.rowid").c_str(), dst.rowid, R.rowid);
    std::fprintf(stderr,
                 "hifir_amd: WARNING: the host copy of level %zu changed between hifamd_add_level and hifamd_finalize -- NOT written by "
                 "this library: %s.  Rebuilt from the imported arrays (their checksums hold).\n",
                 level_no, d.c_str());
    dst = std::move(R);
    const std::vector<uint64_t> again = level_sums(H);
    for (int k = 0; k < 4; ++k)
      if (again[(size_t)(kdst + k)] != H.sums[(size_t)(kdst + k)]) return false;  // (not the arrays that were sealed)
    ++repaired;
    return true;
  };
  if (e_bad && !rebuild("E", H.E, kSumE, H.Er, kSumEr)) other += std::string(other.empty() ? "" : ", ") + "Er";
  if (f_bad && !rebuild("F", H.F, kSumF, H.Fr, kSumFr)) other += std::string(other.empty() ? "" : ", ") + "Fr";
  if (!other.empty())
    throw Error(kHifirError, "internal error: level " + std::to_string(level_no) + ": the host copy changed between hifamd_add_level "
                             "and hifamd_finalize (not written by this library): " + other + " (host copy of the hierarchy is corrupt)");
  return repaired;
}

// ---------------------------------------------------------------------------------------------
// On-disk form of the imported hierarchy: exactly the hifamd_add_level / hifamd_set_dense arguments, so that a
// hierarchy factorized once on a host with the reference can be applied on GPU nodes that do not have it.
// Little-endian, 8-byte aligned records, see include/hifir_amd.h.
// ---------------------------------------------------------------------------------------------
template <class V>
void put_vec(std::FILE *f, const std::vector<V> &v) {
  const int64_t cnt = (int64_t)v.size();
  if (std::fwrite(&cnt, sizeof(cnt), 1, f) != 1) throw Error(kHifirError, "short write");
  if (cnt && std::fwrite(v.data(), sizeof(V), (size_t)cnt, f) != (size_t)cnt) throw Error(kHifirError, "short write");
  const size_t padb = (8 - (cnt * sizeof(V)) % 8) % 8;
  const char zeros[8] = {0};
  if (padb && std::fwrite(zeros, 1, padb, f) != padb) throw Error(kHifirError, "short write");
}
// `limit`: the largest count the file may claim (what the header of the record allows) -- a corrupt count is
// refused BEFORE anything of that size is allocated
template <class V>
void get_vec(std::FILE *f, std::vector<V> &v, int64_t limit) {
  int64_t cnt = 0;
  if (std::fread(&cnt, sizeof(cnt), 1, f) != 1) throw Error(kBadPrec, "truncated hierarchy file");
  if (cnt < 0 || cnt > limit) throw Error(kBadPrec, "corrupt hierarchy file (array length)");
  // ... and by what the file still holds: the record header itself comes from the file (n up to INT32_MAX), so a
  // few-byte hostile or truncated file must not make this allocate gigabytes before the short read is noticed
  if (cnt > 0) {
    const long here = std::ftell(f);
    if (here >= 0 && std::fseek(f, 0, SEEK_END) == 0) {
      const long end = std::ftell(f);
      if (std::fseek(f, here, SEEK_SET) != 0) throw Error(kHifirError, "hierarchy file: seek failed");
      if (end >= here && (uint64_t)cnt > (uint64_t)(end - here) / sizeof(V)) throw Error(kBadPrec, "truncated hierarchy file");
    }
  }
  v.resize((size_t)cnt);
  if (cnt && std::fread(v.data(), sizeof(V), (size_t)cnt, f) != (size_t)cnt) throw Error(kBadPrec, "truncated hierarchy file");
  const size_t padb = (8 - (cnt * sizeof(V)) % 8) % 8;
  char skip[8];
  if (padb && std::fread(skip, 1, padb, f) != padb) throw Error(kBadPrec, "truncated hierarchy file");
}
template <class T>
void put_ccs(std::FILE *f, const Ccs<T> &A) {
  const int64_t hdr[2] = {A.nrows, A.ncols};
  if (std::fwrite(hdr, sizeof(hdr), 1, f) != 1) throw Error(kHifirError, "short write");
  put_vec(f, A.colptr);
  put_vec(f, A.rowind);
  put_vec(f, A.vals);
}
// nrows / ncols: what the level header implies; every length is checked against it and against colptr.back()
// before a single pointer is handed on
template <class T>
void get_ccs(std::FILE *f, Ccs<T> &A, int64_t nrows, int64_t ncols) {
  int64_t hdr[2];
  if (std::fread(hdr, sizeof(hdr), 1, f) != 1) throw Error(kBadPrec, "truncated hierarchy file");
  if (hdr[0] != nrows || hdr[1] != ncols) throw Error(kBadPrec, "inconsistent hierarchy file (matrix shape)");
  A.nrows = nrows;
  A.ncols = ncols;
  get_vec(f, A.colptr, ncols + 1);
  if ((int64_t)A.colptr.size() != ncols + 1 || A.colptr[0] != 0) throw Error(kBadPrec, "inconsistent hierarchy file (column pointer)");
  for (int64_t j = 0; j < ncols; ++j)
    if (A.colptr[(size_t)j + 1] < A.colptr[(size_t)j]) throw Error(kBadPrec, "inconsistent hierarchy file (column pointer)");
  const int64_t nz = A.colptr.back();
  if (nz > (int64_t)std::numeric_limits<int32_t>::max()) throw Error(kBadPrec, "inconsistent hierarchy file (nonzero count)");
  get_vec(f, A.rowind, nz);
  get_vec(f, A.vals, nz);
  if ((int64_t)A.rowind.size() != nz || (int64_t)A.vals.size() != nz) throw Error(kBadPrec, "inconsistent hierarchy file (nonzero count)");
}

template <class T>
void save_hierarchy(std::FILE *f, const HostHierarchy<T> &host) {
  const int64_t nl = (int64_t)host.levels.size(), hd = host.has_dense ? (host.dense.kind == 2 ? 3 : host.dense.kind == 1 ? 2 : 1) : 0;
  if (std::fwrite(&nl, 8, 1, f) != 1 || std::fwrite(&hd, 8, 1, f) != 1) throw Error(kHifirError, "short write");
  for (const auto &H : host.levels) {
    const int64_t hdr[3] = {H.m, H.n, H.F_ncols};
    if (std::fwrite(hdr, sizeof(hdr), 1, f) != 1) throw Error(kHifirError, "short write");
    put_ccs(f, H.L);
    put_ccs(f, H.U);
    put_ccs(f, H.E);
    put_ccs(f, H.F);
    put_vec(f, H.d);
    put_vec(f, H.s);
    put_vec(f, H.t);
    put_vec(f, H.p);
    put_vec(f, H.p_inv);
    put_vec(f, H.q);
    put_vec(f, H.q_inv);
  }
  if (hd) {
    const int64_t nd = host.dense.n;
    const double par = host.dense.kind == 1 ? (double)host.dense.spd : host.dense.rrqr_cond;  // (hd == 2: spd)
    if (std::fwrite(&nd, 8, 1, f) != 1 || std::fwrite(&par, 8, 1, f) != 1) throw Error(kHifirError, "short write");
    put_vec(f, host.dense.mat);
  }
}

// Replays a file into `sink` (add_level / set_dense / set_dense_symm / set_dense_lup with the ABI's argument
// lists).  Every array length is validated against the level header and the column pointers before use.
template <class T, class Sink>
void load_hierarchy(std::FILE *f, Sink &sink) {
  int64_t nl = 0, hd = 0;
  if (std::fread(&nl, 8, 1, f) != 1 || std::fread(&hd, 8, 1, f) != 1) throw Error(kBadPrec, "truncated hierarchy file");
  if (nl < 1 || nl > 4096 || hd < 0 || hd > 3) throw Error(kBadPrec, "corrupt hierarchy file (header)");
  int64_t last_nm = 0;
  for (int64_t l = 0; l < nl; ++l) {
    int64_t hdr[3];
    if (std::fread(hdr, sizeof(hdr), 1, f) != 1) throw Error(kBadPrec, "truncated hierarchy file");
    const int64_t m = hdr[0], n = hdr[1], fn = hdr[2], nm = n - m;
    if (m < 0 || n < m || n <= 0 || n > (int64_t)std::numeric_limits<int32_t>::max() || (fn != 0 && fn != nm))
      throw Error(kBadPrec, "corrupt hierarchy file (level header)");
    Ccs<T> L, U, E, F;
    std::vector<T> d;
    std::vector<double> s, t;
    std::vector<int32_t> p, p_inv, q, q_inv;
    get_ccs(f, L, m, m);
    get_ccs(f, U, m, m);
    get_ccs(f, E, nm, nm ? m : 0);
    get_ccs(f, F, m, fn);
    get_vec(f, d, m);
    get_vec(f, s, n);
    get_vec(f, t, n);
    get_vec(f, p, n);
    get_vec(f, p_inv, n);
    get_vec(f, q, n);
    get_vec(f, q_inv, n);
    if ((int64_t)d.size() != m || (int64_t)s.size() != n || (int64_t)t.size() != n || (int64_t)p.size() != n ||
        (int64_t)q_inv.size() != n || (!p_inv.empty() && (int64_t)p_inv.size() != n) || (!q.empty() && (int64_t)q.size() != n))
      throw Error(kBadPrec, "inconsistent hierarchy file (vector lengths)");
    sink.add_level(m, n, L.colptr.data(), L.rowind.data(), L.vals.data(), U.colptr.data(), U.rowind.data(), U.vals.data(),
                   E.colptr.data(), E.rowind.data(), E.vals.data(), fn, fn ? F.colptr.data() : nullptr, F.rowind.data(),
                   F.vals.data(), d.data(), s.data(), t.data(), p.data(), p_inv.empty() ? nullptr : p_inv.data(),
                   q.empty() ? nullptr : q.data(), q_inv.data());
    last_nm = nm;
  }
  if (hd) {
    int64_t nd = 0;
    double cond = 0.0;
    if (std::fread(&nd, 8, 1, f) != 1 || std::fread(&cond, 8, 1, f) != 1) throw Error(kBadPrec, "truncated hierarchy file");
    if (nd != last_nm || nd <= 0) throw Error(kBadPrec, "inconsistent hierarchy file (dense block size)");
    std::vector<T> mat;
    get_vec(f, mat, nd * nd);
    if ((int64_t)mat.size() != nd * nd) throw Error(kBadPrec, "inconsistent hierarchy file (dense block)");
    if (hd == 3)
      sink.set_dense_lup(nd, mat.data());
    else if (hd == 2) {
      // (the record's double carries the caller's integer spd flag -- its sign matters: a NaN or a value out of int's
      //  range, whose conversion is undefined behaviour, is a corrupt file)
      if (!(std::isfinite(cond) && std::fabs(cond) <= 1e9 && cond == std::floor(cond)))
        throw Error(kBadPrec, "corrupt hierarchy file (spd flag of the symmetric block)");
      sink.set_dense_symm(nd, mat.data(), (int)cond);
    }
    else
      sink.set_dense(nd, mat.data(), cond);
  }
}

// ---------------------------------------------------------------------------------------------
// Optional trailer of a hierarchy file: the ANALYSIS of every level (hifamd_save_ex, flag HIFAMD_SAVE_ANALYSIS) -- the
// wavefront schedules, the band plans and the slot-ordered triangles that analyze_level derives (on a 256^3 grid: 24 s
// of the 45 s a rank spends between hifamd_load and its first apply).  A process that loads such a file with the SAME
// planner options skips that work: ranks other than the one that factorized, restarts.  What is cheap and deterministic
// (block cutting, component streams: plan_dense_blocks / build_cd_streams) is recomputed, not stored.
// Layout, behind the records of save_hierarchy:  "HIFAMDA1", sizeof(T), planner options (doubles), per level a block of
// scalars and vectors, "HIFAMDAF", offset of the trailer, FNV-1a 64 of its bytes.  A loader that does not know the
// trailer never reads it; one that does verifies magic, checksum and options (load_analysis), then level by level every
// size and index range against the factors it has just imported (adopt_analysis: the slot-ordered triangles are REBUILT
// from the imported ones through the stored origins, so the trailer carries no matrix value and fits any factors of the
// same sparsity pattern), and falls back to analyze_level when anything is off -- a damaged trailer costs time; with a
// valid checksum and edited plan arrays it can give wrong numbers, but no index out of range reaches a kernel;
// check_level_invariants at finalize applies to cached plans as to computed ones.
// ---------------------------------------------------------------------------------------------
struct HashIo {
  std::FILE *f;
  uint64_t h = 1469598103934665603ull;
  bool ok = true;
  unsigned char pend[8];
  size_t npend = 0;
  // FNV-1a over 8-byte words of the byte STREAM (independent of how the stream is cut into calls; the arrays are
  // gigabytes on the grids this exists for, hence words); value(): the bytes of an unfinished word are mixed one by one
  void mix(const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    size_t i = 0;
    if (npend) {
      while (npend < 8 && i < n) pend[npend++] = b[i++];
      if (npend < 8) return;
      uint64_t w;
      std::memcpy(&w, pend, 8);
      h = (h ^ w) * 1099511628211ull;
      npend = 0;
    }
    for (; i + 8 <= n; i += 8) {
      uint64_t w;
      std::memcpy(&w, b + i, 8);
      h = (h ^ w) * 1099511628211ull;
    }
    while (i < n) pend[npend++] = b[i++];
  }
  uint64_t value() const {
    uint64_t r = h;
    for (size_t i = 0; i < npend; ++i) r = (r ^ pend[i]) * 1099511628211ull;
    return r;
  }
  void wr(const void *p, size_t n) {
    if (n && std::fwrite(p, 1, n, f) != n) throw Error(kHifirError, "short write");
    mix(p, n);
  }
  bool rd(void *p, size_t n) {
    if (n && std::fread(p, 1, n, f) != n) return ok = false;
    mix(p, n);
    return true;
  }
};
template <class V>
void tput(HashIo &io, const std::vector<V> &v) {
  const int64_t cnt = (int64_t)v.size();
  io.wr(&cnt, 8);
  io.wr(v.data(), (size_t)cnt * sizeof(V));
  const char zeros[8] = {0};
  io.wr(zeros, (8 - ((size_t)cnt * sizeof(V)) % 8) % 8);
}
template <class V>
bool tget(HashIo &io, std::vector<V> &v, int64_t bytes_left) {
  int64_t cnt = 0;
  if (!io.rd(&cnt, 8) || cnt < 0 || (uint64_t)cnt > (uint64_t)bytes_left / sizeof(V)) return io.ok = false;
  v.resize((size_t)cnt);
  char skip[8];
  return io.rd(v.data(), (size_t)cnt * sizeof(V)) && io.rd(skip, (8 - ((size_t)cnt * sizeof(V)) % 8) % 8);
}
// every planner option that shapes a plan, as doubles (a trailer made under other options is ignored)
inline std::vector<double> band_option_words(const BandOptions &o, size_t sizeof_t) {
  return {1.0 /* trailer version */, (double)sizeof_t, (double)kCdDescWords, (double)kCdOwnCap, (double)o.thin_rows, (double)o.band_depth,
          (double)o.max_comp_weight, (double)o.max_wg_rows, (double)o.max_wgs, (double)o.dense_block, (double)o.fuse,
          (double)o.fuse_reorder, (double)o.fuse_max_wgs, (double)o.dense_min_rows, (double)o.cd_rows, o.cd_min_row_nnz,
          (double)o.cd_fuse_max_wgs, (double)o.cd_max_nnz, (double)o.cd_sparse_max_depth, (double)o.cd_sparse_rows,
          (double)o.top_max, (double)o.top_few_wgs, o.dense_max_growth, (double)o.cd_sparse_min_rows};
}
static const char kAnaMagic[8] = {'H', 'I', 'F', 'A', 'M', 'D', 'A', '1'};
static const char kAnaFoot[8] = {'H', 'I', 'F', 'A', 'M', 'D', 'A', 'F'};

// what analyze_level derives and the trailer carries.  The slot-ordered triangles travel as ORIGINS: entry k of the
// slot-ordered row form is entry origin[k] of the row form as imported (ccs_to_csr) -- no matrix value is stored twice,
// and an adopted triangle is by construction the imported one with its rows permuted and reordered inside
template <class T>
struct LevelAnalysis {
  int64_t m = 0, nzL = 0, nzU = 0;
  std::vector<int32_t> originL, originU;
  Schedule Ls, Us;
  BandPlan Lp, Up;  // (core arrays only: block cutting and component streams are rebuilt)
  std::vector<uint8_t> top;
  int64_t top_n = 0;
  int32_t top_bandL = -1, top_bandU = -1;
};
// Ap: slot-ordered (rowid = slot -> row), A0: the same matrix as imported (rowid = identity)
template <class T>
std::vector<int32_t> csr_origin(const Csr<T> &Ap, const Csr<T> &A0) {
  std::vector<int32_t> origin(Ap.col.size());
  std::vector<std::pair<int32_t, int32_t>> byc;
  for (int64_t q = 0; q < Ap.nrows; ++q) {
    const int32_t i = Ap.rowid[(size_t)q];
    byc.clear();
    for (int32_t k = A0.ptr[(size_t)i]; k < A0.ptr[(size_t)i + 1]; ++k) byc.emplace_back(A0.col[(size_t)k], k);
    std::sort(byc.begin(), byc.end());
    for (int32_t k = Ap.ptr[(size_t)q]; k < Ap.ptr[(size_t)q + 1]; ++k) {
      auto it = std::lower_bound(byc.begin(), byc.end(), std::make_pair(Ap.col[(size_t)k], (int32_t)-1));
      if (it == byc.end() || it->first != Ap.col[(size_t)k]) throw Error(kHifirError, "internal error: slot-ordered triangle differs from the imported one");
      origin[(size_t)k] = it->second;
    }
  }
  return origin;
}
// the inverse: false when order is not a permutation or origin is not, row by row, a bijection onto the imported row
template <class T>
bool csr_from_origin(const Csr<T> &A0, const std::vector<int32_t> &order, const std::vector<int32_t> &origin, Csr<T> &out) {
  const int64_t m = A0.nrows, nz = (int64_t)A0.col.size();
  if ((int64_t)order.size() != m || (int64_t)origin.size() != nz || !is_permutation(order, m)) return false;
  out.nrows = m, out.ncols = A0.ncols;
  out.rowid = order;
  out.ptr.assign((size_t)m + 1, 0);
  for (int64_t q = 0; q < m; ++q) {
    const int32_t i = order[(size_t)q];
    out.ptr[(size_t)q + 1] = out.ptr[(size_t)q] + (A0.ptr[(size_t)i + 1] - A0.ptr[(size_t)i]);
  }
  out.col.resize((size_t)nz), out.val.resize((size_t)nz);
  std::vector<uint8_t> seen((size_t)nz, 0);
  for (int64_t q = 0; q < m; ++q) {
    const int32_t i = order[(size_t)q], a = A0.ptr[(size_t)i], e = A0.ptr[(size_t)i + 1];
    for (int32_t k = out.ptr[(size_t)q]; k < out.ptr[(size_t)q + 1]; ++k) {
      const int32_t o = origin[(size_t)k];
      if (o < a || o >= e || seen[(size_t)o]) return false;
      seen[(size_t)o] = 1;
      out.col[(size_t)k] = A0.col[(size_t)o], out.val[(size_t)k] = A0.val[(size_t)o];
    }
  }
  return true;
}
inline void tput_plan(HashIo &io, const BandPlan &P) {
  const int64_t sp = P.cd_sparse ? 1 : 0;
  io.wr(&sp, 8);
  tput(io, P.order), tput(io, P.grp_slot_ptr), tput(io, P.wg_grp_ptr), tput(io, P.band_wg_ptr), tput(io, P.band_prefix);
  tput(io, P.band_fused), tput(io, P.band_dense), tput(io, P.band_cd), tput(io, P.band_old), tput(io, P.srcslot);
  tput(io, P.split), tput(io, P.csplit);
}
inline bool tget_plan(HashIo &io, BandPlan &P, int64_t left) {
  int64_t sp = 0;
  if (!io.rd(&sp, 8) || (sp != 0 && sp != 1)) return io.ok = false;
  P.cd_sparse = sp != 0;
  return tget(io, P.order, left) && tget(io, P.grp_slot_ptr, left) && tget(io, P.wg_grp_ptr, left) && tget(io, P.band_wg_ptr, left) &&
         tget(io, P.band_prefix, left) && tget(io, P.band_fused, left) && tget(io, P.band_dense, left) && tget(io, P.band_cd, left) &&
         tget(io, P.band_old, left) && tget(io, P.srcslot, left) && tget(io, P.split, left) && tget(io, P.csplit, left);
}
template <class T>
void save_analysis(std::FILE *f, const HostHierarchy<T> &host, const BandOptions &opt) {
  const long at = std::ftell(f);
  if (at < 0) throw Error(kHifirError, "hierarchy file: tell failed");
  HashIo io{f};
  io.wr(kAnaMagic, 8);
  const std::vector<double> ow = band_option_words(opt, sizeof(T));
  tput(io, ow);
  const int64_t nl = (int64_t)host.levels.size();
  io.wr(&nl, 8);
  for (const auto &H : host.levels) {
    const int64_t sc[6] = {H.m, (int64_t)H.Lr.col.size(), (int64_t)H.Ur.col.size(), H.top_n, H.top_bandL, H.top_bandU};
    io.wr(sc, sizeof(sc));
    tput(io, csr_origin(H.Lr, ccs_to_csr(H.L, false)));
    tput(io, csr_origin(H.Ur, ccs_to_csr(H.U, true)));
    tput(io, H.Ls.order), tput(io, H.Ls.wf_ptr), tput(io, H.Us.order), tput(io, H.Us.wf_ptr);
    tput_plan(io, H.Lp), tput_plan(io, H.Up);
    tput(io, H.top);
  }
  const int64_t foot[2] = {(int64_t)at, (int64_t)io.value()};
  if (std::fwrite(kAnaFoot, 8, 1, f) != 1 || std::fwrite(foot, 16, 1, f) != 1) throw Error(kHifirError, "short write");
}

// index ranges of a cached plan: everything build_cd_streams / plan_dense_blocks and the upload index with
template <class T>
bool cached_plan_ok(const BandPlan &P, const Csr<T> &A, int64_t m, int64_t nz, const BandOptions &opt) {
  if (A.nrows != m || A.ncols != m || (int64_t)A.ptr.size() != m + 1 || (int64_t)A.col.size() != nz || (int64_t)A.val.size() != nz ||
      (int64_t)A.rowid.size() != m)
    return false;
  if (A.ptr[0] != 0 || A.ptr[(size_t)m] != nz) return false;
  for (int64_t i = 0; i < m; ++i)
    if (A.ptr[(size_t)i + 1] < A.ptr[(size_t)i]) return false;
  for (int64_t k = 0; k < nz; ++k)
    if (A.col[(size_t)k] < 0 || A.col[(size_t)k] >= m) return false;
  if (!is_permutation(A.rowid, m) || P.order != A.rowid) return false;
  auto mono = [](const std::vector<int32_t> &v, int64_t last) {
    if (v.empty() || v[0] != 0 || v.back() != last) return false;
    for (size_t i = 0; i + 1 < v.size(); ++i)
      if (v[i + 1] < v[i]) return false;
    return true;
  };
  if (!mono(P.grp_slot_ptr, m) || !mono(P.wg_grp_ptr, (int64_t)P.grp_slot_ptr.size() - 1) ||
      !mono(P.band_wg_ptr, (int64_t)P.wg_grp_ptr.size() - 1))
    return false;
  const size_t nb = P.band_wg_ptr.size() - 1;
  // (exactly the lengths check_band_plan demands at finalize: a plan that passes here must not fail there)
  auto per_band = [&](const std::vector<uint8_t> &v) { return v.size() == nb; };
  if (!per_band(P.band_prefix) || !per_band(P.band_dense) || !per_band(P.band_fused) || !per_band(P.band_cd) || !per_band(P.band_old))
    return false;
  if ((int64_t)P.split.size() != m || (int64_t)P.srcslot.size() != nz || (int64_t)P.csplit.size() != m) return false;
  for (int64_t s = 0; s < m; ++s) {
    const int32_t a = A.ptr[(size_t)s], e = A.ptr[(size_t)s + 1], sp = P.split[(size_t)s];
    if (sp < a || sp > e) return false;
    if (!P.csplit.empty() && (P.csplit[(size_t)s] < sp || P.csplit[(size_t)s] > e)) return false;
  }
  {  // a nonzero's source slot IS the slot of its column's row: nothing to choose
    std::vector<int32_t> slot_of((size_t)m);
    for (int64_t q = 0; q < m; ++q) slot_of[(size_t)A.rowid[(size_t)q]] = (int32_t)q;
    for (int64_t k = 0; k < nz; ++k)
      if (P.srcslot[(size_t)k] != slot_of[(size_t)A.col[(size_t)k]]) return false;
  }
  // what plan_dense_blocks will lay out for these flags: schemes the options in force do not have are refused, and so
  // is a volume of explicit inverses no plan of THIS triangle can have (flags of a damaged trailer must not make the
  // loader allocate without bound: planned volumes are 13-15 x the nonzeros on the grids measured)
  double inv_elems = 0.0;
  for (size_t b = 0; b < nb; ++b) {
    const bool cd = !P.band_cd.empty() && P.band_cd[b];
    if (cd) {  // (component streams index rows with a byte)
      if (P.csplit.empty() || opt.cd_rows <= 0 || opt.dense_block <= 0 || (P.cd_sparse && opt.cd_sparse_rows <= 0)) return false;
      for (int32_t g = P.band_wg_ptr[b]; g < P.band_wg_ptr[b + 1]; ++g)
        for (int32_t c = P.wg_grp_ptr[(size_t)g]; c < P.wg_grp_ptr[(size_t)g + 1]; ++c) {
          const int32_t rows = P.grp_slot_ptr[(size_t)c + 1] - P.grp_slot_ptr[(size_t)c];
          // (the kernels size their LDS block from the planner OPTIONS: a cached component must fit what they allow)
          if (rows < 1 || rows > 255 || rows > (P.cd_sparse ? opt.cd_sparse_rows : opt.cd_rows)) return false;
          if (!P.cd_sparse) inv_elems += (double)plane_elems(rows, round_up32(rows));
          // a component's own nonzeros [csplit, end) refer to EARLIER rows of the same component
          const int32_t s0 = P.grp_slot_ptr[(size_t)c];
          for (int32_t r = 0; r < rows; ++r)
            for (int32_t k = P.csplit[(size_t)(s0 + r)]; k < A.ptr[(size_t)(s0 + r) + 1]; ++k)
              if (P.srcslot[(size_t)k] < s0 || P.srcslot[(size_t)k] >= s0 + r) return false;
        }
    } else if (P.band_dense[b]) {
      if (opt.dense_block <= 0) return false;
      const int32_t g = P.band_wg_ptr[b];
      const int64_t rows = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)P.band_wg_ptr[b + 1]]] - P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g]];
      const int64_t nblk = (rows + opt.dense_block - 1) / opt.dense_block;
      inv_elems += (double)nblk * (double)plane_elems(std::min<int64_t>(rows, opt.dense_block), round_up32(std::min<int64_t>(rows, opt.dense_block)));
    }
  }
  if (inv_elems > 256.0 * (double)nz + 67108864.0) return false;
  return true;
}
inline bool cached_schedule_ok(const Schedule &S, int64_t m) {
  if ((int64_t)S.order.size() != m || S.wf_ptr.empty() || S.wf_ptr[0] != 0 || S.wf_ptr.back() != m) return false;
  for (size_t i = 0; i + 1 < S.wf_ptr.size(); ++i)
    if (S.wf_ptr[i + 1] < S.wf_ptr[i]) return false;
  return is_permutation(S.order, m);
}

// Reads the trailer of an open hierarchy file, if it has one that fits `opt`; the file position is restored.
// false: no (usable) trailer -- the caller analyzes as usual.
template <class T>
bool load_analysis(std::FILE *f, const BandOptions &opt, std::vector<LevelAnalysis<T>> &out) {
  out.clear();
  const long here = std::ftell(f);
  if (here < 0) return false;
  bool good = false;
  do {
    if (std::fseek(f, 0, SEEK_END) != 0) break;
    const long end = std::ftell(f);
    if (end < here + 24 || std::fseek(f, end - 24, SEEK_SET) != 0) break;
    char foot[8];
    int64_t fo[2];
    if (std::fread(foot, 8, 1, f) != 1 || std::fread(fo, 16, 1, f) != 1 || std::memcmp(foot, kAnaFoot, 8) != 0) break;
    if (fo[0] < here || fo[0] > end - 24 - 8 || std::fseek(f, (long)fo[0], SEEK_SET) != 0) break;
    const int64_t left = end - 24 - fo[0];  // no array of the trailer can be longer than the trailer
    HashIo io{f};
    char magic[8];
    if (!io.rd(magic, 8) || std::memcmp(magic, kAnaMagic, 8) != 0) break;
    std::vector<double> ow;
    if (!tget(io, ow, left) || ow != band_option_words(opt, sizeof(T))) break;
    int64_t nl = 0;
    if (!io.rd(&nl, 8) || nl < 1 || nl > 4096) break;
    out.resize((size_t)nl);
    bool all = true;
    for (int64_t l = 0; l < nl && all; ++l) {
      LevelAnalysis<T> &A = out[(size_t)l];
      int64_t sc[6];
      all = io.rd(sc, sizeof(sc)) && tget(io, A.originL, left) && tget(io, A.originU, left) && tget(io, A.Ls.order, left) &&
            tget(io, A.Ls.wf_ptr, left) && tget(io, A.Us.order, left) && tget(io, A.Us.wf_ptr, left) && tget_plan(io, A.Lp, left) &&
            tget_plan(io, A.Up, left) && tget(io, A.top, left);
      if (!all) break;
      A.m = sc[0], A.nzL = sc[1], A.nzU = sc[2], A.top_n = sc[3], A.top_bandL = (int32_t)sc[4], A.top_bandU = (int32_t)sc[5];
    }
    if (!all || std::ftell(f) != end - 24 || (int64_t)io.value() != fo[1]) break;
    good = true;
  } while (false);
  if (!good) out.clear();
  if (std::fseek(f, here, SEEK_SET) != 0) throw Error(kHifirError, "hierarchy file: seek failed");
  return good;
}

// a level whose analysis came from the trailer: the slot-ordered triangles rebuilt from the imported ones, every size and
// index range checked against them, then the cached arrays in place of analyze_level's and the cheap rest rebuilt.
// false (H untouched): the trailer does not fit this level -- the caller analyzes as usual
template <class T>
bool adopt_analysis(HostLevel<T> &H, LevelAnalysis<T> &A, const BandOptions &opt) {
  const int64_t m = H.m;
  if (A.m != m || A.nzL != (int64_t)H.Lr.col.size() || A.nzU != (int64_t)H.Ur.col.size()) return false;
  Csr<T> Lr, Ur;
  bool okL = false, okU = false;  // (the two triangles side by side, like their analysis)
  par2([&] { okL = csr_from_origin(H.Lr, A.Lp.order, A.originL, Lr) && cached_plan_ok(A.Lp, Lr, m, A.nzL, opt) &&
                   (m == 0 || cached_schedule_ok(A.Ls, m)); },
       [&] { okU = csr_from_origin(H.Ur, A.Up.order, A.originU, Ur) && cached_plan_ok(A.Up, Ur, m, A.nzU, opt) &&
                   (m == 0 || cached_schedule_ok(A.Us, m)); });
  if (!okL || !okU) return false;
  if (!A.top.empty()) {  // the combined top operator: L's last band and U's first hold exactly the flagged rows
    const int64_t nbL = (int64_t)A.Lp.band_wg_ptr.size() - 1, nbU = (int64_t)A.Up.band_wg_ptr.size() - 1;
    if ((int64_t)A.top.size() != m || nbL < 1 || nbU < 1 || A.top_bandL != nbL - 1 || A.top_bandU != 0 || A.top_n < 1 ||
        A.top_n > m || opt.top_max <= 0)
      return false;
    int64_t cnt = 0;
    for (uint8_t t : A.top) cnt += t != 0;
    if (cnt != A.top_n) return false;
    auto band_is_top = [&](const BandPlan &P, const Csr<T> &M, int64_t b) {
      const int32_t s0 = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)P.band_wg_ptr[(size_t)b]]];
      const int32_t s1 = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)P.band_wg_ptr[(size_t)b + 1]]];
      if (s1 - s0 != A.top_n) return false;
      for (int32_t q = s0; q < s1; ++q)
        if (!A.top[(size_t)M.rowid[(size_t)q]]) return false;
      return true;
    };
    if (!band_is_top(A.Lp, Lr, nbL - 1) || !band_is_top(A.Up, Ur, 0)) return false;
  } else if (A.top_n != 0 || A.top_bandL != -1 || A.top_bandU != -1) {
    return false;
  }
  // the cheap rest (block cutting, component streams) is rebuilt on the CANDIDATE plans, before anything of H is
  // replaced: an error there leaves H untouched and the caller analyzes as usual
  int64_t lt = 0, ut = 0;
  try {
    par2(
        [&] {
          lt = plan_dense_blocks<T>(A.Lp, opt);
          build_cd_streams(A.Lp, Lr.ptr);
        },
        [&] {
          ut = plan_dense_blocks<T>(A.Up, opt);
          build_cd_streams(A.Up, Ur.ptr);
        });
  } catch (const Error &) {
    return false;
  }
  H.Lr = std::move(Lr), H.Ur = std::move(Ur);
  H.Ls = std::move(A.Ls), H.Us = std::move(A.Us);
  H.Lp = std::move(A.Lp), H.Up = std::move(A.Up);
  H.top = std::move(A.top);
  H.top_n = A.top_n, H.top_bandL = A.top_bandL, H.top_bandU = A.top_bandU;
  H.Ltinv_elems = lt, H.Utinv_elems = ut;
  return true;
}

}  // namespace hifamd
