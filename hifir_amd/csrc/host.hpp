// host.hpp -- host side of the MI355X-native HIFIR apply path: the imported hierarchy, CCS -> CSR
// conversion in processing order, level scheduling of the triangular factors, and the dense
// last-level factorization (QR with column pivoting, explicit Q^H and R^{-1} for the MFMA GEMMs).
//
// Nothing here runs per apply: it is the one-time "ship the factored hierarchy to HBM" step of
// BASELINE.json's north_star.  Reference data contract: hif::Prec, src/hif/alg/Prec.hpp:82-334.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <thread>
#include <mutex>
#include <cstring>
#include <functional>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

namespace hifamd {

// Host-side parallel loop on plain std::thread (dynamic chunks off an atomic counter).  Deliberately NOT
// OpenMP: the library lives in processes that already carry an OpenMP runtime of their own (PyTorch / MKL).
// Round 1 saw an intermittent host memory corruption with the OpenMP build inside such a process (two
// overwritten row pointers right after the dense factorization); the sanitizer harness of round 2
// (tests/cpp/import_san_test.cpp: ASan/UBSan/TSan, also with this loop on OpenMP and with a second OpenMP
// runtime loaded) did NOT reproduce it, so a second runtime as the cause is a suspicion, not a finding --
// what guards against a recurrence is check_level_invariants at finalize (import.hpp), not this choice.
// f(begin, end) handles [begin, end).
template <class F>
inline void parallel_for(int64_t n, int64_t chunk, F f) {
  if (n <= 0) return;
  if (chunk < 1) chunk = 1;
#ifdef HIFAMD_TEST_OPENMP  // tests/cpp/import_san_test.cpp only: the OpenMP loops the library had before it went thread-only
  {
    const int64_t nchunks_ = (n + chunk - 1) / chunk;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t c = 0; c < nchunks_; ++c) f(c * chunk, std::min(n, (c + 1) * chunk));
    return;
  }
#endif
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 4;
  if (const char *e = std::getenv("HIFIR_AMD_THREADS")) hw = (unsigned)std::max(1, std::atoi(e));
  const int64_t nchunks = (n + chunk - 1) / chunk;
  const unsigned nt = (unsigned)std::min<int64_t>(hw, nchunks);
  if (nt <= 1) {
    f((int64_t)0, n);
    return;
  }
  std::atomic<int64_t> next(0);
  auto worker = [&]() {
    for (;;) {
      const int64_t c = next.fetch_add(1);
      if (c >= nchunks) break;
      f(c * chunk, std::min(n, (c + 1) * chunk));
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(nt - 1);
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(worker);
  worker();
  for (auto &th : pool) th.join();
}


typedef std::complex<double> zdouble;

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

inline double abs_(double x) { return std::fabs(x); }
inline double abs_(const zdouble &x) { return std::abs(x); }
inline double conj_(double x) { return x; }
inline zdouble conj_(const zdouble &x) { return std::conj(x); }
inline double abs1_(double x) { return std::fabs(x); }
inline double abs1_(const zdouble &x) { return std::fabs(x.real()) + std::fabs(x.imag()); }  // BLAS i?amax measure
inline double real_(double x) { return x; }
inline double real_(const zdouble &x) { return x.real(); }

// Column-major (ld = nrows) -> strip-major layout of the dense MFMA kernels: 16-row strips, each stored
// as [k][16 rows] with ldk >= ncols columns per strip (rows beyond nrows and columns beyond ncols are zero).
template <class T>
std::vector<T> to_strip_layout(const T *colmajor, int64_t nrows, int64_t ncols, int64_t ldk = 0) {
  if (ldk < ncols) ldk = ncols;
  const int64_t ns = (nrows + 15) / 16;
  std::vector<T> out((size_t)(ns * ldk * 16), T(0));
  for (int64_t k = 0; k < ncols; ++k)
    for (int64_t r = 0; r < nrows; ++r) out[(size_t)(((r >> 4) * ldk + k) * 16 + (r & 15))] = colmajor[r + k * nrows];
  return out;
}
inline int64_t round_up32(int64_t x) { return (x + 31) & ~(int64_t)31; }

// Operand of the f64 MFMA kernels from a column-major matrix: real -> one strip-major plane; complex ->
// TWO real strip-major planes [re | im] (gfx950 has no complex MFMA: a complex product runs as the two
// real products A_re * X and A_im * X on the interleaved real view of X, recombined by k_zcombine).
inline int64_t plane_elems(int64_t nrows, int64_t ldk) { return ((nrows + 15) / 16) * 16 * ldk; }
inline std::vector<double> mfma_operand(const double *colmajor, int64_t nrows, int64_t ncols, int64_t ldk = 0) {
  return to_strip_layout(colmajor, nrows, ncols, ldk);
}
inline std::vector<double> mfma_operand(const zdouble *colmajor, int64_t nrows, int64_t ncols, int64_t ldk = 0) {
  if (ldk < ncols) ldk = ncols;
  std::vector<double> re((size_t)(nrows * ncols)), im((size_t)(nrows * ncols));
  for (int64_t k = 0; k < nrows * ncols; ++k) {
    re[(size_t)k] = colmajor[k].real();
    im[(size_t)k] = colmajor[k].imag();
  }
  std::vector<double> out = to_strip_layout(re.data(), nrows, ncols, ldk);
  const std::vector<double> pi = to_strip_layout(im.data(), nrows, ncols, ldk);
  out.insert(out.end(), pi.begin(), pi.end());
  return out;
}

// ---------------------------------------------------------------------------------------------
// matrices
// ---------------------------------------------------------------------------------------------
template <class T>
struct Ccs {  // as the reference stores L_B, U_B, E, F (Prec::mat_type = ccs_type, Prec.hpp:86,90)
  int64_t nrows = 0, ncols = 0;
  std::vector<int64_t> colptr;
  std::vector<int32_t> rowind;
  std::vector<T> vals;
  int64_t nnz() const { return colptr.empty() ? 0 : colptr.back(); }
};

template <class T>
struct Csr {  // device-side form: row gather.  ptr is indexed by SLOT (processing order), see rowid
  int64_t nrows = 0, ncols = 0;
  std::vector<int32_t> ptr;    // nrows+1
  std::vector<int32_t> col;    // nnz, per-row order == the reference's per-row accumulation order
  std::vector<T> val;
  std::vector<int32_t> rowid;  // slot -> row id (identity for E, F, A)
};

// CCS -> CSR keeping, inside every row, the order in which the reference's column sweep touches
// that row: ascending column for the forward sweeps (solve_as_strict_lower CompressedStorage.hpp:2268,
// multiply_nt_low :2079), descending column for solve_as_strict_upper (:2357).
template <class T>
Csr<T> ccs_to_csr(const Ccs<T> &A, bool descending_cols) {
  Csr<T> B;
  B.nrows = A.nrows;
  B.ncols = A.ncols;
  const int64_t nz = A.nnz();
  if (nz > (int64_t)std::numeric_limits<int32_t>::max())
    throw Error(4, "matrix has more than 2^31-1 nonzeros: int32 device row pointers overflow");
  B.ptr.assign((size_t)A.nrows + 1, 0);
  for (int64_t k = 0; k < nz; ++k) {
    const int32_t r = A.rowind[(size_t)k];
    if (r < 0 || r >= A.nrows) throw Error(2, "row index out of range in imported CCS matrix");
    ++B.ptr[(size_t)r + 1];
  }
  for (int64_t i = 0; i < A.nrows; ++i) B.ptr[(size_t)i + 1] += B.ptr[(size_t)i];
  B.col.resize((size_t)nz);
  B.val.resize((size_t)nz);
  std::vector<int32_t> fill(B.ptr.begin(), B.ptr.end() - 1);
  auto put = [&](int64_t j) {
    for (int64_t k = A.colptr[(size_t)j]; k < A.colptr[(size_t)j + 1]; ++k) {
      const int32_t r = A.rowind[(size_t)k];
      const int32_t pos = fill[(size_t)r]++;
      B.col[(size_t)pos] = (int32_t)j;
      B.val[(size_t)pos] = A.vals[(size_t)k];
    }
  };
  if (!descending_cols)
    for (int64_t j = 0; j < A.ncols; ++j) put(j);
  else
    for (int64_t j = A.ncols - 1; j >= 0; --j) put(j);
  B.rowid.resize((size_t)A.nrows);
  for (int64_t i = 0; i < A.nrows; ++i) B.rowid[(size_t)i] = (int32_t)i;
  return B;
}

// CSR of A^H straight from the CCS of A: column j of A IS row j of A^T, so no conversion is needed;
// values are conjugated, per-row order is the stored (ascending row index) order, which is the order
// in which the reference's transposed kernels accumulate (CompressedStorage.hpp:2161, :2307, :2399).
template <class T>
Csr<T> adjoint_rows(const Ccs<T> &A) {
  Csr<T> B;
  B.nrows = A.ncols;
  B.ncols = A.nrows;
  if (A.nnz() > (int64_t)std::numeric_limits<int32_t>::max())
    throw Error(4, "matrix has more than 2^31-1 nonzeros: int32 device row pointers overflow");
  B.ptr.resize((size_t)A.ncols + 1);
  for (int64_t j = 0; j <= A.ncols; ++j) B.ptr[(size_t)j] = A.colptr.empty() ? 0 : (int32_t)A.colptr[(size_t)j];
  B.col = A.rowind;
  B.val.resize(A.vals.size());
  for (size_t k = 0; k < A.vals.size(); ++k) B.val[k] = conj_(A.vals[k]);
  B.rowid.resize((size_t)A.ncols);
  for (int64_t i = 0; i < A.ncols; ++i) B.rowid[(size_t)i] = (int32_t)i;
  return B;
}

// ---------------------------------------------------------------------------------------------
// level scheduling
// ---------------------------------------------------------------------------------------------
struct Schedule {
  std::vector<int32_t> order;   // slot -> row id, wavefront by wavefront
  std::vector<int64_t> wf_ptr;  // wavefront w owns slots [wf_ptr[w], wf_ptr[w+1])
  int64_t nwf() const { return (int64_t)wf_ptr.size() - 1; }
};

// Wavefront (dependency depth) of every row of a strict triangle in CSR: a row can be solved as
// soon as every row it references is done.  lower: references j < i, rows visited ascending;
// upper: references j > i, rows visited descending.  Rows inside a wavefront keep that visiting
// order, so slot order is a valid sequential order too.
template <class T>
Schedule level_schedule(const Csr<T> &A, bool lower) {
  const int64_t m = A.nrows;
  std::vector<int32_t> depth((size_t)m, 0);
  int32_t maxd = -1;
  for (int64_t ii = 0; ii < m; ++ii) {
    const int64_t i = lower ? ii : m - 1 - ii;
    int32_t d = 0;
    for (int32_t k = A.ptr[(size_t)i]; k < A.ptr[(size_t)i + 1]; ++k) {
      const int32_t j = A.col[(size_t)k];
      if (lower ? (j >= i) : (j <= i)) throw Error(3, "triangular factor is not strict");
      d = std::max(d, depth[(size_t)j] + 1);
    }
    depth[(size_t)i] = d;
    maxd = std::max(maxd, d);
  }
  Schedule S;
  S.wf_ptr.assign((size_t)(maxd + 2), 0);
  for (int64_t i = 0; i < m; ++i) ++S.wf_ptr[(size_t)depth[(size_t)i] + 1];
  for (size_t w = 0; w + 1 < S.wf_ptr.size(); ++w) S.wf_ptr[w + 1] += S.wf_ptr[w];
  S.order.resize((size_t)m);
  std::vector<int64_t> fill(S.wf_ptr.begin(), S.wf_ptr.end() - 1);
  for (int64_t ii = 0; ii < m; ++ii) {
    const int64_t i = lower ? ii : m - 1 - ii;
    S.order[(size_t)fill[(size_t)depth[(size_t)i]]++] = (int32_t)i;
  }
  return S;
}

// ---------------------------------------------------------------------------------------------
// band plan: how a triangle is actually executed on the GPU
// ---------------------------------------------------------------------------------------------
// A launch per wavefront is latency-bound (a dependent launch costs ~5 us on MI355X whatever its
// size) and the reference's hierarchies have hundreds of wavefronts per triangle.  Instead:
//   * consecutive wavefronts are grouped into BANDS;
//   * inside a band, the rows split into connected COMPONENTS of the dependency graph restricted to
//     the band (the band is a slice of the elimination forest: many small independent subtrees);
//   * components are bin-packed onto workgroups; a workgroup owns a contiguous slot range, walks
//     its rows in dependency order and hands results over through LDS flags -- no communication
//     between workgroups, one launch per band.
// A band that is essentially one component (the thin tail of a triangle) runs on one workgroup,
// after an exact PREFIX pass on the whole chip has folded in every leading nonzero that refers to
// rows before the band.  Accumulation order inside a row never changes.
struct BandPlan {
  std::vector<int32_t> order;         // slot -> row id (bands, then workgroups, then depth order)
  std::vector<int32_t> grp_slot_ptr;  // group g (rows of one depth inside one workgroup) owns these slots
  std::vector<int32_t> wg_grp_ptr;    // workgroup w owns groups [wg_grp_ptr[w], wg_grp_ptr[w+1])
  std::vector<int32_t> band_wg_ptr;   // band b owns workgroups [band_wg_ptr[b], band_wg_ptr[b+1])
  std::vector<uint8_t> band_prefix;   // band b is preceded by the exact prefix pass
  std::vector<uint8_t> band_fused;    // the prefix of band b over sources older than band b-1 rides on band b-1's launch
  // Thin bands solved block by block with explicit inverses of the diagonal blocks (see below):
  std::vector<uint8_t> band_dense;    // band b uses the block-dense scheme
  std::vector<int32_t> band_blk_ptr;  // band b owns blocks [band_blk_ptr[b], band_blk_ptr[b+1])
  std::vector<int32_t> blk_slot0, blk_slot1;  // block -> slot range
  std::vector<int64_t> blk_inv_off;   // block -> offset of its (rows x rows, column-major) inverse
  std::vector<int32_t> srcslot;       // per nonzero (slot order): slot of the source row
  std::vector<int32_t> split;         // per slot: where the band kernel starts in the row
  // Component-dense bands (plan_bands_cd): every group of such a band is one dependency component (<= cd_rows rows,
  // LDS-resident), solved as  x_c = Tinv_c (rhs_c - old-source sums)  with the explicit inverse of its own triangle:
  std::vector<uint8_t> band_cd;       // band b is a component-dense band
  std::vector<uint8_t> band_old;      // band b has nonzeros that refer to earlier bands (else no prefix work exists)
  std::vector<int32_t> csplit;        // per slot: first nonzero whose source lies in the row's own component
  std::vector<int64_t> grp_inv_off;   // per group: offset of the component's inverse (cd bands; -1 elsewhere)
  // what a component's workgroup gathers itself -- the nonzeros [split, csplit) of its rows (sources in the previous
  // band; everything older when no launch carried it) -- as ONE packed stream per component, so that a wave keeps its
  // gathers in flight across row boundaries: entry e = (nonzero index mid_k[e] of the slot-ordered CSR, local row)
  std::vector<int32_t> mid_k;
  std::vector<uint8_t> mid_lrow;
  // per group kCdDescWords int32 words (cd bands; zeros elsewhere): s0, nb, mid0, nmid, inv_off (2 words), then the
  // 16 waves' row chunks wrow[17] (uint8) and their entry offsets wmid[17] (uint16, relative to mid0); sparse plans:
  // words 20..24 = own0, nown, orp0 (into own_rptr), lvl0 (into own_lvl), nlvl
  std::vector<int32_t> cd_desc;
  // cd_sparse: the components' OWN nonzeros are not inverted but solved in LDS, depth level by depth level (triangles
  // with ~2 nonzeros per row: level 0 of a PDE hierarchy -- an inverse would cost 20x the matrix).  Per component the
  // own nonzeros [csplit, end) of its rows as one packed stream (nonzero index own_k, local source row own_lsrc),
  // row offsets own_rptr (nb + 1 per component) and the depth levels as row boundaries own_lvl (nlvl + 1 per component).
  bool cd_sparse = false;
  std::vector<int32_t> own_k;
  std::vector<uint8_t> own_lsrc, own_lvl;
  std::vector<uint16_t> own_rptr;
  int64_t nbands() const { return (int64_t)band_wg_ptr.size() - 1; }
  int64_t nwg() const { return (int64_t)wg_grp_ptr.size() - 1; }
};

constexpr int kCdDescWords = 28;  // 112 bytes per component descriptor (BandPlan::cd_desc)
constexpr int kCdOwnCap = 4096;   // own nonzeros of a sparse component that k_band_cd keeps in LDS

struct BandOptions {
  int64_t thin_rows = 96;    // a wavefront this narrow belongs to a thin run
  int64_t band_depth = 32;   // at most this many wavefronts per band outside thin runs
  int64_t max_comp_weight = 1024;  // ... and a band stops growing before one component gets heavier
                                   // than this many nonzeros (a component is served by ONE compute unit)
  int64_t max_wg_rows = 16384;  // LDS flags per workgroup
  int64_t max_wgs = 1024;    // workgroups per band
  int64_t dense_block = 2048; // rows per diagonal block of a block-dense thin band (0 = scheme off)
  bool fuse = true;            // carried prefixes (band_fused), see finish_band_plan
  bool fuse_reorder = true;    // ... with the rows' nonzeros reordered so that the prefix covers ALL old sources (fast mode)
  int64_t fuse_max_wgs = 512;  // band b-1 must leave compute units idle (96 / 192 / 256 / 512 / 1024: 9.23 / 8.87 / 8.86 / 8.81 / 9.28 ms)
  int64_t dense_min_rows = 96;   // thin bands shorter than this stay on the sequential workgroup
  int64_t cd_rows = 192;         // component-dense bands: rows per component (LDS-resident; 0 = scheme off)
  double cd_min_row_nnz = 4.0;   // ... only for triangles with at least this many nonzeros per row on average
  int64_t cd_fuse_max_wgs = 0;   // a component band with more workgroups than this is not carried by its predecessor (0: always)
  int64_t cd_max_nnz = 0;        // ... and nonzeros per component (0 = no limit): spreads heavy rows over more units
  int64_t cd_sparse_max_depth = 64;  // ... only for triangles with at most this many wavefronts
  int64_t cd_sparse_min_rows = 4096;  // ... only for triangles of at least this many rows
  int64_t cd_sparse_rows = 192;  // sparse-own plans (BandPlan::cd_sparse) for the triangles below cd_min_row_nnz: rows per
                                 // component (0 = those triangles keep the depth-cut flag bands)
  int64_t top_max = 4096;        // combined top operator (choose_top): at most this many rows (0 = off)
  int64_t top_few_wgs = 96;      // ... made of the last bands of L's plan that have at most this many workgroups
  double dense_max_growth = 1e4; // ... and so do bands whose block inverses grow beyond this
};

// A: strict triangle in CSR, natural row order (before any permutation); depth from level_schedule.
template <class T>
BandPlan plan_bands(const Csr<T> &A, const Schedule &S, bool lower, const BandOptions &opt) {
  BandPlan P;
  const int64_t m = A.nrows, nwf = S.nwf();
  std::vector<int32_t> depth((size_t)m), pos((size_t)m);  // pos: rank of the row in level-schedule order
  for (int64_t w = 0; w < nwf; ++w)
    for (int64_t s = S.wf_ptr[(size_t)w]; s < S.wf_ptr[(size_t)w + 1]; ++s) {
      depth[(size_t)S.order[(size_t)s]] = (int32_t)w;
      pos[(size_t)S.order[(size_t)s]] = (int32_t)s;
    }
  P.order.reserve((size_t)m);
  P.grp_slot_ptr.push_back(0);
  P.wg_grp_ptr.push_back(0);
  P.band_wg_ptr.push_back(0);
  std::vector<int32_t> parent((size_t)m), comp_of((size_t)m);
  auto find = [&](int32_t x) {
    while (parent[(size_t)x] != x) {
      parent[(size_t)x] = parent[(size_t)parent[(size_t)x]];
      x = parent[(size_t)x];
    }
    return x;
  };
  int64_t w0 = 0;
  while (w0 < nwf) {
    // ---- band extent
    const bool thin = S.wf_ptr[(size_t)w0 + 1] - S.wf_ptr[(size_t)w0] <= opt.thin_rows;
    int64_t w1 = w0 + 1;
    if (thin) {
      while (w1 < nwf && S.wf_ptr[(size_t)w1 + 1] - S.wf_ptr[(size_t)w1] <= opt.thin_rows &&
             S.wf_ptr[(size_t)w1 + 1] - S.wf_ptr[(size_t)w0] <= opt.max_wg_rows)
        ++w1;
    } else {
      // grow the band one wavefront at a time (incremental union-find) until its heaviest
      // component would exceed what one compute unit should carry
      int64_t totw = 0, maxw = 0;
      auto add_wave = [&](int64_t w) {
        for (int64_t s = S.wf_ptr[(size_t)w]; s < S.wf_ptr[(size_t)w + 1]; ++s) {
          const int32_t i = S.order[(size_t)s];
          parent[(size_t)i] = i;
          comp_of[(size_t)i] = (A.ptr[(size_t)i + 1] - A.ptr[(size_t)i]) + 2;  // weight, valid at roots
          totw += comp_of[(size_t)i];
          maxw = std::max<int64_t>(maxw, comp_of[(size_t)i]);
        }
        if (w == w0) return;
        for (int64_t s = S.wf_ptr[(size_t)w]; s < S.wf_ptr[(size_t)w + 1]; ++s) {
          const int32_t i = S.order[(size_t)s];
          for (int32_t k = A.ptr[(size_t)i]; k < A.ptr[(size_t)i + 1]; ++k) {
            const int32_t j = A.col[(size_t)k];
            if (depth[(size_t)j] >= w0) {
              const int32_t a = find(i), b = find(j);
              if (a != b) {
                parent[(size_t)a] = b;
                comp_of[(size_t)b] += comp_of[(size_t)a];
                maxw = std::max<int64_t>(maxw, comp_of[(size_t)b]);
              }
            }
          }
        }
      };
      add_wave(w0);
      while (w1 < nwf && w1 - w0 < opt.band_depth && S.wf_ptr[(size_t)w1 + 1] - S.wf_ptr[(size_t)w1] > opt.thin_rows) {
        add_wave(w1);
        if (maxw > std::max<int64_t>(opt.max_comp_weight, totw / 512)) break;  // w1 stays out
        ++w1;
      }
    }
    std::vector<std::vector<int32_t>> wg_rows;
    bool prefix = false;
    bool thin_single = false;
    if (thin && w1 - w0 >= 2 && opt.dense_block > 0 &&
        S.wf_ptr[(size_t)w1] - S.wf_ptr[(size_t)w0] >= opt.dense_min_rows) {
      // latency-bound tail: keep it as ONE slot range in depth order (candidate for the block-dense scheme)
      wg_rows.assign(1, std::vector<int32_t>());
      for (int64_t s = S.wf_ptr[(size_t)w0]; s < S.wf_ptr[(size_t)w1]; ++s) wg_rows[0].push_back(S.order[(size_t)s]);
      prefix = true;
      thin_single = true;
    }
    for (; !thin_single;) {  // (re)try with a shallower band if a component does not fit one workgroup
      const int64_t s0 = S.wf_ptr[(size_t)w0], s1 = S.wf_ptr[(size_t)w1];
      // ---- connected components of the dependency graph restricted to the band
      for (int64_t s = s0; s < s1; ++s) parent[(size_t)S.order[(size_t)s]] = S.order[(size_t)s];
      for (int64_t s = s0; s < s1; ++s) {
        const int32_t i = S.order[(size_t)s];
        for (int32_t k = A.ptr[(size_t)i]; k < A.ptr[(size_t)i + 1]; ++k) {
          const int32_t j = A.col[(size_t)k];
          if (depth[(size_t)j] >= w0) {
            const int32_t a = find(i), b = find(j);
            if (a != b) parent[(size_t)a] = b;
          }
        }
      }
      std::vector<int32_t> roots;
      std::vector<int64_t> weight, crows;
      for (int64_t s = s0; s < s1; ++s) {
        const int32_t i = S.order[(size_t)s], r = find(i);
        if (r == i) {
          comp_of[(size_t)i] = (int32_t)roots.size();
          roots.push_back(i);
          weight.push_back(0);
          crows.push_back(0);
        }
      }
      for (int64_t s = s0; s < s1; ++s) {
        const int32_t i = S.order[(size_t)s], c = comp_of[(size_t)find(i)];
        comp_of[(size_t)i] = c;  // (roots keep their own id: comp_of[root] was set above)
        weight[(size_t)c] += (A.ptr[(size_t)i + 1] - A.ptr[(size_t)i]) + 2;
        crows[(size_t)c] += 1;
      }
      const int64_t ncomp = (int64_t)roots.size();
      int64_t maxrows = 0, totw = 0, maxw = 0;
      for (int64_t c = 0; c < ncomp; ++c) {
        maxrows = std::max(maxrows, crows[(size_t)c]);
        totw += weight[(size_t)c];
        maxw = std::max(maxw, weight[(size_t)c]);
      }
      if (maxrows > opt.max_wg_rows && w1 - w0 > 1) {
        w1 = w0 + std::max<int64_t>(1, (w1 - w0) / 2);
        continue;
      }
      if (maxrows > opt.max_wg_rows)
        throw Error(4, "a single wavefront component exceeds the per-workgroup row limit");
      // ---- bin-pack components onto workgroups (largest first onto the lightest workgroup)
      int64_t nw = std::min<int64_t>(ncomp, std::max<int64_t>(1, std::min<int64_t>(opt.max_wgs, (s1 - s0) / 16)));
      nw = std::max(nw, (s1 - s0 + opt.max_wg_rows - 1) / opt.max_wg_rows);
      std::vector<int64_t> idx((size_t)ncomp);
      for (int64_t c = 0; c < ncomp; ++c) idx[(size_t)c] = c;
      std::sort(idx.begin(), idx.end(), [&](int64_t x, int64_t y) { return weight[(size_t)x] > weight[(size_t)y]; });
      std::vector<int64_t> lrows((size_t)nw, 0);
      std::vector<int32_t> wg_of((size_t)ncomp);
      // min-heap on load
      std::vector<std::pair<int64_t, int64_t>> heap;
      for (int64_t g = 0; g < nw; ++g) heap.push_back({0, g});
      auto cmp = [](const std::pair<int64_t, int64_t> &x, const std::pair<int64_t, int64_t> &y) { return x > y; };
      std::make_heap(heap.begin(), heap.end(), cmp);
      for (int64_t q = 0; q < ncomp; ++q) {
        const int64_t c = idx[(size_t)q];
        // lightest workgroup that still has room for the component's rows
        std::vector<std::pair<int64_t, int64_t>> parked;
        for (;;) {
          if (heap.empty()) {  // nobody has room left: open another workgroup
            lrows.push_back(0);
            heap.push_back({0, nw++});
          }
          std::pop_heap(heap.begin(), heap.end(), cmp);
          auto top = heap.back();
          heap.pop_back();
          if (lrows[(size_t)top.second] + crows[(size_t)c] <= opt.max_wg_rows) {
            wg_of[(size_t)c] = (int32_t)top.second;
            lrows[(size_t)top.second] += crows[(size_t)c];
            top.first += weight[(size_t)c];
            heap.push_back(top);
            std::push_heap(heap.begin(), heap.end(), cmp);
            break;
          }
          parked.push_back(top);
        }
        for (auto &pk : parked) {
          heap.push_back(pk);
          std::push_heap(heap.begin(), heap.end(), cmp);
        }
      }
      wg_rows.assign((size_t)nw, std::vector<int32_t>());
      for (int64_t s = s0; s < s1; ++s) {  // level-schedule order: depth-major, so each list is depth-sorted
        const int32_t i = S.order[(size_t)s];
        wg_rows[(size_t)wg_of[(size_t)comp_of[(size_t)i]]].push_back(i);
      }
      // one workgroup carrying most of a sizeable band: let the whole chip do the independent prefixes first
      prefix = (maxw * 4 > totw) && totw > 4096;
      break;
    }
    for (auto &rows : wg_rows) {
      if (rows.empty()) continue;
      int32_t cur = -1;
      for (int32_t i : rows) {
        if (depth[(size_t)i] != cur) {
          if (cur != -1) P.grp_slot_ptr.push_back((int32_t)P.order.size());
          cur = depth[(size_t)i];
        }
        P.order.push_back(i);
      }
      P.grp_slot_ptr.push_back((int32_t)P.order.size());
      P.wg_grp_ptr.push_back((int32_t)P.grp_slot_ptr.size() - 1);
    }
    P.band_wg_ptr.push_back((int32_t)P.wg_grp_ptr.size() - 1);
    P.band_prefix.push_back(prefix ? 1 : 0);
    P.band_dense.push_back(thin_single ? 1 : 0);
    w0 = w1;
  }
  (void)lower;
  (void)pos;
  return P;
}

// ---------------------------------------------------------------------------------------------
// Component-dense plan.  The depth-cut bands above stop growing as soon as ONE component gets heavy, so the upper
// part of a triangle -- where the elimination forest has merged into a few hundred subtrees -- costs dozens of
// short launches, and its top (few rows per wavefront, hundreds of wavefronts) a chain of 2,048-row blocks.
// Here the triangle is cut along its SUBTREES instead: rows are visited in dependency order and joined to the
// components of the rows they need (union-find) as long as a component stays within cd_rows rows; a row whose
// component would grow beyond that is deferred to the next pass, and with it everything that needs it.  One pass =
// one band = one launch; its components are independent of each other, each is served by ONE workgroup that keeps
// the component's right-hand sides in LDS and applies the explicit inverse of the component's own (small) triangle
// on the matrix cores -- no dependent step, no polling inside a launch, any depth.  Measured on the reference's
// 1M-row hierarchies: 6-7 passes per triangle instead of 20-40 depth bands, and what is left over for the chain
// of 2,048-row blocks shrinks from ~9,000 rows to a few hundred.
// An upper triangle fans OUT from the root of the forest: it is planned on its reversed dependency graph (row j
// "needs" every row i that reads it -- the CCS columns), which is an in-forest again, and the passes then run in
// reverse order: a row's sources lie in the same component or in a pass that ran earlier.
// ---------------------------------------------------------------------------------------------
template <class T>
BandPlan plan_bands_cd(const Csr<T> &A, const Schedule &S, bool lower, const BandOptions &opt_,
                       const std::vector<uint8_t> *top = nullptr, bool sparse = false) {
  BandOptions opt = opt_;
  if (sparse) {  // own nonzeros stay sparse and LDS-resident: cap them, and the rows, per component
    opt.cd_rows = opt_.cd_sparse_rows;
    opt.cd_max_nnz = kCdOwnCap - 64;
  }
  BandPlan P;
  P.cd_sparse = sparse;
  const int64_t m = A.nrows;
  std::vector<int32_t> depth((size_t)m);
  for (int64_t w = 0; w < S.nwf(); ++w)
    for (int64_t s = S.wf_ptr[(size_t)w]; s < S.wf_ptr[(size_t)w + 1]; ++s) depth[(size_t)S.order[(size_t)s]] = (int32_t)w;
  // "needs" lists of the planning graph: the rows themselves (lower), or the transposed pattern (upper)
  std::vector<int32_t> tptr, tcol;
  const int32_t *nptr = A.ptr.data(), *ncol = A.col.data();
  if (!lower) {
    tptr.assign((size_t)m + 1, 0);
    for (size_t k = 0; k < A.col.size(); ++k) ++tptr[(size_t)A.col[k] + 1];
    for (int64_t i = 0; i < m; ++i) tptr[(size_t)i + 1] += tptr[(size_t)i];
    tcol.resize(A.col.size());
    std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
    for (int64_t i = 0; i < m; ++i)
      for (int32_t k = A.ptr[(size_t)i]; k < A.ptr[(size_t)i + 1]; ++k) tcol[(size_t)fill[(size_t)A.col[(size_t)k]]++] = (int32_t)i;
    nptr = tptr.data(), ncol = tcol.data();
  }
  // visiting order: any topological order of the planning graph.  lower: the level schedule; upper (reversed
  // graph, needs point to SMALLER rows): ascending row number
  std::vector<int32_t> remaining((size_t)m), next;
  if (lower)
    remaining = S.order;
  else
    for (int64_t i = 0; i < m; ++i) remaining[(size_t)i] = (int32_t)i;
  // a forced TOP set (choose_top): those rows are never planned into passes; they form the rest band, whatever its
  // size, and the passes run until every other row is placed.  The set is closed (nothing outside it needs... see
  // choose_top), so no planned row ever waits for a top row in the planning graph.
  std::vector<int32_t> top_rows;
  if (top) {
    std::vector<int32_t> keep;
    keep.reserve(remaining.size());
    for (int32_t i : remaining) ((*top)[(size_t)i] ? top_rows : keep).push_back(i);
    remaining.swap(keep);
  }
  std::vector<int32_t> pass((size_t)m, -1), parent((size_t)m), crows((size_t)m);
  std::vector<int64_t> cnnz((size_t)m);
  std::vector<uint8_t> deferred((size_t)m, 0);
  auto find = [&](int32_t x) {
    while (parent[(size_t)x] != x) {
      parent[(size_t)x] = parent[(size_t)parent[(size_t)x]];
      x = parent[(size_t)x];
    }
    return x;
  };
  const int64_t rest_rows = opt.dense_block > 0 ? opt.dense_block : 0;  // what one block of the dense chain takes
  std::vector<std::vector<int32_t>> pass_rows;                           // rows of every pass, visiting order
  std::vector<int32_t> roots;
  int32_t np = 0;
  while (!remaining.empty() && (top || (int64_t)remaining.size() > rest_rows)) {
    next.clear();
    std::vector<int32_t> mine;
    for (int32_t i : remaining) {
      parent[(size_t)i] = i;
      crows[(size_t)i] = 1;
      bool dfr = false;
      roots.clear();
      int64_t r = 1, wsum = A.ptr[(size_t)i + 1] - A.ptr[(size_t)i];
      for (int32_t k = nptr[i]; k < nptr[i + 1] && !dfr; ++k) {
        const int32_t j = ncol[k];
        if (pass[(size_t)j] >= 0 && pass[(size_t)j] < np) continue;  // finished in an earlier pass
        if (top && (*top)[(size_t)j]) throw Error(4, "internal error: the top set is not closed");
        if (deferred[(size_t)j]) {
          dfr = true;
          break;
        }
        const int32_t rt = find(j);
        if (std::find(roots.begin(), roots.end(), rt) == roots.end()) {
          roots.push_back(rt);
          r += crows[(size_t)rt];
          wsum += cnnz[(size_t)rt];
        }
      }
      if (dfr || r > opt.cd_rows || (opt.cd_max_nnz > 0 && wsum > opt.cd_max_nnz && !roots.empty())) {
        deferred[(size_t)i] = 1;
        next.push_back(i);
        continue;
      }
      for (int32_t rt : roots) parent[(size_t)rt] = i;
      crows[(size_t)i] = (int32_t)r;
      cnnz[(size_t)i] = wsum;
      pass[(size_t)i] = np;
      mine.push_back(i);
    }
    for (int32_t i : next) deferred[(size_t)i] = 0;
    if (mine.empty()) throw Error(4, "internal error: component-dense planner made no progress");
    // a pass that filled fewer than four components means the rest is (nearly) one chain: a launch per <= cd_rows rows
    // would lose against the dense block chain's launch pair per dense_block rows
    const bool degenerate = (int64_t)mine.size() < 4 * opt.cd_rows;
    pass_rows.push_back(std::move(mine));
    ++np;
    remaining.swap(next);
    if (degenerate && opt.dense_block > 0 && !top) break;
  }
  if (!remaining.empty() && opt.dense_block <= 0) {  // no dense chain available (cannot happen: cd needs dense_block > 0)
    throw Error(4, "internal error: component-dense plan without the dense block chain");
  }
  if (top) remaining = top_rows;  // (the rest band IS the top set)
  // ---- emit bands in EXECUTION order: lower: pass 0, 1, ..., rest;  upper: rest, pass np-1, ..., 0
  P.grp_slot_ptr.push_back(0);
  P.wg_grp_ptr.push_back(0);
  P.band_wg_ptr.push_back(0);
  std::vector<int32_t> comp_id((size_t)m, -1);
  auto emit_rest = [&]() {
    if (remaining.empty()) return;
    // slot ranges in dependency (depth) order, cut into blocks of dense_block rows by plan_dense_blocks; one band
    // per max_wg_rows rows (a band whose inverses grow too much falls back to ONE flag workgroup)
    std::vector<int32_t> rows(remaining);
    std::stable_sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) {
      return depth[(size_t)a] != depth[(size_t)b] ? depth[(size_t)a] < depth[(size_t)b] : (lower ? a < b : a > b);
    });
    for (size_t at = 0; at < rows.size(); at += (size_t)opt.max_wg_rows) {
      const size_t end = std::min(rows.size(), at + (size_t)opt.max_wg_rows);
      int32_t cur = -1;
      for (size_t q = at; q < end; ++q) {
        const int32_t i = rows[q];
        if (depth[(size_t)i] != cur) {
          if (cur != -1) P.grp_slot_ptr.push_back((int32_t)P.order.size());
          cur = depth[(size_t)i];
        }
        P.order.push_back(i);
      }
      P.grp_slot_ptr.push_back((int32_t)P.order.size());
      P.wg_grp_ptr.push_back((int32_t)P.grp_slot_ptr.size() - 1);
      P.band_wg_ptr.push_back((int32_t)P.wg_grp_ptr.size() - 1);
      P.band_prefix.push_back(1);
      P.band_dense.push_back(1);
      P.band_cd.push_back(0);
    }
  };
  auto emit_pass = [&](int32_t p) {
    const std::vector<int32_t> &rows = pass_rows[(size_t)p];
    // components of the pass: union-find roots were the LAST row joined; recompute ids by a second sweep
    for (int32_t i : rows) parent[(size_t)i] = i;
    for (int32_t i : rows)
      for (int32_t k = nptr[i]; k < nptr[i + 1]; ++k) {
        const int32_t j = ncol[k];
        if (pass[(size_t)j] != p) continue;
        const int32_t a = find(i), b = find(j);
        if (a != b) parent[(size_t)a] = b;
      }
    std::vector<std::vector<int32_t>> crow;
    for (int32_t i : rows) {
      const int32_t rt = find(i);
      if (comp_id[(size_t)rt] < 0) {
        comp_id[(size_t)rt] = (int32_t)crow.size();
        crow.emplace_back();
      }
      crow[(size_t)comp_id[(size_t)rt]].push_back(i);
    }
    for (int32_t i : rows) comp_id[(size_t)find(i)] = -1;  // (reset for the next pass)
    // tiny components (a few rows: leaves, rows without in-pass sources) are packed together: a union of independent
    // components is a valid block (block-diagonal inverse), and one 32-row block costs less than 32 one-row blocks
    {
      std::vector<std::vector<int32_t>> packed;
      std::vector<int32_t> bag;
      // (sparse plans bag everything up to half a component; a pass with few rows gets smaller bags, so that its
      //  gathers are spread over the whole chip instead of a few dozen workgroups)
      const size_t small = sparse ? (size_t)opt.cd_rows / 2 : 8;
      const size_t bagcap = sparse ? std::max<size_t>(16, std::min<size_t>((size_t)opt.cd_rows, rows.size() / 256)) : 32;
      int64_t bagw = 0;
      for (auto &r : crow) {
        if (r.size() > small) {
          packed.push_back(std::move(r));
          continue;
        }
        int64_t rw = 0;
        for (int32_t i : r) rw += A.ptr[(size_t)i + 1] - A.ptr[(size_t)i];
        if (!bag.empty() && (bag.size() + r.size() > bagcap || (sparse && bagw + rw > opt.cd_max_nnz))) {
          bagw = 0;
          packed.push_back(std::move(bag));
          bag.clear();
        }
        bagw += rw;
        bag.insert(bag.end(), r.begin(), r.end());
      }
      if (!bag.empty()) packed.push_back(std::move(bag));
      crow.swap(packed);
    }
    const int64_t nc = (int64_t)crow.size();
    std::vector<int64_t> cw((size_t)nc, 0);
    for (int64_t c = 0; c < nc; ++c)
      for (int32_t i : crow[(size_t)c]) cw[(size_t)c] += (A.ptr[(size_t)i + 1] - A.ptr[(size_t)i]) + 2 + (int64_t)crow[(size_t)c].size() / 2;
    // heaviest components first: the hardware hands workgroups out in order, so the long ones start first
    std::vector<int64_t> idx((size_t)nc);
    for (int64_t c = 0; c < nc; ++c) idx[(size_t)c] = c;
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return cw[(size_t)a] > cw[(size_t)b]; });
    // workgroups: one component each up to max_wgs; beyond that the lightest are chained onto shared workgroups
    const int64_t nw = std::min<int64_t>(nc, std::max<int64_t>(1, 8 * opt.max_wgs));  // (the hardware balances them)
    std::vector<std::vector<int64_t>> wgc((size_t)nw);
    {
      std::vector<std::pair<int64_t, int64_t>> heap;
      for (int64_t g = 0; g < nw; ++g) heap.push_back({0, g});
      auto cmp = [](const std::pair<int64_t, int64_t> &x, const std::pair<int64_t, int64_t> &y) { return x > y; };
      std::make_heap(heap.begin(), heap.end(), cmp);
      for (int64_t q = 0; q < nc; ++q) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        auto top = heap.back();
        heap.pop_back();
        wgc[(size_t)top.second].push_back(idx[(size_t)q]);
        top.first += cw[(size_t)idx[(size_t)q]];
        heap.push_back(top);
        std::push_heap(heap.begin(), heap.end(), cmp);
      }
    }
    for (int64_t g = 0; g < nw; ++g) {
      for (int64_t c : wgc[(size_t)g]) {
        std::vector<int32_t> &r = crow[(size_t)c];
        // inside a component: the triangle's own dependency order (its inverse is then lower triangular)
        std::stable_sort(r.begin(), r.end(), [&](int32_t a, int32_t b) {
          return depth[(size_t)a] != depth[(size_t)b] ? depth[(size_t)a] < depth[(size_t)b] : (lower ? a < b : a > b);
        });
        for (int32_t i : r) P.order.push_back(i);
        P.grp_slot_ptr.push_back((int32_t)P.order.size());
      }
      P.wg_grp_ptr.push_back((int32_t)P.grp_slot_ptr.size() - 1);
    }
    P.band_wg_ptr.push_back((int32_t)P.wg_grp_ptr.size() - 1);
    P.band_prefix.push_back(0);
    P.band_dense.push_back(0);
    P.band_cd.push_back(1);
  };
  P.order.reserve((size_t)m);
  if (lower) {
    for (int32_t p = 0; p < np; ++p) emit_pass(p);
    emit_rest();
  } else {
    emit_rest();
    for (int32_t p = np - 1; p >= 0; --p) emit_pass(p);
  }
  if ((int64_t)P.order.size() != m) throw Error(4, "internal error: component-dense plan lost rows");
  return P;
}

// After the CSR has been permuted into the band plan's slot order: source slots and split points.
// Block-dense thin bands.  A thin band is a short, deep, single-component triangular system: hundreds
// of dependent steps of a few rows each, i.e. pure latency (~2 us per step through LDS flags).  Cut
// into diagonal blocks of <= dense_block rows it becomes, per block,
//   (1) an UPDATE with everything outside the block (rows before the band and earlier blocks; all
//       finished, so it is an ordinary parallel sparse kernel), and
//   (2) ONE dense product with the explicit inverse of the block's unit triangle on the f64 matrix
//       cores -- no dependent steps at all inside a block.
// The inverses are formed here (forward substitution on the identity, O(rows * nnz_in_block)).
// This changes the order of summation inside thin bands (tolerance-level differences, 1e-15 relative
// on the hierarchies measured); a band whose inverse entries grow beyond dense_max_growth keeps the
// sequential scheme, which is backward stable for any factor.
inline void put_operand(double *base, int64_t, int64_t idx, double v) { base[idx] = v; }
inline void put_operand(double *base, int64_t plane, int64_t idx, const zdouble &v) {
  base[idx] = v.real();
  base[plane + idx] = v.imag();
}

// Packed gather streams and descriptors of the component-dense bands (BandPlan::mid_k / mid_lrow / cd_desc); call it
// after plan_dense_blocks (the descriptors carry the inverse offsets).  The
// rows of a component are dealt to the workgroup's 16 waves as CONTIGUOUS chunks balanced by (entries + rows); a wave
// initialises its rows' right-hand sides in LDS and subtracts its entries, no other wave touches those rows in phase 1.
// chunk boundaries of a component's packed stream (descriptor words 6.. wrow[17] uint8, 11.. wmid[17] uint16): wave w
// takes rows while the cost so far (an entry = 1, a row = 2) stays within its share; at most 64 rows per wave (its row
// ids sit one per lane)
inline void cd_balance_chunks(int32_t *dsc, const std::vector<int32_t> &rowstart, int32_t nb, int32_t nmid) {
  constexpr int NW = 16;
  uint8_t *wrow = reinterpret_cast<uint8_t *>(&dsc[6]);
  uint16_t *wmid = reinterpret_cast<uint16_t *>(&dsc[11]);
  const double total = (double)nmid + 2.0 * nb;
  int32_t r = 0;
  for (int w = 0; w < NW; ++w) {
    wrow[w] = (uint8_t)r;
    wmid[w] = (uint16_t)rowstart[(size_t)r];
    const double goal = (w == NW - 1) ? total + 1.0 : total * (w + 1) / NW;
    const int32_t first = r;
    while (r < nb && r - first < 64 && (double)rowstart[(size_t)r + 1] + 2.0 * (r + 1) <= goal + 1e-9) ++r;
  }
  if (r < nb) {  // (a chunk ran into the 64-row limit: spread the rows evenly instead -- nb <= 255: <= 16 rows per wave)
    for (int w = 0; w < NW; ++w) {
      const int32_t rr = (int32_t)((int64_t)nb * w / NW);
      wrow[w] = (uint8_t)rr, wmid[w] = (uint16_t)rowstart[(size_t)rr];
    }
  }
  wrow[NW] = (uint8_t)nb;
  wmid[NW] = (uint16_t)nmid;
}

// S5 fused into the second L solve of a level (prec_solve.hpp:397-399 + :406): the streams of build_cd_streams with the
// row's F entries appended -- a gather from the child's solution v[m + k] IS an "older source" of the row, and
// rhs = s b[p] - sum L_old x - sum F v has the form the kernel already computes.  Sources are row numbers relative to
// the L solve's vector w; the arena keeps v right behind w, so v[m + k] is row src_row0 + k with src_row0 = n + m.
// Possible when every row of the triangle is first touched by the main phase of a component band (no prefix pass, no
// carried prefix, no dense block band, no combined top): cd_f_fusable.
inline bool cd_f_fusable(const BandPlan &P) {
  if (P.band_cd.empty() || P.nbands() == 0) return false;
  for (int64_t b = 0; b < P.nbands(); ++b) {
    if (!P.band_cd[(size_t)b] || P.band_dense[(size_t)b] || P.band_prefix[(size_t)b]) return false;
    if (!P.band_fused.empty() && P.band_fused[(size_t)b]) return false;
  }
  return true;
}
template <class T>
struct CdFusedStreams {
  std::vector<int32_t> desc, col;
  std::vector<T> val;
  std::vector<uint8_t> lrow;
};
template <class T>
bool build_cd_streams_fused(const BandPlan &P, const Csr<T> &Lr, const Csr<T> &Fr, int64_t src_row0, CdFusedStreams<T> &S) {
  S.desc = P.cd_desc;
  S.col.clear(), S.val.clear(), S.lrow.clear();
  if (src_row0 + Fr.ncols > (int64_t)std::numeric_limits<int32_t>::max() / 64) return false;
  for (int64_t b = 0; b < P.nbands(); ++b)
    for (int32_t g = P.band_wg_ptr[(size_t)b]; g < P.band_wg_ptr[(size_t)b + 1]; ++g)
      for (int32_t c = P.wg_grp_ptr[(size_t)g]; c < P.wg_grp_ptr[(size_t)g + 1]; ++c) {
        const int32_t s0 = P.grp_slot_ptr[(size_t)c], nb = P.grp_slot_ptr[(size_t)c + 1] - s0;
        const int64_t mid0 = (int64_t)S.col.size();
        std::vector<int32_t> rowstart((size_t)nb + 1, 0);
        for (int32_t r = 0; r < nb; ++r) {
          for (int32_t k = P.split[(size_t)(s0 + r)]; k < P.csplit[(size_t)(s0 + r)]; ++k) {
            S.col.push_back(Lr.col[(size_t)k]);
            S.val.push_back(Lr.val[(size_t)k]);
            S.lrow.push_back((uint8_t)r);
          }
          const int32_t i = Lr.rowid[(size_t)(s0 + r)];
          for (int32_t k = Fr.ptr[(size_t)i]; k < Fr.ptr[(size_t)i + 1]; ++k) {
            S.col.push_back((int32_t)(src_row0 + Fr.col[(size_t)k]));
            S.val.push_back(Fr.val[(size_t)k]);
            S.lrow.push_back((uint8_t)r);
          }
          rowstart[(size_t)r + 1] = (int32_t)((int64_t)S.col.size() - mid0);
        }
        const int32_t nmid = rowstart[(size_t)nb];
        if (nmid > 65535 || mid0 + nmid > (int64_t)std::numeric_limits<int32_t>::max()) return false;  // (keep S5 separate)
        int32_t *dsc = &S.desc[(size_t)c * kCdDescWords];
        dsc[2] = (int32_t)mid0, dsc[3] = nmid;
        cd_balance_chunks(dsc, rowstart, nb, nmid);
      }
  return true;
}

inline void build_cd_streams(BandPlan &P, const std::vector<int32_t> &aptr /* row pointer of the slot-ordered CSR */) {
  const size_t ngrp = P.grp_slot_ptr.size() - 1;
  P.cd_desc.assign(ngrp * (size_t)kCdDescWords, 0);
  P.mid_k.clear();
  P.mid_lrow.clear();
  P.own_k.clear();
  P.own_lsrc.clear();
  P.own_lvl.clear();
  P.own_rptr.clear();
  if (P.band_cd.empty()) return;
  for (int64_t b = 0; b < P.nbands(); ++b) {
    if (!P.band_cd[(size_t)b]) continue;
    for (int32_t g = P.band_wg_ptr[(size_t)b]; g < P.band_wg_ptr[(size_t)b + 1]; ++g)
      for (int32_t c = P.wg_grp_ptr[(size_t)g]; c < P.wg_grp_ptr[(size_t)g + 1]; ++c) {
        const int32_t s0 = P.grp_slot_ptr[(size_t)c], nb = P.grp_slot_ptr[(size_t)c + 1] - s0;
        if (nb > 255) throw Error(4, "internal error: component larger than 255 rows");
        const int64_t mid0 = (int64_t)P.mid_k.size();
        std::vector<int32_t> rowstart((size_t)nb + 1, 0);
        for (int32_t r = 0; r < nb; ++r) {
          for (int32_t k = P.split[(size_t)(s0 + r)]; k < P.csplit[(size_t)(s0 + r)]; ++k) {
            P.mid_k.push_back(k);
            P.mid_lrow.push_back((uint8_t)r);
          }
          rowstart[(size_t)r + 1] = (int32_t)((int64_t)P.mid_k.size() - mid0);
        }
        const int32_t nmid = rowstart[(size_t)nb];
        if (nmid > 65535 || mid0 + nmid > (int64_t)std::numeric_limits<int32_t>::max())
          throw Error(4, "internal error: component gather stream too long");
        int32_t *dsc = &P.cd_desc[(size_t)c * kCdDescWords];
        dsc[0] = s0, dsc[1] = nb, dsc[2] = (int32_t)mid0, dsc[3] = nmid;
        std::memcpy(&dsc[4], &P.grp_inv_off[(size_t)c], 8);
        cd_balance_chunks(dsc, rowstart, nb, nmid);
        if (!P.cd_sparse) continue;
        // own stream: the rows' nonzeros inside the component, row offsets, and the depth levels (a row's level =
        // 1 + the deepest own source; rows are in dependency order, so one sweep suffices -- and levels must be
        // contiguous row ranges: the rows were sorted by the triangle's depth, which dominates the own depth only
        // up to ties, so the level of a row is forced to be at least its predecessor's)
        const int64_t own0 = (int64_t)P.own_k.size(), orp0 = (int64_t)P.own_rptr.size(), lvl0 = (int64_t)P.own_lvl.size();
        std::vector<int32_t> lev((size_t)nb, 0);
        int32_t cur = 0;
        P.own_lvl.push_back(0);
        for (int32_t r = 0; r < nb; ++r) {
          P.own_rptr.push_back((uint16_t)((int64_t)P.own_k.size() - own0));
          int32_t need = 0;
          for (int32_t k = P.csplit[(size_t)(s0 + r)]; k < aptr[(size_t)(s0 + r) + 1]; ++k) {
            const int32_t q = P.srcslot[(size_t)k] - s0;
            P.own_k.push_back(k);
            P.own_lsrc.push_back((uint8_t)q);
            need = std::max(need, lev[(size_t)q] + 1);
          }
          if (need > cur) {  // a new level starts at this row
            cur = need;
            P.own_lvl.push_back((uint8_t)r);
          }
          lev[(size_t)r] = cur;
        }
        P.own_rptr.push_back((uint16_t)((int64_t)P.own_k.size() - own0));
        P.own_lvl.push_back((uint8_t)nb);
        const int64_t nown = (int64_t)P.own_k.size() - own0, nlvl = (int64_t)P.own_lvl.size() - lvl0 - 1;
        if (nown > kCdOwnCap) throw Error(4, "internal error: sparse component with too many own nonzeros");
        dsc[20] = (int32_t)own0, dsc[21] = (int32_t)nown, dsc[22] = (int32_t)orp0, dsc[23] = (int32_t)lvl0, dsc[24] = (int32_t)nlvl;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Sparse-own U bands with streamed SINKS (round 4; kernel k_band_us).  Level 0 of a PDE hierarchy: two thirds of a
// component's rows are sinks of the U solve INSIDE the component -- no row of the component reads their result (their column
// of the component's own block is empty).  Such a row needs no LDS slot: the component's other ("black") rows are solved
// level by level in LDS as k_band_cd<false, sparse> does, then every sink is streamed -- right-hand side in, its own entries
// gathered from the black rows in LDS, result out.  LDS per component: black rows x 512 B instead of all rows x 512 B (33 KB
// instead of 98 KB on the reference's 1M-row hierarchy) => two workgroups per compute unit, which the band kernels of level 0
// are bound by.  Per row the arithmetic and its order are k_band_cd's (right-hand side, outside entries in CSR
// order, own entries in CSR order): the same bits.
// The component's rows are renumbered black first (relative order kept: dependency order), sinks behind:
//   desc   : per component 8 words: s0 (first slot, = the plan's), nb, nbk (black rows), own0, orp0, lvl0, nlvl, 0
//   rowid  : per NEW slot the row id;  oslot: per new slot the plan's slot (for arrays kept in plan order)
//   mptr / mcol / mval : per new slot its outside entries [split, csplit) (source ROW, value), CSR order
//   own_val / own_src  : per new slot its own entries (value, NEW local index of the source: always a black row)
//   own_rptr : per component nb + 1 offsets relative to own0;  own_lvl : per component nlvl + 1 black-row boundaries
// band_ok[b]: every workgroup of band b owns exactly one component and the band touches its rows first;
// band_nbk[b] / band_own[b]: most black rows / own entries of one component of the band (LDS sizing).
// ---------------------------------------------------------------------------------------------
template <class T>
struct UsPlan {
  std::vector<int32_t> desc, rowid, oslot, mptr, mcol;
  std::vector<T> mval, own_val;
  std::vector<uint8_t> own_src, own_lvl;
  std::vector<uint16_t> own_rptr;
  std::vector<uint8_t> band_ok;
  std::vector<int32_t> band_nbk, band_own, band_c0;
  int64_t sinks = 0, blacks = 0;
  bool any = false;
};
constexpr int kUsDescWords = 8;

template <class T>
void build_us_plan(const BandPlan &P, const Csr<T> &A /* slot order */, UsPlan<T> &U) {
  U = UsPlan<T>();
  const int64_t nb_bands = P.nbands();
  if (!P.cd_sparse || P.band_cd.empty() || nb_bands <= 0) return;
  const int64_t m = A.nrows;
  const size_t ngrp = P.grp_slot_ptr.size() - 1;
  U.desc.assign(ngrp * (size_t)kUsDescWords, 0);
  U.rowid.assign((size_t)m, 0);
  U.oslot.assign((size_t)m, 0);
  for (int64_t sl = 0; sl < m; ++sl) U.rowid[(size_t)sl] = A.rowid[(size_t)sl], U.oslot[(size_t)sl] = (int32_t)sl;
  U.mptr.assign((size_t)m + 1, 0);
  U.band_ok.assign((size_t)nb_bands, 0);
  U.band_nbk.assign((size_t)nb_bands, 0);
  U.band_own.assign((size_t)nb_bands, 0);
  U.band_c0.assign((size_t)nb_bands, 0);
  std::vector<int32_t> mcnt((size_t)m, 0);
  // pass 1: the new order of every qualifying component
  std::vector<int32_t> newpos((size_t)m, -1);  // plan slot -> new slot (identity outside qualifying bands)
  for (int64_t sl = 0; sl < m; ++sl) newpos[(size_t)sl] = (int32_t)sl;
  for (int64_t b = 0; b < nb_bands; ++b) {
    if (!P.band_cd[(size_t)b]) continue;
    if (!P.band_dense.empty() && P.band_dense[(size_t)b]) continue;
    if (!P.band_prefix.empty() && P.band_prefix[(size_t)b]) continue;
    if (!P.band_fused.empty() && P.band_fused[(size_t)b]) continue;
    const int32_t g0 = P.band_wg_ptr[(size_t)b], g1 = P.band_wg_ptr[(size_t)b + 1];
    const int32_t c0 = P.wg_grp_ptr[(size_t)g0], c1 = P.wg_grp_ptr[(size_t)g1];
    if (c1 - c0 != g1 - g0 || c1 <= c0) continue;  // (bags: the component kernel of the plan keeps them)
    bool ok = true;
    for (int32_t c = c0; c < c1 && ok; ++c) ok = P.grp_slot_ptr[(size_t)c + 1] - P.grp_slot_ptr[(size_t)c] <= 255;
    if (!ok) continue;
    U.band_ok[(size_t)b] = 1;
    U.band_c0[(size_t)b] = c0;
    U.any = true;
    for (int32_t c = c0; c < c1; ++c) {
      const int32_t s0 = P.grp_slot_ptr[(size_t)c], nb = P.grp_slot_ptr[(size_t)c + 1] - s0;
      std::vector<uint8_t> read((size_t)nb, 0);
      for (int32_t r = 0; r < nb; ++r)
        for (int32_t k = P.csplit[(size_t)(s0 + r)]; k < A.ptr[(size_t)(s0 + r) + 1]; ++k) read[(size_t)(P.srcslot[(size_t)k] - s0)] = 1;
      int32_t nbk = 0;
      for (int32_t r = 0; r < nb; ++r)
        if (read[(size_t)r]) newpos[(size_t)(s0 + r)] = s0 + nbk++;
      int32_t q = nbk;
      for (int32_t r = 0; r < nb; ++r)
        if (!read[(size_t)r]) newpos[(size_t)(s0 + r)] = s0 + q++;
      int32_t *dsc = &U.desc[(size_t)c * kUsDescWords];
      dsc[0] = s0, dsc[1] = nb, dsc[2] = nbk;
      U.band_nbk[(size_t)b] = std::max(U.band_nbk[(size_t)b], nbk);
      U.blacks += nbk;
      U.sinks += nb - nbk;
    }
  }
  if (!U.any) return;
  for (int64_t sl = 0; sl < m; ++sl) {
    const int32_t ns = newpos[(size_t)sl];
    U.rowid[(size_t)ns] = A.rowid[(size_t)sl];
    U.oslot[(size_t)ns] = (int32_t)sl;
  }
  // pass 2: outside entries per new slot
  for (int64_t ns = 0; ns < m; ++ns) {
    const int32_t sl = U.oslot[(size_t)ns];
    U.mptr[(size_t)ns + 1] = U.mptr[(size_t)ns] + (P.csplit[(size_t)sl] - P.split[(size_t)sl]);
  }
  U.mcol.resize((size_t)U.mptr[(size_t)m]);
  U.mval.resize((size_t)U.mptr[(size_t)m]);
  for (int64_t ns = 0; ns < m; ++ns) {
    const int32_t sl = U.oslot[(size_t)ns];
    int32_t o = U.mptr[(size_t)ns];
    for (int32_t k = P.split[(size_t)sl]; k < P.csplit[(size_t)sl]; ++k, ++o) U.mcol[(size_t)o] = A.col[(size_t)k], U.mval[(size_t)o] = A.val[(size_t)k];
  }
  // pass 3: own entries, row offsets and black levels per component (new order)
  for (int64_t b = 0; b < nb_bands; ++b) {
    if (!U.band_ok[(size_t)b]) continue;
    const int32_t g0 = P.band_wg_ptr[(size_t)b], g1 = P.band_wg_ptr[(size_t)b + 1];
    for (int32_t c = P.wg_grp_ptr[(size_t)g0]; c < P.wg_grp_ptr[(size_t)g1]; ++c) {
      int32_t *dsc = &U.desc[(size_t)c * kUsDescWords];
      const int32_t s0 = dsc[0], nb = dsc[1], nbk = dsc[2];
      const int64_t own0 = (int64_t)U.own_val.size(), orp0 = (int64_t)U.own_rptr.size(), lvl0 = (int64_t)U.own_lvl.size();
      std::vector<int32_t> lev((size_t)nb, 0);
      int32_t cur = 0;
      U.own_lvl.push_back(0);
      for (int32_t r = 0; r < nb; ++r) {  // r: NEW local index
        const int32_t sl = U.oslot[(size_t)(s0 + r)];
        U.own_rptr.push_back((uint16_t)((int64_t)U.own_val.size() - own0));
        int32_t need = 0;
        for (int32_t k = P.csplit[(size_t)sl]; k < A.ptr[(size_t)sl + 1]; ++k) {
          const int32_t q = newpos[(size_t)P.srcslot[(size_t)k]] - s0;  // (a black row: somebody reads it)
          U.own_val.push_back(A.val[(size_t)k]);
          U.own_src.push_back((uint8_t)q);
          need = std::max(need, lev[(size_t)q] + 1);
        }
        if (r < nbk) {
          if (need > cur) {
            cur = need;
            U.own_lvl.push_back((uint8_t)r);
          }
          lev[(size_t)r] = cur;
        }
      }
      U.own_rptr.push_back((uint16_t)((int64_t)U.own_val.size() - own0));
      U.own_lvl.push_back((uint8_t)nbk);
      const int64_t nown = (int64_t)U.own_val.size() - own0, nlvl = (int64_t)U.own_lvl.size() - lvl0 - 1;
      if (nown > 65535) {  // (uint16 offsets: such a component keeps the plan's kernel)
        U.band_ok[(size_t)b] = 0;
      }
      dsc[3] = (int32_t)own0, dsc[4] = (int32_t)orp0, dsc[5] = (int32_t)lvl0, dsc[6] = (int32_t)nlvl;
      U.band_own[(size_t)b] = std::max<int32_t>(U.band_own[(size_t)b], (int32_t)nown);
    }
  }
}

// what k_band_us indexes with, re-derived from the arrays themselves (finalize calls it before the upload: no index out of
// range reaches the kernel, whatever built the plan)
template <class T>
void check_us_plan(const BandPlan &P, const Csr<T> &A, const UsPlan<T> &U) {
  auto fail = [](const char *why) { throw Error(4, std::string("internal error: streamed-sink plan: ") + why); };
  const int64_t m = A.nrows;
  if ((int64_t)U.rowid.size() != m || (int64_t)U.oslot.size() != m || (int64_t)U.mptr.size() != m + 1) fail("array lengths");
  std::vector<uint8_t> seen((size_t)m, 0);
  for (int64_t ns = 0; ns < m; ++ns) {
    const int32_t sl = U.oslot[(size_t)ns];
    if (sl < 0 || sl >= m || seen[(size_t)sl]) fail("slot permutation");
    seen[(size_t)sl] = 1;
    if (U.rowid[(size_t)ns] != A.rowid[(size_t)sl]) fail("row ids");
    if (U.mptr[(size_t)ns + 1] < U.mptr[(size_t)ns]) fail("outside-entry offsets");
  }
  if ((size_t)U.mptr[(size_t)m] != U.mcol.size() || U.mcol.size() != U.mval.size()) fail("outside-entry arrays");
  for (int32_t c : U.mcol)
    if (c < 0 || c >= A.ncols) fail("outside-entry source");
  for (int64_t b = 0; b < P.nbands(); ++b) {
    if (!U.band_ok[(size_t)b]) continue;
    const int32_t g0 = P.band_wg_ptr[(size_t)b], g1 = P.band_wg_ptr[(size_t)b + 1];
    if (U.band_c0[(size_t)b] != P.wg_grp_ptr[(size_t)g0] || P.wg_grp_ptr[(size_t)g1] - P.wg_grp_ptr[(size_t)g0] != g1 - g0) fail("one component per workgroup");
    for (int32_t c = P.wg_grp_ptr[(size_t)g0]; c < P.wg_grp_ptr[(size_t)g1]; ++c) {
      const int32_t *dsc = &U.desc[(size_t)c * kUsDescWords];
      const int32_t s0 = dsc[0], nb = dsc[1], nbk = dsc[2], own0 = dsc[3], orp0 = dsc[4], lvl0 = dsc[5], nlvl = dsc[6];
      if (s0 != P.grp_slot_ptr[(size_t)c] || nb != P.grp_slot_ptr[(size_t)c + 1] - s0 || nb < 1 || nb > 255 || nbk < 0 || nbk > nb) fail("descriptor");
      if (nbk > U.band_nbk[(size_t)b]) fail("black rows exceed the band's LDS size");
      if (orp0 < 0 || (size_t)orp0 + (size_t)nb + 1 > U.own_rptr.size() || lvl0 < 0 || nlvl < 1 || (size_t)lvl0 + (size_t)nlvl + 1 > U.own_lvl.size()) fail("offset arrays");
      const int32_t nown = U.own_rptr[(size_t)orp0 + (size_t)nb];
      if (own0 < 0 || (size_t)own0 + (size_t)nown > U.own_val.size() || U.own_val.size() != U.own_src.size() || nown > U.band_own[(size_t)b]) fail("own entries");
      if (U.own_lvl[(size_t)lvl0] != 0 || U.own_lvl[(size_t)lvl0 + (size_t)nlvl] != nbk) fail("level ends");
      std::vector<int32_t> lev((size_t)nb, -1);
      for (int32_t lv = 0; lv < nlvl; ++lv) {
        const int32_t a = U.own_lvl[(size_t)lvl0 + (size_t)lv], e = U.own_lvl[(size_t)lvl0 + (size_t)lv + 1];
        if (e < a) fail("levels not monotone");
        for (int32_t r = a; r < e; ++r) lev[(size_t)r] = lv;
      }
      for (int32_t r = 0; r < nb; ++r) {
        const int32_t eb = U.own_rptr[(size_t)orp0 + (size_t)r], ee = U.own_rptr[(size_t)orp0 + (size_t)r + 1];
        if (ee < eb || ee > nown) fail("own offsets");
        for (int32_t e = eb; e < ee; ++e) {
          const int32_t q = U.own_src[(size_t)own0 + (size_t)e];
          if (q >= nbk) fail("an own entry reads a sink");
          if (r < nbk && !(lev[(size_t)q] < lev[(size_t)r])) fail("a black row reads a row of its own or a later level");
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Tile form of what a dense-own component band walks (round 4; kernel k_band_ct).  The entries [split, csplit) of a
// component's rows -- sources finished by earlier launches -- are a small sparse matrix (rows x distinct sources) whose
// rows share most of their sources (measured on the reference's 1M-row hierarchies: a 16-row strip of a component
// meets 11-13 entries per 4 distinct sources).  Per 16-row strip the distinct sources, oldest first, are cut into groups
// of four; a group is ONE v_mfma_f64_16x16x4 step per 16-column slice: A = the 16 x 4 coefficient tile (zeros where a row
// lacks the source), B = the four gathered source rows.  Every distinct source of a strip is gathered once, and the
// entry walk of k_band_cd (a dozen vector / scalar instructions per entry and wave) becomes one matrix instruction per
// dozen entries.  The summation order differs from the row loops' (tolerance-level; fast mode only).
//   sptr  : per component S + 1 tile offsets (S = strips of the component), block of component c at desc word 20
//   src   : 4 source ROW numbers per tile (a short last group repeats its last source with zero coefficients)
//   coef  : 64 per tile, element (k << 4) | r = coefficient of row 16 s + r for source k
//   desc words 22, 23: four uint16 masks -- the strips each of the kernel's four waves owns (balanced by tiles)
// ---------------------------------------------------------------------------------------------
struct CtTiles {
  std::vector<int32_t> sptr, src;
  std::vector<double> coef;
  std::vector<int32_t> desc;  // copy of BandPlan::cd_desc with words 20, 22, 23 filled for the dense-own components
  int64_t ntiles = 0;
  int32_t max_wave_tiles = 0;  // most tiles one wave of one component walks (the serial part of a band)
  std::vector<int32_t> band_wave_tiles;  // ... per band
};

// (complex data, round 4: a tile's coefficients are TWO real tiles [re 64 | im 64] -- four real matrix instructions per tile)
template <class T>
void build_ct_tiles(const BandPlan &P, const Csr<T> &A /* slot order */, CtTiles &Tl) {
  constexpr int64_t kTile = sizeof(T) == sizeof(double) ? 64 : 128;  // doubles per tile
  Tl = CtTiles();
  if (P.band_cd.empty() || P.cd_sparse || P.cd_desc.empty()) return;
  Tl.desc = P.cd_desc;
  const int64_t ngrp = (int64_t)P.grp_slot_ptr.size() - 1;
  // which components take part, their strips, and (pass 1) the tiles of every strip
  std::vector<uint8_t> is_cd((size_t)ngrp, 0);
  Tl.band_wave_tiles.assign((size_t)P.nbands(), 0);
  std::vector<int32_t> band_of((size_t)ngrp, -1);
  for (int64_t b = 0; b < P.nbands(); ++b) {
    if (!P.band_cd[(size_t)b]) continue;
    for (int32_t c = P.wg_grp_ptr[(size_t)P.band_wg_ptr[(size_t)b]]; c < P.wg_grp_ptr[(size_t)P.band_wg_ptr[(size_t)b + 1]]; ++c)
      is_cd[(size_t)c] = 1, band_of[(size_t)c] = (int32_t)b;
  }
  std::vector<int64_t> sp0((size_t)ngrp + 1, 0);  // first sptr entry of every component
  for (int64_t c = 0; c < ngrp; ++c) {
    const int32_t nb = P.grp_slot_ptr[(size_t)c + 1] - P.grp_slot_ptr[(size_t)c];
    sp0[(size_t)c + 1] = sp0[(size_t)c] + (is_cd[(size_t)c] ? (nb + 15) / 16 + 1 : 0);
  }
  if (sp0[(size_t)ngrp] > (int64_t)std::numeric_limits<int32_t>::max()) throw Error(4, "component tile index overflows int32");
  std::vector<int32_t> cnt((size_t)sp0[(size_t)ngrp], 0);  // tiles per strip (slot s + 1 of the component's block)
  auto strip_sources = [&](int32_t r0, int32_t r1, std::vector<int32_t> &u) {  // distinct source slots, ascending
    u.clear();
    for (int32_t s = r0; s < r1; ++s)
      for (int32_t k = P.split[(size_t)s]; k < P.csplit[(size_t)s]; ++k) u.push_back(P.srcslot[(size_t)k]);
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
  };
  parallel_for(ngrp, 64, [&](int64_t c0, int64_t c1) {
    std::vector<int32_t> u;
    for (int64_t c = c0; c < c1; ++c) {
      if (!is_cd[(size_t)c]) continue;
      const int32_t a = P.grp_slot_ptr[(size_t)c], e = P.grp_slot_ptr[(size_t)c + 1];
      for (int32_t r0 = a, s = 0; r0 < e; r0 += 16, ++s) {
        strip_sources(r0, std::min(e, r0 + 16), u);
        cnt[(size_t)(sp0[(size_t)c] + s + 1)] = (int32_t)((u.size() + 3) / 4);
      }
    }
  });
  // absolute tile offsets
  Tl.sptr.assign(cnt.size(), 0);
  int64_t total = 0;
  for (int64_t c = 0; c < ngrp; ++c) {
    if (!is_cd[(size_t)c]) continue;
    const int64_t b0 = sp0[(size_t)c], b1 = sp0[(size_t)c + 1];
    for (int64_t q = b0; q < b1; ++q) {
      total += cnt[(size_t)q];  // (cnt[b0] == 0: the block's first entry is the component's first tile)
      if (total > (int64_t)std::numeric_limits<int32_t>::max() / 64) throw Error(4, "component tile index overflows int32");
      Tl.sptr[(size_t)q] = (int32_t)total;
    }
  }
  Tl.ntiles = total;
  Tl.src.assign((size_t)(4 * total), 0);
  Tl.coef.assign((size_t)(kTile * total), 0.0);
  std::vector<int32_t> wave_max((size_t)ngrp, 0);
  parallel_for(ngrp, 64, [&](int64_t c0, int64_t c1) {
    std::vector<int32_t> u;
    for (int64_t c = c0; c < c1; ++c) {
      if (!is_cd[(size_t)c]) continue;
      const int32_t a = P.grp_slot_ptr[(size_t)c], e = P.grp_slot_ptr[(size_t)c + 1];
      const int32_t S = (e - a + 15) / 16;
      for (int32_t r0 = a, s = 0; r0 < e; r0 += 16, ++s) {
        const int32_t r1 = std::min(e, r0 + 16);
        strip_sources(r0, r1, u);
        const int64_t t0 = Tl.sptr[(size_t)(sp0[(size_t)c] + s)];
        const int64_t nt = ((int64_t)u.size() + 3) / 4;
        for (int64_t q = 0; q < 4 * nt; ++q)  // source ROW numbers (what the kernel gathers from), short group padded
          Tl.src[(size_t)(4 * t0 + q)] = A.rowid[(size_t)u[(size_t)std::min<int64_t>(q, (int64_t)u.size() - 1)]];
        for (int32_t sl = r0; sl < r1; ++sl)
          for (int32_t k = P.split[(size_t)sl]; k < P.csplit[(size_t)sl]; ++k) {
            const int64_t q = std::lower_bound(u.begin(), u.end(), P.srcslot[(size_t)k]) - u.begin();
            const size_t at = (size_t)(kTile * (t0 + q / 4) + ((q & 3) << 4) + (sl - r0));
            Tl.coef[at] += real_(A.val[(size_t)k]);
            if (kTile == 128) Tl.coef[at + 64] += reinterpret_cast<const double *>(&A.val[(size_t)k])[sizeof(T) / sizeof(double) - 1];
          }
      }
      // strips to waves: heaviest first onto the lightest of the four waves (a strip costs its tiles + its 4 row loads)
      int32_t *dsc = &Tl.desc[(size_t)c * kCdDescWords];
      dsc[20] = (int32_t)sp0[(size_t)c];
      uint16_t mask[4] = {0, 0, 0, 0};
      int64_t load[4] = {0, 0, 0, 0}, tiles_w[4] = {0, 0, 0, 0};
      std::vector<int32_t> order((size_t)S);
      for (int32_t s = 0; s < S; ++s) order[(size_t)s] = s;
      auto tiles_of = [&](int32_t s) { return (int64_t)cnt[(size_t)(sp0[(size_t)c] + s + 1)]; };
      std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return tiles_of(x) > tiles_of(y); });
      for (int32_t s : order) {
        int wmin = 0;
        for (int w_ = 1; w_ < 4; ++w_)
          if (load[w_] < load[wmin]) wmin = w_;
        mask[wmin] = (uint16_t)(mask[wmin] | (1u << s));
        load[wmin] += tiles_of(s) + 2;
        tiles_w[wmin] += tiles_of(s);
      }
      dsc[22] = (int32_t)((uint32_t)mask[0] | ((uint32_t)mask[1] << 16));
      dsc[23] = (int32_t)((uint32_t)mask[2] | ((uint32_t)mask[3] << 16));
      wave_max[(size_t)c] = (int32_t)std::max(std::max(tiles_w[0], tiles_w[1]), std::max(tiles_w[2], tiles_w[3]));
    }
  });
  for (int64_t c = 0; c < ngrp; ++c) {
    if (!is_cd[(size_t)c]) continue;
    Tl.max_wave_tiles = std::max(Tl.max_wave_tiles, wave_max[(size_t)c]);
    int32_t &bw = Tl.band_wave_tiles[(size_t)band_of[(size_t)c]];
    bw = std::max(bw, wave_max[(size_t)c]);
  }
}

// Pass 1 (import time, cheap): cut every candidate band into blocks and lay their MFMA operands out back to
// back; returns the total number of doubles.  blk_inv_off counts doubles; complex blocks hold two planes.
template <class T>
int64_t plan_dense_blocks(BandPlan &P, const BandOptions &opt) {
  const int64_t nplanes = sizeof(T) == sizeof(double) ? 1 : 2;
  P.band_blk_ptr.assign(1, 0);
  P.blk_slot0.clear();
  P.blk_slot1.clear();
  P.blk_inv_off.clear();
  int64_t total = 0;
  P.grp_inv_off.assign(P.grp_slot_ptr.size() - 1, -1);
  for (int64_t b = 0; b < P.nbands(); ++b) {
    if (!P.band_cd.empty() && P.band_cd[(size_t)b] && P.cd_sparse) {
      // sparse-own components: solved in LDS, nothing to invert
    } else if (!P.band_cd.empty() && P.band_cd[(size_t)b]) {  // component-dense band: one inverse per component (group)
      for (int32_t g = P.band_wg_ptr[(size_t)b]; g < P.band_wg_ptr[(size_t)b + 1]; ++g)
        for (int32_t c = P.wg_grp_ptr[(size_t)g]; c < P.wg_grp_ptr[(size_t)g + 1]; ++c) {
          const int32_t r0 = P.grp_slot_ptr[(size_t)c], r1 = P.grp_slot_ptr[(size_t)c + 1];
          P.blk_slot0.push_back(r0);
          P.blk_slot1.push_back(r1);
          P.blk_inv_off.push_back(total);
          P.grp_inv_off[(size_t)c] = total;
          total += nplanes * plane_elems(r1 - r0, round_up32(r1 - r0));
        }
    } else if (P.band_dense[(size_t)b]) {
      const int32_t g = P.band_wg_ptr[(size_t)b];
      const int32_t s0 = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g]], s1 = P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)g + 1]];
      for (int32_t r0 = s0; r0 < s1; r0 += (int32_t)opt.dense_block) {
        const int32_t r1 = std::min<int32_t>(s1, r0 + (int32_t)opt.dense_block);
        P.blk_slot0.push_back(r0);
        P.blk_slot1.push_back(r1);
        P.blk_inv_off.push_back(total);
        total += nplanes * plane_elems(r1 - r0, round_up32(r1 - r0));
      }
    }
    P.band_blk_ptr.push_back((int32_t)P.blk_slot0.size());
  }
  return total;
}
inline int64_t dense_block_elems(int64_t nb, bool cplx) { return (cplx ? 2 : 1) * plane_elems(nb, round_up32(nb)); }

// Pass 2 (finalize time, one block at a time into a caller-provided buffer of dense_block_elems doubles that
// is reused -- and shipped to the device -- block after block): the explicit inverse of block q's unit
// triangle written straight into operand layout.  Returns the largest |entry| (the caller compares it with
// dense_max_growth and lets an unstable band fall back to the sequential workgroup).
template <class T>
double build_dense_block(const BandPlan &P, const Csr<T> &A, size_t q, double *ops, bool serial = false) {
  const int32_t r0 = P.blk_slot0[q], nb = P.blk_slot1[q] - r0;
  const int64_t ldk = round_up32(nb), plane = plane_elems(nb, ldk);
  std::memset(ops, 0, sizeof(double) * (size_t)dense_block_elems(nb, sizeof(T) != sizeof(double)));
  // the block's own strict triangle, row by row (local column, value)
  std::vector<int32_t> bp((size_t)nb + 1, 0), bq;
  std::vector<T> bv;
  for (int32_t r = 0; r < nb; ++r) {
    const int32_t s = r0 + r;
    for (int32_t k = A.ptr[(size_t)s]; k < A.ptr[(size_t)s + 1]; ++k) {
      const int32_t c = P.srcslot[(size_t)k] - r0;
      if (c >= 0) {
        bq.push_back(c);
        bv.push_back(A.val[(size_t)k]);
      }
    }
    bp[(size_t)r + 1] = (int32_t)bq.size();
  }
  double growth = 1.0;
  // forward substitution on the identity, CB columns at a time: the CB right-hand sides of a chunk sit
  // side by side (row-major nb x CB scratch), so every update y[r][:] -= T(r,q) * y[q][:] is one
  // contiguous SIMD axpy; chunks are independent (one per thread).  y[q][j] is zero for q < c0 + j, hence
  // no per-column test is needed beyond q >= c0.
  constexpr int32_t CB = 16;
  std::mutex gmx;
  auto body = [&](int64_t cb0, int64_t cb1) {
    std::vector<T> Yc((size_t)nb * CB);
    double g = 1.0;
    for (int64_t cb = cb0; cb < cb1; ++cb) {
      const int32_t c0 = (int32_t)cb * CB;
      const int32_t cw = std::min<int32_t>(CB, nb - c0);
      std::fill(Yc.begin() + (size_t)c0 * CB, Yc.end(), T(0));
      for (int32_t j = 0; j < cw; ++j) Yc[(size_t)(c0 + j) * CB + j] = T(1);
      for (int32_t r = c0 + 1; r < nb; ++r) {
        T *yr = &Yc[(size_t)r * CB];
        for (int32_t k = bp[(size_t)r]; k < bp[(size_t)r + 1]; ++k) {
          const int32_t c = bq[(size_t)k];
          if (c < c0) continue;
          const T a = bv[(size_t)k];
          const T *yq = &Yc[(size_t)c * CB];
          for (int32_t j = 0; j < CB; ++j) yr[j] -= a * yq[j];
        }
      }
      for (int32_t r = c0; r < nb; ++r) {  // element (r, c0 + j) of the strip-major operand
        const int64_t base = ((int64_t)(r >> 4) * ldk + c0) * 16 + (r & 15);
        for (int32_t j = 0; j < cw && c0 + j <= r; ++j) {
          const T val = Yc[(size_t)r * CB + j];
          put_operand(ops, plane, base + (int64_t)j * 16, val);
          g = std::max(g, abs_(val));
        }
      }
    }
    std::lock_guard<std::mutex> lk(gmx);
    growth = std::max(growth, g);
  };
  if (serial)  // (the caller runs many small blocks in parallel)
    body(0, (nb + CB - 1) / CB);
  else
    parallel_for((nb + CB - 1) / CB, 1, body);
  return growth;
}

template <class T>
void finish_band_plan(BandPlan &P, Csr<T> &A /* rows in slot order */, const BandOptions &opt = BandOptions()) {
  const int64_t m = A.nrows;
  std::vector<int32_t> slot_of((size_t)m);
  for (int64_t s = 0; s < m; ++s) slot_of[(size_t)A.rowid[(size_t)s]] = (int32_t)s;
  P.srcslot.resize(A.col.size());
  for (size_t k = 0; k < A.col.size(); ++k) P.srcslot[k] = slot_of[(size_t)A.col[k]];
  P.split.resize((size_t)m);
  // Carried prefixes.  A narrow band keeps 20-100 of the 256 compute units busy, and what bounds it is the rate at
  // which ONE unit gets the gathers of its heaviest component issued.  Most of those gathers read rows
  // that were finished long ago.  So when band b-1 is narrow, its LAUNCH carries extra workgroups that run, on the
  // idle units, the exact prefix pass of band b over the sources older than band b-1 (they depend on nothing band
  // b-1 computes); band b then starts every row at split[] and gathers only the sources inside band b-1 and inside
  // itself.  No launch is added.  Fast mode lists the old sources first in every row of such a band (relative order
  // kept; the summation order of the row changes, tolerance-level like the block-dense bands); exact mode keeps the
  // reference's order and carries only the leading run of old sources.
  const int64_t nb_ = P.nbands();
  P.band_fused.assign((size_t)nb_, 0);
  P.band_old.assign((size_t)nb_, 0);
  if (P.band_cd.size() != (size_t)nb_) P.band_cd.assign((size_t)nb_, 0);
  P.csplit.assign((size_t)m, 0);
  auto band_first_slot = [&](int64_t b) { return P.grp_slot_ptr[(size_t)P.wg_grp_ptr[(size_t)P.band_wg_ptr[(size_t)b]]]; };
  if (opt.fuse)
    for (int64_t b = 1; b < nb_; ++b) {
      const int64_t wgs_prev = P.band_wg_ptr[(size_t)b] - P.band_wg_ptr[(size_t)b - 1];
      if (P.band_cd[(size_t)b]) {  // a component-dense band is carried by ANY preceding band kernel (its workgroups are
        // handed out behind the band's own) -- unless the band fills the chip several times over by itself: then its
        // own workgroups gather the old sources faster than the few carried ones of a narrow predecessor would
        const int64_t wgs_b = P.band_wg_ptr[(size_t)b + 1] - P.band_wg_ptr[(size_t)b];
        if (!P.band_dense[(size_t)b - 1] && (opt.cd_fuse_max_wgs <= 0 || wgs_b <= opt.cd_fuse_max_wgs)) P.band_fused[(size_t)b] = 1;
      } else if (!P.band_dense[(size_t)b] && !P.band_prefix[(size_t)b] && !P.band_dense[(size_t)b - 1] &&
                 !P.band_cd[(size_t)b - 1] && wgs_prev <= opt.fuse_max_wgs)
        P.band_fused[(size_t)b] = 1;
    }
  // ... and only where there IS something to carry: a band none of whose nonzeros is older than its predecessor would
  // pay a pass over all its rows (the carrying pass does the first touch) for nothing
  for (int64_t b = 1; b < nb_; ++b) {
    if (!P.band_fused[(size_t)b]) continue;
    const int32_t prev0 = band_first_slot(b - 1), s0b = band_first_slot(b);
    const int32_t s1b = b + 1 < nb_ ? band_first_slot(b + 1) : (int32_t)m;
    bool older = false;
    for (int32_t k = A.ptr[(size_t)s0b]; k < A.ptr[(size_t)s1b] && !older; ++k) older = P.srcslot[(size_t)k] < prev0;
    if (!older) P.band_fused[(size_t)b] = 0;
  }
  std::vector<int32_t> idx, tcol, tsrc;
  std::vector<T> tval;
  // stable partition of the row's nonzeros [k0, k1) by class (0, 1, 2, ...): relative order inside a class is kept
  auto reorder_row = [&](int32_t k0, int32_t k1, const std::function<int(int32_t)> &cls, int ncls) {
    bool sorted = true;
    int last = 0;
    for (int32_t k = k0; k < k1 && sorted; ++k) {
      const int c = cls(P.srcslot[(size_t)k]);
      if (c < last) sorted = false;
      last = c;
    }
    if (sorted) return;
    idx.clear();
    for (int c = 0; c < ncls; ++c)
      for (int32_t k = k0; k < k1; ++k)
        if (cls(P.srcslot[(size_t)k]) == c) idx.push_back(k);
    tcol.resize(idx.size());
    tsrc.resize(idx.size());
    tval.resize(idx.size());
    for (size_t q = 0; q < idx.size(); ++q) {
      tcol[q] = A.col[(size_t)idx[q]];
      tsrc[q] = P.srcslot[(size_t)idx[q]];
      tval[q] = A.val[(size_t)idx[q]];
    }
    for (size_t q = 0; q < idx.size(); ++q) {
      A.col[(size_t)k0 + q] = tcol[q];
      P.srcslot[(size_t)k0 + q] = tsrc[q];
      A.val[(size_t)k0 + q] = tval[q];
    }
  };
  for (int64_t b = 0; b < nb_; ++b) {
    const int32_t band0 = band_first_slot(b);
    const int32_t prev0 = b > 0 ? band_first_slot(b - 1) : 0;
    const bool cd = P.band_cd[(size_t)b] != 0;
    for (int32_t g = P.band_wg_ptr[(size_t)b]; g < P.band_wg_ptr[(size_t)b + 1]; ++g) {
      // the unit whose rows may depend on each other: the whole workgroup (flag / dense bands), or ONE component
      // = group of a component-dense band
      const int32_t c0 = P.wg_grp_ptr[(size_t)g], c1 = P.wg_grp_ptr[(size_t)g + 1];
      for (int32_t c = c0; c < c1; c = cd ? c + 1 : c1) {
        const int32_t s0 = P.grp_slot_ptr[(size_t)c], s1 = P.grp_slot_ptr[(size_t)(cd ? c + 1 : c1)];
        for (int32_t s = s0; s < s1; ++s) {
          const int32_t k0 = A.ptr[(size_t)s], k1 = A.ptr[(size_t)s + 1];
          // Block-dense bands (fast mode only; their summation order differs from the reference's anyway):
          // a row lists ALL nonzeros whose sources were finished before the band first (relative order kept),
          // so that the chip-wide prefix pass folds every one of them in and [split, end) holds in-band
          // sources only.  The rows of the band's first block then need nothing but that prefix before their
          // block product (Engine::launch_trsv delivers it straight into the product's right-hand side).
          // Bands with a carried prefix (fast mode): the same with "older than band b-1" as the criterion.
          // Component-dense bands: [older than band b-1 | band b-1 | own component]; the first part is carried by the
          // previous launch (when fused), the second is gathered by the component's workgroup, the third IS the inverse.
          if (cd) {
            const bool fz = P.band_fused[(size_t)b] != 0;
            reorder_row(k0, k1, [&](int32_t q) { return q >= s0 ? 2 : ((fz && q < prev0) ? 0 : 1); }, 3);
          } else if (P.band_dense[(size_t)b]) {
            reorder_row(k0, k1, [&](int32_t q) { return q < s0 ? 0 : 1; }, 2);
          } else if (P.band_fused[(size_t)b] && opt.fuse_reorder) {
            reorder_row(k0, k1, [&](int32_t q) { return q < prev0 ? 0 : 1; }, 2);
          }
          int32_t kk = k0;
          if (P.band_prefix[(size_t)b])
            while (kk < k1 && P.srcslot[(size_t)kk] < s0) ++kk;
          else if (P.band_fused[(size_t)b])
            while (kk < k1 && P.srcslot[(size_t)kk] < prev0) ++kk;
          P.split[(size_t)s] = kk;
          int32_t kc = k1;
          if (cd) {
            kc = kk;
            while (kc < k1 && P.srcslot[(size_t)kc] < s0) ++kc;
          }
          P.csplit[(size_t)s] = kc;
          for (int32_t k = k0; k < k1; ++k) {
            const int32_t q = P.srcslot[(size_t)k];  // either before the band, or earlier in this unit
            if (q < band0) P.band_old[(size_t)b] = 1;
            if (!(q < band0 || (q >= s0 && q < s)))
              throw Error(4, "internal error: band plan violates the dependency order");
            if (cd && ((k < kc) != (q < s0)))
              throw Error(4, "internal error: component-dense row is not partitioned into old and own sources");
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The narrow top of a level.  L's solve ENDS in the top of the elimination forest and U's BEGINS there: a handful of
// poorly parallel passes plus the dense rest band on either side -- about ten launches that keep a few compute units
// busy.  For a set T of rows that is closed under both triangles (every row that reads a T row in L is in T; every row a
// T row reads in U is in T) the three steps collapse into ONE dense product
//     z_T = U_TT^{-1} D_T^{-1} L_TT^{-1} (w_T - L_TC x_C) = G t_T ,
// after which U continues with the rows outside T.  choose_top takes the last bands of a first plan of L (whole bands,
// hence closed in L) up to top_max rows and closes the set under U and L; it gives up (empty result) when the closure
// outgrows top_max -- for the (structurally symmetric) factors of a PDE hierarchy nothing has to be added.
// ---------------------------------------------------------------------------------------------
template <class T>
std::vector<uint8_t> choose_top(const Csr<T> &Lr, const Csr<T> &Ur, const BandPlan &Lp0, int64_t top_max, int64_t few_wgs) {
  const int64_t m = Lr.nrows, nb = Lp0.nbands();
  std::vector<uint8_t> top;
  if (m == 0 || nb < 2 || top_max <= 0) return top;
  auto band_slot = [&](int64_t b) { return Lp0.grp_slot_ptr[(size_t)Lp0.wg_grp_ptr[(size_t)Lp0.band_wg_ptr[(size_t)b]]]; };
  // last bands of L's plan while they are narrow and fit
  int64_t b0 = nb;
  while (b0 > 1) {
    const int64_t cand = b0 - 1;
    const int64_t rows_after = m - band_slot(cand);
    const int64_t wgs = Lp0.band_wg_ptr[(size_t)cand + 1] - Lp0.band_wg_ptr[(size_t)cand];
    if (rows_after > top_max || (wgs > few_wgs && !Lp0.band_dense[(size_t)cand])) break;
    b0 = cand;
  }
  if (b0 >= nb) return top;
  top.assign((size_t)m, 0);
  std::vector<int32_t> work;
  int64_t cnt = 0;
  for (int64_t s = band_slot(b0); s < m; ++s) top[(size_t)Lp0.order[(size_t)s]] = 1, work.push_back(Lp0.order[(size_t)s]), ++cnt;
  // closure: U-sources of T rows join T; so do the rows that read a T row in L (L's columns = transposed pattern)
  std::vector<int32_t> tptr((size_t)m + 1, 0), tcol(Lr.col.size());
  for (size_t k = 0; k < Lr.col.size(); ++k) ++tptr[(size_t)Lr.col[k] + 1];
  for (int64_t i = 0; i < m; ++i) tptr[(size_t)i + 1] += tptr[(size_t)i];
  {
    std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
    for (int64_t i = 0; i < m; ++i)
      for (int32_t k = Lr.ptr[(size_t)i]; k < Lr.ptr[(size_t)i + 1]; ++k) tcol[(size_t)fill[(size_t)Lr.col[(size_t)k]]++] = (int32_t)i;
  }
  while (!work.empty()) {
    const int32_t i = work.back();
    work.pop_back();
    auto add = [&](int32_t j) {
      if (!top[(size_t)j]) top[(size_t)j] = 1, work.push_back(j), ++cnt;
    };
    for (int32_t k = Ur.ptr[(size_t)i]; k < Ur.ptr[(size_t)i + 1]; ++k) add(Ur.col[(size_t)k]);  // what i reads in U
    for (int32_t k = tptr[(size_t)i]; k < tptr[(size_t)i + 1]; ++k) add(tcol[(size_t)k]);          // who reads i in L
    if (cnt > top_max) return std::vector<uint8_t>();
  }
  return top;
}

// G = U_TT^{-1} D_T^{-1} L_TT^{-1}, column-major |T| x |T|, rows and columns in the slot order of L's top band
// [r0L, r0L + nt).  Lr / Ur: the slot-ordered triangles after finish_band_plan (per-nonzero source slots in the plans);
// U's top band is [r0U, r0U + nt).  Returns the largest |entry| (growth check, like the block inverses).
template <class T>
double build_top_operator(const Csr<T> &Lr, const BandPlan &Lp, int32_t r0L, const Csr<T> &Ur, const BandPlan &Up, int32_t r0U,
                          int64_t nt, const std::vector<T> &d, std::vector<T> &G) {
  G.assign((size_t)(nt * nt), T(0));
  // U-local index of every L-local row (same row set, two slot orders)
  std::vector<int32_t> u_of_row((size_t)Lr.nrows, -1), l2u((size_t)nt), u2l((size_t)nt);
  for (int64_t r = 0; r < nt; ++r) u_of_row[(size_t)Ur.rowid[(size_t)(r0U + r)]] = (int32_t)r;
  for (int64_t r = 0; r < nt; ++r) {
    const int32_t u = u_of_row[(size_t)Lr.rowid[(size_t)(r0L + r)]];
    if (u < 0) throw Error(4, "internal error: the top bands of L and U hold different rows");
    l2u[(size_t)r] = u;
    u2l[(size_t)u] = (int32_t)r;
  }
  double growth = 0.0;
  std::mutex gmx;
  constexpr int64_t CB = 16;  // columns side by side: every update is a contiguous axpy
  parallel_for((nt + CB - 1) / CB, 1, [&](int64_t cb0, int64_t cb1) {
    std::vector<T> Y((size_t)(nt * CB)), Z((size_t)(nt * CB));
    double g = 0.0;
    for (int64_t cb = cb0; cb < cb1; ++cb) {
      const int64_t c0 = cb * CB, cw = std::min<int64_t>(CB, nt - c0);
      std::fill(Y.begin(), Y.end(), T(0));
      for (int64_t j = 0; j < cw; ++j) Y[(size_t)((c0 + j) * CB + j)] = T(1);
      // forward: L_TT y = e (unit diagonal), L slot order; rows before c0 stay zero
      for (int64_t r = c0; r < nt; ++r) {
        T *yr = &Y[(size_t)(r * CB)];
        const int32_t s = r0L + (int32_t)r;
        for (int32_t k = Lr.ptr[(size_t)s]; k < Lr.ptr[(size_t)s + 1]; ++k) {
          const int32_t q = Lp.srcslot[(size_t)k] - r0L;
          if (q < c0) continue;  // (outside T, or a row that is still zero)
          const T a = Lr.val[(size_t)k];
          const T *yq = &Y[(size_t)((int64_t)q * CB)];
          for (int64_t j = 0; j < CB; ++j) yr[j] -= a * yq[j];
        }
      }
      // scale by D^{-1} and move to U's order
      for (int64_t r = 0; r < nt; ++r) {
        const T dr = d[(size_t)Lr.rowid[(size_t)(r0L + r)]];
        T *zr = &Z[(size_t)((int64_t)l2u[(size_t)r] * CB)];
        const T *yr = &Y[(size_t)(r * CB)];
        for (int64_t j = 0; j < CB; ++j) zr[j] = yr[j] / dr;
      }
      // backward: U_TT z = y (unit diagonal), U slot order (its own dependency order)
      for (int64_t r = 0; r < nt; ++r) {
        T *zr = &Z[(size_t)(r * CB)];
        const int32_t s = r0U + (int32_t)r;
        for (int32_t k = Ur.ptr[(size_t)s]; k < Ur.ptr[(size_t)s + 1]; ++k) {
          const int32_t q = Up.srcslot[(size_t)k] - r0U;
          if (q < 0 || q >= nt) throw Error(4, "internal error: a top row of U reads a row outside the top set");
          const T a = Ur.val[(size_t)k];
          const T *zq = &Z[(size_t)((int64_t)q * CB)];
          for (int64_t j = 0; j < CB; ++j) zr[j] -= a * zq[j];
        }
      }
      for (int64_t u = 0; u < nt; ++u) {
        const int64_t r = u2l[(size_t)u];
        for (int64_t j = 0; j < cw; ++j) {
          const T val = Z[(size_t)(u * CB + j)];
          G[(size_t)(r + (c0 + j) * nt)] = val;
          g = std::max(g, abs_(val));
        }
      }
    }
    std::lock_guard<std::mutex> lk(gmx);
    growth = std::max(growth, g);
  });
  return growth;
}

// Physically permute the CSR rows into slot order so that the matrix streams through HBM in the
// order the kernels consume it (coalesced index/value reads for every batch width).
template <class T>
Csr<T> permute_rows(const Csr<T> &A, const std::vector<int32_t> &order) {
  Csr<T> B;
  B.nrows = A.nrows;
  B.ncols = A.ncols;
  B.ptr.assign((size_t)A.nrows + 1, 0);
  B.col.resize(A.col.size());
  B.val.resize(A.val.size());
  B.rowid = order;
  int32_t pos = 0;
  for (int64_t s = 0; s < A.nrows; ++s) {
    const int32_t i = order[(size_t)s];
    B.ptr[(size_t)s] = pos;
    for (int32_t k = A.ptr[(size_t)i]; k < A.ptr[(size_t)i + 1]; ++k, ++pos) {
      B.col[(size_t)pos] = A.col[(size_t)k];
      B.val[(size_t)pos] = A.val[(size_t)k];
    }
  }
  B.ptr[(size_t)A.nrows] = pos;
  return B;
}

// ---------------------------------------------------------------------------------------------
// Tiled form of a Schur coupling block (E, F) for the matrix cores.  Consecutive rows of E / F share most of their
// columns (measured on the reference's 1M-row hierarchies: a block of 16 rows touches 4-6x fewer DISTINCT source rows
// than it has nonzeros), but a row-gather kernel fetches a 512-byte source row per nonzero.  Here 16 rows form a
// block; the block's distinct columns, ascending, are cut into groups of 4; a group is one v_mfma_f64_16x16x4 step per
// 16-column tile: A = the 16 x 4 coefficient tile (zeros where a row lacks the column), B = the 4 gathered source
// rows.  Every distinct source row of a block is gathered ONCE.  The summation order differs from the reference's
// row loops (tolerance-level; fast mode only).
// ---------------------------------------------------------------------------------------------
struct SpmmTiles {
  int64_t nrows = 0, nblk = 0;
  int rb = 1;                     // 16-row tiles per block (a block's distinct columns are gathered once for all of them)
  std::vector<int32_t> blk_gptr;  // nblk + 1: the groups of block b
  std::vector<int32_t> ucol;      // 4 per group (padded with a repeated column whose coefficients are zero)
  std::vector<double> coef;       // 64 rb per group: element rb_ * 64 + (k << 4) | r  ->  A(16 (rb b + rb_) + r, ucol[4 g + k])
  double reuse = 0.0;             // nonzeros per distinct (block, column) pair
};

inline SpmmTiles build_spmm_tiles(const Csr<double> &A, int rb = 1) {
  SpmmTiles Tl;
  Tl.nrows = A.nrows;
  Tl.rb = rb;
  const int64_t brows = 16 * (int64_t)rb;
  Tl.nblk = (A.nrows + brows - 1) / brows;
  Tl.blk_gptr.assign((size_t)Tl.nblk + 1, 0);
  // pass 1: distinct columns per block (rows are sorted by column: a merge via sort of the block's columns)
  std::vector<std::vector<int32_t>> ucols((size_t)Tl.nblk);
  int64_t distinct = 0;
  parallel_for(Tl.nblk, 256, [&](int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; ++b) {
      const int64_t r0 = brows * b, r1 = std::min<int64_t>(A.nrows, r0 + brows);
      std::vector<int32_t> &u = ucols[(size_t)b];
      u.assign(A.col.begin() + A.ptr[(size_t)r0], A.col.begin() + A.ptr[(size_t)r1]);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
    }
  });
  for (int64_t b = 0; b < Tl.nblk; ++b) {
    distinct += (int64_t)ucols[(size_t)b].size();
    Tl.blk_gptr[(size_t)b + 1] = Tl.blk_gptr[(size_t)b] + (int32_t)((ucols[(size_t)b].size() + 3) / 4);
  }
  Tl.reuse = distinct ? (double)A.col.size() / (double)distinct : 0.0;
  const int64_t ng = Tl.blk_gptr[(size_t)Tl.nblk];
  Tl.ucol.assign((size_t)(4 * ng), 0);
  Tl.coef.assign((size_t)(64 * rb * ng), 0.0);
  parallel_for(Tl.nblk, 256, [&](int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; ++b) {
      const std::vector<int32_t> &u = ucols[(size_t)b];
      const int64_t g0 = Tl.blk_gptr[(size_t)b], g1 = Tl.blk_gptr[(size_t)b + 1];
      for (int64_t q = 0; q < 4 * (g1 - g0); ++q)
        Tl.ucol[(size_t)(4 * g0 + q)] = u.empty() ? 0 : u[(size_t)std::min<int64_t>(q, (int64_t)u.size() - 1)];
      const int64_t r0 = brows * b, r1 = std::min<int64_t>(A.nrows, r0 + brows);
      for (int64_t i = r0; i < r1; ++i)
        for (int32_t k = A.ptr[(size_t)i]; k < A.ptr[(size_t)i + 1]; ++k) {
          const int64_t q = std::lower_bound(u.begin(), u.end(), A.col[(size_t)k]) - u.begin();  // position in the block's list
          const int64_t rl = i - r0;
          Tl.coef[(size_t)(64 * rb * (g0 + q / 4) + 64 * (rl >> 4) + ((q & 3) << 4) + (rl & 15))] += A.val[(size_t)k];
        }
    }
  });
  return Tl;
}
// complex coupling blocks (round 4): the same blocks and groups; a group's coefficients as TWO real tiles [re 64 | im 64]
// (gfx950 has no complex MFMA: a complex tile product is four real ones, k_spmm_tile_z)
inline SpmmTiles build_spmm_tiles(const Csr<zdouble> &A, int = 1) {
  SpmmTiles Tl;
  Tl.nrows = A.nrows;
  Tl.rb = 1;
  Tl.nblk = (A.nrows + 15) / 16;
  Tl.blk_gptr.assign((size_t)Tl.nblk + 1, 0);
  std::vector<std::vector<int32_t>> ucols((size_t)Tl.nblk);
  int64_t distinct = 0;
  parallel_for(Tl.nblk, 256, [&](int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; ++b) {
      const int64_t r0 = 16 * b, r1 = std::min<int64_t>(A.nrows, r0 + 16);
      std::vector<int32_t> &u = ucols[(size_t)b];
      u.assign(A.col.begin() + A.ptr[(size_t)r0], A.col.begin() + A.ptr[(size_t)r1]);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
    }
  });
  for (int64_t b = 0; b < Tl.nblk; ++b) {
    distinct += (int64_t)ucols[(size_t)b].size();
    Tl.blk_gptr[(size_t)b + 1] = Tl.blk_gptr[(size_t)b] + (int32_t)((ucols[(size_t)b].size() + 3) / 4);
  }
  Tl.reuse = distinct ? (double)A.col.size() / (double)distinct : 0.0;
  const int64_t ng = Tl.blk_gptr[(size_t)Tl.nblk];
  Tl.ucol.assign((size_t)(4 * ng), 0);
  Tl.coef.assign((size_t)(128 * ng), 0.0);
  parallel_for(Tl.nblk, 256, [&](int64_t b0, int64_t b1) {
    for (int64_t b = b0; b < b1; ++b) {
      const std::vector<int32_t> &u = ucols[(size_t)b];
      const int64_t g0 = Tl.blk_gptr[(size_t)b], g1 = Tl.blk_gptr[(size_t)b + 1];
      for (int64_t q = 0; q < 4 * (g1 - g0); ++q)
        Tl.ucol[(size_t)(4 * g0 + q)] = u.empty() ? 0 : u[(size_t)std::min<int64_t>(q, (int64_t)u.size() - 1)];
      const int64_t r0 = 16 * b, r1 = std::min<int64_t>(A.nrows, r0 + 16);
      for (int64_t i = r0; i < r1; ++i)
        for (int32_t k = A.ptr[(size_t)i]; k < A.ptr[(size_t)i + 1]; ++k) {
          const int64_t q = std::lower_bound(u.begin(), u.end(), A.col[(size_t)k]) - u.begin();
          const size_t at = (size_t)(128 * (g0 + q / 4) + ((q & 3) << 4) + (i - r0));
          Tl.coef[at] += A.val[(size_t)k].real();
          Tl.coef[at + 64] += A.val[(size_t)k].imag();
        }
    }
  });
  return Tl;
}

// ---------------------------------------------------------------------------------------------
// one level + the whole hierarchy (host copy)
// ---------------------------------------------------------------------------------------------
template <class T>
struct HostLevel {
  int64_t m = 0, n = 0, F_ncols = 0;
  bool E_void = false;  // adjoint levels only: the original level had no F
  Ccs<T> L, U, E, F;  // as imported
  std::vector<T> d;
  std::vector<double> s, t;
  std::vector<int32_t> p, p_inv, q, q_inv;
  // derived
  Csr<T> Lr, Ur, Er, Fr;
  Schedule Ls, Us;   // plain level schedules (wavefronts), kept for queries
  BandPlan Lp, Up;   // what the device executes
  int64_t Ltinv_elems = 0, Utinv_elems = 0;  // doubles of block-inverse operands (built and shipped at finalize)
  // Combined top operator (choose_top / build_top_operator): the rows of `top` -- the narrow top of the elimination
  // forest, where L ends and U begins -- are solved by ONE dense product  z_T = U_TT^{-1} D_T^{-1} L_TT^{-1} t_T
  std::vector<uint8_t> top;         // per row; empty = no combined operator
  int64_t top_n = 0;                // |T|
  int32_t top_bandL = -1, top_bandU = -1;  // the bands of Lp / Up that hold exactly the rows of T
  // checksums of the arrays above as they stood when the level had been imported and analyzed (import.hpp seal_level /
  // verify_level): hifamd_finalize tells a host copy that changed in between -- twice in four rounds two row pointers of E
  // did, cause unshown -- from the one it made, names the array, and rebuilds E / F's row forms from the imported arrays
  std::vector<uint64_t> sums;
};

// Dense last level: A P = Q R (GEQP3 semantics), numerical rank as QRCP::factorize decides it
// (small_scale/QRCP.hpp:107-179).  For the device the factors are expanded once into two explicit
// column-major operators so that the apply is two MFMA GEMMs plus a row scatter:
//   QH   = Q^H                    (n x n)
//   Rinv = R^{-1} upper triangle  (n x n; the leading rk x rk block is R(1:rk,1:rk)^{-1})
template <class T>
struct HostDense {
  int64_t n = 0, rank = 0;
  std::vector<T> mat;          // the unfactored block as imported (kept for hifamd_save)
  double rrqr_cond = 0.0;
  std::vector<T> qr, tau;      // GEQP3 layout
  std::vector<int32_t> jpvt0;  // 0-based column permutation
  std::vector<T> QH, Rinv;     // explicit operators, column-major
  // adjoint apply (QRCP::_solve_t, QRCP.hpp:413-452): z = Q(:,1:rk) * (R(1:rk,1:rk)^{-H} * (P^T c)(1:rk))
  std::vector<T> Q, RinvH;     // Q = (Q^H)^H, RinvH = (R^{-1})^H (lower triangular), column-major
  // symmetric / Hermitian last level (the reference's SYEIG solver, small_scale/SYEIG.hpp): kind = 1.
  // A = V diag(w) V^H, truncated in the order `trunc`.  The device then reuses the two dense operators:
  //   QH := diag(1/w) V^H with rows in truncation order,  Q := V with columns in truncation order
  // (solve = Q(:,1:rk) * (QH(1:rk,:) c)); SymMul := diag(w) V^H (rows in truncation order) for the product.
  // kind = 2: the reference built with HIF_DENSE_MODE=0 uses LU with partial pivoting (small_scale/LUP.hpp)
  // for the last level: QH := A^{-1} (on the adjoint engine its plain transpose: LUP::solve passes 'T' to
  // ?getrs also for complex data, LUP.hpp:150), SymMul := A (adjoint: A^H, ?gemv 'C', LUP.hpp:187) for the product.
  int kind = 0, spd = 0;
  std::vector<double> w;
  std::vector<int32_t> trunc;
  std::vector<T> evec, SymMul;
};


template <class T>
double col_norm(const T *x, int64_t n) {
  double scale = 0.0, ssq = 1.0;
  auto acc = [&](double a) {
    a = std::fabs(a);
    if (a == 0.0) return;
    if (scale < a) {
      const double r = scale / a;
      ssq = 1.0 + ssq * r * r;
      scale = a;
    } else {
      const double r = a / scale;
      ssq += r * r;
    }
  };
  for (int64_t i = 0; i < n; ++i) {
    acc(real_(x[i]));
    if (sizeof(T) == sizeof(zdouble)) acc(reinterpret_cast<const double *>(x + i)[1]);
  }
  return scale * std::sqrt(ssq);
}

// Householder QR with column pivoting (LAPACK ?geqp3 semantics: pivot on the largest remaining
// partial column norm, norms downdated and recomputed on cancellation).
template <class T>
void qr_colpiv(int64_t n, std::vector<T> &A, std::vector<int32_t> &jpvt0, std::vector<T> &tau) {
  std::vector<double> vn1((size_t)n), vn2((size_t)n);
  jpvt0.resize((size_t)n);
  tau.assign((size_t)n, T(0));
  const double tol3z = std::sqrt(std::numeric_limits<double>::epsilon() * 0.5);
  for (int64_t j = 0; j < n; ++j) {
    jpvt0[(size_t)j] = (int32_t)j;
    vn1[(size_t)j] = vn2[(size_t)j] = col_norm(&A[(size_t)(j * n)], n);
  }
  std::vector<T> w((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    int64_t pvt = i;
    for (int64_t j = i + 1; j < n; ++j)
      if (vn1[(size_t)j] > vn1[(size_t)pvt]) pvt = j;
    if (pvt != i) {
      std::swap_ranges(A.begin() + pvt * n, A.begin() + pvt * n + n, A.begin() + i * n);
      std::swap(jpvt0[(size_t)pvt], jpvt0[(size_t)i]);
      vn1[(size_t)pvt] = vn1[(size_t)i];
      vn2[(size_t)pvt] = vn2[(size_t)i];
    }
    T *v = &A[(size_t)(i + i * n)];
    const int64_t len = n - i;
    // reflector: H = I - tau v v^H, v[0] = 1, H^H x = beta e1 with beta real
    const double xnorm = col_norm(v + 1, len - 1);
    const T alpha = v[0];
    const double ar = real_(alpha), ai = sizeof(T) == sizeof(zdouble) ? reinterpret_cast<const double *>(&alpha)[1] : 0.0;
    if (xnorm == 0.0 && ai == 0.0) {
      tau[(size_t)i] = T(0);
    } else {
      double beta = std::sqrt(ar * ar + ai * ai + xnorm * xnorm);
      if (ar >= 0.0) beta = -beta;
      T tq;
      if (sizeof(T) == sizeof(zdouble)) {
        double *tp = reinterpret_cast<double *>(&tq);
        tp[0] = (beta - ar) / beta;
        tp[1] = -ai / beta;
      } else
        tq = T((beta - ar) / beta);
      tau[(size_t)i] = tq;
      const T sc = T(1.0) / (alpha - T(beta));
      for (int64_t r = 1; r < len; ++r) v[r] *= sc;
      v[0] = T(beta);
    }
    if (i + 1 < n && tau[(size_t)i] != T(0)) {
      const T beta_keep = v[0];
      v[0] = T(1);
      const T ctau = conj_(tau[(size_t)i]);
      for (int64_t j = i + 1; j < n; ++j) {
        T *c = &A[(size_t)(i + j * n)];
        T dot = T(0);
        for (int64_t r = 0; r < len; ++r) dot += conj_(v[r]) * c[r];
        dot *= ctau;
        for (int64_t r = 0; r < len; ++r) c[r] -= v[r] * dot;
      }
      v[0] = beta_keep;
    }
    for (int64_t j = i + 1; j < n; ++j) {
      if (vn1[(size_t)j] == 0.0) continue;
      double temp = abs_(A[(size_t)(i + j * n)]) / vn1[(size_t)j];
      temp = std::max(0.0, 1.0 - temp * temp);
      const double r = vn1[(size_t)j] / vn2[(size_t)j];
      if (temp * r * r <= tol3z) {
        vn1[(size_t)j] = (i + 1 < n) ? col_norm(&A[(size_t)(i + 1 + j * n)], n - i - 1) : 0.0;
        vn2[(size_t)j] = vn1[(size_t)j];
      } else
        vn1[(size_t)j] *= std::sqrt(temp);
    }
  }
}

// Incremental condition estimation step (LAPACK ?laic1), used only when the diagonal filter of
// QRCP::factorize fires (QRCP.hpp:150-163 -> _est_rank_2norm :333-364).
template <class T>
void laic1(int job, int64_t j, const T *x, double sest, const T *w, T gamma, double &sestpr, T &s, T &c) {
  const double eps = std::numeric_limits<double>::epsilon() * 0.5;
  T alpha = T(0);
  for (int64_t i = 0; i < j; ++i) alpha += conj_(x[i]) * w[i];
  const double absalp = abs_(alpha), absgam = abs_(gamma), absest = std::fabs(sest);
  auto nrm = [](const T &a, const T &b) { return std::sqrt(abs_(a) * abs_(a) + abs_(b) * abs_(b)); };
  if (job == 1) {
    if (sest == 0.0) {
      const double s1 = std::max(absgam, absalp);
      if (s1 == 0.0) {
        s = T(0), c = T(1), sestpr = 0.0;
      } else {
        s = alpha / s1, c = gamma / s1;
        const double tmp = nrm(s, c);
        s /= tmp, c /= tmp, sestpr = s1 * tmp;
      }
    } else if (absgam <= eps * absest) {
      s = T(1), c = T(0);
      const double tmp = std::max(absest, absalp), s1 = absest / tmp, s2 = absalp / tmp;
      sestpr = tmp * std::sqrt(s1 * s1 + s2 * s2);
    } else if (absalp <= eps * absest) {
      if (absgam <= absest)
        s = T(1), c = T(0), sestpr = absest;
      else
        s = T(0), c = T(1), sestpr = absgam;
    } else if (absest <= eps * absalp || absest <= eps * absgam) {
      const double s1 = absgam, s2 = absalp;
      if (s1 <= s2) {
        const double tmp = s1 / s2, scl = std::sqrt(1.0 + tmp * tmp);
        sestpr = s2 * scl, s = (alpha / s2) / scl, c = (gamma / s2) / scl;
      } else {
        const double tmp = s2 / s1, scl = std::sqrt(1.0 + tmp * tmp);
        sestpr = s1 * scl, s = (alpha / s1) / scl, c = (gamma / s1) / scl;
      }
    } else {
      const T z1 = alpha / absest, z2 = gamma / absest;
      const double b = (1.0 - abs_(z1) * abs_(z1) - abs_(z2) * abs_(z2)) * 0.5, cc = abs_(z1) * abs_(z1);
      const double t = b > 0.0 ? cc / (b + std::sqrt(b * b + cc)) : std::sqrt(b * b + cc) - b;
      const T sine = -z1 / t, cosine = -z2 / (1.0 + t);
      const double tmp = nrm(sine, cosine);
      s = sine / tmp, c = cosine / tmp, sestpr = std::sqrt(t + 1.0) * absest;
    }
    return;
  }
  if (sest == 0.0) {
    sestpr = 0.0;
    T sine, cosine;
    if (std::max(absgam, absalp) == 0.0)
      sine = T(1), cosine = T(0);
    else
      sine = -conj_(gamma), cosine = conj_(alpha);
    const double s1 = std::max(abs_(sine), abs_(cosine));
    s = sine / s1, c = cosine / s1;
    const double tmp = nrm(s, c);
    s /= tmp, c /= tmp;
  } else if (absgam <= eps * absest) {
    s = T(0), c = T(1), sestpr = absgam;
  } else if (absalp <= eps * absest) {
    if (absgam <= absest)
      s = T(0), c = T(1), sestpr = absgam;
    else
      s = T(1), c = T(0), sestpr = absest;
  } else if (absest <= eps * absalp || absest <= eps * absgam) {
    const double s1 = absgam, s2 = absalp;
    if (s1 <= s2) {
      const double tmp = s1 / s2, scl = std::sqrt(1.0 + tmp * tmp);
      sestpr = absest * (tmp / scl), s = -(conj_(gamma) / s2) / scl, c = (conj_(alpha) / s2) / scl;
    } else {
      const double tmp = s2 / s1, scl = std::sqrt(1.0 + tmp * tmp);
      sestpr = absest / scl, s = -(conj_(gamma) / s1) / scl, c = (conj_(alpha) / s1) / scl;
    }
  } else {
    const T z1 = alpha / absest, z2 = gamma / absest;
    const double a1 = abs_(z1), a2 = abs_(z2), a12 = abs_(conj_(z1) * z2);
    const double norma = std::max(1.0 + a1 * a1 + a12, a12 + a2 * a2);
    const double test = 1.0 + 2.0 * (a1 - a2) * (a1 + a2);
    T sine, cosine;
    if (test >= 0.0) {
      const double b = (a1 * a1 + a2 * a2 + 1.0) * 0.5, cc = a2 * a2;
      const double t = cc / (b + std::sqrt(std::fabs(b * b - cc)));
      sine = z1 / (1.0 - t), cosine = -z2 / t;
      sestpr = std::sqrt(t + 4.0 * eps * eps * norma) * absest;
    } else {
      const double b = (a2 * a2 + a1 * a1 - 1.0) * 0.5, cc = a1 * a1;
      const double t = b >= 0.0 ? -cc / (b + std::sqrt(b * b + cc)) : b - std::sqrt(b * b + cc);
      sine = -z1 / t, cosine = -z2 / (1.0 + t);
      sestpr = std::sqrt(1.0 + t + 4.0 * eps * eps * norma) * absest;
    }
    const double tmp = nrm(sine, cosine);
    s = sine / tmp, c = cosine / tmp;
  }
}

// explicit Q^H and R^{-1} from the GEQP3 factors (also used to rebuild them for the adjoint operators)
template <class T>
void dense_explicit_ops(HostDense<T> &D) {
  const int64_t n = D.n;
  const T *A = D.qr.data();
  // explicit Q^H: start from I and apply H(0)^H, H(1)^H, ... to its columns (Q^H = H(n-1)^H...H(0)^H);
  // the columns are independent of each other
  D.QH.assign((size_t)(n * n), T(0));
  for (int64_t j = 0; j < n; ++j) D.QH[(size_t)(j + j * n)] = T(1);
  parallel_for(n, 8, [&](int64_t j0, int64_t j1) {
    for (int64_t j = j0; j < j1; ++j)
      for (int64_t i = 0; i < n; ++i) {
        const T ctau = conj_(D.tau[(size_t)i]);
        if (ctau == T(0)) continue;
        const T *v = A + i + i * n;  // v[0] is implicit 1
        T *c = &D.QH[(size_t)(i + j * n)];
        T dot = c[0];
        for (int64_t r = 1; r < n - i; ++r) dot += conj_(v[r]) * c[r];
        dot *= ctau;
        c[0] -= dot;
        for (int64_t r = 1; r < n - i; ++r) c[r] -= v[r] * dot;
      }
  });
  // explicit R^{-1} (upper triangular), column by column: R X = I  =>  back substitution
  D.Rinv.assign((size_t)(n * n), T(0));
  parallel_for(n, 8, [&](int64_t j0, int64_t j1) {
    for (int64_t j = j0; j < j1; ++j) {
      T *x = &D.Rinv[(size_t)(j * n)];
      x[j] = T(1);
      for (int64_t k = j; k >= 0; --k) {  // column-oriented sweep: contiguous reads of R(:,k)
        const T rkk = A[k + k * n];
        if (rkk == T(0)) {  // exactly singular pivot: such columns lie beyond any usable rank
          for (int64_t i = 0; i <= j; ++i) x[i] = T(0);
          break;
        }
        x[k] /= rkk;
        const T xk = x[k];
        const T *rk = A + k * n;
        for (int64_t i = 0; i < k; ++i) x[i] -= rk[i] * xk;
      }
    }
  });
}

// adjoint operators: conjugate transposes of the two explicit factors
template <class T>
void dense_adjoint_ops(HostDense<T> &D) {
  const int64_t n = D.n;
  if (D.QH.empty() || D.Rinv.empty()) dense_explicit_ops(D);
  D.Q.resize((size_t)(n * n));
  D.RinvH.resize((size_t)(n * n));
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < n; ++i) {
      D.Q[(size_t)(i + j * n)] = conj_(D.QH[(size_t)(j + i * n)]);
      D.RinvH[(size_t)(i + j * n)] = conj_(D.Rinv[(size_t)(j + i * n)]);
    }
}

template <class T>
void dense_factorize(HostDense<T> &D, const T *mat_colmajor, int64_t n, double rrqr_cond) {
  D.n = n;
  D.mat.assign(mat_colmajor, mat_colmajor + n * n);
  D.rrqr_cond = rrqr_cond;
  D.qr.assign(mat_colmajor, mat_colmajor + n * n);
  qr_colpiv(n, D.qr, D.jpvt0, D.tau);
  const double eps = std::numeric_limits<double>::epsilon();
  const double diag_tol = std::sqrt(eps), cond_tol = 1.0 / std::pow(eps, 2.0 / 3.0);
  const double cond_thres = rrqr_cond <= 0.0 ? cond_tol : rrqr_cond;
  const T *A = D.qr.data();
  bool cond_test = false;
  const double diag_eps = diag_tol * abs_(A[0]);
  for (int64_t i = n; i != 0; --i)
    if (abs_(A[(i - 1) + (i - 1) * n]) < diag_eps) {
      cond_test = true;
      break;
    }
  D.rank = n;
  if (cond_test) {
    std::vector<T> x((size_t)n, T(0)), y((size_t)n, T(0));
    x[0] = y[0] = T(1);
    double smax = abs_(A[0]), smin = smax, sminpr = 0, smaxpr = 0;
    T s1, c1, s2, c2;
    int64_t rk = 0;
    for (; rk < n; ++rk) {
      laic1(2, rk, x.data(), smin, A + rk * n, A[rk + rk * n], sminpr, s1, c1);
      laic1(1, rk, y.data(), smax, A + rk * n, A[rk + rk * n], smaxpr, s2, c2);
      if (!(smaxpr <= sminpr * cond_thres)) break;
      for (int64_t i = 0; i < rk; ++i) x[(size_t)i] *= s1, y[(size_t)i] *= s2;
      x[(size_t)rk] = c1, y[(size_t)rk] = c2;
      smin = sminpr, smax = smaxpr;
    }
    D.rank = rk;
  }
  dense_explicit_ops(D);
}

// ---------------------------------------------------------------------------------------------
// Symmetric / Hermitian eigendecomposition A = V diag(w) V^H, w ascending (what LAPACK ?syev / ?heev
// 'V','L' returns to SYEIG::factorize, SYEIG.hpp:107-129): Householder reduction of the lower
// triangle to a REAL symmetric tridiagonal matrix (the unblocked ?sytd2 / ?hetd2 recurrence: the
// reflectors are generated with a real beta, so the off-diagonal is real also for complex input),
// explicit Q = H(0) H(1) ... H(n-2), then implicit QL with Wilkinson shifts on (d, e), the plane rotations
// accumulated into Q.  A: column-major n x n, only the lower triangle is read; on exit the eigenvectors.
// ---------------------------------------------------------------------------------------------
template <class T>
void herm_eig(int64_t n, std::vector<T> &A, std::vector<double> &w) {
  w.assign((size_t)n, 0.0);
  if (n == 0) return;
  std::vector<double> d((size_t)n, 0.0), e((size_t)n, 0.0);
  std::vector<T> tau((size_t)n, T(0)), x((size_t)n);
  auto at = [&](int64_t i, int64_t j) -> T & { return A[(size_t)(i + j * n)]; };
  // work on a full Hermitian copy (upper := conj(lower)) so that the rank-2 updates are plain loops
  for (int64_t j = 0; j < n; ++j) {
    at(j, j) = T(real_(at(j, j)));
    for (int64_t i = j + 1; i < n; ++i) at(j, i) = conj_(at(i, j));
  }
  for (int64_t i = 0; i + 1 < n; ++i) {
    const int64_t len = n - i - 1;  // the reflector acts on rows/columns i+1 .. n-1
    T *v = &at(i + 1, i);
    // ?larfg: H^H (alpha; xrest) = (beta; 0), H = I - tau v v^H, v[0] = 1, beta real
    const double xnorm = col_norm(v + 1, len - 1);
    const T alpha = v[0];
    const double ar = real_(alpha), ai = sizeof(T) == sizeof(zdouble) ? reinterpret_cast<const double *>(&alpha)[1] : 0.0;
    T taui = T(0);
    double beta = ar;
    if (!(xnorm == 0.0 && ai == 0.0)) {
      beta = std::sqrt(ar * ar + ai * ai + xnorm * xnorm);
      if (ar >= 0.0) beta = -beta;
      if (sizeof(T) == sizeof(zdouble)) {
        double *tp = reinterpret_cast<double *>(&taui);
        tp[0] = (beta - ar) / beta;
        tp[1] = -ai / beta;
      } else
        taui = T((beta - ar) / beta);
      const T sc = T(1.0) / (alpha - T(beta));
      for (int64_t r = 1; r < len; ++r) v[r] *= sc;
    }
    e[(size_t)i] = beta;
    if (taui != T(0)) {
      v[0] = T(1);
      // x = taui * A22 v;  x += (-1/2 taui (x^H v)) v;  A22 -= v x^H + x v^H
      parallel_for(len, 64, [&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
          T acc = T(0);
          for (int64_t c = 0; c < len; ++c) acc += conj_(at(i + 1 + c, i + 1 + r)) * v[c];  // row r of A22 = conj of column r
          x[(size_t)r] = taui * acc;
        }
      });
      T dot = T(0);
      for (int64_t r = 0; r < len; ++r) dot += conj_(x[(size_t)r]) * v[r];
      const T al = T(-0.5) * taui * dot;
      for (int64_t r = 0; r < len; ++r) x[(size_t)r] += al * v[r];
      parallel_for(len, 64, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; ++c) {
          const T cvc = conj_(v[c]), cxc = conj_(x[(size_t)c]);
          T *col = &at(i + 1, i + 1 + c);
          for (int64_t r = 0; r < len; ++r) col[r] -= v[r] * cxc + x[(size_t)r] * cvc;
        }
      });
    }
    d[(size_t)i] = real_(at(i, i));
    tau[(size_t)i] = taui;
  }
  d[(size_t)(n - 1)] = real_(at(n - 1, n - 1));
  // explicit Q = H(0) ... H(n-2) in a separate array (the reflector vectors live below the subdiagonal of A)
  std::vector<T> Q((size_t)(n * n), T(0));
  for (int64_t j = 0; j < n; ++j) Q[(size_t)(j + j * n)] = T(1);
  for (int64_t i = n - 2; i >= 0; --i) {
    const T taui = tau[(size_t)i];
    if (taui == T(0)) continue;
    const int64_t len = n - i - 1;
    const T *v = &at(i + 1, i);  // v[0] is 1 (set above)
    parallel_for(n - i - 1, 16, [&](int64_t j0, int64_t j1) {  // only columns i+1.. are not yet unit vectors e_j with j <= i
      for (int64_t jj = j0; jj < j1; ++jj) {
        T *c = &Q[(size_t)((i + 1) + (i + 1 + jj) * n)];
        T dot = T(0);
        for (int64_t r = 0; r < len; ++r) dot += conj_(v[r]) * c[r];
        dot *= taui;
        for (int64_t r = 0; r < len; ++r) c[r] -= v[r] * dot;
      }
    });
  }
  // implicit QL (EISPACK tql2 recurrence) on the real tridiagonal (d, e), rotations applied to Q's columns
  const double eps = std::numeric_limits<double>::epsilon();
  std::vector<int64_t> rot_i;
  std::vector<double> rot_s, rot_c;
  for (int64_t l = 0; l < n; ++l) {
    int iter = 0;
    int64_t m;
    do {
      for (m = l; m + 1 < n; ++m) {
        const double dd = std::fabs(d[(size_t)m]) + std::fabs(d[(size_t)m + 1]);
        if (std::fabs(e[(size_t)m]) <= eps * dd) break;
      }
      if (m != l) {
        if (iter++ == 80) throw Error(4, "symmetric eigensolver: the QL iteration did not converge");
        double g = (d[(size_t)l + 1] - d[(size_t)l]) / (2.0 * e[(size_t)l]);
        double r = std::hypot(g, 1.0);
        g = d[(size_t)m] - d[(size_t)l] + e[(size_t)l] / (g + std::copysign(r, g));
        double sn = 1.0, cs = 1.0, p = 0.0;
        int64_t i;
        for (i = m - 1; i >= l; --i) {
          double f = sn * e[(size_t)i];
          const double b = cs * e[(size_t)i];
          r = std::hypot(f, g);
          e[(size_t)i + 1] = r;
          if (r == 0.0) {
            d[(size_t)i + 1] -= p;
            e[(size_t)m] = 0.0;
            break;
          }
          sn = f / r;
          cs = g / r;
          g = d[(size_t)i + 1] - p;
          r = (d[(size_t)i] - g) * sn + 2.0 * cs * b;
          p = sn * r;
          d[(size_t)i + 1] = g + p;
          g = cs * r - b;
          rot_i.push_back(i);
          rot_s.push_back(sn);
          rot_c.push_back(cs);
        }
        // the sweep's plane rotations, in order, on the eigenvector columns: row chunks in parallel
        if (!rot_i.empty()) {
          parallel_for(n, 512, [&](int64_t k0, int64_t k1) {
            for (size_t q = 0; q < rot_i.size(); ++q) {
              T *zi = &Q[(size_t)(rot_i[q] * n)], *zi1 = &Q[(size_t)((rot_i[q] + 1) * n)];
              const double sq = rot_s[q], cq = rot_c[q];
              for (int64_t k = k0; k < k1; ++k) {
                const T fz = zi1[k];
                zi1[k] = sq * zi[k] + cq * fz;
                zi[k] = cq * zi[k] - sq * fz;
              }
            }
          });
          rot_i.clear();
          rot_s.clear();
          rot_c.clear();
        }
        if (r == 0.0 && i >= l) continue;
        d[(size_t)l] -= p;
        e[(size_t)l] = g;
        e[(size_t)m] = 0.0;
      }
    } while (m != l);
  }
  // ascending eigenvalues, eigenvectors permuted along
  std::vector<int64_t> ord((size_t)n);
  for (int64_t i = 0; i < n; ++i) ord[(size_t)i] = i;
  std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return d[(size_t)a] < d[(size_t)b]; });
  for (int64_t j = 0; j < n; ++j) {
    w[(size_t)j] = d[(size_t)ord[(size_t)j]];
    std::copy(Q.begin() + ord[(size_t)j] * n, Q.begin() + ord[(size_t)j] * n + n, A.begin() + j * n);
  }
}

// SYEIG::factorize (SYEIG.hpp:107-175): eigendecomposition, then the truncation order and numerical rank
// for positive definite (spd > 0), negative definite (spd < 0) or indefinite (spd == 0) blocks.
template <class T>
void dense_symm_ops(HostDense<T> &D) {
  const int64_t n = D.n;
  D.QH.assign((size_t)(n * n), T(0));
  D.Q.assign((size_t)(n * n), T(0));
  D.SymMul.assign((size_t)(n * n), T(0));
  for (int64_t i = 0; i < n; ++i) {
    const int64_t c = D.trunc[(size_t)i];
    const double wi = D.w[(size_t)c];
    for (int64_t k = 0; k < n; ++k) {
      const T vkc = D.evec[(size_t)(k + c * n)];
      D.QH[(size_t)(i + k * n)] = conj_(vkc) / wi;      // work[trunc[i]] /= w[trunc[i]]   (:194-195)
      D.SymMul[(size_t)(i + k * n)] = conj_(vkc) * wi;  // work[trunc[i]] *= w[trunc[i]]   (:268-269)
      D.Q[(size_t)(k + i * n)] = vkc;
    }
  }
}

template <class T>
void dense_factorize_symm(HostDense<T> &D, const T *mat_colmajor, int64_t n, int spd) {
  D.kind = 1;
  D.spd = spd;
  D.n = n;
  D.mat.assign(mat_colmajor, mat_colmajor + n * n);
  D.evec.assign(mat_colmajor, mat_colmajor + n * n);
  herm_eig(n, D.evec, D.w);
  D.trunc.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) D.trunc[(size_t)i] = (int32_t)i;
  const double EPS = std::pow(std::numeric_limits<double>::epsilon(), 2.0 / 3.0);
  double wmax = 0.0;
  for (double v : D.w) wmax = std::max(wmax, std::fabs(v));
  const double thres = EPS * wmax;
  int64_t rank = n;
  if (spd > 0) {
    for (int64_t i = n - 1; i >= 0; --i)
      if (D.w[(size_t)i] <= 0.0 || std::fabs(D.w[(size_t)i]) <= thres)
        --rank;
      else
        break;
  } else if (spd < 0) {
    std::reverse(D.trunc.begin(), D.trunc.end());
    for (int64_t i = 0; i < n; ++i)
      if (D.w[(size_t)i] >= 0.0 || std::fabs(D.w[(size_t)i]) <= thres)
        --rank;
      else
        break;
  } else {
    std::stable_sort(D.trunc.begin(), D.trunc.end(),
                     [&](int32_t a, int32_t b) { return std::fabs(D.w[(size_t)a]) > std::fabs(D.w[(size_t)b]); });
    for (int64_t i = n - 1; i >= 0; --i)
      if (std::fabs(D.w[(size_t)D.trunc[(size_t)i]]) <= thres)
        --rank;
      else
        break;
  }
  D.rank = rank;
  dense_symm_ops(D);
}

// LUP::factorize (LUP.hpp:100-119: ?getrf) followed by the explicit inverse (column j = solution of A x = e_j by
// the ?getrs sequence: row interchanges, unit-lower forward sweep, upper backward sweep).  A: column-major n x n,
// overwritten by its LU factors; ipiv 0-based.  Returns the 1-based index of the first exactly zero pivot, else 0.
template <class T>
int64_t lu_partial_pivot(int64_t n, std::vector<T> &A, std::vector<int32_t> &ipiv) {
  ipiv.assign((size_t)n, 0);
  int64_t info = 0;
  for (int64_t k = 0; k < n; ++k) {
    int64_t pv = k;
    double best = abs1_(A[(size_t)(k + k * n)]);
    for (int64_t i = k + 1; i < n; ++i) {  // ?getf2 / i?amax: largest |re| + |im| (complex) or |x| (real)
      const double a = abs1_(A[(size_t)(i + k * n)]);
      if (a > best) best = a, pv = i;
    }
    ipiv[(size_t)k] = (int32_t)pv;
    if (A[(size_t)(pv + k * n)] == T(0)) {
      if (!info) info = k + 1;
      continue;
    }
    if (pv != k)
      for (int64_t j = 0; j < n; ++j) std::swap(A[(size_t)(k + j * n)], A[(size_t)(pv + j * n)]);
    const T piv = A[(size_t)(k + k * n)];
    for (int64_t i = k + 1; i < n; ++i) A[(size_t)(i + k * n)] /= piv;
    parallel_for(n - k - 1, 32, [&](int64_t j0, int64_t j1) {
      for (int64_t jj = j0; jj < j1; ++jj) {
        const int64_t j = k + 1 + jj;
        const T ukj = A[(size_t)(k + j * n)];
        if (ukj == T(0)) continue;
        T *c = &A[(size_t)(j * n)];
        const T *l = &A[(size_t)(k * n)];
        for (int64_t i = k + 1; i < n; ++i) c[i] -= l[i] * ukj;
      }
    });
  }
  return info;
}

template <class T>
void dense_lup_ops(HostDense<T> &D, bool adjoint) {
  const int64_t n = D.n;
  const T *LU = D.qr.data();
  std::vector<T> inv((size_t)(n * n), T(0));
  parallel_for(n, 8, [&](int64_t j0, int64_t j1) {
    for (int64_t j = j0; j < j1; ++j) {
      T *x = &inv[(size_t)(j * n)];
      x[j] = T(1);
      for (int64_t k = 0; k < n; ++k)
        if (D.jpvt0[(size_t)k] != k) std::swap(x[k], x[D.jpvt0[(size_t)k]]);
      for (int64_t k = 0; k < n; ++k) {  // L y = P e_j
        const T xk = x[k];
        if (xk == T(0)) continue;
        const T *l = LU + k * n;
        for (int64_t i = k + 1; i < n; ++i) x[i] -= l[i] * xk;
      }
      for (int64_t k = n - 1; k >= 0; --k) {  // U x = y
        x[k] /= LU[k + k * n];
        const T xk = x[k];
        const T *u = LU + k * n;
        for (int64_t i = 0; i < k; ++i) x[i] -= u[i] * xk;
      }
    }
  });
  D.QH.assign((size_t)(n * n), T(0));
  D.SymMul.assign((size_t)(n * n), T(0));
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < n; ++i) {
      D.QH[(size_t)(i + j * n)] = adjoint ? inv[(size_t)(j + i * n)] : inv[(size_t)(i + j * n)];  // 'T', not 'C'
      D.SymMul[(size_t)(i + j * n)] = adjoint ? conj_(D.mat[(size_t)(j + i * n)]) : D.mat[(size_t)(i + j * n)];
    }
}

template <class T>
void dense_factorize_lup(HostDense<T> &D, const T *mat_colmajor, int64_t n) {
  D.kind = 2;
  D.n = n;
  D.mat.assign(mat_colmajor, mat_colmajor + n * n);
  D.qr.assign(mat_colmajor, mat_colmajor + n * n);
  const int64_t info = lu_partial_pivot(n, D.qr, D.jpvt0);
  if (info)  // LUP::factorize only warns (rank = info - 1) and ?getrs would then divide by zero (LUP.hpp:108-116)
    throw Error(3, "dense block is exactly singular (zero pivot in the LU factorization): use the QRCP last level");
  D.rank = n;
  dense_lup_ops(D, false);
}

template <class T>
struct HostHierarchy {
  std::vector<HostLevel<T>> levels;
  bool has_dense = false;
  HostDense<T> dense;
  // the user's matrix for IR (0-based CRS after normalisation)
  Csr<T> A;
  bool has_A = false;
};

}  // namespace hifamd
