// libhifir_amd_shim.cpp -- the drop-in libhifir: every symbol of the reference's C ABI
// (libhifir/include/libhifir.h:240-362, 611-740, 1231-1341) with identical signatures and status codes, with
// lhf?Apply / lhf?Solve served from HBM by the hand-written gfx950 path of include/hifir_amd.h.
//
//   lhf?Create / Setup / Refactorize : factorization stays on the host (the reference's header-only C++ path,
//                                      HIF::factorize, src/hif/builder.hpp:388 -- BASELINE north_star), then every
//                                      hif::Prec of M.precs() is handed to hifamd_add_level / hifamd_set_dense*
//                                      (Prec.hpp:309-323) and the hierarchy is shipped to HBM once
//   lhf?Apply / lhf?Solve            : hifamd_apply_batch with the operator / rank rules of libhifir.cpp:447-472
//   lhf?Update                       : the borrowed user matrix is (re)uploaded for iterative refinement
//   lhf?ApplyBatch, lhfSetDevices,
//   lhf?Save/LoadHierarchy           : additive (include/libhifir_amd_ext.h)
//   s / c / sd / cz families         : the single-precision hierarchy is factorized on the host by the reference in single
//                                      precision (its own templates), WIDENED exactly to fp64 / complex fp64 on the way to
//                                      HBM and applied there by the same kernels: lhfsd* / lhfcz* (single hierarchy, double
//                                      vectors, libhifir.cpp:1192-1284) run the reference's own mixed arithmetic in the
//                                      sparse stages; lhfs* / lhfc* convert b on the way in and round x on the way out
//                                      (the apply itself is carried out in the wider type: never less accurate than the
//                                      reference's float loops; compared with the reference's results at float tolerance)
//
// This file includes the reference's OWN declaration header (-I$(REF)/libhifir/include), so a signature that
// drifted from the reference's would not compile.  Nothing of libhifir.cpp is reproduced: the per-type blocks there
// are four hand-written copies; here one template serves all types.  Build: shim/Makefile (needs the reference
// headers for the host factorization, like oracle/Makefile; the result travels to GPU nodes as a binary).
#ifndef HIF_THROW
#  define HIF_THROW  // hif_error -> std::runtime_error (caught at this boundary), as libhifir.cpp:35-37 builds it
#endif
#include <hifir.hpp>

#include "libhifir.h"
#include "libhifir_amd_ext.h"
#include "hifir_amd.h"

#include <cfloat>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

namespace {

// process-global, unsynchronised, returned once then cleared -- exactly the reference's convention
// (libhifir.cpp:45-53, :224-229): slot 1 collects, slot 0 keeps the string alive for the caller
std::string g_msg[2];
void set_msg(const std::string &m) { g_msg[1] = m; }
LhfStatus fail(const std::string &m) {
  set_msg(m);
  return LHF_HIFIR_ERROR;
}
// a HifAmdStatus IS an LhfStatus (same values, hifir_amd.h:35-41); the message travels along
LhfStatus from_amd(HifAmdStatus st) {
  if (st != HIFAMD_SUCCESS) {
    const char *m = hifamd_last_error();
    set_msg(m ? m : "hifir_amd error");
  }
  return (LhfStatus)st;
}

std::vector<int> g_devices;  // lhfSetDevices; empty = the current device

typedef std::complex<double> zdbl;
typedef std::complex<float> cflt;

// the type a value travels in on the device: single-precision data is widened (exactly) on the way to HBM
template <class V>
struct Wide {
  typedef V type;
  static const bool single = false;
};
template <>
struct Wide<float> {
  typedef double type;
  static const bool single = true;
};
template <>
struct Wide<cflt> {
  typedef zdbl type;
  static const bool single = true;
};
// -> pointer to n values of type W: the source itself when it already is W, a widened copy in buf otherwise
template <class W>
const W *widen(const W *src, std::size_t, std::vector<W> &) {
  return src;
}
template <class N, class W>
N narrow(const W &w) {
  return N(w);
}
template <>
cflt narrow<cflt, zdbl>(const zdbl &w) {
  return cflt((float)w.real(), (float)w.imag());
}
template <class W, class S>
const W *widen(const S *src, std::size_t n, std::vector<W> &buf) {
  buf.resize(n);
  for (std::size_t i = 0; i < n; ++i) buf[i] = W(src[i]);
  return (n && src) ? buf.data() : nullptr;
}

template <class V>
struct MatrixRec {  // never owns: aliases the caller's arrays (libhifir.cpp:316-321)
  LhfIndPtr *indptr;
  LhfInt *indices;
  V *vals;
  std::size_t n;
  bool rowmajor;
};

template <class V>
struct HifRec {
  typedef hif::HIF<V, LhfInt, LhfIndPtr> hif_t;
  hif_t *M;                    // the host factorization (NULL for a handle made by lhf?LoadHierarchy)
  MatrixRec<V> *A;             // borrowed (libhifir.cpp:413)
  MatrixRec<typename Wide<V>::type> *Aw;  // single-precision handles: the double matrix of lhfsdUpdate / lhfczUpdate (Ad / Az)
  const void *resident;        // which of the two the devices hold for iterative refinement
  std::vector<HifAmdHdl> gpu;  // one resident copy of the hierarchy per device of lhfSetDevices
  double rrqr_cond;
  std::size_t nrows;
};

template <class V>
struct ValueTag;
template <>
struct ValueTag<double> {
  static const HifAmdValueType vt = HIFAMD_D;
};
template <>
struct ValueTag<std::complex<double>> {
  static const HifAmdValueType vt = HIFAMD_Z;
};
template <>
struct ValueTag<float> {
  static const HifAmdValueType vt = HIFAMD_D;
};
template <>
struct ValueTag<std::complex<float>> {
  static const HifAmdValueType vt = HIFAMD_Z;
};

// ---- matrices ----------------------------------------------------------------------------------------------
template <class Rec, class V>
Rec *matrix_create(int is_rowmajor, std::size_t n, const LhfIndPtr *indptr, const LhfInt *indices, const V *vals) {
  Rec *m = new (std::nothrow) Rec();
  if (!m) return nullptr;
  m->rowmajor = is_rowmajor != 0;
  m->indptr = nullptr, m->indices = nullptr, m->vals = nullptr, m->n = 0;
  if (n && indptr && indices && vals) {
    m->indptr = const_cast<LhfIndPtr *>(indptr);
    m->indices = const_cast<LhfInt *>(indices);
    m->vals = const_cast<V *>(vals);
    m->n = n;
  }
  return m;
}
template <class Rec, class V>
LhfStatus matrix_wrap(Rec *m, std::size_t n, const LhfIndPtr *indptr, const LhfInt *indices, const V *vals) {
  if (!m) return LHF_NULL_OBJ;
  if (!(n && indptr && indices && vals)) return LHF_NULL_OBJ;
  m->indptr = const_cast<LhfIndPtr *>(indptr);
  m->indices = const_cast<LhfInt *>(indices);
  m->vals = const_cast<V *>(vals);
  m->n = n;
  return LHF_SUCCESS;
}
template <class Rec>
std::size_t matrix_nnz(const Rec *m) {
  return (m && m->indptr) ? (std::size_t)(m->indptr[m->n] - m->indptr[0]) : 0;
}
// MatrixMarket -> the caller's (pre-sized) arrays, through the reference's own reader (CompressedStorage.hpp:864)
template <class V, class Rec>
LhfStatus matrix_read(const char *fname, Rec *m) {
  if (!m) return LHF_NULL_OBJ;
  try {
    if (m->rowmajor) {
      const auto A = hif::CRS<V, LhfInt, LhfIndPtr>::from_mm(fname);
      if (A.nrows() != m->n || A.ncols() != m->n) return LHF_MISMATCHED_SIZES;
      std::copy(A.row_start().cbegin(), A.row_start().cend(), m->indptr);
      std::copy(A.col_ind().cbegin(), A.col_ind().cend(), m->indices);
      std::copy(A.vals().cbegin(), A.vals().cend(), m->vals);
    } else {
      const auto A = hif::CCS<V, LhfInt, LhfIndPtr>::from_mm(fname);
      if (A.nrows() != m->n || A.ncols() != m->n) return LHF_MISMATCHED_SIZES;
      std::copy(A.col_start().cbegin(), A.col_start().cend(), m->indptr);
      std::copy(A.row_ind().cbegin(), A.row_ind().cend(), m->indices);
      std::copy(A.vals().cbegin(), A.vals().cend(), m->vals);
    }
  } catch (const std::exception &e) {
    return fail(e.what());
  }
  return LHF_SUCCESS;
}
template <class V>
LhfStatus vector_read(const char *fname, std::size_t n, V *v) {
  try {
    const auto vec = hif::Array<V>::from_mm(fname);
    if (vec.size() != n) return LHF_MISMATCHED_SIZES;
    std::copy(vec.cbegin(), vec.cend(), v);
  } catch (const std::exception &e) {
    return fail(e.what());
  }
  return LHF_SUCCESS;
}

// ---- hierarchy -> HBM ----------------------------------------------------------------------------------------
hif::Params params_from(const double par[]) {  // the LHF_* slots of libhifir.h:94-117 onto hif::Options
  hif::Params p = hif::get_default_params();
  p.tau_L = par[LHF_DROPTOL_L], p.tau_U = par[LHF_DROPTOL_U];
  p.kappa_d = par[LHF_COND_D], p.kappa = par[LHF_COND];
  p.alpha_L = par[LHF_ALPHA_L], p.alpha_U = par[LHF_ALPHA_U];
  p.verbose = (int)par[LHF_VERBOSE], p.reorder = (int)par[LHF_REORDER];
  p.symm_pre_lvls = (int)par[LHF_SYMMPRELVLS], p.threads = (int)par[LHF_THREADS];
  p.rrqr_cond = par[LHF_RRQR_COND], p.pivot = (int)par[LHF_PIVOT], p.beta = par[LHF_BETA];
  p.is_symm = (int)par[LHF_ISSYMM], p.no_pre = (int)par[LHF_NOPRE];
  p.nzp_thres = par[LHF_NZP_THRES], p.dense_thres = (int)par[LHF_DENSE_THRES];
  return p;
}

template <class V>
void release_gpu(HifRec<V> *h) {
  for (HifAmdHdl g : h->gpu) hifamd_destroy(g);
  h->gpu.clear();
}

template <class Arr>
const typename Arr::value_type *data_or_null(const Arr &a) {
  return a.size() ? a.data() : nullptr;
}

// the user's matrix on every device, CRS (a column-major handle is transposed here; values are not conjugated)
template <class V, class MV>
LhfStatus upload_matrix_of(HifRec<V> *h, const MatrixRec<MV> *A) {
  typedef typename Wide<V>::type W;
  if (!A || !A->indptr || h->gpu.empty() || A->n != h->nrows) return LHF_SUCCESS;  // (checked again when refining)
  const std::size_t n = A->n;
  std::vector<int64_t> ip;
  std::vector<int32_t> ix;
  std::vector<MV> vv;
  std::vector<W> wide;
  const int64_t *ipp = (const int64_t *)A->indptr;
  const int32_t *ixp = A->indices;
  const MV *vp = A->vals;
  if (!A->rowmajor) {
    const LhfIndPtr base = A->indptr[0];
    const std::size_t nz = (std::size_t)(A->indptr[n] - base);
    ip.assign(n + 1, 0), ix.resize(nz), vv.resize(nz);
    for (std::size_t k = 0; k < nz; ++k) ++ip[(std::size_t)(A->indices[k] - base) + 1];
    for (std::size_t i = 0; i < n; ++i) ip[i + 1] += ip[i];
    std::vector<int64_t> fill(ip.begin(), ip.end() - 1);
    for (std::size_t j = 0; j < n; ++j)
      for (LhfIndPtr k = A->indptr[j] - base; k < A->indptr[j + 1] - base; ++k) {
        const int64_t pos = fill[(std::size_t)(A->indices[k] - base)]++;
        ix[(std::size_t)pos] = (int32_t)j;
        vv[(std::size_t)pos] = A->vals[k];
      }
    ipp = ip.data(), ixp = ix.data(), vp = vv.data();
  }
  const std::size_t nz = (std::size_t)(A->indptr[n] - A->indptr[0]);
  const W *wp = widen<W>(vp, nz, wide);
  for (HifAmdHdl g : h->gpu) {
    const LhfStatus st = from_amd(hifamd_set_matrix(g, (int64_t)n, ipp, ixp, wp));
    if (st != LHF_SUCCESS) return st;
  }
  h->resident = A;
  return LHF_SUCCESS;
}
template <class V>
LhfStatus upload_matrix(HifRec<V> *h) {
  h->resident = nullptr;
  return upload_matrix_of(h, h->A);
}

// HIFIR_AMD_MAX_NRHS (1 .. 64, default 64): the widest batch a handle of this process will be asked for -- what ship() and
// hif_load() pass to hifamd_finalize.  A single-vector lhf?Solve user may say 1; lhf?ApplyBatch with more columns than
// that is tiled.  (In this build the work arena stays 64 columns wide whatever is asked for: lhf?GetResidentBytes tells.)
static int64_t shim_max_nrhs() {
  const char *e = std::getenv("HIFIR_AMD_MAX_NRHS");
  if (!e || !*e) return 64;
  const long v = std::strtol(e, nullptr, 10);
  return v < 1 ? 1 : (v > 64 ? 64 : v);
}

// M.precs() -> hifamd_add_level / hifamd_set_dense* -> hifamd_finalize, once per device
template <class V>
LhfStatus ship(HifRec<V> *h) {
  release_gpu(h);
  if (h->M->empty()) return fail("MILU-Prec is empty!");
  h->nrows = h->M->nrows();
  std::vector<int> devs = g_devices;
  if (devs.empty()) devs.push_back(-1);
  for (int dev : devs) {
    HifAmdHdl g = nullptr;
    LhfStatus st = from_amd(hifamd_create(ValueTag<V>::vt, dev, &g));
    if (st != LHF_SUCCESS) return st;
    h->gpu.push_back(g);
    typedef typename Wide<V>::type W;
    for (const auto &p : h->M->precs()) {
      // (a single-precision hierarchy: every value widened exactly; the buffers live until the level is imported)
      std::vector<W> bL, bU, bE, bF, bd, bm;
      std::vector<double> bs, bt;
      st = from_amd(hifamd_add_level(
          g, (int64_t)p.m, (int64_t)p.n, (const int64_t *)data_or_null(p.L_B.col_start()), data_or_null(p.L_B.row_ind()),
          widen<W>(data_or_null(p.L_B.vals()), p.L_B.vals().size(), bL), (const int64_t *)data_or_null(p.U_B.col_start()),
          data_or_null(p.U_B.row_ind()), widen<W>(data_or_null(p.U_B.vals()), p.U_B.vals().size(), bU),
          (const int64_t *)data_or_null(p.E.col_start()), data_or_null(p.E.row_ind()),
          widen<W>(data_or_null(p.E.vals()), p.E.vals().size(), bE), (int64_t)(p.F.col_start().size() ? p.F.ncols() : 0),
          (const int64_t *)data_or_null(p.F.col_start()), data_or_null(p.F.row_ind()),
          widen<W>(data_or_null(p.F.vals()), p.F.vals().size(), bF), widen<W>(data_or_null(p.d_B), p.d_B.size(), bd),
          widen<double>(data_or_null(p.s), p.s.size(), bs), widen<double>(data_or_null(p.t), p.t.size(), bt), data_or_null(p.p),
          data_or_null(p.p_inv), data_or_null(p.q), data_or_null(p.q_inv)));
      if (st != LHF_SUCCESS) return st;
      // the UNFACTORED Schur complement of the last level (Prec::inquire_or_export_dense, Prec.hpp:275-303)
      if (!p.dense_solver.empty()) {
        const auto &mat = p.dense_solver.mat_backup();
        const W *mp = widen<W>(mat.array().data(), mat.array().size(), bm);
        if (std::strcmp(p.dense_solver.method(), "LUP") == 0)  // a reference built with HIF_DENSE_MODE=0
          st = from_amd(hifamd_set_dense_lup(g, (int64_t)mat.nrows(), mp));
        else {
          // (QRCP.hpp:110-118: without a user threshold the rank test uses eps^(-2/3) of the hierarchy's OWN scalar type)
          double cond = h->rrqr_cond;
          if (Wide<V>::single && cond <= 0.0) cond = (double)(float)(1.0 / std::pow((double)FLT_EPSILON, 2. / 3));
          st = from_amd(hifamd_set_dense(g, (int64_t)mat.nrows(), mp, cond));
        }
      } else if (!p.symm_dense_solver.empty()) {
        const auto &mat = p.symm_dense_solver.mat_backup();
        st = from_amd(hifamd_set_dense_symm(g, (int64_t)mat.nrows(), widen<W>(mat.array().data(), mat.array().size(), bm), 0));
      }
      if (st != LHF_SUCCESS) return st;
    }
    st = from_amd(hifamd_finalize(g, shim_max_nrhs()));
    if (st != LHF_SUCCESS) return st;
  }
  return upload_matrix(h);
}

template <class V>
LhfStatus factorize_and_ship(HifRec<V> *h, const MatrixRec<V> *S, const double params[]) {
  try {
    if (!h->M) {
      h->M = new typename HifRec<V>::hif_t();
    }
    h->rrqr_cond = params ? params[LHF_RRQR_COND] : 0.0;
    if (params) {
      if (S->rowmajor)
        h->M->template factorize<true>(S->n, S->indptr, S->indices, S->vals, params_from(params));
      else
        h->M->template factorize<false>(S->n, S->indptr, S->indices, S->vals, params_from(params));
    } else {
      if (S->rowmajor)
        h->M->template factorize<true>(S->n, S->indptr, S->indices, S->vals);
      else
        h->M->template factorize<false>(S->n, S->indptr, S->indices, S->vals);
    }
  } catch (const std::exception &e) {
    return fail(e.what());
  }
  return ship(h);
}

template <class Hif, class Mat, class V>
LhfStatus hif_setup(Hif *h, Mat *A, Mat *S, const double params[]) {
  if (!h) return LHF_NULL_OBJ;
  h->A = A ? A : S;  // libhifir.cpp:413
  Mat *fac = S ? S : A;
  if (!fac) return LHF_NULL_OBJ;
  return factorize_and_ship<V>(h, fac, params);
}

template <class Hif, class Mat, class V>
Hif *hif_create(Mat *A, Mat *S, const double params[]) {
  Hif *h = new (std::nothrow) Hif();
  if (!h) return nullptr;
  h->M = nullptr, h->A = nullptr, h->Aw = nullptr, h->resident = nullptr, h->rrqr_cond = 0.0, h->nrows = 0;
  const LhfStatus st = hif_setup<Hif, Mat, V>(h, A, S, params);
  if (st == LHF_HIFIR_ERROR) {  // (a NULL matrix leaves an empty handle behind, as libhifir.cpp:383-396 does)
    release_gpu(h);
    delete h->M;
    delete h;
    return nullptr;
  }
  return h;
}

template <class Hif>
LhfStatus hif_destroy(Hif *h) {
  if (h) {
    release_gpu(h);
    delete h->M;
    delete h;
  }
  return LHF_SUCCESS;
}

template <class Hif, class Mat>
LhfStatus hif_update(Hif *h, Mat *A) {
  if (!h) return LHF_NULL_OBJ;
  h->A = A;
  if (h->A && !h->gpu.empty() && h->A->n != h->nrows) return LHF_MISMATCHED_SIZES;  // libhifir.cpp:423-424
  return upload_matrix(h);
}

// the device half of lhf?Apply: vectors already in the type the device computes in
template <class V>
LhfStatus run_apply(HifRec<V> *h, LhfOperationType op, const typename Wide<V>::type *B, std::size_t nrhs, std::size_t ldb,
                    bool refine, int nirs, const double *betas, int64_t rnk, typename Wide<V>::type *X, std::size_t ldx,
                    int *ir_status) {
  int *status = (refine && betas) ? ir_status : nullptr;  // only the bounded variant reports (libhifir.cpp:189-203)
  const std::size_t nd = std::min(h->gpu.size(), nrhs);
  if (nd <= 1)
    return from_amd(hifamd_apply_batch(h->gpu[0], (HifAmdOp)op, B, (int64_t)ldb, X, (int64_t)ldx, (int64_t)nrhs,
                                       refine ? nirs : 1, refine ? betas : nullptr, rnk, status));
  // RHS-sharded over the devices of lhfSetDevices: contiguous column blocks, one host thread per device
  std::vector<HifAmdStatus> st(nd, HIFAMD_SUCCESS);
  std::vector<std::string> msg(nd);
  std::vector<std::thread> th;
  for (std::size_t d = 0; d < nd; ++d) {
    const std::size_t base = nrhs / nd, rem = nrhs % nd;
    const std::size_t c0 = d * base + std::min(d, rem), nc = base + (d < rem ? 1 : 0);
    th.emplace_back([=, &st, &msg] {
      st[d] = hifamd_apply_batch(h->gpu[d], (HifAmdOp)op, B + c0, (int64_t)ldb, X + c0, (int64_t)ldx, (int64_t)nc,
                                 refine ? nirs : 1, refine ? betas : nullptr, rnk, status ? status + 2 * c0 : nullptr);
      if (st[d] != HIFAMD_SUCCESS) {  // (the library's message is per thread)
        const char *m = hifamd_last_error();
        msg[d] = m ? m : "hifir_amd error";
      }
    });
  }
  for (auto &t : th) t.join();
  for (std::size_t d = 0; d < nd; ++d)
    if (st[d] != HIFAMD_SUCCESS) {
      set_msg(msg[d]);
      return (LhfStatus)st[d];
    }
  return LHF_SUCCESS;
}
// (vectors of the device's type go straight through; single-precision ones are widened into a dense batch [n][nrhs] on the
// way in and rounded on the way out)
template <class V>
LhfStatus run_apply_any(std::true_type, HifRec<V> *h, LhfOperationType op, const typename Wide<V>::type *B, std::size_t nrhs,
                        std::size_t ldb, bool refine, int nirs, const double *betas, int64_t rnk, typename Wide<V>::type *X,
                        std::size_t ldx, int *ir_status) {
  return run_apply(h, op, B, nrhs, ldb, refine, nirs, betas, rnk, X, ldx, ir_status);
}
template <class V, class BV>
LhfStatus run_apply_any(std::false_type, HifRec<V> *h, LhfOperationType op, const BV *B, std::size_t nrhs, std::size_t ldb,
                        bool refine, int nirs, const double *betas, int64_t rnk, BV *X, std::size_t ldx, int *ir_status) {
  typedef typename Wide<V>::type W;
  const std::size_t n = h->nrows;
  std::vector<W> Bw(n * nrhs), Xw(n * nrhs);
  for (std::size_t i = 0; i < n; ++i)
    for (std::size_t c = 0; c < nrhs; ++c) Bw[i * nrhs + c] = W(B[i * ldb + c]);
  const LhfStatus st = run_apply(h, op, Bw.data(), nrhs, nrhs, refine, nirs, betas, rnk, Xw.data(), nrhs, ir_status);
  if (st != LHF_SUCCESS) return st;
  for (std::size_t i = 0; i < n; ++i)
    for (std::size_t c = 0; c < nrhs; ++c) X[i * ldx + c] = narrow<BV>(Xw[i * nrhs + c]);
  return LHF_SUCCESS;
}

// lhf?Apply for nrhs columns (nrhs = 1, ld = 1: the reference's entry point), libhifir.cpp:447-472
// BV: the type of the caller's vectors -- V itself, or (lhfsd* / lhfcz*, :1192-1284) the wide type of a single-precision
// hierarchy
template <class V, class BV>
LhfStatus hif_apply(HifRec<V> *h, LhfOperationType op, const BV *B, std::size_t nrhs, std::size_t ldb, int nirs,
                    const double *betas, int rank, BV *X, std::size_t ldx, int *ir_status) {
  if (!h) return LHF_NULL_OBJ;
  if (h->gpu.empty()) return fail("MILU-Prec is empty!");
  if (op != LHF_S && op != LHF_SH && op != LHF_M && op != LHF_MH) return fail("unknown operation tag");
  const bool prod = (op == LHF_M || op == LHF_MH);
  const bool refine = !prod && nirs > 1;
  int64_t rnk = 0;  // the direct solve and the product never see `rank` (libhifir.cpp:459-461 pass no r: numerical rank)
  if (Wide<V>::single && h->M && h->M->schur_size() && h->M->schur_rank() < h->M->schur_size())
    rnk = (int64_t)h->M->schur_rank();  // (the numerical rank the reference's single-precision QRCP found, not the wide one's)
  if (refine) {
    // the matrix of the residual: A for vectors of the hierarchy's own type, Ad / Az for the mixed entry points
    const bool own = std::is_same<BV, V>::value;
    const std::size_t an = own ? (h->A ? h->A->n : 0) : (h->Aw ? h->Aw->n : 0);
    const void *want = own ? (const void *)h->A : (const void *)h->Aw;
    if (!want) return LHF_NULL_OBJ;
    if (h->nrows != an) return LHF_MISMATCHED_SIZES;
    if (h->resident != want) {
      const LhfStatus su = own ? upload_matrix_of(h, h->A) : upload_matrix_of(h, h->Aw);
      if (su != LHF_SUCCESS) return su;
    }
    rnk = rank == LHF_DEFAULT_RANK ? -1 : rank;  // :453-455
  }
  return run_apply_any<V>(typename std::is_same<BV, typename Wide<V>::type>::type(), h, op, B, nrhs, ldb, refine, nirs, betas, rnk, X,
                          ldx, ir_status);
}

// lhf?ApplyBatchDev: the blocks of an RHS-sharded batch stay in the HBM of their devices (block d on replica d, i.e. on
// device ids[d] of lhfSetDevices); direct operators only; enqueue-and-return on the replicas' own streams
template <class V>
LhfStatus hif_apply_dev(HifRec<V> *h, LhfOperationType op, int nblocks, const V *const *B, const std::size_t *ncols,
                        const std::size_t *ldb, V *const *X, const std::size_t *ldx) {
  if (!h) return LHF_NULL_OBJ;
  if (h->gpu.empty()) return fail("MILU-Prec is empty!");
  if (op != LHF_S && op != LHF_SH && op != LHF_M && op != LHF_MH) return fail("unknown operation tag");
  if (nblocks < 1 || !B || !X || !ncols || !ldb || !ldx) return LHF_NULL_OBJ;
  if ((std::size_t)nblocks > h->gpu.size()) return LHF_MISMATCHED_SIZES;  // one replica (lhfSetDevices) per block
  for (int d = 0; d < nblocks; ++d) {
    if (!ncols[d]) continue;  // (an empty block: nothing to do on that device)
    if (!B[d] || !X[d]) return LHF_NULL_OBJ;
    if (ldb[d] < ncols[d] || ldx[d] < ncols[d]) return LHF_MISMATCHED_SIZES;
    const LhfStatus st = from_amd(hifamd_apply_batch_dev(h->gpu[(std::size_t)d], (HifAmdOp)op, B[d], (int64_t)ldb[d], X[d],
                                                         (int64_t)ldx[d], (int64_t)ncols[d], 0, nullptr));
    if (st != LHF_SUCCESS) return st;
  }
  return LHF_SUCCESS;
}
template <class V>
LhfStatus hif_gather_dev(HifRec<V> *h, int nblocks, const V *const *X, const std::size_t *ncols, const std::size_t *ldx,
                         V *dst, std::size_t ldd) {
  if (!h) return LHF_NULL_OBJ;
  if (h->gpu.empty()) return fail("MILU-Prec is empty!");
  if (nblocks < 1 || !X || !ncols || !ldx || !dst) return LHF_NULL_OBJ;
  if ((std::size_t)nblocks > h->gpu.size()) return LHF_MISMATCHED_SIZES;
  std::size_t col0 = 0;
  for (int d = 0; d < nblocks; ++d) {
    if (ncols[d]) {
      const LhfStatus st = from_amd(hifamd_copy_columns_dev(h->gpu[(std::size_t)d], X[d], (int64_t)ldx[d], (int64_t)ncols[d], dst,
                                                            (int64_t)ldd, (int64_t)col0));
      if (st != LHF_SUCCESS) return st;
    }
    col0 += ncols[d];
  }
  return LHF_SUCCESS;
}
template <class V>
LhfStatus hif_sync_devices(HifRec<V> *h) {
  if (!h) return LHF_NULL_OBJ;
  for (HifAmdHdl g : h->gpu) {
    const LhfStatus st = from_amd(hifamd_sync(g));
    if (st != LHF_SUCCESS) return st;
  }
  return LHF_SUCCESS;
}

template <class V>
void hif_stats(const HifRec<V> *h, std::size_t stats[]) {  // the nine slots of lhf?GetStats (libhifir.h:700-716)
  for (int i = 0; i < 9; ++i) stats[i] = 0;
  if (!h) return;
  if (h->M && !h->M->empty()) {
    const auto &M = *h->M;
    stats[0] = M.nnz(), stats[1] = M.stats(0), stats[2] = M.stats(1), stats[3] = M.stats(4), stats[4] = M.stats(5);
    stats[5] = M.levels(), stats[6] = M.rank(), stats[7] = M.schur_rank(), stats[8] = M.schur_size();
  } else if (!h->gpu.empty()) {  // loaded handle: what the hierarchy itself tells (the deferral counters are factorization history)
    HifAmdHdl g = h->gpu[0];
    stats[0] = (std::size_t)hifamd_nnz(g), stats[5] = (std::size_t)hifamd_levels(g);
    stats[7] = (std::size_t)hifamd_schur_rank(g), stats[8] = (std::size_t)hifamd_schur_size(g);
    stats[6] = (std::size_t)hifamd_nrows(g) - (stats[8] - stats[7]);
  }
}

// what one replica of the handle keeps in HBM (bytes): [0] factors + plan arrays, [1] explicit operators (block inverses,
// combined tops, tail operator), [2] coefficient tiles of the component bands, [3] work arena, [4] arena columns,
// [5] the batch width the handle was finalized for
template <class V>
LhfStatus hif_resident(const HifRec<V> *h, std::size_t out[6]) {
  if (!h || !out) return fail("NULL argument");
  for (int i = 0; i < 6; ++i) out[i] = 0;
  if (h->gpu.empty()) return fail("the handle has no hierarchy in HBM yet (lhf?Setup / lhf?LoadHierarchy)");
  double v[19];
  const int nv = hifamd_stats_ext(h->gpu[0], v, 19);
  if (nv < 19) return fail("library too old for lhf?GetResidentBytes");
  out[0] = (std::size_t)v[17], out[1] = (std::size_t)(v[2] + v[3] + v[4]), out[2] = (std::size_t)v[16], out[3] = (std::size_t)v[14];
  out[4] = (std::size_t)v[15], out[5] = (std::size_t)v[18];
  return LHF_SUCCESS;
}

template <class Hif, class V>
Hif *hif_load(const char *path) {
  if (!path) {
    set_msg("NULL path");
    return nullptr;
  }
  Hif *h = new (std::nothrow) Hif();
  if (!h) return nullptr;
  h->M = nullptr, h->A = nullptr, h->Aw = nullptr, h->resident = nullptr, h->rrqr_cond = 0.0, h->nrows = 0;
  std::vector<int> devs = g_devices;
  if (devs.empty()) devs.push_back(-1);
  for (int dev : devs) {
    HifAmdHdl g = nullptr;
    LhfStatus st = from_amd(hifamd_load(path, dev, &g));
    if (st == LHF_SUCCESS) {
      h->gpu.push_back(g);
      if (hifamd_value_type(g) != (int)ValueTag<V>::vt) {
        set_msg("the hierarchy file holds the other value type (real vs complex)");
        st = LHF_BAD_PREC;
      }
    }
    if (st == LHF_SUCCESS) st = from_amd(hifamd_finalize(g, shim_max_nrhs()));
    if (st != LHF_SUCCESS) {
      hif_destroy(h);
      return nullptr;
    }
  }
  h->nrows = (std::size_t)hifamd_nrows(h->gpu[0]);
  return h;
}

}  // namespace

// ---- the opaque C structs of libhifir.h ------------------------------------------------------------------------
struct LhfdMatrix : MatrixRec<double> {};
struct LhfsMatrix : MatrixRec<float> {};
struct LhfzMatrix : MatrixRec<std::complex<double>> {};
struct LhfcMatrix : MatrixRec<std::complex<float>> {};
struct LhfdHif : HifRec<double> {};
struct LhfzHif : HifRec<std::complex<double>> {};
struct LhfsHif : HifRec<float> {};
struct LhfcHif : HifRec<std::complex<float>> {};

extern "C" {

void lhfGetVersions(int versions[]) {
  versions[0] = HIF_GLOBAL_VERSION, versions[1] = HIF_MAJOR_VERSION, versions[2] = HIF_MINOR_VERSION;
}
// declared by the reference (libhifir.h:245,250) but defined nowhere in libhifir.cpp: harmless no-ops here
void lhfEnableWarning(void) {}
void lhfDisableWarning(void) {}

const char *lhfGetErrorMsg(void) {  // one shot: hand the pending message out and clear it (libhifir.cpp:224-229)
  g_msg[0].swap(g_msg[1]);
  g_msg[1].clear();
  return g_msg[0].empty() ? NULL : g_msg[0].c_str();
}

LhfStatus lhfSetDefaultParams(double params[]) {
  const hif::Params &d = hif::DEFAULT_PARAMS;
  params[LHF_DROPTOL_L] = d.tau_L, params[LHF_DROPTOL_U] = d.tau_U;
  params[LHF_COND_D] = d.kappa_d, params[LHF_COND] = d.kappa;
  params[LHF_ALPHA_L] = d.alpha_L, params[LHF_ALPHA_U] = d.alpha_U;
  params[LHF_VERBOSE] = d.verbose, params[LHF_REORDER] = d.reorder;
  params[LHF_SYMMPRELVLS] = d.symm_pre_lvls, params[LHF_THREADS] = d.threads;
  params[LHF_RRQR_COND] = d.rrqr_cond, params[LHF_PIVOT] = d.pivot, params[LHF_BETA] = d.beta;
  params[LHF_ISSYMM] = d.is_symm, params[LHF_NOPRE] = d.no_pre;
  params[LHF_NZP_THRES] = d.nzp_thres, params[LHF_DENSE_THRES] = d.dense_thres;
  return LHF_SUCCESS;
}
LhfStatus lhfSetDroptol(const double droptol, double params[]) {
  params[LHF_DROPTOL_L] = params[LHF_DROPTOL_U] = droptol;
  return LHF_SUCCESS;
}
LhfStatus lhfSetAlpha(const double alpha, double params[]) {
  params[LHF_ALPHA_L] = params[LHF_ALPHA_U] = alpha;
  return LHF_SUCCESS;
}
LhfStatus lhfSetKappa(const double kappa, double params[]) {
  params[LHF_COND] = params[LHF_COND_D] = kappa;
  return LHF_SUCCESS;
}

LhfStatus lhfQueryMmFile(const char *fname, int *is_sparse, int *is_real, size_t *nrows, size_t *ncols, size_t *nnz) {
  std::FILE *f = fname ? std::fopen(fname, "r") : nullptr;
  if (!f) return LHF_NULL_OBJ;
  LhfStatus st = LHF_SUCCESS;
  try {
    bool sparse = false, real = false;
    int type_id = 0;
    hif::internal::mm_read_firstline(f, sparse, real, type_id);
    *is_sparse = sparse, *is_real = real;
    if (sparse) {
      hif::internal::mm_read_sparse_size(f, *nrows, *ncols, *nnz);
    } else {
      *nnz = 0;
      hif::internal::mm_read_dense_size(f, *nrows, *ncols);
    }
  } catch (const std::exception &e) {
    st = fail(e.what());
  }
  std::fclose(f);
  return st;
}

// ---- additive: devices ---------------------------------------------------------------------------------------
int lhfGetDeviceCount(void) { return hifamd_device_count(); }
LhfStatus lhfSetDevices(const int *ids, int n) {
  if (!ids || n <= 0) {
    g_devices.clear();
    return LHF_SUCCESS;
  }
  const int cnt = hifamd_device_count();
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= cnt) return LHF_MISMATCHED_SIZES;
  g_devices.assign(ids, ids + n);
  return LHF_SUCCESS;
}

// ---- the per-type entry points ---------------------------------------------------------------------------------
// T: type letter, V: C++ value type, CV: the C value type of the header
#define LHF_MATRIX_API(T, V, CV)                                                                                      \
  Lhf##T##MatrixHdl lhf##T##CreateMatrix(const int is_rowmajor, const size_t n, const LhfIndPtr *indptr,              \
                                         const LhfInt *indices, const CV *vals) {                                     \
    return matrix_create<Lhf##T##Matrix, V>(is_rowmajor, n, indptr, indices, (const V *)vals);                        \
  }                                                                                                                   \
  LhfStatus lhf##T##DestroyMatrix(Lhf##T##MatrixHdl mat) {                                                            \
    delete mat;                                                                                                       \
    return LHF_SUCCESS;                                                                                               \
  }                                                                                                                   \
  size_t lhf##T##GetMatrixSize(const Lhf##T##MatrixHdl mat) { return mat ? mat->n : 0; }                              \
  size_t lhf##T##GetMatrixNnz(const Lhf##T##MatrixHdl mat) { return matrix_nnz(mat); }                                \
  LhfStatus lhf##T##ReadSparse(const char *fname, Lhf##T##MatrixHdl mat) { return matrix_read<V>(fname, mat); }       \
  LhfStatus lhf##T##ReadVector(const char *fname, const size_t n, CV *v) { return vector_read<V>(fname, n, (V *)v); } \
  LhfStatus lhf##T##WrapMatrix(Lhf##T##MatrixHdl mat, const size_t n, const LhfIndPtr *indptr, const LhfInt *indices, \
                               const CV *vals) {                                                                      \
    return matrix_wrap<Lhf##T##Matrix, V>(mat, n, indptr, indices, (const V *)vals);                                  \
  }

LHF_MATRIX_API(d, double, double)
LHF_MATRIX_API(s, float, float)
LHF_MATRIX_API(z, zdbl, double _Complex)
LHF_MATRIX_API(c, cflt, float _Complex)

#define LHF_HIF_API(T, V, CV)                                                                                         \
  Lhf##T##HifHdl lhf##T##Create(const Lhf##T##MatrixHdl A, const Lhf##T##MatrixHdl S, const double params[]) {        \
    return hif_create<Lhf##T##Hif, Lhf##T##Matrix, V>(A, S, params);                                                  \
  }                                                                                                                   \
  LhfStatus lhf##T##Destroy(Lhf##T##HifHdl hif) { return hif_destroy(hif); }                                          \
  LhfStatus lhf##T##Setup(Lhf##T##HifHdl hif, const Lhf##T##MatrixHdl A, const Lhf##T##MatrixHdl S,                   \
                          const double params[]) {                                                                    \
    return hif_setup<Lhf##T##Hif, Lhf##T##Matrix, V>(hif, A, S, params);                                              \
  }                                                                                                                   \
  LhfStatus lhf##T##Update(Lhf##T##HifHdl hif, const Lhf##T##MatrixHdl A) { return hif_update(hif, A); }              \
  LhfStatus lhf##T##Refactorize(Lhf##T##HifHdl hif, const Lhf##T##MatrixHdl S, const double params[]) {               \
    if (!hif || !S) return LHF_NULL_OBJ;                                                                              \
    return factorize_and_ship<V>(hif, S, params);                                                                     \
  }                                                                                                                   \
  LhfStatus lhf##T##Apply(const Lhf##T##HifHdl hif, const LhfOperationType op, const CV *b, const int nirs,           \
                          const double *betas, const int rank, CV *x, int *ir_status) {                               \
    return hif_apply<V, V>(hif, op, (const V *)b, 1, 1, nirs, betas, rank, (V *)x, 1, ir_status);                           \
  }                                                                                                                   \
  LhfStatus lhf##T##Solve(const Lhf##T##HifHdl hif, const CV *b, CV *x) {                                             \
    return hif_apply<V, V>(hif, LHF_S, (const V *)b, 1, 1, 1, nullptr, 0, (V *)x, 1, nullptr);                              \
  }                                                                                                                   \
  LhfStatus lhf##T##ApplyBatch(const Lhf##T##HifHdl hif, const LhfOperationType op, const CV *B, const size_t nrhs,   \
                               const size_t ldb, const int nirs, const double *betas, const int rank, CV *X,          \
                               const size_t ldx, int *ir_status) {                                                    \
    if (!nrhs || ldb < nrhs || ldx < nrhs) return hif ? LHF_MISMATCHED_SIZES : LHF_NULL_OBJ;                          \
    return hif_apply<V, V>(hif, op, (const V *)B, nrhs, ldb, nirs, betas, rank, (V *)X, ldx, ir_status);                    \
  }                                                                                                                   \
  LhfStatus lhf##T##ApplyBatchDev(const Lhf##T##HifHdl hif, const LhfOperationType op, const int nblocks,             \
                                  const CV *const *B_dev, const size_t *ncols, const size_t *ldb, CV *const *X_dev,   \
                                  const size_t *ldx) {                                                                \
    return hif_apply_dev<V>(hif, op, nblocks, (const V *const *)B_dev, ncols, ldb, (V *const *)X_dev, ldx);           \
  }                                                                                                                   \
  LhfStatus lhf##T##GatherBatchDev(const Lhf##T##HifHdl hif, const int nblocks, const CV *const *X_dev,               \
                                   const size_t *ncols, const size_t *ldx, CV *dst_dev, const size_t ldd) {           \
    return hif_gather_dev<V>(hif, nblocks, (const V *const *)X_dev, ncols, ldx, (V *)dst_dev, ldd);                   \
  }                                                                                                                   \
  LhfStatus lhf##T##SyncDevices(const Lhf##T##HifHdl hif) { return hif_sync_devices<V>(hif); }                        \
  LhfStatus lhf##T##GetStats(const Lhf##T##HifHdl hif, size_t stats[]) {                                              \
    hif_stats<V>(hif, stats);                                                                                         \
    return LHF_SUCCESS;                                                                                               \
  }                                                                                                                   \
  size_t lhf##T##GetNnz(const Lhf##T##HifHdl hif) {                                                                   \
    size_t s[9];                                                                                                      \
    hif_stats<V>(hif, s);                                                                                             \
    return s[0];                                                                                                      \
  }                                                                                                                   \
  size_t lhf##T##GetLevels(const Lhf##T##HifHdl hif) {                                                                \
    size_t s[9];                                                                                                      \
    hif_stats<V>(hif, s);                                                                                             \
    return s[5];                                                                                                      \
  }                                                                                                                   \
  size_t lhf##T##GetSchurSize(const Lhf##T##HifHdl hif) {                                                             \
    size_t s[9];                                                                                                      \
    hif_stats<V>(hif, s);                                                                                             \
    return s[8];                                                                                                      \
  }                                                                                                                   \
  size_t lhf##T##GetSchurRank(const Lhf##T##HifHdl hif) {                                                             \
    size_t s[9];                                                                                                      \
    hif_stats<V>(hif, s);                                                                                             \
    return s[7];                                                                                                      \
  }                                                                                                                   \
  LhfStatus lhf##T##SaveHierarchy(const Lhf##T##HifHdl hif, const char *path) {                                       \
    if (!hif) return LHF_NULL_OBJ;                                                                                    \
    if (hif->gpu.empty()) return fail("MILU-Prec is empty!");                                                         \
    /* (HIFIR_AMD_SAVE_ANALYSIS=1: with the analysis trailer -- the other ranks of a job skip the host analysis) */     \
    const char *ana = std::getenv("HIFIR_AMD_SAVE_ANALYSIS");                                                          \
    return from_amd(hifamd_save_ex(hif->gpu[0], path, (ana && ana[0] && ana[0] != '0') ? HIFAMD_SAVE_ANALYSIS : 0));   \
  }                                                                                                                   \
  Lhf##T##HifHdl lhf##T##LoadHierarchy(const char *path) { return hif_load<Lhf##T##Hif, V>(path); }                   \
  LhfStatus lhf##T##GetResidentBytes(const Lhf##T##HifHdl hif, size_t bytes[6]) {                                     \
    if (!hif) return LHF_NULL_OBJ;                                                                                    \
    return hif_resident<V>(hif, bytes);                                                                               \
  }

LHF_HIF_API(d, double, double)
LHF_HIF_API(z, zdbl, double _Complex)

// single precision: the reference's twelve entry points per family (no additive ones), and the mixed sd / cz entry points
// that apply a single-precision hierarchy to double vectors (libhifir.cpp:1185-1284)
#define LHF_SINGLE_API(T, V, CV)                                                                                      \
  Lhf##T##HifHdl lhf##T##Create(const Lhf##T##MatrixHdl A, const Lhf##T##MatrixHdl S, const double params[]) {        \
    return hif_create<Lhf##T##Hif, Lhf##T##Matrix, V>(A, S, params);                                                  \
  }                                                                                                                   \
  LhfStatus lhf##T##Destroy(Lhf##T##HifHdl hif) { return hif_destroy(hif); }                                          \
  LhfStatus lhf##T##Setup(Lhf##T##HifHdl hif, const Lhf##T##MatrixHdl A, const Lhf##T##MatrixHdl S,                   \
                          const double params[]) {                                                                    \
    return hif_setup<Lhf##T##Hif, Lhf##T##Matrix, V>(hif, A, S, params);                                              \
  }                                                                                                                   \
  LhfStatus lhf##T##Update(Lhf##T##HifHdl hif, const Lhf##T##MatrixHdl A) { return hif_update(hif, A); }              \
  LhfStatus lhf##T##Refactorize(Lhf##T##HifHdl hif, const Lhf##T##MatrixHdl S, const double params[]) {               \
    if (!hif || !S) return LHF_NULL_OBJ;                                                                              \
    return factorize_and_ship<V>(hif, S, params);                                                                     \
  }                                                                                                                   \
  LhfStatus lhf##T##Apply(const Lhf##T##HifHdl hif, const LhfOperationType op, const CV *b, const int nirs,           \
                          const double *betas, const int rank, CV *x, int *ir_status) {                               \
    return hif_apply<V, V>(hif, op, (const V *)b, 1, 1, nirs, betas, rank, (V *)x, 1, ir_status);                     \
  }                                                                                                                   \
  LhfStatus lhf##T##Solve(const Lhf##T##HifHdl hif, const CV *b, CV *x) {                                             \
    return hif_apply<V, V>(hif, LHF_S, (const V *)b, 1, 1, 1, nullptr, 0, (V *)x, 1, nullptr);                        \
  }                                                                                                                   \
  LhfStatus lhf##T##GetStats(const Lhf##T##HifHdl hif, size_t stats[]) {                                              \
    hif_stats<V>(hif, stats);                                                                                         \
    return LHF_SUCCESS;                                                                                               \
  }                                                                                                                   \
  size_t lhf##T##GetNnz(const Lhf##T##HifHdl hif) {                                                                   \
    size_t s[9];                                                                                                      \
    hif_stats<V>(hif, s);                                                                                             \
    return s[0];                                                                                                      \
  }                                                                                                                   \
  size_t lhf##T##GetLevels(const Lhf##T##HifHdl hif) {                                                                \
    size_t s[9];                                                                                                      \
    hif_stats<V>(hif, s);                                                                                             \
    return s[5];                                                                                                      \
  }                                                                                                                   \
  size_t lhf##T##GetSchurSize(const Lhf##T##HifHdl hif) {                                                             \
    size_t s[9];                                                                                                      \
    hif_stats<V>(hif, s);                                                                                             \
    return s[8];                                                                                                      \
  }                                                                                                                   \
  size_t lhf##T##GetSchurRank(const Lhf##T##HifHdl hif) {                                                             \
    size_t s[9];                                                                                                      \
    hif_stats<V>(hif, s);                                                                                             \
    return s[7];                                                                                                      \
  }

LHF_SINGLE_API(s, float, float)
LHF_SINGLE_API(c, cflt, float _Complex)

// (the double matrix of the mixed entry points is only remembered here, :1185-1190; it reaches the devices when an lhfsd /
// lhfcz refinement asks for it)
LhfStatus lhfsdUpdate(LhfsHifHdl hif, LhfdMatrixHdl A) {
  if (!hif) return LHF_NULL_OBJ;
  hif->Aw = A;
  if (hif->resident && hif->resident != (const void *)hif->A) hif->resident = nullptr;
  if (A && !hif->gpu.empty() && A->n != hif->nrows) return LHF_MISMATCHED_SIZES;
  return LHF_SUCCESS;
}
LhfStatus lhfsdApply(const LhfsHifHdl hif, const LhfOperationType op, const double *b, const int nirs, const double *betas,
                     const int rank, double *x, int *ir_status) {
  return hif_apply<float, double>(hif, op, b, 1, 1, nirs, betas, rank, x, 1, ir_status);
}
LhfStatus lhfsdSolve(const LhfsHifHdl hif, const double *b, double *x) {
  return hif_apply<float, double>(hif, LHF_S, b, 1, 1, 1, nullptr, 0, x, 1, nullptr);
}
LhfStatus lhfczUpdate(LhfcHifHdl hif, LhfzMatrixHdl A) {
  if (!hif) return LHF_NULL_OBJ;
  hif->Aw = A;
  if (hif->resident && hif->resident != (const void *)hif->A) hif->resident = nullptr;
  if (A && !hif->gpu.empty() && A->n != hif->nrows) return LHF_MISMATCHED_SIZES;
  return LHF_SUCCESS;
}
LhfStatus lhfczApply(const LhfcHifHdl hif, const LhfOperationType op, const double _Complex *b, const int nirs,
                     const double *betas, const int rank, double _Complex *x, int *ir_status) {
  return hif_apply<cflt, zdbl>(hif, op, (const zdbl *)b, 1, 1, nirs, betas, rank, (zdbl *)x, 1, ir_status);
}
LhfStatus lhfczSolve(const LhfcHifHdl hif, const double _Complex *b, double _Complex *x) {
  return hif_apply<cflt, zdbl>(hif, LHF_S, (const zdbl *)b, 1, 1, 1, nullptr, 0, (zdbl *)x, 1, nullptr);
}

}  // extern "C"
