// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin extern "C" driver around the *real* reference (hifirworks/hifir, header-only C++11),
// compiled from the sources where they lie under /root/reference/src by oracle/Makefile into
// oracle/_ref/libhifref.so (git-ignored; travels to the GPU box as a prebuilt binary only).
// Nothing from the reference is copied into this repository: this file only *calls* its public
// API (hif::HIF<>::factorize/solve/mmultiply/hifir, hif::Prec fields) so that
//   * the CPU restatement in oracle/hif_oracle.c can be validated against the real thing, and
//   * golden fixtures (tests/golden/) can be generated from it (tests/golden/make_golden.py).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// Reference API used (file:line relative to /root/reference):
//   HIF::factorize<IsCrs>(n, indptr, indices, vals, params)  src/hif/builder.hpp:388
//   HIF::solve(b, x, trans, r)                               src/hif/builder.hpp:410
//   HIF::mmultiply(x, y, trans, r)                           src/hif/builder.hpp:503
//   HIF::hifir(A, b, N, x) / (A, b, N, betas, x)             src/hif/builder.hpp:459,482
//   Prec public fields m,n,L_B,d_B,U_B,E,F,s,t,p,p_inv,q,q_inv,dense_solver
//                                                            src/hif/alg/Prec.hpp:309-323
//   CCS::col_start()/row_ind()/vals()                        src/hif/ds/CompressedStorage.hpp:1910-1915
//   CCS kernels solve_as_strict_lower/upper, multiply_nt_low :2268,:2357,:2079
//   QRCP::mat_backup(), rank()                               src/hif/small_scale/QRCP.hpp
#include <complex>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#define HIF_THROW 1
#include <hifir.hpp>
// the reference's right-preconditioned GMRES driver (an example header, not library code):
// gmres_hif(A, b, M, restart, rtol, maxit, verbose, full_rank)   examples/advanced/gmres.hpp:19-123
#include <gmres.hpp>

namespace {

std::string g_err;

template <class T>
struct Ref {
  typedef hif::HIF<T, int, std::ptrdiff_t> hif_t;
  typedef typename hif_t::prec_type prec_t;
  typedef hif::CRS<T, int, std::ptrdiff_t> crs_t;
  hif_t M;
  std::vector<const prec_t *> lv;
  // the user matrix (own copy, for hifir)
  std::vector<std::ptrdiff_t> ip;
  std::vector<int> ind;
  std::vector<T> val;
  size_t n = 0;
};

// params[]: 0 tau, 1 kappa(_d), 2 alpha, 3 dense_thres (<=0: keep default), 4 rrqr_cond,
//           5 is_symm (symm_level_factorize, builder.hpp:540-541), 6 spd (Options::spd, SYEIG truncation)
hif::Params make_params(const double *params) {
  hif::Params p = hif::DEFAULT_PARAMS;
  p.verbose = hif::VERBOSE_NONE;
  if (params) {
    if (params[0] > 0) p.tau_L = p.tau_U = params[0];
    if (params[1] > 0) p.kappa = p.kappa_d = params[1];
    if (params[2] > 0) p.alpha_L = p.alpha_U = params[2];
    if (params[3] > 0) p.dense_thres = (int)params[3];
    if (params[4] > 0) p.rrqr_cond = params[4];
    p.is_symm = params[5] != 0.0 ? 1 : 0;
    p.spd = (int)params[6];
  }
  return p;
}

template <class T>
void *do_factorize(size_t n, const int64_t *indptr, const int *indices, const T *vals,
                   const double *params) {
  try {
    auto *r = new Ref<T>();
    r->n = n;
    r->ip.assign(indptr, indptr + n + 1);
    r->ind.assign(indices, indices + indptr[n]);
    r->val.assign(vals, vals + indptr[n]);
    hif::Params p = make_params(params);
    r->M.template factorize<true>(n, r->ip.data(), r->ind.data(), r->val.data(), p);
    for (const auto &pr : r->M.precs()) r->lv.push_back(&pr);
    return r;
  } catch (const std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}

template <class T>
void level_sizes(void *h, int l, int64_t *out) {
  auto *r = (Ref<T> *)h;
  const auto &p = *r->lv[l];
  out[0] = p.m;
  out[1] = p.n;
  out[2] = p.L_B.nnz();
  out[3] = p.U_B.nnz();
  out[4] = p.E.nnz();
  out[5] = p.F.nnz();
  out[6] = p.dense_solver.empty() ? 0 : (int64_t)p.dense_solver.mat_backup().nrows();
  out[7] = p.dense_solver.empty() ? 0 : (int64_t)p.dense_solver.rank();
  out[10] = 0;
  if (!p.dense_solver.empty() && std::strcmp(p.dense_solver.method(), "LUP") == 0) out[10] = 2;  // HIF_DENSE_MODE=0 build
  if (p.dense_solver.empty() && !p.symm_dense_solver.empty()) {  // is_symm: Prec::symm_dense_solver (SYEIG)
    out[6] = (int64_t)p.symm_dense_solver.mat_backup().nrows();
    out[7] = (int64_t)p.symm_dense_solver.rank();
    out[10] = 1;
  }
  out[8] = p.E.ncols();  // number of columns of E (== m when nm>0)
  out[9] = p.F.ncols();  // number of columns of F (== nm when present)
}

template <class M, class T>
void export_ccs(const M &A, int64_t *colptr, int *rowind, T *vals) {
  const size_t nc = A.ncols();
  if (A.col_start().size() == 0) {
    for (size_t j = 0; j <= nc; ++j) colptr[j] = 0;
    return;
  }
  for (size_t j = 0; j <= nc; ++j) colptr[j] = (int64_t)A.col_start()[j];
  const size_t nz = A.nnz();
  for (size_t k = 0; k < nz; ++k) {
    rowind[k] = A.row_ind()[k];
    vals[k] = A.vals()[k];
  }
}

// which: 0 L_B, 1 U_B, 2 E, 3 F   (CCS, exactly as the reference stores them)
template <class T>
void level_ccs(void *h, int l, int which, int64_t *colptr, int *rowind, T *vals) {
  auto *r = (Ref<T> *)h;
  const auto &p = *r->lv[l];
  switch (which) {
    case 0: export_ccs(p.L_B, colptr, rowind, vals); break;
    case 1: export_ccs(p.U_B, colptr, rowind, vals); break;
    case 2: export_ccs(p.E, colptr, rowind, vals); break;
    default: export_ccs(p.F, colptr, rowind, vals); break;
  }
}

template <class T>
void level_vectors(void *h, int l, T *d, double *s, double *t, int *p, int *p_inv, int *q,
                   int *q_inv) {
  auto *r = (Ref<T> *)h;
  const auto &pr = *r->lv[l];
  for (size_t i = 0; i < pr.m; ++i) d[i] = pr.d_B[i];
  for (size_t i = 0; i < pr.n; ++i) {
    s[i] = pr.s[i];
    t[i] = pr.t[i];
    p[i] = pr.p[i];
    p_inv[i] = pr.p_inv[i];
    q[i] = pr.q[i];
    q_inv[i] = pr.q_inv[i];
  }
}

template <class T>
void level_dense(void *h, int l, T *mat) {
  auto *r = (Ref<T> *)h;
  const auto &pr = *r->lv[l];
  if (pr.dense_solver.empty()) {  // symmetric branch of Prec::inquire_or_export_dense (Prec.hpp:294-303)
    const auto &a = pr.symm_dense_solver.mat_backup().array();
    for (size_t i = 0; i < a.size(); ++i) mat[i] = a[i];
    return;
  }
  const auto &a = pr.dense_solver.mat_backup().array();
  for (size_t i = 0; i < a.size(); ++i) mat[i] = a[i];
}

template <class T>
int do_solve(void *h, const T *b, T *x, int64_t rank, bool tran = false) {
  auto *r = (Ref<T> *)h;
  try {
    hif::Array<T> bb(r->n, const_cast<T *>(b), true), xx(r->n, x, true);
    r->M.solve(bb, xx, tran, (size_t)rank);  // tran: prec_solve_tran, alg/prec_solve.hpp:542-612
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}

template <class T>
int do_mmultiply(void *h, const T *x, T *y, int64_t rank, bool tran = false) {
  auto *r = (Ref<T> *)h;
#ifdef HIFREF_LUP_BUILD
  // With HIF_DENSE_MODE=0 the reference's product does not compile: prec_prod calls dense_solver.multiply(y, r)
  // (alg/prec_prod.hpp:85) but LUP only has multiply(x, y, r, tran) (small_scale/LUP.hpp:181-200).
  (void)r, (void)x, (void)y, (void)rank, (void)tran;
  g_err = "HIF::mmultiply is not instantiable when the reference is built with HIF_DENSE_MODE=0";
  return 1;
#else
  try {
    hif::Array<T> xx(r->n, const_cast<T *>(x), true), yy(r->n, y, true);
    r->M.mmultiply(xx, yy, tran, (size_t)rank);  // tran: prec_prod_tran, alg/prec_prod.hpp:148-235
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
#endif
}

template <class T>
int do_hifir(void *h, const T *b, int nirs, const double *betas, T *x, int *ir_status) {
  auto *r = (Ref<T> *)h;
  try {
    typename Ref<T>::crs_t A(r->n, r->n, r->ip.data(), r->ind.data(), r->val.data(), true);
    hif::Array<T> bb(r->n, const_cast<T *>(b), true), xx(r->n, x, true);
    if (!betas) {
      r->M.hifir(A, bb, (size_t)nirs, xx);
      if (ir_status) ir_status[0] = nirs, ir_status[1] = -1;
    } else {
      auto st = r->M.hifir(A, bb, (size_t)nirs, betas, xx);
      if (ir_status) ir_status[0] = (int)st.first, ir_status[1] = st.second;
    }
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}

// gmres_hif (examples/advanced/gmres.hpp:19-123) on the handle's own matrix; out[0] = flag
// (0 converged / 1 stagnated / 2 reached maxit), out[1] = iterations
template <class T>
int do_gmres(void *h, const T *b, int restart, double rtol, int maxit, int full_rank, T *x, int *out) {
  auto *r = (Ref<T> *)h;
  try {
    typename Ref<T>::crs_t A(r->n, r->n, r->ip.data(), r->ind.data(), r->val.data(), true);
    hif::Array<T> bb(r->n, const_cast<T *>(b), true);
    auto res = gmres_hif(A, bb, r->M, restart, rtol, maxit, 0, full_rank != 0);
    const hif::Array<T> &xs = std::get<0>(res);
    for (size_t i = 0; i < r->n; ++i) x[i] = xs[i];
    out[0] = std::get<1>(res);
    out[1] = std::get<2>(res);
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}

// fgmres_hifir (examples/advanced/gmres.hpp:127-231): flexible GMRES whose inner preconditioner is
// HIF::hifir with 2^outer refinement sweeps; out[0] = flag, out[1] = iterations, out[2] = number of sweeps
template <class T>
int do_fgmres(void *h, const T *b, int restart, double rtol, int maxit, int full_rank, T *x, int *out) {
  auto *r = (Ref<T> *)h;
  try {
    typename Ref<T>::crs_t A(r->n, r->n, r->ip.data(), r->ind.data(), r->val.data(), true);
    hif::Array<T> bb(r->n, const_cast<T *>(b), true);
    auto res = fgmres_hifir(A, bb, r->M, restart, rtol, maxit, 0, full_rank != 0);
    const hif::Array<T> &xs = std::get<0>(res);
    for (size_t i = 0; i < r->n; ++i) x[i] = xs[i];
    out[0] = std::get<1>(res);
    out[1] = std::get<2>(res);
    out[2] = std::get<3>(res);
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}

// y = A x with the reference CRS kernel (serial multiply_nt; mt_mv.hpp partitions rows only)
template <class T>
void do_spmv(size_t n, const int64_t *indptr, const int *indices, const T *vals, const T *x, T *y) {
  std::vector<std::ptrdiff_t> ip(indptr, indptr + n + 1);
  hif::CRS<T, int, std::ptrdiff_t> A(n, n, ip.data(), const_cast<int *>(indices),
                                      const_cast<T *>(vals), true);
  A.multiply_nt_low(x, (size_t)0, n, y);
}

// raw CCS kernels on caller data (unit-level validation of the restatement)
// op: 0 solve_as_strict_lower, 1 solve_as_strict_upper, 2 multiply_nt_low (y = A x),
//     3 solve_as_strict_lower_tran, 4 solve_as_strict_upper_tran, 5 multiply_t_low (y = A^H x)
template <class T>
void do_ccs_kernel(int op, size_t nrows, size_t ncols, const int64_t *colptr, const int *rowind,
                   const T *vals, const T *x, T *y) {
  std::vector<std::ptrdiff_t> cp(colptr, colptr + ncols + 1);
  hif::CCS<T, int, std::ptrdiff_t> A(nrows, ncols, cp.data(), const_cast<int *>(rowind),
                                      const_cast<T *>(vals), true);
  if (op == 2) {
    A.multiply_nt_low(x, y);
  } else if (op == 5) {
    A.multiply_t_low(x, y);
  } else {
    hif::Array<T> yy(nrows, y, true);
    if (op == 0)
      A.solve_as_strict_lower(yy);
    else if (op == 1)
      A.solve_as_strict_upper(yy);
    else if (op == 3)
      A.solve_as_strict_lower_tran(yy);
    else
      A.solve_as_strict_upper_tran(yy);
  }
}

// dense last level alone: hif::QRCP<T> (small_scale/QRCP.hpp:50): set_matrix, factorize, solve /
// multiply with an explicit rank argument (0 = numerical rank). mat is column-major n x n.
template <class T>
int do_qrcp(size_t n, const T *mat, double rrqr_cond, int op, const T *b, int64_t rank_in, T *x,
            int64_t *rank_out) {
  try {
    hif::DenseMatrix<T> D(n, n);
    for (size_t i = 0; i < n * n; ++i) D.array()[i] = mat[i];
    hif::QRCP<T> qr;
    qr.set_matrix(std::move(D));
    hif::Options o = hif::get_default_options();
    o.verbose = hif::VERBOSE_NONE;
    o.rrqr_cond = rrqr_cond;
    qr.factorize(o);
    *rank_out = (int64_t)qr.rank();
    hif::Array<T> xx(n, x, true);
    for (size_t i = 0; i < n; ++i) x[i] = b[i];
    if (op == 0)
      qr.solve(xx, (size_t)rank_in);
    else if (op == 2)
      qr.solve(xx, (size_t)rank_in, true);  // _solve_t, small_scale/QRCP.hpp:413-452
    else if (op == 3)
      qr.multiply(xx, (size_t)rank_in, true);  // _multiply_t, small_scale/QRCP.hpp:502-541
    else
      qr.multiply(xx, (size_t)rank_in);
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}

}  // namespace

typedef std::complex<double> zt;

extern "C" {

const char *hifref_error(void) { return g_err.c_str(); }

void *hifref_d_factorize(size_t n, const int64_t *ip, const int *ind, const double *v,
                         const double *params) {
  return do_factorize<double>(n, ip, ind, v, params);
}
void *hifref_z_factorize(size_t n, const int64_t *ip, const int *ind, const void *v,
                         const double *params) {
  return do_factorize<zt>(n, ip, ind, (const zt *)v, params);
}
void hifref_d_destroy(void *h) { delete (Ref<double> *)h; }
void hifref_z_destroy(void *h) { delete (Ref<zt> *)h; }
int hifref_d_nlevels(void *h) { return (int)((Ref<double> *)h)->lv.size(); }
int hifref_z_nlevels(void *h) { return (int)((Ref<zt> *)h)->lv.size(); }
int64_t hifref_d_nnz(void *h) { return (int64_t)((Ref<double> *)h)->M.nnz(); }
int64_t hifref_z_nnz(void *h) { return (int64_t)((Ref<zt> *)h)->M.nnz(); }
void hifref_d_level_sizes(void *h, int l, int64_t *o) { level_sizes<double>(h, l, o); }
void hifref_z_level_sizes(void *h, int l, int64_t *o) { level_sizes<zt>(h, l, o); }
void hifref_d_level_ccs(void *h, int l, int w, int64_t *cp, int *ri, double *v) {
  level_ccs<double>(h, l, w, cp, ri, v);
}
void hifref_z_level_ccs(void *h, int l, int w, int64_t *cp, int *ri, void *v) {
  level_ccs<zt>(h, l, w, cp, ri, (zt *)v);
}
void hifref_d_level_vectors(void *h, int l, double *d, double *s, double *t, int *p, int *pi,
                            int *q, int *qi) {
  level_vectors<double>(h, l, d, s, t, p, pi, q, qi);
}
void hifref_z_level_vectors(void *h, int l, void *d, double *s, double *t, int *p, int *pi, int *q,
                            int *qi) {
  level_vectors<zt>(h, l, (zt *)d, s, t, p, pi, q, qi);
}
void hifref_d_level_dense(void *h, int l, double *m) { level_dense<double>(h, l, m); }
void hifref_z_level_dense(void *h, int l, void *m) { level_dense<zt>(h, l, (zt *)m); }
int hifref_d_solve(void *h, const double *b, double *x, int64_t rank) {
  return do_solve<double>(h, b, x, rank);
}
int hifref_z_solve(void *h, const void *b, void *x, int64_t rank) {
  return do_solve<zt>(h, (const zt *)b, (zt *)x, rank);
}
int hifref_d_solve_tran(void *h, const double *b, double *x, int64_t rank) {
  return do_solve<double>(h, b, x, rank, true);
}
int hifref_z_solve_tran(void *h, const void *b, void *x, int64_t rank) {
  return do_solve<zt>(h, (const zt *)b, (zt *)x, rank, true);
}
int hifref_d_mmultiply(void *h, const double *x, double *y, int64_t rank) {
  return do_mmultiply<double>(h, x, y, rank);
}
int hifref_z_mmultiply(void *h, const void *x, void *y, int64_t rank) {
  return do_mmultiply<zt>(h, (const zt *)x, (zt *)y, rank);
}
int hifref_d_mmultiply_tran(void *h, const double *x, double *y, int64_t rank) {
  return do_mmultiply<double>(h, x, y, rank, true);
}
int hifref_z_mmultiply_tran(void *h, const void *x, void *y, int64_t rank) {
  return do_mmultiply<zt>(h, (const zt *)x, (zt *)y, rank, true);
}
// HIF::nsp / HIF::nsp_tran (builder.hpp:491-492): constant-mode null-space filter applied inside solve
// (NspFilter::set_nsp_const, NspFilter.hpp:118-125); start > end removes the filter
void hifref_d_set_nsp_const(void *h, int tran, int64_t start, int64_t end) {
  auto *r = (Ref<double> *)h;
  hif::NspFilterPtr f;
  if (start <= end || end < 0) {
    f = hif::create_nsp_filter();
    f->set_nsp_const((size_t)start, end < 0 ? (size_t)-1 : (size_t)end);
  }
  (tran ? r->M.nsp_tran : r->M.nsp) = f;
}
int hifref_d_hifir(void *h, const double *b, int nirs, const double *betas, double *x, int *st) {
  return do_hifir<double>(h, b, nirs, betas, x, st);
}
int hifref_z_hifir(void *h, const void *b, int nirs, const double *betas, void *x, int *st) {
  return do_hifir<zt>(h, (const zt *)b, nirs, betas, (zt *)x, st);
}
int hifref_d_gmres(void *h, const double *b, int restart, double rtol, int maxit, int full_rank, double *x,
                   int *out) {
  return do_gmres<double>(h, b, restart, rtol, maxit, full_rank, x, out);
}
int hifref_d_fgmres(void *h, const double *b, int restart, double rtol, int maxit, int full_rank, double *x,
                    int *out) {
  return do_fgmres<double>(h, b, restart, rtol, maxit, full_rank, x, out);
}
void hifref_d_spmv(size_t n, const int64_t *ip, const int *ind, const double *v, const double *x,
                   double *y) {
  do_spmv<double>(n, ip, ind, v, x, y);
}
void hifref_z_spmv(size_t n, const int64_t *ip, const int *ind, const void *v, const void *x,
                   void *y) {
  do_spmv<zt>(n, ip, ind, (const zt *)v, (const zt *)x, (zt *)y);
}
int hifref_d_qrcp(size_t n, const double *mat, double cond, int op, const double *b, int64_t rin,
                  double *x, int64_t *rout) {
  return do_qrcp<double>(n, mat, cond, op, b, rin, x, rout);
}
int hifref_z_qrcp(size_t n, const void *mat, double cond, int op, const void *b, int64_t rin,
                  void *x, int64_t *rout) {
  return do_qrcp<zt>(n, (const zt *)mat, cond, op, (const zt *)b, rin, (zt *)x, rout);
}
void hifref_d_ccs_kernel(int op, size_t nr, size_t nc, const int64_t *cp, const int *ri,
                         const double *v, const double *x, double *y) {
  do_ccs_kernel<double>(op, nr, nc, cp, ri, v, x, y);
}
void hifref_z_ccs_kernel(int op, size_t nr, size_t nc, const int64_t *cp, const int *ri,
                         const void *v, const void *x, void *y) {
  do_ccs_kernel<zt>(op, nr, nc, cp, ri, (const zt *)v, (const zt *)x, (zt *)y);
}

}  // extern "C"
