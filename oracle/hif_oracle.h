/* oracle/hif_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's preconditioner-apply hot path (HIFIR v0.2.0,
 * /root/reference): prec_solve / prec_solve_tran + their CCS kernels + dense QRCP last level + prec_prod (round-trip
 * checker) + iterative refinement + CRS SpMV.  Every function in hif_oracle_impl.inc cites the
 * reference file:line it follows.  PARITY STATUS: PINNED -- validated (tests/test_oracle_vs_ref.py,
 * run where oracle/_ref/libhifref.so exists) bit-for-bit on all sparse stages against the real
 * reference compiled -O2 -fno-fast-math -ffp-contract=off, to 1e-12 on the dense level, and against
 * the reference's own MATLAB known-answer dense vectors (tests/golden/kat_*.json, from
 * tests/test_sss_qrcp.cpp / test_qrcp_cmplx.cpp).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liborc.so.
 *
 * Two instantiations: orc_d_* (double) and orc_z_* (double _Complex, C99 layout == std::complex).
 * Matrices are handed over exactly as hif::Prec stores them (CCS, Prec.hpp:309-323); the oracle
 * copies them.
 */
#ifndef HIF_ORACLE_H
#define HIF_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DECL(P, T)                                                                              \
  void *orc_##P##_create(int nlevels);                                                              \
  void orc_##P##_destroy(void *h);                                                                  \
  /* one hif::Prec: CCS L_B,U_B (m cols), E (m cols, nm rows), F (F_ncols cols, m rows) */          \
  int orc_##P##_set_level(void *h, int l, int64_t m, int64_t n, const int64_t *Lcp, const int *Lri, \
                          const T *Lv, const int64_t *Ucp, const int *Uri, const T *Uv,             \
                          const int64_t *Ecp, const int *Eri, const T *Ev, int64_t F_ncols,         \
                          const int64_t *Fcp, const int *Fri, const T *Fv, const T *d,              \
                          const double *s, const double *t, const int *p, const int *p_inv,         \
                          const int *q, const int *q_inv);                                          \
  /* unfactored column-major nd x nd block of the last level; factorizes by QRCP */                 \
  int orc_##P##_set_dense(void *h, int64_t nd, const T *mat, double rrqr_cond);                     \
  /* the same block of an is_symm hierarchy (Prec::symm_dense_solver): SYEIG (small_scale/SYEIG.hpp) */ \
  int orc_##P##_set_dense_symm(void *h, int64_t nd, const T *mat, int spd);                         \
  /* ... and of a reference built with HIF_DENSE_MODE=0: LU with partial pivoting (small_scale/LUP.hpp) */ \
  int orc_##P##_set_dense_lup(void *h, int64_t nd, const T *mat);                                   \
  int64_t orc_##P##_dense_rank(void *h);                                                            \
  int64_t orc_##P##_work_size(void *h);                                                             \
  int orc_##P##_solve(void *h, const T *b, T *x, int64_t rank);                                     \
  /* B,X row-interleaved [n][nrhs] (CompressedStorage.hpp:2127); column-by-column solve */          \
  int orc_##P##_solve_batch(void *h, const T *B, T *X, int64_t nrhs, int64_t rank, int threads);    \
  /* x = M^{-H} b: HIF::solve(b, x, trans = true) -> prec_solve_tran (alg/prec_solve.hpp:542-612) */  \
  int orc_##P##_solve_tran(void *h, const T *b, T *x, int64_t rank);                                \
  int orc_##P##_solve_tran_batch(void *h, const T *B, T *X, int64_t nrhs, int64_t rank,             \
                                 int threads);                                                      \
  int orc_##P##_mmultiply(void *h, const T *x, T *y, int64_t rank);                                 \
  /* y = M^H x: HIF::mmultiply(x, y, trans = true) -> prec_prod_tran (alg/prec_prod.hpp:148-235) */   \
  int orc_##P##_mmultiply_tran(void *h, const T *x, T *y, int64_t rank);                            \
  int orc_##P##_hifir(void *h, int64_t n, const int64_t *ip, const int *ind, const T *v,            \
                      const T *b, int nirs, const double *betas, int64_t rank, T *x,                \
                      int *ir_status);                                                              \
  void orc_##P##_crs_mv(int64_t n, const int64_t *ip, const int *ind, const T *v, const T *x,       \
                        T *y);                                                                      \
  /* raw CCS kernels; op 0 strict-lower solve, 1 strict-upper solve, 2 y = A x, 3 / 4 the two    */   \
  /* solves with the conjugate transpose, 5 y = A^H x */                                            \
  void orc_##P##_ccs_kernel(int op, int64_t nrows, int64_t ncols, const int64_t *cp,                \
                            const int *ri, const T *v, const T *x, T *y);                           \
  /* the same three with nrhs interleaved right-hand sides, [n][nrhs] */                            \
  void orc_##P##_ccs_kernel_mrhs(int op, int64_t nrows, int64_t ncols, const int64_t *cp,           \
                                 const int *ri, const T *v, int64_t nrhs, const T *x, T *y);        \
  /* dense block alone: QRCP factor + (op 0) solve / (op 1) multiply / (op 2) solve with A^H /   */  \
  /* (op 3) multiply with A^H */                                                                    \
  int orc_##P##_qrcp(int64_t n, const T *mat, double rrqr_cond, int op, const T *b,                 \
                     int64_t rank_in, T *x, int64_t *rank_out);                                     \
  /* symmetric dense block alone: SYEIG factor + (op 0) solve / (op 1) multiply */                  \
  int orc_##P##_syeig(int64_t n, const T *mat, int spd, int op, const T *b, int64_t rank_in, T *x,  \
                      int64_t *rank_out, double *w_out);                                            \
  /* LU block alone: (op 0) solve / (op 1) multiply / (op 2) solve 'T' / (op 3) multiply 'C' */     \
  int orc_##P##_lup(int64_t n, const T *mat, int op, const T *b, T *x);

ORC_DECL(d, double)
#ifndef __cplusplus
ORC_DECL(z, double _Complex)
#endif

#ifdef __cplusplus
}
#endif
#endif
