/* hifir_amd.h -- C ABI of the MI355X-native HIFIR preconditioner-apply path.
 *
 * Plain C, plain pointers and sizes, no C++/torch types.  This is the drop-in boundary for the
 * reference's hot path: the reference keeps factorizing on the host, hands each level of its
 * hif::Precs list to this library once (hifamd_add_level / hifamd_set_dense, the field set of
 * hif::Prec::export_sparse_data + inquire_or_export_dense), and from then on every
 * HIF::solve / HIF::hifir / lhf?Solve / lhf?Apply(LHF_S) call is served from HBM by hand-written
 * gfx950 kernels.  INTEGRATION.md shows the reference-side binding.
 *
 * All file:line citations are relative to the reference tree (HIFIR v0.2.0).
 *
 * Conventions
 *   - every function returns a HifAmdStatus (values mirror LhfStatus, libhifir/include/libhifir.h:148-154);
 *     the message of the last failure on the calling thread is returned by hifamd_last_error()
 *     (cf. lhfGetErrorMsg, libhifir.h:255).
 *   - value type: double (HIFAMD_D) or double complex (HIFAMD_Z, C99 layout == std::complex<double>);
 *     index type int32 (LhfInt), pointer type int64 (LhfIndPtr = ptrdiff_t), libhifir.h:47-83.
 *   - a handle is NOT thread-safe (same rule as the reference: HIF::solve mutates `mutable _prec_work`,
 *     src/hif/builder.hpp:579); distinct handles may be used from distinct threads.
 *   - multi-RHS blocks are row-interleaved [n][nrhs] with an explicit row stride (ld, in elements),
 *     i.e. the layout of hif::Array<std::array<T,Nrhs>> (src/hif/ds/CompressedStorage.hpp:2127).
 *   - there is no CPU fallback: without a usable HIP device every compute entry point fails with
 *     HIFAMD_HIFIR_ERROR.
 */
#ifndef HIFIR_AMD_H
#define HIFIR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum HifAmdStatus {
  HIFAMD_SUCCESS = 0,      /* LHF_SUCCESS */
  HIFAMD_NULL_OBJ,         /* LHF_NULL_OBJ */
  HIFAMD_MISMATCHED_SIZES, /* LHF_MISMATCHED_SIZES */
  HIFAMD_BAD_PREC,         /* LHF_BAD_PREC */
  HIFAMD_HIFIR_ERROR       /* LHF_HIFIR_ERROR */
} HifAmdStatus;

typedef enum HifAmdValueType { HIFAMD_D = 0, HIFAMD_Z = 1 } HifAmdValueType;

/* operator tags of lhf?Apply (LhfOperationType, libhifir/include/libhifir.h:159-164), same values */
typedef enum HifAmdOp {
  HIFAMD_S = 0, /* x = M^{-1} b   (prec_solve,      alg/prec_solve.hpp:332-412) */
  HIFAMD_SH,    /* x = M^{-H} b   (prec_solve_tran, alg/prec_solve.hpp:542-612) */
  HIFAMD_M,     /* x = M b        (prec_prod,       alg/prec_prod.hpp:55-147)  */
  HIFAMD_MH     /* x = M^H b      (prec_prod_tran,  alg/prec_prod.hpp:148-235) */
} HifAmdOp;

typedef struct HifAmdPrec *HifAmdHdl; /* opaque: one multilevel preconditioner resident in HBM */

/* ---- library ------------------------------------------------------------------------------ */
const char *hifamd_version(void);
/* message of the last error on this thread, or NULL; cleared by the call (libhifir.cpp:224-229) */
const char *hifamd_last_error(void);
/* number of visible HIP devices (0 without a GPU; never fails) */
int hifamd_device_count(void);

/* ---- lifecycle ---------------------------------------------------------------------------- */
/* replaces `new HIF<>` in lhf?Create (libhifir.cpp:383-396); device = HIP ordinal, -1 = current */
HifAmdStatus hifamd_create(HifAmdValueType vt, int device, HifAmdHdl *out);
HifAmdStatus hifamd_destroy(HifAmdHdl h); /* NULL-safe, frees HBM (cf. lhf?Destroy, libhifir.h:619) */

/* ---- hierarchy import (host pointers; data is copied) ------------------------------------- */
/* One call per hif::Prec, in list order (src/hif/alg/Prec.hpp:309-323).  The four matrices are
 * CCS exactly as the reference stores them (Prec::mat_type = ccs_type, Prec.hpp:86,90):
 *   L_B, U_B : m x m strict triangles, implicit unit diagonal, sorted row indices
 *   E        : (n-m) x m,   F : m x F_ncols (F_ncols == n-m, or 0 when absent; prec_solve.hpp:395)
 * d: m values; s,t: n REAL scalings (Prec.hpp:96-99); p, q_inv: n 0-based permutations.
 * p_inv and q are only needed by HIFAMD_SH / HIFAMD_M / HIFAMD_MH and may be NULL otherwise (with q the plain solve
 * also writes its output permutation from inside the last triangular kernel instead of a separate pass). */
HifAmdStatus hifamd_add_level(HifAmdHdl h, int64_t m, int64_t n,
                              const int64_t *L_colptr, const int32_t *L_rowind, const void *L_vals,
                              const int64_t *U_colptr, const int32_t *U_rowind, const void *U_vals,
                              const int64_t *E_colptr, const int32_t *E_rowind, const void *E_vals,
                              int64_t F_ncols,
                              const int64_t *F_colptr, const int32_t *F_rowind, const void *F_vals,
                              const void *d, const double *s, const double *t,
                              const int32_t *p, const int32_t *p_inv,
                              const int32_t *q, const int32_t *q_inv);
/* The UNFACTORED column-major nd x nd Schur complement of the last level, as
 * Prec::inquire_or_export_dense hands it out (Prec.hpp:275-293).  Factorized here on the host by
 * QR with column pivoting + rank determination (QRCP::factorize, small_scale/QRCP.hpp:107-179);
 * rrqr_cond <= 0 selects the reference default eps^(-2/3) (QRCP.hpp:110-117). */
HifAmdStatus hifamd_set_dense(HifAmdHdl h, int64_t nd, const void *mat_colmajor, double rrqr_cond);
/* The same block of a hierarchy that the reference factorized with is_symm (symm_level_factorize,
 * alg/symm_factor.hpp:654-657 fills Prec::symm_dense_solver instead of dense_solver; the export is the
 * symmetric branch of Prec::inquire_or_export_dense, Prec.hpp:294-303).  Factorized here on the host by a
 * symmetric / Hermitian eigendecomposition with the truncation rules of SYEIG::factorize
 * (small_scale/SYEIG.hpp:107-175): spd > 0 positive definite, < 0 negative definite, 0 indefinite
 * (Options::spd).  Only the lower triangle is read.  Solve, conjugate-transpose solve and product follow
 * SYEIG::solve / multiply (SYEIG.hpp:181-200, 256-273), incl. the run-time `rank` argument. */
HifAmdStatus hifamd_set_dense_symm(HifAmdHdl h, int64_t nd, const void *mat_colmajor, int spd);
/* The same block when the reference was built with HIF_DENSE_MODE=0 (macros.hpp:100-105, small_scale/solver.hpp:49):
 * its last level is then LU with partial pivoting (small_scale/LUP.hpp).  Factorized here on the host
 * (?getrf semantics, LUP::factorize :100-119) and applied as one product with the explicit inverse; the `rank`
 * argument is ignored as in LUP::solve (:141), the conjugate-transpose apply passes 'T' to the solve and 'C' to the
 * product exactly like LUP.hpp:150,187.  An exactly singular block is refused (HIFAMD_BAD_PREC). */
HifAmdStatus hifamd_set_dense_lup(HifAmdHdl h, int64_t nd, const void *mat_colmajor);
/* Converts CCS -> schedule-ordered CSR, level-schedules the triangular factors, ships everything to
 * HBM and sizes the work arena for batches of up to max_nrhs (the reference sizes its work buffer
 * on first use only, builder.hpp:414-416 -- not replicated). */
HifAmdStatus hifamd_finalize(HifAmdHdl h, int64_t max_nrhs);

/* ---- on-disk form of an imported hierarchy ---------------------------------- */
/* hifamd_save writes exactly what hifamd_add_level / hifamd_set_dense received (before or after
 * hifamd_finalize); hifamd_load creates a handle and replays those calls -- the caller then calls
 * hifamd_finalize.  Lets a hierarchy factorized once on a host that has the reference be applied on
 * GPU nodes that do not.  Format (little-endian): "HIFAMD1\0", int64 value type, int64 #levels, int64
 * has_dense; per level int64 {m, n, F_ncols}, the four CCS matrices as int64 {nrows, ncols} + three
 * counted arrays (int64 count, data, zero padding to 8 bytes), then d, s, t, p, p_inv, q, q_inv as
 * counted arrays; the dense block as int64 nd, double rrqr_cond and a counted column-major array. */
HifAmdStatus hifamd_save(HifAmdHdl h, const char *path);
HifAmdStatus hifamd_load(const char *path, int device, HifAmdHdl *out);
/* hifamd_save with options.  HIFAMD_SAVE_ANALYSIS appends the ANALYSIS of every level (wavefront schedules, band plans,
 * slot-ordered triangles: what hifamd_add_level derives on the host, e.g. 24 s of the 45 s between load and first apply
 * on a 256^3 grid) as a checksummed trailer behind the records above.  hifamd_load adopts it when it was made with the
 * planner options in force (HIFIR_AMD_* environment) and fits the factors' sparsity pattern (it holds no matrix values),
 * verifies every size and index range first, and analyzes as usual otherwise (also with HIFIR_AMD_LOAD_ANALYSIS=0): a trailer changes how long a load
 * takes, never its result -- hifamd_stats_ext slot 12 tells how many levels came from it.  A reader that does not know
 * the trailer ignores it.  No reference counterpart. */
#define HIFAMD_SAVE_ANALYSIS 1
HifAmdStatus hifamd_save_ex(HifAmdHdl h, const char *path, int flags);

/* ---- queries (cf. lhf?GetLevels/GetNnz/GetSchurSize/GetSchurRank, libhifir.h:722-740) ------ */
int hifamd_value_type(HifAmdHdl h);     /* HIFAMD_D / HIFAMD_Z of the handle (what hifamd_load found in the file); -1 for NULL */
int hifamd_device(HifAmdHdl h);         /* HIP ordinal the handle is bound to (after hifamd_finalize; before: as created, -1 = current) */
int64_t hifamd_nrows(HifAmdHdl h);
int64_t hifamd_levels(HifAmdHdl h);     /* counts the dense block as a level (builder.hpp:141-147) */
int64_t hifamd_nnz(HifAmdHdl h);        /* Prec::nnz summed (Prec.hpp:170-176) */
int64_t hifamd_schur_size(HifAmdHdl h);
int64_t hifamd_schur_rank(HifAmdHdl h);
/* stats[0..15]: 0 sum n_l, 1 sum m_l, 2 nnz(L)+nnz(U), 3 nnz(E)+nnz(F), 4 dense n, 5 B_mat bytes,
 * 6 B_vec bytes per RHS (SURVEY 8d formula), 7 #wavefronts L (all levels), 8 #wavefronts U,
 * 9 kernel launches per apply at the last captured batch width, 10 sparse levels, 11 bands (L+U, all
 * levels), 12 band workgroups, 13 seconds spent in hifamd_finalize, 14 bytes of explicit operators resident in HBM
 * (block inverses + combined top operators + the tail operator), 15 milliseconds of the last hipGraph capture */
HifAmdStatus hifamd_stats(HifAmdHdl h, double *stats16);
/* Set-up and operator accounting (additive, no reference counterpart); returns the number of values it knows, writes
 * min(cap, that) of them: 0 finalize seconds, 1 last graph capture ms, 2 bytes of block inverses, 3 bytes of combined top
 * operators, 4 bytes of the tail operator, 5 rows of the tail operator (0: the recursion runs), 6 its first level,
 * 7 relative difference product vs recursion on the finalize-time probe, 8 max |entry| of the tail operator, 9 why it
 * was rejected (0 not, 1 not finite, 2 growth, 3 probe, 4 an error while it was formed), 10 / 11 the probe and growth
 * limits in force (HIFIR_AMD_TAIL_PROBE_TOL, HIFIR_AMD_TAIL_GROWTH), 12 levels whose analysis came from the trailer of the
 * file the handle was loaded from (hifamd_save_ex), 13 host seconds spent analyzing the levels (or adopting their
 * analysis), 14 bytes of the work arena (w + v of every level), 15 its width in columns (this build: always 64 -- the
 * fast kernels address a 64-column arena; max_nrhs bounds the batch width a call may pass, not the arena), 16 bytes of
 * the component bands' coefficient tiles, 17 bytes of the factors with their plan arrays, 18 max_nrhs of hifamd_finalize,
 * 19 / 20 rows (all levels) whose L / U result the FIRST solve of a level does not store because nothing reads it from
 * memory (real handles, sparse-own levels; HIFIR_AMD_SKIP_ROWS=0: none), 21 arrays of the host copy that hifamd_finalize found
 * changed since hifamd_add_level -- not by this library -- and rebuilt from the imported arrays (a warning names them on stderr;
 * anything it cannot rebuild is refused).
 * -1 for a NULL handle. */
int hifamd_stats_ext(HifAmdHdl h, double *out, int cap);
/* Per-level sizes (what the SURVEY 8(d) byte formula needs level by level): 0 m, 1 n, 2 nnz(L_B), 3 nnz(U_B), 4 nnz(E),
 * 5 nnz(F), 6 / 7 wavefronts of L / U, 8 / 9 launches ("bands") of the L / U plan, 10 rows of the combined top operator.
 * Returns the number of values it knows (-1: NULL handle or no such level). */
int hifamd_level_stats(HifAmdHdl h, int level, double *out, int cap);
/* Which level and stage every kernel launch of the LAST batched apply belongs to, in launch order (for attributing a
 * kernel trace): out[i] = 16 * level + stage, stage 1 S1 gather, 2 first LDU solve (with the fused S1), 3 S3 product with
 * E, 4 dense block / tail operator, 5 S5 product with F, 6 second LDU solve (with the fused S5 / S7), 7 S7 scatter
 * (prec_solve.hpp:359-411).  Returns the number of launches; writes min(cap, that). */
int hifamd_launch_map(HifAmdHdl h, int32_t *out, int cap);
/* level schedule of one triangular factor (host-side analysis; usable without a GPU):
 * which = 0 (L_B) / 1 (U_B).  *nwf = number of wavefronts; if order != NULL it receives the m row
 * ids in processing order and wf_ptr (nwf+1 entries) the wavefront boundaries into it. */
HifAmdStatus hifamd_level_schedule(HifAmdHdl h, int level, int which, int64_t *nwf,
                                   int32_t *order, int64_t *wf_ptr);

/* ---- apply: x = M^{-1} b ------------------------------------------------------------------ */
/* rank: 0 = numerical rank of the dense level, <0 or > size = full (QRCP.hpp:376-377) */
/* HIF::solve (builder.hpp:409-423) / lhf?Solve (libhifir.h:698): host pointers, one RHS */
HifAmdStatus hifamd_solve(HifAmdHdl h, const void *b, void *x, int64_t rank);
/* batched, HOST pointers: B, X are [n][nrhs] row-interleaved with row strides ldb, ldx */
HifAmdStatus hifamd_solve_batch(HifAmdHdl h, const void *B, int64_t ldb, void *X, int64_t ldx,
                                int64_t nrhs, int64_t rank);
/* batched, DEVICE pointers on the handle's device; enqueued on `stream` (a hipStream_t, NULL = the
 * handle's own stream) and NOT synchronized.  B and X must not alias (libhifir Ownership note). */
HifAmdStatus hifamd_solve_batch_dev(HifAmdHdl h, const void *dB, int64_t ldb, void *dX, int64_t ldx,
                                    int64_t nrhs, int64_t rank, void *stream);

/* ---- outer matrix, SpMV and iterative refinement ------------------------------------------ */
/* user matrix A in CRS (copied to HBM): what lhf?Setup borrows for IR (libhifir.cpp:413);
 * 0- or 1-based like the reference (builder.hpp:311-329) */
HifAmdStatus hifamd_set_matrix(HifAmdHdl h, int64_t n, const int64_t *indptr, const int32_t *indices,
                               const void *vals);
/* Y = A X, row dot-products in CRS order (CRS::multiply_nt_low, CompressedStorage.hpp:1109-1127;
 * mt::multiply_nt, utils/mt_mv.hpp:58-73); device pointers, [n][nrhs] */
HifAmdStatus hifamd_spmv_batch_dev(HifAmdHdl h, const void *dX, int64_t ldx, void *dY, int64_t ldy,
                                   int64_t nrhs, void *stream);
/* HIF::hifir (builder.hpp:459-489, alg/IterRefine.hpp:77-165) for nrhs columns at once; host
 * pointers.  betas == NULL: fixed nirs sweeps.  betas = {lower, upper}: per-column relative
 * residual test; ir_status (2*nrhs ints, may be NULL) gets {iterations, flag} per column with
 * flag 0 converged / 1 diverged / -1 reached nirs (IterRefine.hpp:119-120). */
HifAmdStatus hifamd_hifir_batch(HifAmdHdl h, const void *B, int64_t ldb, void *X, int64_t ldx,
                                int64_t nrhs, int nirs, const double *betas, int64_t rank,
                                int *ir_status);
/* same with device pointers for B and X (ir_status stays a host array) */
HifAmdStatus hifamd_hifir_batch_dev(HifAmdHdl h, const void *dB, int64_t ldb, void *dX, int64_t ldx,
                                    int64_t nrhs, int nirs, const double *betas, int64_t rank,
                                    int *ir_status);

/* ---- null-space filter (HIF::nsp / HIF::nsp_tran, builder.hpp:419-422, 491-492) --------------- */
/* Constant mode (NspFilter::set_nsp_const, NspFilter.hpp:118-125): after every HIFAMD_S (op = HIFAMD_S)
 * or HIFAMD_SH (op = HIFAMD_SH) apply -- inside iterative refinement and GMRES too -- each column loses
 * the mean of its rows [start, end); end < 0 means "to the last row"; start > end >= 0 removes the
 * filter.  Unlike the reference (builder.hpp:439) batched applies are filtered as well.  Needs a
 * finalized handle. */
HifAmdStatus hifamd_set_nsp_const(HifAmdHdl h, HifAmdOp op, int64_t start, int64_t end);

/* ---- lhf?Apply with an operator tag (libhifir.h:685, libhifir.cpp:447-472), batched ----------- */
/* op = HIFAMD_S / HIFAMD_SH: nirs <= 1 direct apply (ir_status, if given, gets {1, -1} per column);
 * nirs > 1: iterative refinement with A (HIFAMD_S) or A^H (HIFAMD_SH, IterRefine.hpp:93-96).
 * op = HIFAMD_M / HIFAMD_MH: the multilevel product (HIF::mmultiply, builder.hpp:503-513), always direct.
 * rank = -2 (LHF_DEFAULT_RANK) -> full rank for products and when refining, numerical rank otherwise
 * (libhifir.cpp:453-455).
 * The adjoint hierarchy (U^H, L^H, F^H, E^H, conj(d), (t,q) in, (s,p_inv) out, A^H solve on the dense
 * block) is analysed and shipped to HBM on the first HIFAMD_SH call; it needs the q and p_inv arrays
 * in hifamd_add_level (so do the product operators).  Host pointers. */
HifAmdStatus hifamd_apply_batch(HifAmdHdl h, HifAmdOp op, const void *B, int64_t ldb, void *X, int64_t ldx,
                                int64_t nrhs, int nirs, const double *betas, int64_t rank, int *ir_status);
/* direct apply with device pointers, enqueued on `stream` and not synchronized */
HifAmdStatus hifamd_apply_batch_dev(HifAmdHdl h, HifAmdOp op, const void *dB, int64_t ldb, void *dX, int64_t ldx,
                                    int64_t nrhs, int64_t rank, void *stream);

/* ---- right-preconditioned restarted GMRES, batched (the caller of the hot path) ------------- */
/* The reference's driver gmres_hif (examples/advanced/gmres.hpp:19-123: x0 = 0, modified Gram-Schmidt,
 * Givens rotations, stop on |y_{j+1}| / ||b|| <= rtol) for nrhs columns in lock step; all vectors
 * stay in HBM -- and so do the Hessenberg columns, rotations, residuals and per-column iteration state;
 * one batched apply, one SpMM and j + 2 fused axpy/dot passes per inner step serve every column, and the host
 * reads back three integers per step.  Needs hifamd_set_matrix.  Real and complex handles: the Hessenberg
 * entries are Hermitian products sum conj(q_i) v_i (for real data that IS the example's hif::inner; for
 * complex data the example's sum conj(v_i) q_i is the conjugate and would not orthogonalize), the rotations
 * follow gmres.hpp:75-83 with their conjugates.  rank: 0 numerical rank (the example's default),
 * -1 full.  Per column: flags[c] = 0 converged / 1 stagnated / 2 reached maxit, iters[c] = inner
 * iterations (either may be NULL).  Host pointers; the _dev variant takes device pointers for B, X. */
HifAmdStatus hifamd_gmres_batch(HifAmdHdl h, const void *B, int64_t ldb, void *X, int64_t ldx, int64_t nrhs,
                                int restart, double rtol, int maxit, int64_t rank, int *flags, int *iters);
HifAmdStatus hifamd_gmres_batch_dev(HifAmdHdl h, const void *dB, int64_t ldb, void *dX, int64_t ldx, int64_t nrhs,
                                    int restart, double rtol, int maxit, int64_t rank, int *flags, int *iters);

/* The flexible variant fgmres_hifir (examples/advanced/gmres.hpp:127-231): the preconditioner of outer cycle
 * k is iterative refinement with 2^k sweeps (HIF::hifir), the preconditioned basis is kept and x is updated
 * from it.  sweeps[c] (may be NULL) = refinement sweeps spent on column c (the driver's num_mv). */
HifAmdStatus hifamd_fgmres_batch(HifAmdHdl h, const void *B, int64_t ldb, void *X, int64_t ldx, int64_t nrhs,
                                 int restart, double rtol, int maxit, int64_t rank, int *flags, int *iters,
                                 int *sweeps);

/* ---- instrumentation ---------------------------------------------------------------------- */
/* Average device time (ms) of the last `hifamd_solve_batch_dev`-shaped graph over `reps` replays,
 * measured with HIP events on the handle's stream (the stream the kernels run on). */
HifAmdStatus hifamd_time_apply(HifAmdHdl h, const void *dB, int64_t ldb, void *dX, int64_t ldx,
                               int64_t nrhs, int64_t rank, int warmup, int reps, double *ms_avg);
HifAmdStatus hifamd_sync(HifAmdHdl h);
/* Stream-ordered copy of an [n][ncols] block (device memory of the handle's device, row stride lds) into the columns
 * [col0, col0 + ncols) of dst (row stride ldd; device memory of ANY device of the process: a peer copy), enqueued on
 * the handle's own stream behind whatever was enqueued there with stream = NULL -- the gather of an RHS-sharded batch
 *  without a host round trip.  Not synchronized (hifamd_sync). */
HifAmdStatus hifamd_copy_columns_dev(HifAmdHdl h, const void *src, int64_t lds, int64_t ncols, void *dst, int64_t ldd,
                                     int64_t col0);
/* development aid: checksums of the device-resident arrays of a finalized handle (per level: the nine
 * arrays of L, U, E, F, then d, s, t, p, q_inv; finally Q^H, R^{-1}, jpvt and the rank); returns how many */
int hifamd_debug_checksums(HifAmdHdl h, uint64_t *out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* HIFIR_AMD_H */
