/* libhifir_amd_ext.h -- the ADDITIVE exports of the drop-in libhifir (shim/libhifir_amd_shim.cpp).
 *
 * The shim exports every symbol of the reference's C ABI with identical signatures (it compiles against the
 * reference's own libhifir/include/libhifir.h, so the compiler checks that) and serves lhf?Apply / lhf?Solve from
 * HBM through include/hifir_amd.h.  The entry points below have no reference counterpart ("additive
 * exports the build needs"); include this header AFTER libhifir.h.
 */
#ifndef LIBHIFIR_AMD_EXT_H
#define LIBHIFIR_AMD_EXT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* HIP ordinals on which handles created AFTERWARDS keep a resident copy of their hierarchy (default: the current
 * device only).  lhf?ApplyBatch shards the right-hand-side columns over them -- contiguous column blocks, one host
 * thread per device, no data-path collective; lhf?Apply / lhf?Solve use the first one.  The same
 * ordinal may be listed more than once (independent replicas on one GPU: rehearses the sharding on a 1-GPU box).
 * n <= 0 or ids == NULL restores the default.  LHF_MISMATCHED_SIZES for an ordinal the process cannot see. */
LhfStatus lhfSetDevices(const int *ids, int n);
/* number of HIP devices visible to the process (0: every Create / Setup fails with LHF_HIFIR_ERROR -- no CPU fallback) */
int lhfGetDeviceCount(void);

/* lhf?Apply (libhifir.h:685, :997) for nrhs right-hand sides at once.  B and X are row-interleaved [n][nrhs]
 * blocks with row strides ldb, ldx (in elements): the layout of hif::Array<std::array<T, Nrhs>>
 * (src/hif/ds/CompressedStorage.hpp:2127).  Same operator / nirs / betas / rank rules as lhf?Apply
 * (libhifir.cpp:447-472); ir_status, if given with betas, receives {iterations, flag} per column (2 * nrhs ints). */
LhfStatus lhfdApplyBatch(const LhfdHifHdl hif, const LhfOperationType op, const double *B, const size_t nrhs,
                         const size_t ldb, const int nirs, const double *betas, const int rank, double *X,
                         const size_t ldx, int *ir_status);
LhfStatus lhfzApplyBatch(const LhfzHifHdl hif, const LhfOperationType op, const double _Complex *B, const size_t nrhs,
                         const size_t ldb, const int nirs, const double *betas, const int rank, double _Complex *X,
                         const size_t ldx, int *ir_status);

/* The same for a batch that is ALREADY sharded over the devices of lhfSetDevices and stays there (a Krylov solver on
 * several GPUs): block d -- B_dev[d], X_dev[d]: device pointers into the HBM of device ids[d], row-interleaved
 * [n][ncols[d]] with row strides ldb[d], ldx[d] -- is applied by the handle's replica d.  Direct operators only
 * (LHF_S / LHF_SH with the numerical rank, LHF_M / LHF_MH; no refinement); every block is ENQUEUED on its replica's own
 * stream and the call returns: no PCIe traffic, no host threads.  nblocks <= the number of devices given to
 * lhfSetDevices before the handle was set up; a block of 0 columns is skipped.
 * lhf?GatherBatchDev enqueues, behind those applies, peer copies of the blocks into the consecutive column ranges of
 * ONE [n][ldd] block dst_dev (device memory of any device; SURVEY 8(e): the gather at the end of a batch);
 * lhf?SyncDevices waits for everything enqueued on the handle's replicas. */
LhfStatus lhfdApplyBatchDev(const LhfdHifHdl hif, const LhfOperationType op, const int nblocks, const double *const *B_dev,
                            const size_t *ncols, const size_t *ldb, double *const *X_dev, const size_t *ldx);
LhfStatus lhfzApplyBatchDev(const LhfzHifHdl hif, const LhfOperationType op, const int nblocks,
                            const double _Complex *const *B_dev, const size_t *ncols, const size_t *ldb,
                            double _Complex *const *X_dev, const size_t *ldx);
LhfStatus lhfdGatherBatchDev(const LhfdHifHdl hif, const int nblocks, const double *const *X_dev, const size_t *ncols,
                             const size_t *ldx, double *dst_dev, const size_t ldd);
LhfStatus lhfzGatherBatchDev(const LhfzHifHdl hif, const int nblocks, const double _Complex *const *X_dev,
                             const size_t *ncols, const size_t *ldx, double _Complex *dst_dev, const size_t ldd);
LhfStatus lhfdSyncDevices(const LhfdHifHdl hif);
LhfStatus lhfzSyncDevices(const LhfzHifHdl hif);

/* Export / import of the factored hierarchy in the on-disk format of hifamd_save / hifamd_load (include/hifir_amd.h):
 * factorize once where the host factorization is affordable, apply on GPU nodes without refactorizing.  A loaded
 * handle serves Apply / Solve / the size queries; lhf?Refactorize gives it a host factorization again.
 * With HIFIR_AMD_SAVE_ANALYSIS=1 in the environment the file also carries the host analysis of every level
 * (hifamd_save_ex, HIFAMD_SAVE_ANALYSIS): the ranks that load it skip that work. */
LhfStatus lhfdSaveHierarchy(const LhfdHifHdl hif, const char *path);
LhfStatus lhfzSaveHierarchy(const LhfzHifHdl hif, const char *path);
LhfdHifHdl lhfdLoadHierarchy(const char *path);
LhfzHifHdl lhfzLoadHierarchy(const char *path);

/* What ONE replica of the handle keeps resident in HBM, in bytes (the sibling of lhf?GetStats, libhifir.h:700-716, for
 * the device side): bytes[0] factors with their plan arrays, [1] explicit operators (block inverses, combined top
 * operators, tail operator), [2] coefficient tiles of the component bands, [3] work arena, [4] columns of that arena,
 * [5] the widest batch the handle was finalized for -- HIFIR_AMD_MAX_NRHS in the environment of lhf?Setup /
 * lhf?LoadHierarchy (1 .. 64, default 64).  In this build the arena is 64 columns wide whatever was asked for (the
 * fast kernels address a 64-column arena); a narrower request only bounds the tile width of lhf?ApplyBatch. */
LhfStatus lhfdGetResidentBytes(const LhfdHifHdl hif, size_t bytes[6]);
LhfStatus lhfzGetResidentBytes(const LhfzHifHdl hif, size_t bytes[6]);

#ifdef __cplusplus
}
#endif
#endif /* LIBHIFIR_AMD_EXT_H */
