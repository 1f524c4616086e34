// kernels.hip.hpp -- hand-written gfx950 (CDNA4, wave64) kernels of the HIFIR apply path.
//
// Vector layout everywhere: row-interleaved [n][R], R = 2^logR <= 64 right-hand sides per row
// (the layout of hif::Array<std::array<T,Nrhs>>, ds/CompressedStorage.hpp:2127).  A wave64 is split
// into G = 64/R row groups of R lanes: lane = g*R + c handles RHS column c of the g-th row the wave
// currently owns.  R = 64: one wave per row, one coalesced 512-B access per touched row;
// R = 1: one lane per row.  Inside a row every lane walks the nonzeros SEQUENTIALLY in the
// reference's accumulation order with separate multiply and subtract/add (the TU is compiled with
// -ffp-contract=off), so every sparse stage is bit-identical to the reference's scalar loops for
// every batch width.
//
// Reference loops restated as row gathers (file:line relative to the reference tree):
//   k_trsv_*  L: CCS::solve_as_strict_lower   ds/CompressedStorage.hpp:2268-2279 (+ mrhs :2287)
//             D: y[i] /= d[i]                 alg/prec_solve.hpp:219 (fused: v[i] = w[i] / d[i])
//             U: CCS::solve_as_strict_upper   ds/CompressedStorage.hpp:2357-2369 (+ mrhs :2377)
//   k_spmm_epi    CCS::multiply_nt_low :2079 fused with  y = s[p]*b[p] - y  prec_solve.hpp:366-368,397-399
//   k_gather_scale   work[i] = s[p[i]]*b[p[i]]           alg/prec_solve.hpp:359,402
//   k_scatter_scale  y[i] = t[i]*work[q_inv[i]]          alg/prec_solve.hpp:411
//   k_dense_gemm     QRCP::_solve_nt (ormqr, trsv, perm) small_scale/QRCP.hpp:371-411 on f64 MFMA
//   k_crs_spmm       CRS::multiply_nt_low(x,istart,len,y) ds/CompressedStorage.hpp:1109-1127
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hifamd {

struct cplx {
  double x, y;
};

// ---- value-type arithmetic, spelled out so that no contraction / reassociation can happen ------
__device__ __forceinline__ double vzero(double) { return 0.0; }
__device__ __forceinline__ cplx vzero(cplx) { return cplx{0.0, 0.0}; }
__device__ __forceinline__ double vmul(double a, double b) { return a * b; }
__device__ __forceinline__ cplx vmul(cplx a, cplx b) {
  return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ double vsub(double a, double b) { return a - b; }
__device__ __forceinline__ cplx vsub(cplx a, cplx b) { return cplx{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ double vadd(double a, double b) { return a + b; }
__device__ __forceinline__ cplx vadd(cplx a, cplx b) { return cplx{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ double vscale(double s, double a) { return s * a; }
__device__ __forceinline__ cplx vscale(double s, cplx a) { return cplx{s * a.x, s * a.y}; }
__device__ __forceinline__ double vdiv(double a, double b) { return a / b; }
__device__ __forceinline__ cplx vdiv(cplx a, cplx b) {
  // Smith's algorithm (robust against overflow like libgcc's __divdc3; tolerance-level parity)
  if (fabs(b.x) >= fabs(b.y)) {
    const double r = b.y / b.x, den = b.x + b.y * r;
    return cplx{(a.x + a.y * r) / den, (a.y - a.x * r) / den};
  }
  const double r = b.x / b.y, den = b.x * r + b.y;
  return cplx{(a.x * r + a.y) / den, (a.y * r - a.x) / den};
}

// lane decomposition
struct LaneMap {
  int g, c, G;
};
__device__ __forceinline__ LaneMap lane_map(int logR) {
  const int lane = threadIdx.x & 63;
  LaneMap m;
  m.g = lane >> logR;
  m.c = lane & ((1 << logR) - 1);
  m.G = 64 >> logR;
  return m;
}

// ---------------------------------------------------------------------------------------------
// S1: w[i] = s[p[i]] * b[p[i]],  rows [0, cnt)
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_gather_scale(const T *__restrict__ bin, int64_t ldb, int nrhs,
                                                      const int32_t *__restrict__ p,
                                                      const double *__restrict__ s, int64_t cnt,
                                                      T *__restrict__ w, int logR) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < cnt; i += nwaves * lm.G) {
    const int32_t src = p[i];
    T val = vzero(T());
    if (lm.c < nrhs) val = vscale(s[src], bin[(int64_t)src * ldb + lm.c]);
    w[(i << logR) + lm.c] = val;
  }
}

// ---------------------------------------------------------------------------------------------
// S7: y[i] = t[i] * v[q_inv[i]],  rows [0, n)
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_scatter_scale(const T *__restrict__ v,
                                                       const int32_t *__restrict__ qinv,
                                                       const double *__restrict__ t, int64_t n,
                                                       T *__restrict__ yout, int64_t ldy, int nrhs,
                                                       int logR) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < n; i += nwaves * lm.G) {
    if (lm.c < nrhs) {
      const int64_t src = qinv[i];
      yout[i * ldy + lm.c] = vscale(t[i], v[(src << logR) + lm.c]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// one row of a triangular solve.  LOWER: acc = w[i] - sum_asc L(i,j) w[j]; w[i] = acc; v[i] = acc/d[i]
//                                 UPPER: acc = v[i] - sum_desc U(i,j) v[j]; v[i] = acc
// (the CSR row already lists its columns in the order the reference's column sweep meets them)
// ---------------------------------------------------------------------------------------------
template <class T, bool LOWER>
__device__ __forceinline__ void trsv_row(int64_t slot, const int32_t *__restrict__ ptr,
                                         const int32_t *__restrict__ col, const T *__restrict__ val,
                                         const int32_t *__restrict__ rowid, const T *__restrict__ d,
                                         T *w, T *v, int logR, int c) {
  const int64_t i = rowid[slot];
  const int32_t k0 = ptr[slot], k1 = ptr[slot + 1];
  T *x = LOWER ? w : v;
  T acc = x[(i << logR) + c];
  int32_t k = k0;
  for (; k + 4 <= k1; k += 4) {  // 4 independent gathers in flight, accumulated in order
    const int32_t j0 = col[k], j1 = col[k + 1], j2 = col[k + 2], j3 = col[k + 3];
    const T a0 = val[k], a1 = val[k + 1], a2 = val[k + 2], a3 = val[k + 3];
    const T x0 = x[((int64_t)j0 << logR) + c], x1 = x[((int64_t)j1 << logR) + c];
    const T x2 = x[((int64_t)j2 << logR) + c], x3 = x[((int64_t)j3 << logR) + c];
    acc = vsub(acc, vmul(a0, x0));
    acc = vsub(acc, vmul(a1, x1));
    acc = vsub(acc, vmul(a2, x2));
    acc = vsub(acc, vmul(a3, x3));
  }
  for (; k < k1; ++k) acc = vsub(acc, vmul(val[k], x[((int64_t)col[k] << logR) + c]));
  x[(i << logR) + c] = acc;
  if (LOWER) v[(i << logR) + c] = vdiv(acc, d[i]);
}

// wide wavefront: slots [s0, s1) are mutually independent -> any number of workgroups
template <class T, bool LOWER>
__global__ void __launch_bounds__(256) k_trsv_wide(int64_t s0, int64_t s1, const int32_t *__restrict__ ptr,
                                                   const int32_t *__restrict__ col,
                                                   const T *__restrict__ val,
                                                   const int32_t *__restrict__ rowid,
                                                   const T *__restrict__ d, T *w, T *v, int logR) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t slot = s0 + wave * lm.G + lm.g; slot < s1; slot += nwaves * lm.G)
    trsv_row<T, LOWER>(slot, ptr, col, val, rowid, d, w, v, logR, lm.c);
}

// run of thin wavefronts [wf0, wf1): ONE workgroup of 16 waves walks them with a workgroup barrier
// between consecutive wavefronts (all waves share this CU's L1, so workgroup scope suffices) --
// replaces (wf1 - wf0) dependent kernel boundaries by barriers.
template <class T, bool LOWER>
__global__ void __launch_bounds__(1024) k_trsv_seq(int32_t wf0, int32_t wf1,
                                                   const int32_t *__restrict__ wfptr,
                                                   const int32_t *__restrict__ ptr,
                                                   const int32_t *__restrict__ col,
                                                   const T *__restrict__ val,
                                                   const int32_t *__restrict__ rowid,
                                                   const T *__restrict__ d, T *w, T *v, int logR) {
  const LaneMap lm = lane_map(logR);
  const int wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  for (int32_t wf = wf0; wf < wf1; ++wf) {
    const int32_t s0 = wfptr[wf], s1 = wfptr[wf + 1];
    for (int32_t slot = s0 + wave * lm.G + lm.g; slot < s1; slot += nwaves * lm.G)
      trsv_row<T, LOWER>(slot, ptr, col, val, rowid, d, w, v, logR, lm.c);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// S3 / S5:  out[i] = s[p[roff+i]] * b[p[roff+i]] - sum_asc A(i,j) x[j],  rows [0, nrows)
// (accumulate from 0.0 in ascending column order, THEN subtract from the scaled rhs: exactly
//  y = E*work followed by y = s*b - y of prec_solve.hpp:366-368 / :397-399)
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_spmm_epi(int64_t nrows, const int32_t *__restrict__ ptr,
                                                  const int32_t *__restrict__ col,
                                                  const T *__restrict__ val, const T *__restrict__ x,
                                                  const T *__restrict__ bin, int64_t ldb, int nrhs,
                                                  const int32_t *__restrict__ p,
                                                  const double *__restrict__ s, int64_t roff,
                                                  T *__restrict__ out, int logR) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < nrows; i += nwaves * lm.G) {
    const int32_t k0 = ptr[i], k1 = ptr[i + 1];
    T acc = vzero(T());
    int32_t k = k0;
    for (; k + 4 <= k1; k += 4) {
      const int32_t j0 = col[k], j1 = col[k + 1], j2 = col[k + 2], j3 = col[k + 3];
      const T a0 = val[k], a1 = val[k + 1], a2 = val[k + 2], a3 = val[k + 3];
      const T x0 = x[((int64_t)j0 << logR) + lm.c], x1 = x[((int64_t)j1 << logR) + lm.c];
      const T x2 = x[((int64_t)j2 << logR) + lm.c], x3 = x[((int64_t)j3 << logR) + lm.c];
      acc = vadd(acc, vmul(x0, a0));
      acc = vadd(acc, vmul(x1, a1));
      acc = vadd(acc, vmul(x2, a2));
      acc = vadd(acc, vmul(x3, a3));
    }
    for (; k < k1; ++k) acc = vadd(acc, vmul(x[((int64_t)col[k] << logR) + lm.c], val[k]));
    const int32_t src = p[roff + i];
    T rhs = vzero(T());
    if (lm.c < nrhs) rhs = vscale(s[src], bin[(int64_t)src * ldb + lm.c]);
    out[(i << logR) + lm.c] = vsub(rhs, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// outer CRS SpMM: y = A x (RESID = false) or r = b - A x (RESID = true), tmp = 0; tmp += a*x
// ---------------------------------------------------------------------------------------------
template <class T, bool RESID>
__global__ void __launch_bounds__(256) k_crs_spmm(int64_t nrows, const int32_t *__restrict__ ptr,
                                                  const int32_t *__restrict__ col,
                                                  const T *__restrict__ val, const T *__restrict__ x,
                                                  int64_t ldx, const T *__restrict__ b, int64_t ldb,
                                                  T *__restrict__ y, int64_t ldy, int nrhs, int logR) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < nrows; i += nwaves * lm.G) {
    if (lm.c >= nrhs) continue;
    const int32_t k0 = ptr[i], k1 = ptr[i + 1];
    T acc = vzero(T());
    for (int32_t k = k0; k < k1; ++k) acc = vadd(acc, vmul(val[k], x[(int64_t)col[k] * ldx + lm.c]));
    if (RESID) acc = vsub(b[i * ldb + lm.c], acc);
    y[i * ldy + lm.c] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// BLAS-1 helpers of iterative refinement (alg/IterRefine.hpp:99,103,147,156-157), [n][nrhs] blocks
// ---------------------------------------------------------------------------------------------
// op 0: y = 0 | 1: y = x | 2: y += x | 3: y = x + z
template <class T>
__global__ void __launch_bounds__(256) k_vec_op(int op, int64_t n, int nrhs, T *y, int64_t ldy,
                                                const T *x, int64_t ldx, const T *z, int64_t ldz) {
  const int64_t total = n * nrhs;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / nrhs;
    const int c = (int)(e - i * nrhs);
    T r;
    if (op == 0)
      r = vzero(T());
    else if (op == 1)
      r = x[i * ldx + c];
    else if (op == 2)
      r = vadd(y[i * ldy + c], x[i * ldx + c]);
    else
      r = vadd(x[i * ldx + c], z[i * ldz + c]);
    y[i * ldy + c] = r;
  }
}

__device__ __forceinline__ double vabs2(double a) { return a * a; }
__device__ __forceinline__ double vabs2(cplx a) { return a.x * a.x + a.y * a.y; }

// per-column sum of squares; partial[block][c] then a tiny second pass on the host side of the API
template <class T>
__global__ void __launch_bounds__(256) k_colnorm2_partial(int64_t n, int nrhs, const T *x, int64_t ldx,
                                                          double *partial /* [gridDim.x][nrhs] */) {
  __shared__ double sm[256];
  // thread t owns column c = t % nrhs_pad over a strided set of rows
  const int cpad = nrhs;  // nrhs <= 64 here
  const int rows_per_pass = 256 / cpad;
  const int c = threadIdx.x % cpad, rloc = threadIdx.x / cpad;
  double acc = 0.0;
  if (rloc < rows_per_pass)
    for (int64_t i = (int64_t)blockIdx.x * rows_per_pass + rloc; i < n; i += (int64_t)gridDim.x * rows_per_pass)
      acc += vabs2(x[i * ldx + c]);
  sm[threadIdx.x] = (rloc < rows_per_pass) ? acc : 0.0;
  __syncthreads();
  if (threadIdx.x < cpad) {
    double tot = 0.0;
    for (int r = 0; r < rows_per_pass; ++r) tot += sm[r * cpad + threadIdx.x];
    partial[(int64_t)blockIdx.x * nrhs + threadIdx.x] = tot;
  }
}

// ---------------------------------------------------------------------------------------------
// dense last level on the f64 matrix cores: Out[rowmap(i)] = sum_{k=kbeg(i)}^{kend-1} A(i,k) X[k]
//   A column-major (lda), rows >= mrows_valid are treated as zero rows (rank truncation);
//   upper != 0: A is upper triangular, k starts at the row tile's first row;
//   rowmap != NULL: output row permutation (jpvt scatter of QRCP.hpp:400-404).
// One wave owns a 16-row strip x all ceil(R/16) column tiles; v_mfma_f64_16x16x4_f64 lane maps:
//   A: lane l holds A[i = l&15][k = l>>4]; B: B[k = l>>4][j = l&15]; C/D: col = l&15, row = (l>>4) + 4*reg.
// ---------------------------------------------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_dense_gemm_d(int mrows_total, int mrows_valid, int kend, int upper,
                                                      const double *__restrict__ A, int lda,
                                                      const double *__restrict__ X, int logR,
                                                      const int32_t *__restrict__ rowmap,
                                                      double *__restrict__ Out) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((((int64_t)blockIdx.x * blockDim.x) + threadIdx.x) >> 6);
  const int i0 = wave * 16;
  if (i0 >= mrows_total) return;
  const int R = 1 << logR;
  const int ntile = (R + 15) >> 4;
  const int arow = i0 + (lane & 15);
  const int kq = lane >> 4;
  const bool arow_ok = arow < mrows_valid;
  v4f64 acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
  const int kbeg = upper ? i0 : 0;  // i0 is a multiple of 16, hence of 4
  for (int k0 = kbeg; k0 < kend; k0 += 4) {
    const int k = k0 + kq;
    const bool kok = k < kend;
    const double a = (arow_ok && kok) ? A[(int64_t)k * lda + arow] : 0.0;
    for (int t = 0; t < ntile; ++t) {
      const int colx = t * 16 + (lane & 15);
      const double b = (kok && colx < R) ? X[((int64_t)k << logR) + colx] : 0.0;
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
  for (int t = 0; t < ntile; ++t) {
    const int colx = t * 16 + (lane & 15);
    if (colx >= R) continue;
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + kq + 4 * r;
      if (row < mrows_total) {
        const int orow = rowmap ? rowmap[row] : row;
        Out[((int64_t)orow << logR) + colx] = acc[t][r];
      }
    }
  }
}

// complex dense level: plain wave-per-row-strip VALU version (no complex MFMA on gfx950);
// one lane per (row, column) pair of a 64/R-row strip, k sequential.
__global__ void __launch_bounds__(256) k_dense_gemm_z(int mrows_total, int mrows_valid, int kend, int upper,
                                                      const cplx *__restrict__ A, int lda,
                                                      const cplx *__restrict__ X, int logR,
                                                      const int32_t *__restrict__ rowmap,
                                                      cplx *__restrict__ Out) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < mrows_total; i += nwaves * lm.G) {
    cplx acc{0.0, 0.0};
    if (i < mrows_valid)
      for (int k = upper ? (int)i : 0; k < kend; ++k)
        acc = vadd(acc, vmul(A[(int64_t)k * lda + i], X[((int64_t)k << logR) + lm.c]));
    const int64_t orow = rowmap ? rowmap[i] : i;
    Out[(orow << logR) + lm.c] = acc;
  }
}

}  // namespace hifamd
