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
//             D: y[i] /= d[i]                 alg/prec_solve.hpp:219 (fused into the first U kernel that touches a row)
//             U: CCS::solve_as_strict_upper   ds/CompressedStorage.hpp:2357-2369 (+ mrhs :2377)
//   k_trsv_band_p the band form at R = 64 (trsv_band_r64): next row's head behind the last gathers, one poll per item,
//                 carried prefixes of the next band on the idle compute units (host.hpp finish_band_plan)
//   k_spmm_epi    CCS::multiply_nt_low :2079 fused with  y = s[p]*b[p] - y  prec_solve.hpp:366-368,397-399
//                 (R = 64: spmm_stream_r64, the same item/batch pipeline)
//   k_gather_scale   work[i] = s[p[i]]*b[p[i]]           alg/prec_solve.hpp:359,402
//   k_scatter_scale  y[i] = t[i]*work[q_inv[i]]          alg/prec_solve.hpp:411
//   k_dense_gemm     QRCP::_solve_nt (ormqr, trsv, perm) small_scale/QRCP.hpp:371-411 on f64 MFMA
//   k_crs_spmm       CRS::multiply_nt_low(x,istart,len,y) ds/CompressedStorage.hpp:1109-1127
//   k_thin_update + k_tri_gemm_d   the thin tail of a triangle in block-dense form (host.hpp plan_dense_blocks)
//   k_gather_div / k_prod_rows / k_spmm_prod / k_scatter_div   prec_prod, alg/prec_prod.hpp:55-147
//   k_zcombine       complex products as two real MFMA products
//   k_gm_step / k_gm_finish / k_gm_backsolve / k_gm_combine / k_gm_colop   device-resident Arnoldi process of GMRES
//   k_colsum_partial / k_sub_colmean   null-space-filter BLAS-1
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace hifamd {

struct cplx {
  double x, y;
};
typedef double v4f64 __attribute__((ext_vector_type(4)));  // accumulator tile of v_mfma_f64_16x16x4_f64

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

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Rows that are neighbours in
// the matrix share most of the vector rows they gather, so the streaming kernels renumber their
// workgroups such that every XCD walks ONE contiguous window of rows (a gathered row is then fetched
// into one L2 instead of up to eight).
__device__ int g_xcd_remap = 1;
__device__ __forceinline__ unsigned xcd_block() {
  const unsigned g = gridDim.x, b = blockIdx.x;
  if (!g_xcd_remap) return b;
  const unsigned q = g >> 3, r = g & 7u, x = b & 7u;
  return x * q + (x < r ? x : r) + (b >> 3);
}

__device__ __forceinline__ double vdivr(double a, double r) { return a / r; }
__device__ __forceinline__ cplx vdivr(cplx a, double r) { return cplx{a.x / r, a.y / r}; }

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

// A caller-owned vector block as a kernel argument: either a plain pointer, or a pointer that the kernel
// reads from a device slot at run time (+ a column offset).  The latter lets ONE captured hipGraph serve
// every (B, X) pair of a given shape: the slot is rewritten before each replay instead of re-capturing
// the ~550 nodes whenever a Krylov solver hands over a different pair of vectors.
template <class T>
struct IoPtr {
  T *direct;
  T *const *slot;
  int64_t off;
  __device__ __forceinline__ T *get() const { return slot ? (*slot + off) : direct; }
};

// S1 fused into the triangular solve: the kernel that touches a row of L FIRST takes its right-hand side straight from
// the level's input, rhs[i] = s[p[i]] * b[p[i]] (prec_solve.hpp:359), instead of from a w[] that k_gather_scale wrote --
// one launch and one write + read of the level's rows less.  bin.direct == nullptr && bin.slot == nullptr: not fused.
template <class T>
struct FirstL {
  IoPtr<const T> bin;
  int64_t ldb;
  int nrhs;
  const int32_t *p;
  const double *s;
  __device__ __forceinline__ bool on() const { return bin.direct != nullptr || bin.slot != nullptr; }
};
template <class T>
__device__ __forceinline__ T first_l_rhs(const T *__restrict__ bin, const FirstL<T> &f, int32_t i, int lane) {
  const int32_t src = f.p[i];
  const T b = bin[(int64_t)src * f.ldb + min(lane, f.nrhs - 1)];
  return lane < f.nrhs ? vscale(f.s[src], b) : vzero(T());
}

// What a level's FIRST solve may leave out, per slot of a sparse-own triangle (engine.hip build_row_flags; NULL otherwise):
//   bit 0  nobody reads this row's result from memory in this solve: it is not stored (L: a row without entries whose
//          value only its own component uses; U: a row no other component and no column of E refers to);
//   bit 1  (U) the L solve did not store the row: its right-hand side is s[p[i]] * b[p[i]] (FirstL), the product the L
//          kernel would have stored.
// (p[i] and s[p[i]] in slot order beside the flags -- no dependent gathers -- measured SLOWER: the U band 244 -> 316 us.)
struct RowSkip {
  const uint8_t *flag;
};

// S7 fused into the LAST band of a level's final U solve: y[i] = t[i] * v[q_inv[i]] (prec_solve.hpp:411) is, for a row
// r that this band finishes, y[q[r]] = t[q[r]] * v_r -- the band writes the level's output itself and v[r], which
// nothing reads any more, not at all; k_scatter_scale_list serves the rows of the other bands and of the child.
template <class T>
struct LastU {
  IoPtr<T> out;
  int64_t ldy;
  int nrhs;
  const int32_t *q;
  const double *t;
  __device__ __forceinline__ bool on() const { return out.direct != nullptr || out.slot != nullptr; }
};

// ---------------------------------------------------------------------------------------------
// S1: w[i] = s[p[i]] * b[p[i]],  rows [0, cnt)
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_gather_scale(IoPtr<const T> bin_, int64_t ldb, int nrhs,
                                                      const int32_t *__restrict__ p,
                                                      const double *__restrict__ s, int64_t cnt,
                                                      T *__restrict__ w, int logR) {
  const T *__restrict__ bin = bin_.get();
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < cnt; i += nwaves * lm.G) {
    const int32_t src = p[i];
    T val = vzero(T());
    if (lm.c < nrhs) val = vscale(s[src], bin[(int64_t)src * ldb + lm.c]);
    w[(i << logR) + lm.c] = val;
  }
}

// ---------------------------------------------------------------------------------------------
// out[i] = in[map[i]], rows [0, n) of an arena block: the P^T x of the adjoint dense solve
// (QRCP::_solve_t, QRCP.hpp:428-433)
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_row_gather(const T *__restrict__ in, const int32_t *__restrict__ map,
                                                    int64_t n, T *__restrict__ out, int logR) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < n; i += nwaves * lm.G)
    out[(i << logR) + lm.c] = in[((int64_t)map[i] << logR) + lm.c];
}

// ---------------------------------------------------------------------------------------------
// S7: y[i] = t[i] * v[q_inv[i]],  rows [0, n)
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_scatter_scale(const T *__restrict__ v,
                                                       const int32_t *__restrict__ qinv,
                                                       const double *__restrict__ t, int64_t n,
                                                       IoPtr<T> yout_, int64_t ldy, int nrhs,
                                                       int logR) {
  T *__restrict__ yout = yout_.get();
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < n; i += nwaves * lm.G) {
    if (lm.c < nrhs) {
      const int64_t src = qinv[i];
      yout[i * ldy + lm.c] = vscale(t[i], v[(src << logR) + lm.c]);
    }
  }
}

// the same for a LIST of output rows (the rows a fused last U band -- LastU -- does not write itself); R = 64
template <class T>
__global__ void __launch_bounds__(256) k_scatter_scale_list(const T *__restrict__ v, const int32_t *__restrict__ qinv,
                                                            const double *__restrict__ t, const int32_t *__restrict__ list,
                                                            int64_t cnt, IoPtr<T> yout_, int64_t ldy, int nrhs) {
  T *__restrict__ yout = yout_.get();
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // four rows per trip: the three dependent hops of a row (list -> q_inv, t -> v row) overlap across the four
  constexpr int UR = 4;
  for (int64_t k0 = wave * UR; k0 < cnt; k0 += nwaves * UR) {
    int64_t i_[UR];
    int32_t q_[UR];
    double t_[UR];
    T x_[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) i_[u] = list[min(k0 + u, cnt - 1)];
#pragma unroll
    for (int u = 0; u < UR; ++u) q_[u] = qinv[i_[u]], t_[u] = t[i_[u]];
#pragma unroll
    for (int u = 0; u < UR; ++u) x_[u] = v[((int64_t)q_[u] << 6) + min(lane, nrhs - 1)];  // (a narrow batch reads the columns it has)
#pragma unroll
    for (int u = 0; u < UR; ++u)
      if (k0 + u < cnt && lane < nrhs) yout[i_[u] * ldy + lane] = vscale(t_[u], x_[u]);
  }
}

// ---------------------------------------------------------------------------------------------
// one row of a triangular solve.  LOWER: acc = w[i] - sum_asc L(i,j) w[j]; w[i] = acc
//                                 UPPER: acc = w[i]/d[i] (first touch) or v[i]; acc -= sum_desc U(i,j) v[j]; v[i] = acc
// (the CSR row already lists its columns in the order the reference's column sweep meets them)
// ---------------------------------------------------------------------------------------------
template <class T, bool LOWER, bool PREFIX>
__device__ __forceinline__ void trsv_row(int64_t slot, const int32_t *__restrict__ ptr,
                                         const int32_t *__restrict__ split,
                                         const int32_t *__restrict__ col, const T *__restrict__ val,
                                         const int32_t *__restrict__ rowid, const T *__restrict__ d,
                                         T *w, T *v, int logR, int c, bool first_u) {
  const int64_t i = rowid[slot];
  const int32_t k0 = ptr[slot], k1 = PREFIX ? split[slot] : ptr[slot + 1];
  T *x = LOWER ? w : v;
  // U's right-hand side is D^{-1} (L^{-1} w): the kernel that touches a U row FIRST divides (prec_solve.hpp:219)
  T acc = (!LOWER && first_u) ? vdiv(w[(i << logR) + c], d[i]) : x[(i << logR) + c];
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
}

// ---------------------------------------------------------------------------------------------
// R = 64 software pipeline of the level-scheduled solves: a wave owns rows s_first, s_first+stride,
// ... (slot order) and streams them as ITEMS of up to 64 nonzeros.  The 64 lanes fetch an item's
// (column, value, source-slot) triples with ONE coalesced load each and broadcast them with
// v_readlane; the NEXT item (next 64 nonzeros of the row, or the head of the wave's next row with
// its right-hand side and pivot) is always in flight while the current one is consumed, and the
// row headers run two rows ahead on the scalar unit.  Gathers go out eight 512-byte rows at a time;
// accumulation stays in the reference's order.  In place: x[i] holds the rhs on entry and the
// solution on exit.  UPPER rows start from rhs_u[i] / d[i] when this kernel is the first to touch them (first_u).
//   MODE 0: all dependencies were finished by earlier launches                   (k_trsv_wide)
//   MODE 2: dependencies with srcslot >= slot0 are acquired through LDS flags     (k_trsv_band)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t rl32(int32_t v, int idx) { return __builtin_amdgcn_readlane(v, idx); }
__device__ __forceinline__ double rl64(double v, int idx) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffLL), idx);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), idx);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double rlv(double v, int idx) { return rl64(v, idx); }
__device__ __forceinline__ cplx rlv(cplx v, int idx) { return cplx{rl64(v.x, idx), rl64(v.y, idx)}; }
__device__ __forceinline__ int32_t rfl(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// compiler barrier that "touches" a value: nothing that depends on it is scheduled above this point
__device__ __forceinline__ void pin_here(double &v) { asm volatile("" : "+v"(v)::"memory"); }
__device__ __forceinline__ void pin_here(cplx &v) { asm volatile("" : "+v"(v.x), "+v"(v.y)::"memory"); }

// Development probe (make PROBE=1): wall_clock64 stamps of every wave of k_trsv_band -- kernel entry, start
// of work, exit, and per row (the wave's first two) start / first gather / last accumulation / flag, number
// of nonzeros and poll sleeps.  tests/probe_summary.py and tests/probe_rows.py read the dump
// (HIFIR_AMD_PROBE_OUT); profiles/r01_band_*_timestamps.* were made with it.  Compiled out by default.
#ifdef HIFAMD_PROBE
#define HIFAMD_PROBE_ARG , unsigned long long *tsw = nullptr
#define HIFAMD_STAMP(k) \
  if (tsw && lane == 0 && prow < 2) tsw[4 + prow * 6 + (k)] = wall_clock64();
#define HIFAMD_STAMP_VAL(k, v) \
  if (tsw && lane == 0 && prow < 2) tsw[4 + prow * 6 + (k)] = (unsigned long long)(v);
#else
#define HIFAMD_PROBE_ARG
#define HIFAMD_STAMP(k)
#define HIFAMD_STAMP_VAL(k, v)
#endif
// make PROBE=2 (-DHIFAMD_PROBE -DHIFAMD_PROBE_BATCH): instead of the per-row stamps, the wave's FIRST row records the
// timeline of its leading full batches in words 4..15 of its record: (gathers issued, accumulated) x up to six batches
#if defined(HIFAMD_PROBE) && defined(HIFAMD_PROBE_BATCH)
#undef HIFAMD_STAMP
#undef HIFAMD_STAMP_VAL
#define HIFAMD_STAMP(k)
#define HIFAMD_STAMP_VAL(k, v)
#define HIFAMD_STAMP_BATCH(e)                                                                   \
  if (tsw && lane == 0 && prow == 0 && pbatch < 6) {                                            \
    tsw[4 + 2 * pbatch + (e)] = wall_clock64();                                                 \
    pbatch += (e);                                                                              \
  }
#else
#define HIFAMD_STAMP_BATCH(e)
#endif

template <class T, int MODE, bool LOWER, bool PREFIX>
__device__ __forceinline__ bool trsv_stream_r64(int32_t s_first, int32_t s_end, int32_t stride,
                                                const int32_t *__restrict__ ptr, const int32_t *__restrict__ split,
                                                const int32_t *__restrict__ col,
                                                const T *__restrict__ val, const int32_t *__restrict__ srcslot,
                                                const int32_t *__restrict__ rowid, const T *__restrict__ d, T *x,
                                                const T *__restrict__ rhs_u, int lane, int *flag, int32_t slot0,
                                                unsigned *errflag, bool first_u, T *tb = nullptr, int32_t tb_s0 = 0,
                                                int32_t tb_s1 = 0, const T *fl_bin = nullptr,
                                                const FirstL<T> *fl = nullptr HIFAMD_PROBE_ARG) {
  int32_t s = rfl(s_first);
  if (s >= s_end) return true;
  const bool first_l = LOWER && fl_bin != nullptr;  // this kernel is the first to touch its L rows: S1 fused (FirstL)
#ifdef HIFAMD_PROBE
  int prow = 0;
  bool pfirst = false;
#endif
  // nonzero range of a row: the whole row, its PREFIX [ptr, split) (dependencies finished before the
  // thin run started; exact partial sum, no division yet) or, in a thin run, the rest [split, end)
#define HIFAMD_KBEG(sl) rfl((MODE == 2) ? split[sl] : ptr[sl])
#define HIFAMD_KEND(sl) rfl(PREFIX ? split[sl] : ptr[(sl) + 1])
  int32_t i_c = rfl(rowid[s]), k_c = HIFAMD_KBEG(s), e_c = HIFAMD_KEND(s);
  int32_t s_n = s + stride;
  bool has_n = s_n < s_end;
  int32_t i_n = 0, k_n = 0, e_n = 0;
  if (has_n) {
    i_n = rfl(rowid[s_n]);
    k_n = HIFAMD_KBEG(s_n);
    e_n = HIFAMD_KEND(s_n);
  }
  int32_t colv = 0, ssv = 0;
  T valv = vzero(T());
  {
    const int32_t kk = k_c + lane;
    if (kk < e_c) {
      colv = col[kk];
      valv = val[kk];
      if (MODE == 2) ssv = srcslot[kk];
    }
  }
  // U rows start from D^{-1} times the L solution (rhs_u = w) when this kernel is the first to touch them
  const bool div_u = !LOWER && first_u;
  T acc = div_u ? vdiv(rhs_u[((int64_t)i_c << 6) + lane], d[i_c])
                : (first_l ? first_l_rhs(fl_bin, *fl, i_c, lane) : x[((int64_t)i_c << 6) + lane]);
  // head of the wave's next row: its first item, right-hand side, pivot, and the header two rows
  // ahead.  MODE 0 issues it while the current row is consumed; MODE 2 only AFTER the current row's
  // flag is up, because a workgroup-scope release waits for every outstanding vector-memory
  // operation of the wave (vmcnt(0)) and these cold loads would sit on the dependency chain.
#define HIFAMD_PREFETCH_NEXT_ROW()                          \
  do {                                                      \
    const int32_t kk_ = k_n + lane;                         \
    if (kk_ < e_n) {                                        \
      colv2 = col[kk_];                                     \
      valv2 = val[kk_];                                     \
      if (MODE == 2) ssv2 = srcslot[kk_];                   \
    }                                                       \
    acc2 = div_u ? vdiv(rhs_u[((int64_t)i_n << 6) + lane], d[i_n])                                  \
                 : (first_l ? first_l_rhs(fl_bin, *fl, i_n, lane) : x[((int64_t)i_n << 6) + lane]); \
    s_nn = s_n + stride;                                    \
    has_nn = s_nn < s_end;                                  \
    if (has_nn) {                                           \
      i_nn = rfl(rowid[s_nn]);                              \
      k_nn = HIFAMD_KBEG(s_nn);                             \
      e_nn = HIFAMD_KEND(s_nn);                             \
    }                                                       \
  } while (0)
  for (;;) {
    const int32_t cnt = min(64, e_c - k_c);  // <= 0 for an empty row
    const bool row_done = (k_c + 64 >= e_c);
    // ---- prefetch the next item (and, at a row end, the header two rows ahead)
    int32_t colv2 = 0, ssv2 = 0;
    T valv2 = vzero(T()), acc2 = vzero(T());
    int32_t s_nn = 0, i_nn = 0, k_nn = 0, e_nn = 0;
    bool has_nn = false;
    if (!row_done) {
      const int32_t kk = k_c + 64 + lane;
      if (kk < e_c) {
        colv2 = col[kk];
        valv2 = val[kk];
        if (MODE == 2) ssv2 = srcslot[kk];
      }
    } else if (has_n && MODE != 2) {
      HIFAMD_PREFETCH_NEXT_ROW();
    }
    // ---- consume the current item in order.  MODE 2: the item's source slots sit one per lane, so
    // ONE LDS instruction polls the flags of all its in-run dependencies; the ready prefix of the
    // item is consumed (eight gathers per batch) and only then is the rest polled again.
    int32_t t = 0;
    unsigned spins = 0;
#ifdef HIFAMD_PROBE
    if (k_c == HIFAMD_KBEG(s)) {
      HIFAMD_STAMP(0)
      HIFAMD_STAMP_VAL(4, e_c - k_c)
      pfirst = false;
    }
#endif
    while (t < cnt) {
      int32_t lim = cnt;
      if (MODE == 2) {
        const bool inrun = lane >= t && lane < cnt && ssv >= slot0;
        bool rdy = true;
        if (inrun) rdy = __hip_atomic_load(&flag[ssv - slot0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
        const unsigned long long pending = ~__ballot(rdy);
        if (pending) lim = min(cnt, (int32_t)__builtin_ctzll(pending));
        if (lim <= t) {
          __builtin_amdgcn_s_sleep(1);
          if ((++spins & 4095u) == 0 && spins > (1u << 24)) {
            if (lane == 0) __hip_atomic_store(errflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
          }
          continue;
        }
      }
#ifdef HIFAMD_PROBE
      if (!pfirst) {
        HIFAMD_STAMP(1)
        pfirst = true;
      }
#endif
      while (t < lim) {
        const int nb = min(8, lim - t);
        int32_t j[8];
        T a[8], xv[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const int idx = min(t + b, 63);
          j[b] = rl32(colv, idx);
          a[b] = rlv(valv, idx);
        }
#pragma unroll
        for (int b = 0; b < 8; ++b)
          if (b < nb) xv[b] = x[((int64_t)j[b] << 6) + lane];
#pragma unroll
        for (int b = 0; b < 8; ++b)
          if (b < nb) acc = vsub(acc, vmul(a[b], xv[b]));
        t += nb;
      }
    }
    if (row_done) {
      HIFAMD_STAMP(2)
      // PREFIX pass of a block-dense band: the rows of the band's FIRST block have no other contribution
      // before their block product, so their partial sums go straight into the product's right-hand side
      // (saves that block's k_thin_update launch)
      if (PREFIX && tb && s < tb_s1)
        tb[((int64_t)(s - tb_s0) << 6) + lane] = acc;
      else
        x[((int64_t)i_c << 6) + lane] = acc;
      if (MODE == 2) {  // release: the row's stores (all 64 lanes) are complete before its flag goes up
        if (lane == 0) __hip_atomic_store(&flag[s - slot0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
#ifdef HIFAMD_PROBE
      HIFAMD_STAMP(3)
      HIFAMD_STAMP_VAL(5, spins)
      ++prow;
#endif
      if (!has_n) break;
      if (MODE == 2) HIFAMD_PREFETCH_NEXT_ROW();
      s = s_n;
      i_c = i_n;
      k_c = k_n;
      e_c = e_n;
      acc = acc2;
      s_n = s_nn;
      has_n = has_nn;
      i_n = i_nn;
      k_n = k_nn;
      e_n = e_nn;
    } else {
      k_c += 64;
    }
    colv = colv2;
    valv = valv2;
    ssv = ssv2;
  }
  return true;
}

// wide wavefront: slots [s0, s1) are mutually independent -> any number of workgroups
template <class T, bool LOWER, bool PREFIX>
__global__ void __launch_bounds__(256) k_trsv_wide(int64_t s0, int64_t s1, const int32_t *__restrict__ ptr,
                                                   const int32_t *__restrict__ split,
                                                   const int32_t *__restrict__ col,
                                                   const T *__restrict__ val,
                                                   const int32_t *__restrict__ rowid,
                                                   const T *__restrict__ d, T *w, T *v, int logR, int first_u,
                                                   T *tb, int32_t tb_s1, FirstL<T> fl) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  if (logR == 6) {  // (tb: R = 64 only; rows [s0, tb_s1) deliver their result to tb, see trsv_stream_r64)
    const T *fl_bin = (LOWER && fl.on()) ? fl.bin.get() : nullptr;
    trsv_stream_r64<T, 0, LOWER, PREFIX>((int32_t)(s0 + wave), (int32_t)s1, (int32_t)nwaves, ptr, split, col, val,
                                         nullptr, rowid, d, LOWER ? w : v, w, threadIdx.x & 63, nullptr, 0, nullptr,
                                         first_u != 0, tb, (int32_t)s0, tb_s1, fl_bin, &fl);
    return;
  }
  for (int64_t slot = s0 + wave * lm.G + lm.g; slot < s1; slot += nwaves * lm.G)
    trsv_row<T, LOWER, PREFIX>(slot, ptr, split, col, val, rowid, d, w, v, logR, lm.c, first_u != 0);
}

// ---------------------------------------------------------------------------------------------
// One BAND of a triangle (host.hpp BandPlan): workgroup wg0 + blockIdx.x owns the slot range of its
// groups, i.e. a set of whole connected components of the band's dependency graph.  Dependencies on
// rows before the band were finished by earlier launches (plain loads); dependencies inside the
// workgroup's own range are handed over through one LDS flag per row with workgroup-scope
// release/acquire (all waves share this CU's L1/L2 path).  No workgroup ever waits for another
// one, so any number of them may run in any order.  In place: x[i] holds the rhs on entry and the
// solution on exit (LOWER also writes v[i] = x[i] / d[i]).
// R = 64: rows are dealt round-robin over the 16 waves in SLOT order (a row only waits for smaller
// slots of the same workgroup, so this cannot deadlock) and streamed through trsv_stream_r64.
// R < 64: a wave takes G = 64/R rows of ONE group (same depth, hence independent) at a time.
// If the band had a PREFIX pass (k_trsv_wide<.., true>), rows start at split[slot].
// ---------------------------------------------------------------------------------------------
#define HIFAMD_TAIL_MAX 16384
template <class T, bool LOWER>
__global__ void __launch_bounds__(1024) k_trsv_band(int32_t wg0, const int32_t *__restrict__ wg_grp_ptr,
                                                    const int32_t *__restrict__ grp_slot_ptr,
                                                    const int32_t *__restrict__ ptr,
                                                    const int32_t *__restrict__ split,
                                                    const int32_t *__restrict__ col, const T *__restrict__ val,
                                                    const int32_t *__restrict__ srcslot,
                                                    const int32_t *__restrict__ rowid, const T *__restrict__ d,
                                                    T *w, T *v, int logR, unsigned *errflag, int first_u
#ifdef HIFAMD_PROBE
                                                    ,
                                                    unsigned long long *ts, int probe_id
#endif
) {
  __shared__ int flag[HIFAMD_TAIL_MAX];
  const LaneMap lm = lane_map(logR);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#ifdef HIFAMD_PROBE  // [launch][workgroup < 256][wave][16]: 0 entry, 2 start of work, 3 exit, 4.. two rows x 6
  unsigned long long *tsw =
      (ts && blockIdx.x < 256) ? ts + (((size_t)probe_id * 256 + blockIdx.x) * 16 + wave) * 16 : nullptr;
  if (tsw && lane == 0) tsw[0] = wall_clock64();
#endif
  const int32_t wf0 = wg_grp_ptr[wg0 + blockIdx.x], wf1 = wg_grp_ptr[wg0 + blockIdx.x + 1];
  const int32_t *wfptr = grp_slot_ptr;
  const int32_t slot0 = wfptr[wf0], slot1 = wfptr[wf1];
  for (int t = threadIdx.x; t < slot1 - slot0; t += blockDim.x) flag[t] = 0;
  __syncthreads();
  T *x = LOWER ? w : v;
  if (logR == 6) {
#ifdef HIFAMD_PROBE
    if (tsw && lane == 0) tsw[2] = wall_clock64();
    trsv_stream_r64<T, 2, LOWER, false>(slot0 + wave, slot1, nw, ptr, split, col, val, srcslot, rowid, d, x, w, lane,
                                        flag, slot0, errflag, first_u != 0, nullptr, 0, 0, tsw);
    if (tsw && lane == 0) tsw[3] = wall_clock64();
#else
    trsv_stream_r64<T, 2, LOWER, false>(slot0 + wave, slot1, nw, ptr, split, col, val, srcslot, rowid, d, x, w, lane,
                                        flag, slot0, errflag, first_u != 0);
#endif
    return;
  }
  for (int32_t wf = wf0; wf < wf1; ++wf) {
    const int32_t s0 = wfptr[wf], s1 = wfptr[wf + 1];
    for (int32_t base = s0 + wave * lm.G; base < s1; base += nw * lm.G) {
      const int32_t slot = base + lm.g;
      const bool active = slot < s1;
      int64_t i = 0;
      int32_t k = 0, k1 = 0;
      T acc = vzero(T());
      if (active) {
        i = rowid[slot];
        k = split[slot];  // [ptr, split) was folded in by the prefix pass
        k1 = ptr[slot + 1];
        acc = (!LOWER && first_u) ? vdiv(w[(i << logR) + lm.c], d[i]) : x[(i << logR) + lm.c];
      }
      while (__any(k < k1)) {
        const int nb = min(4, k1 - k);
        int32_t j[4], ss[4];
        T a[4], xv[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          j[b] = 0;
          ss[b] = -1;
          a[b] = vzero(T());
          if (b < nb) {
            j[b] = col[k + b];
            a[b] = val[k + b];
            ss[b] = srcslot[k + b];
          }
        }
        unsigned spins = 0;
        for (;;) {  // acquire every in-run dependency of this batch (rows of earlier wavefronts only)
          bool ok = true;
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (b < nb && ss[b] >= slot0)
              ok = ok && (__hip_atomic_load(&flag[ss[b] - slot0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0);
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(1);
          if ((++spins & 4095u) == 0 && spins > (1u << 24)) {
            if (lane == 0) __hip_atomic_store(errflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
          }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (b < nb) xv[b] = x[((int64_t)j[b] << logR) + lm.c];
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (b < nb) acc = vsub(acc, vmul(a[b], xv[b]));
        if (nb > 0) k += nb;
      }
      if (active) {
        x[(i << logR) + lm.c] = acc;
      }
      // release: every lane's stores are complete before the row's flag goes up
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (active && lm.c == 0)
        __hip_atomic_store(&flag[slot - slot0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The band form of the R = 64 pipeline (k_trsv_band_p, the default at R = 64): dependencies with srcslot >= slot0 are
// acquired through LDS flags.  Per row and wave the first version (trsv_stream_r64 MODE 2, still selectable with
// HIFIR_AMD_BAND_PIPE=0) pays two serial memory round trips -- the row's gathers, then, behind the row's flag, the head
// of the wave's next row -- plus scalar-memory waits for the row headers.  This version overlaps them:
//  * row headers (row id, nonzero range, pivot) of the wave's next 64 rows sit one per lane in registers (one vector
//    load per array and 64 rows, the following block is requested half a block ahead) and are broadcast with
//    v_readlane: no scalar-memory wait inside the loop;
//  * the head of the wave's NEXT row (first item, right-hand side) goes out right BEHIND the last gathers of the
//    current row.  Loads return in order: issued in front of the gathers these cold loads would delay them and with
//    them the row's flag; issued after the flag they add a full memory round trip to every row;
//  * that last batch, the trailing loads, the accumulation, the row's store and its flag are ONE basic block of
//    straight-line code per batch size (0, 1, 2, 3, 4, 6 or 8 gathers; slots beyond the row's end read one valid word
//    that is ignored, addresses are selected instead of branched on).  The compiler then counts its waits exactly
//    (e.g. s_waitcnt vmcnt(11) ... vmcnt(4) in front of eight accumulations: only the gathers are waited for) and has
//    no block boundary at which to copy a register that a trailing load is still going to write.  The next row's
//    data is first touched behind the flag (pin_here).  The batch sizes matter: with eight gathers always, the
//    padding loads of the one- and two-nonzero rows of the wide first bands cost more gather slots of the compute
//    unit than the overlap saves (measured: 10.92 -> 11.16 ms; with the sizes above 10.04 -> 9.65 ms).
// Longer rows run their leading full batches through plain loops in front of that block.
// Accumulation order is the reference's (exact mode stays bit-exact).  In place: x[i] holds the rhs on entry, the
// solution on exit; UPPER rows start from rhs_u[i] / d[i] when this kernel is the first to touch them (first_u).
// ---------------------------------------------------------------------------------------------
template <class T, bool LOWER>
__device__ __forceinline__ bool trsv_band_r64(const int32_t s_first_, const int32_t s_end, const int32_t stride,
                                              const int32_t *__restrict__ ptr, const int32_t *__restrict__ split,
                                              const int32_t *__restrict__ col, const T *__restrict__ val,
                                              const int32_t *__restrict__ srcslot,
                                              const int32_t *__restrict__ rowid, const T *__restrict__ d, T *x,
                                              const T *__restrict__ rhs_u, const int lane, int *flag,
                                              const int32_t slot0, unsigned *errflag, const bool first_u,
                                              const T *fl_bin = nullptr, const FirstL<T> *fl = nullptr HIFAMD_PROBE_ARG) {
#ifdef HIFAMD_PROBE
  int prow = 0;
  int pbatch = 0;
  (void)pbatch;
#endif
  const int32_t s_first = rfl(s_first_);  // (wave-uniform, which the compiler cannot see from threadIdx.x >> 6)
  const bool div_u = !LOWER && first_u;
  // S1 fused (FirstL): this kernel touches its L rows first and reads rhs[i] = s[p[i]] * b[p[i]]; the row's source row
  // p[i] and scale s[p[i]] travel in the header blocks, so the loop still issues ONE right-hand-side load per row
  const bool first_l = LOWER && first_u && fl_bin != nullptr;
  const T *rhs = div_u ? rhs_u : (first_l ? fl_bin : (const T *)x);
  const int64_t rstride = first_l ? fl->ldb : 64;
  const int rlane = first_l ? min(lane, fl->nrhs - 1) : lane;
  // header blocks: lane l holds the header of the wave's row number (64 * block + l)
  int32_t h_i = 0, h_k = 0, h_e = 0, g_i = 0, g_k = 0, g_e = 0, h_p = 0, g_p = 0;
  T h_d = vzero(T()), g_d = vzero(T());
  double h_s = 0.0, g_s = 0.0;
#define HIFAMD_LOAD_HDR(hi, hk, he, hd, hp, hs, base)    \
  {                                                      \
    const int32_t sl_ = (base) + lane * stride;          \
    if (sl_ < s_end) {                                   \
      hi = rowid[sl_];                                   \
      hk = split[sl_];                                   \
      he = ptr[sl_ + 1];                                 \
      if (div_u) hd = d[hi];                             \
      if (first_l) {                                     \
        hp = fl->p[hi];                                  \
        hs = fl->s[hp];                                  \
      }                                                  \
    }                                                    \
  }
  // right-hand side of row (id i, source row pr, scale sc) as loaded / as used
#define HIFAMD_RHS_AT(i, pr) rhs[(int64_t)(first_l ? (pr) : (i)) * rstride + rlane]
#define HIFAMD_RHS_FIX(val, sc) (first_l ? (lane < fl->nrhs ? vscale((sc), (val)) : vzero(T())) : (val))
  // wait until the leading `hi` nonzeros of the current item (item_cnt of them valid) have their in-band sources
  // finished.  ONE LDS instruction polls the flags of the whole item (source slots sit one per lane) and `ready`
  // remembers how many leading nonzeros were found ready, so that a row whose sources all finished before the
  // band -- phase 1 of every band -- polls once, not once per batch.
#define HIFAMD_BAND_POLL(hi)                                                                                      \
  if (ready < (hi)) {                                                                                             \
    unsigned spins_ = 0;                                                                                          \
    for (;;) {                                                                                                    \
      bool rdy_ = true;                                                                                           \
      if (lane < item_cnt && ssv >= slot0)                                                                        \
        rdy_ = __hip_atomic_load(&flag[ssv - slot0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;        \
      const unsigned long long pend_ = ~__ballot(rdy_);                                                           \
      ready = pend_ ? (int32_t)__builtin_ctzll(pend_) : 64;                                                       \
      if (ready >= (hi)) break;                                                                                   \
      __builtin_amdgcn_s_sleep(1);                                                                                \
      if ((++spins_ & 4095u) == 0 && spins_ > (1u << 24)) {                                                       \
        if (lane == 0) __hip_atomic_store(errflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);               \
        return false;                                                                                             \
      }                                                                                                           \
    }                                                                                                             \
  }
  // a full batch of eight nonzeros starting at lane t of the current item (plain: leading part of longer rows)
#define HIFAMD_BAND_FULL8(t)                                                                 \
  {                                                                                          \
    int32_t j_[8];                                                                           \
    T a_[8], xv_[8];                                                                         \
    _Pragma("unroll") for (int b = 0; b < 8; ++b) {                                          \
      j_[b] = rl32(colv, (t) + b);                                                           \
      a_[b] = rlv(valv, (t) + b);                                                            \
    }                                                                                        \
    _Pragma("unroll") for (int b = 0; b < 8; ++b) xv_[b] = x[((int64_t)j_[b] << 6) + lane];  \
    HIFAMD_STAMP_BATCH(0) /* (all eight gathers issued) */                                   \
    _Pragma("unroll") for (int b = 0; b < 8; ++b) acc = vsub(acc, vmul(a_[b], xv_[b]));      \
    HIFAMD_STAMP_BATCH(1) /* (all eight accumulated) */                                      \
  }
  HIFAMD_LOAD_HDR(h_i, h_k, h_e, h_d, h_p, h_s, s_first)  // (requested first: in flight while the flags are cleared)
  // the workgroup's flags: one per slot of its range [slot0, s_end)
  for (int t_ = (int)threadIdx.x; t_ < s_end - slot0; t_ += (int)blockDim.x) flag[t_] = 0;
  __syncthreads();
  if (s_first >= s_end) return true;
  int hidx = 0;
  int32_t s = s_first;
  int32_t i_c = rl32(h_i, 0), k_c = rl32(h_k, 0), e_c = rl32(h_e, 0);
  int32_t colv = col[k_c + lane], ssv = srcslot[k_c + lane];  // (80 padding elements: DevCsr::upload)
  T valv = val[k_c + lane];
  T acc = HIFAMD_RHS_AT(i_c, rl32(h_p, 0));
  acc = HIFAMD_RHS_FIX(acc, rl64(h_s, 0));
  if (div_u) acc = vdiv(acc, rlv(h_d, 0));
  for (;;) {  // one row per iteration
    // header of the wave's next row: registers only.  Without a next row the trailing loads re-read
    // the current row (valid, ignored).
    const int32_t s_n = s + stride;
    const bool has_n = s_n < s_end;
    int32_t i_n = i_c, k_n = k_c, e_n = e_c, p_n = rl32(h_p, hidx);
    T d_n = vzero(T());
    double s_n2 = rl64(h_s, hidx);
    if (has_n) {
      if (hidx < 63) {
        i_n = rl32(h_i, hidx + 1);
        k_n = rl32(h_k, hidx + 1);
        e_n = rl32(h_e, hidx + 1);
        if (div_u) d_n = rlv(h_d, hidx + 1);
        if (first_l) p_n = rl32(h_p, hidx + 1), s_n2 = rl64(h_s, hidx + 1);
      } else {
        i_n = rl32(g_i, 0);
        k_n = rl32(g_k, 0);
        e_n = rl32(g_e, 0);
        if (div_u) d_n = rlv(g_d, 0);
        if (first_l) p_n = rl32(g_p, 0), s_n2 = rl64(g_s, 0);
      }
    }
    // the following header block is requested half a block ahead
    if (hidx == 32 && s + 32 * stride < s_end) HIFAMD_LOAD_HDR(g_i, g_k, g_e, g_d, g_p, g_s, s + 32 * stride)
    // ---- long rows: full 64-nonzero items in front of the row's last item
    while (e_c - k_c > 64) {
      const int32_t c2 = col[k_c + 64 + lane], s2 = srcslot[k_c + 64 + lane];
      const T v2 = val[k_c + 64 + lane];
      {
        const int32_t item_cnt = 64;
        int32_t ready = 0;
        for (int t = 0; t < 64; t += 8) {
          HIFAMD_BAND_POLL(t + 8)
          HIFAMD_BAND_FULL8(t)
        }
      }
      colv = c2;
      valv = v2;
      ssv = s2;
      k_c += 64;
    }
    // ---- the row's last item: leading full batches ...
    const int32_t cnt = e_c - k_c;  // <= 64, <= 0 for a row without nonzeros in this band
    const int32_t item_cnt = cnt;
    int32_t ready = 0, t = 0;
    HIFAMD_STAMP(0)
    HIFAMD_STAMP_VAL(4, cnt)
    while (cnt - t > 8) {
      HIFAMD_BAND_POLL(t + 8)
      HIFAMD_BAND_FULL8(t)
      t += 8;
    }
    // ---- ... and the last batch (0..8 nonzeros) with everything behind it, one basic block per batch size
    // (1, 2, 4 or 8 gathers issued: the rows of the wide first bands have one or two nonzeros, and every padding
    // load costs a slot of the compute unit's gather rate)
    const int nb = max(cnt - t, 0);
    HIFAMD_STAMP(1)  // (leading batches done)
    HIFAMD_BAND_POLL(t + nb)
    HIFAMD_STAMP(2)  // (the last batch's sources are finished)
    {
      const T *xrow = x + lane;
      const T *xdummy = x + ((int64_t)i_c << 6);  // one valid word for the slots beyond nb
#define HIFAMD_BAND_FINAL(NBT)                                                                            \
  {                                                                                                       \
    const T *pj[NBT ? NBT : 1];                                                                           \
    T a[NBT ? NBT : 1], xv[NBT ? NBT : 1];                                                                                    \
    _Pragma("unroll") for (int b = 0; b < NBT; ++b) {                                                     \
      const int idx = min(t + b, 63);                                                                     \
      const int32_t jb = rl32(colv, idx);                                                                 \
      a[b] = rlv(valv, idx);                                                                              \
      pj[b] = (b < nb) ? xrow + ((int64_t)jb << 6) : xdummy;                                              \
    }                                                                                                     \
    _Pragma("unroll") for (int b = 0; b < NBT; ++b) xv[b] = *pj[b];                                       \
    colv = col[k_n + lane];                                                                               \
    valv = val[k_n + lane];                                                                               \
    ssv = srcslot[k_n + lane];                                                                            \
    T acc2 = HIFAMD_RHS_AT(i_n, p_n);                                                                     \
    asm volatile("" ::: "memory"); /* pins the trailing loads HERE */                                     \
    _Pragma("unroll") for (int b = 0; b < NBT; ++b) {                                                     \
      const T pr = vmul(a[b], xv[b]);                                                                     \
      if (b < nb) acc = vsub(acc, pr);                                                                    \
    }                                                                                                     \
    x[((int64_t)i_c << 6) + lane] = acc;                                                                  \
    /* release: the row's stores (all 64 lanes) are ordered before its flag */                            \
    if (lane == 0) __hip_atomic_store(&flag[s - slot0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); \
    pin_here(acc2); /* the next row's data is first touched HERE, behind the flag */                      \
    acc = div_u ? vdiv(acc2, d_n) : HIFAMD_RHS_FIX(acc2, s_n2);                                           \
  }
      if (nb <= 0)
        HIFAMD_BAND_FINAL(0)
      else if (nb <= 1)
        HIFAMD_BAND_FINAL(1)
      else if (nb <= 2)
        HIFAMD_BAND_FINAL(2)
      else if (nb <= 3)
        HIFAMD_BAND_FINAL(3)
      else if (nb <= 4)
        HIFAMD_BAND_FINAL(4)
      else if (nb <= 6)
        HIFAMD_BAND_FINAL(6)
      else
        HIFAMD_BAND_FINAL(8)
#undef HIFAMD_BAND_FINAL
    }
    HIFAMD_STAMP(3)  // (row stored, flag raised, next row's head has arrived)
#ifdef HIFAMD_PROBE
    ++prow;
#endif
    if (!has_n) break;
    s = s_n;
    i_c = i_n;
    k_c = k_n;
    e_c = e_n;
    if (++hidx == 64) {
      hidx = 0;
      h_i = g_i;
      h_k = g_k;
      h_e = g_e;
      h_d = g_d;
      h_p = g_p;
      h_s = g_s;
    }
  }
#undef HIFAMD_LOAD_HDR
#undef HIFAMD_RHS_AT
#undef HIFAMD_RHS_FIX
#undef HIFAMD_BAND_POLL
#undef HIFAMD_BAND_FULL8
  return true;
}

template <class T, bool LOWER>
__global__ void __launch_bounds__(1024) k_trsv_band_p(int32_t wg0, const int32_t *__restrict__ wg_slot,
                                                      const int32_t *__restrict__ ptr,
                                                      const int32_t *__restrict__ split,
                                                      const int32_t *__restrict__ col, const T *__restrict__ val,
                                                      const int32_t *__restrict__ srcslot,
                                                      const int32_t *__restrict__ rowid, const T *__restrict__ d,
                                                      T *w, T *v, unsigned *errflag, int first_u, int32_t n_band,
                                                      int32_t ps0, int32_t ps1, FirstL<T> fl
#ifdef HIFAMD_PROBE
                                                      ,
                                                      unsigned long long *ts, int probe_id
#endif
) {
  __shared__ int flag[HIFAMD_TAIL_MAX];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // workgroups beyond the band's own: the carried prefix of the NEXT band, slots [ps0, ps1), over the sources older than
  // this band (host.hpp finish_band_plan) -- independent of everything this launch computes, on otherwise idle units
  const T *fl_bin = (LOWER && fl.on()) ? fl.bin.get() : nullptr;
  if ((int32_t)blockIdx.x >= n_band) {
    const int32_t pw = __builtin_amdgcn_readfirstlane(((int32_t)blockIdx.x - n_band) * nw + wave);
    trsv_stream_r64<T, 0, LOWER, true>(ps0 + pw, ps1, ((int32_t)gridDim.x - n_band) * nw, ptr, split, col, val, nullptr, rowid,
                                       d, LOWER ? w : v, w, lane, nullptr, 0, nullptr, true, nullptr, 0, 0, fl_bin, &fl);
    return;
  }
#ifdef HIFAMD_PROBE  // same record layout as k_trsv_band: 0 entry, 2 start of work, 3 exit, 4.. two rows x 6
  unsigned long long *tsw =
      (ts && blockIdx.x < 256) ? ts + (((size_t)probe_id * 256 + blockIdx.x) * 16 + wave) * 16 : nullptr;
  if (tsw && lane == 0) tsw[0] = tsw[2] = wall_clock64();
#endif
  // the workgroup's slot range in ONE load level (wg_slot[g] = grp_slot_ptr[wg_grp_ptr[g]])
  const int32_t slot0 = wg_slot[wg0 + blockIdx.x], slot1 = wg_slot[wg0 + blockIdx.x + 1];
#ifdef HIFAMD_PROBE
  trsv_band_r64<T, LOWER>(slot0 + wave, slot1, nw, ptr, split, col, val, srcslot, rowid, d, LOWER ? w : v, w, lane, flag,
                          slot0, errflag, first_u != 0, fl_bin, &fl, tsw);
  if (tsw && lane == 0) tsw[3] = wall_clock64();
#else
  trsv_band_r64<T, LOWER>(slot0 + wave, slot1, nw, ptr, split, col, val, srcslot, rowid, d, LOWER ? w : v, w, lane, flag,
                          slot0, errflag, first_u != 0, fl_bin, &fl);
#endif
}

// ---------------------------------------------------------------------------------------------
// Block-dense thin bands (host.hpp build_dense_blocks), step (1) for the block of slots [r0, r1):
//   t[s - r0] = x[row(s)] - sum over the row's nonzeros in [split, end) whose source slot lies
//               BEFORE the block (rows before the band or in earlier blocks: all finished).
// Step (2) is k_dense_gemm with the block's explicit inverse.  One row per wave group.
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_thin_update(int32_t r0, int32_t r1, const int32_t *__restrict__ ptr,
                                                     const int32_t *__restrict__ split,
                                                     const int32_t *__restrict__ col, const T *__restrict__ val,
                                                     const int32_t *__restrict__ srcslot,
                                                     const int32_t *__restrict__ rowid, const T *__restrict__ x,
                                                     T *__restrict__ tbuf, int logR) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t s = r0 + wave * lm.G + lm.g; s < r1; s += nwaves * lm.G) {
    const int64_t i = rowid[s];
    const int32_t k0 = split[s], k1 = ptr[s + 1];
    T acc = x[(i << logR) + lm.c];
    int32_t k = k0;
    for (; k + 4 <= k1; k += 4) {
      const int32_t j0 = col[k], j1 = col[k + 1], j2 = col[k + 2], j3 = col[k + 3];
      const int32_t q0 = srcslot[k], q1 = srcslot[k + 1], q2 = srcslot[k + 2], q3 = srcslot[k + 3];
      const T a0 = val[k], a1 = val[k + 1], a2 = val[k + 2], a3 = val[k + 3];
      const T x0 = x[((int64_t)j0 << logR) + lm.c], x1 = x[((int64_t)j1 << logR) + lm.c];
      const T x2 = x[((int64_t)j2 << logR) + lm.c], x3 = x[((int64_t)j3 << logR) + lm.c];
      if (q0 < r0) acc = vsub(acc, vmul(a0, x0));
      if (q1 < r0) acc = vsub(acc, vmul(a1, x1));
      if (q2 < r0) acc = vsub(acc, vmul(a2, x2));
      if (q3 < r0) acc = vsub(acc, vmul(a3, x3));
    }
    for (; k < k1; ++k)
      if (srcslot[k] < r0) acc = vsub(acc, vmul(val[k], x[((int64_t)col[k] << logR) + lm.c]));
    tbuf[((s - r0) << logR) + lm.c] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// R = 64 software pipeline of the Schur products (k_spmm_epi): a wave owns rows row0, row0 + stride, ...
// and streams each as items of up to 64 nonzeros, exactly like trsv_stream_r64 (MODE 0): the lanes fetch
// an item's (column, value) pairs with one coalesced load each and broadcast them with v_readlane,
// gathers go out eight 512-byte rows at a time, and while an item is consumed the next one (or the first
// item, scaled right-hand side and header of the wave's next row) is already in flight.  The sum starts
// from 0.0 and runs in ascending column order; the scaled right-hand side is combined at the end, as
// y = E*work; y = s*b - y does (prec_solve.hpp:366-368, :397-399).
// ---------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void spmm_stream_r64(int64_t row0, int64_t nrows, int64_t stride,
                                                const int32_t *__restrict__ ptr, const int32_t *__restrict__ col,
                                                const T *__restrict__ val, const T *__restrict__ x,
                                                const T *__restrict__ bin, int64_t ldb, int nrhs,
                                                const int32_t *__restrict__ p, const double *__restrict__ s,
                                                int64_t roff, T *__restrict__ out, int lane) {
  if (row0 >= nrows) return;
  int64_t i = row0;
  int32_t k_c = rfl(ptr[i]), e_c = rfl(ptr[i + 1]);
  int64_t i_n = i + stride;
  bool has_n = i_n < nrows;
  int32_t k_n = 0, e_n = 0;
  if (has_n) {
    k_n = rfl(ptr[i_n]);
    e_n = rfl(ptr[i_n + 1]);
  }
  int32_t colv = 0;
  T valv = vzero(T());
  if (k_c + lane < e_c) {
    colv = col[k_c + lane];
    valv = val[k_c + lane];
  }
  T rhs = vzero(T());
  {
    const int32_t src = rfl(p[roff + i]);
    if (lane < nrhs) rhs = vscale(s[src], bin[(int64_t)src * ldb + lane]);
  }
  T acc = vzero(T());
  for (;;) {
    const int32_t cnt = min(64, e_c - k_c);  // <= 0 for an empty row
    const bool row_done = (k_c + 64 >= e_c);
    int32_t colv2 = 0;
    T valv2 = vzero(T()), rhs2 = vzero(T());
    int64_t i_nn = 0;
    int32_t k_nn = 0, e_nn = 0;
    bool has_nn = false;
    if (!row_done) {
      const int32_t kk = k_c + 64 + lane;
      if (kk < e_c) {
        colv2 = col[kk];
        valv2 = val[kk];
      }
    } else if (has_n) {
      const int32_t kk = k_n + lane;
      if (kk < e_n) {
        colv2 = col[kk];
        valv2 = val[kk];
      }
      const int32_t src = rfl(p[roff + i_n]);
      if (lane < nrhs) rhs2 = vscale(s[src], bin[(int64_t)src * ldb + lane]);
      i_nn = i_n + stride;
      has_nn = i_nn < nrows;
      if (has_nn) {
        k_nn = rfl(ptr[i_nn]);
        e_nn = rfl(ptr[i_nn + 1]);
      }
    }
    int32_t t = 0;
    while (t < cnt) {
      const int nb = min(8, cnt - t);
      int32_t j[8];
      T a[8], xv[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int idx = min(t + b, 63);
        j[b] = rl32(colv, idx);
        a[b] = rlv(valv, idx);
      }
#pragma unroll
      for (int b = 0; b < 8; ++b)
        if (b < nb) xv[b] = x[((int64_t)j[b] << 6) + lane];
#pragma unroll
      for (int b = 0; b < 8; ++b)
        if (b < nb) acc = vadd(acc, vmul(xv[b], a[b]));
      t += nb;
    }
    if (row_done) {
      out[(i << 6) + lane] = vsub(rhs, acc);
      if (!has_n) break;
      i = i_n;
      k_c = k_n;
      e_c = e_n;
      rhs = rhs2;
      acc = vzero(T());
      i_n = i_nn;
      has_n = has_nn;
      k_n = k_nn;
      e_n = e_nn;
    } else {
      k_c += 64;
    }
    colv = colv2;
    valv = valv2;
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
                                                  IoPtr<const T> bin_, int64_t ldb, int nrhs,
                                                  const int32_t *__restrict__ p,
                                                  const double *__restrict__ s, int64_t roff,
                                                  T *__restrict__ out, int logR) {
  const T *__restrict__ bin = bin_.get();
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  if (logR == 6) {
    spmm_stream_r64<T>(__builtin_amdgcn_readfirstlane((int)wave), nrows, nwaves, ptr, col, val, x, bin, ldb, nrhs, p, s,
                       roff, out, threadIdx.x & 63);
    return;
  }
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
// S3 / S5 for a NARROW batch in the 64-column arena (round 3): k_spmm_epi at R = 64 spends a whole wave -- and a 512-byte
// (complex: 1 KB) gather -- on every nonzero, whatever the batch width.  Here a wave is G = 64 / R row groups of R lanes
// (R = 16 or 32 >= nrhs): G rows side by side, each lane group walking ITS row's nonzeros in the reference's order
// (column and value are fetched per lane; the R lanes of a group read the same address), eight gathers of R columns in
// flight.  Arena rows stay 64 columns apart.  Same arithmetic as k_spmm_epi: a column's bits do not depend on the width.
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_spmm_epi_narrow(int64_t nrows, const int32_t *__restrict__ ptr,
                                                         const int32_t *__restrict__ col, const T *__restrict__ val,
                                                         const T *__restrict__ x, IoPtr<const T> bin_, int64_t ldb, int nrhs,
                                                         const int32_t *__restrict__ p, const double *__restrict__ s,
                                                         int64_t roff, T *__restrict__ out, int logR) {
  const T *__restrict__ bin = bin_.get();
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < nrows; i += nwaves * lm.G) {
    const int32_t k0 = ptr[i], k1 = ptr[i + 1];
    T acc = vzero(T());
    int32_t k = k0;
    for (; k + 8 <= k1; k += 8) {
      int32_t j_[8];
      T a_[8], x_[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) j_[u] = col[k + u], a_[u] = val[k + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) x_[u] = x[((int64_t)j_[u] << 6) + lm.c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = vadd(acc, vmul(x_[u], a_[u]));
    }
    for (; k < k1; ++k) acc = vadd(acc, vmul(x[((int64_t)col[k] << 6) + lm.c], val[k]));
    const int32_t src = p[roff + i];
    T rhs = vzero(T());
    if (lm.c < nrhs) rhs = vscale(s[src], bin[(int64_t)src * ldb + lm.c]);
    out[(i << 6) + lm.c] = vsub(rhs, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// S3 / S5 on the matrix cores (host.hpp SpmmTiles): out[i] = s[p[roff+i]] * b[p[roff+i]] - sum_j A(i,j) x[j] with the
// rows in blocks of 16 and each block's distinct columns in groups of 4.  One wave per block: per group ONE coalesced
// load of the 16 x 4 coefficient tile (A operand), the 4 source rows gathered as the B operands of the four
// 16-column tiles (lane l fetches x[ucol[4g + (l >> 4)]][16 ct + (l & 15)]), four MFMAs.  Two groups in flight.
// Every distinct source row of a block is fetched once, not once per nonzero.  R = 64, real data, fast mode.
// ---------------------------------------------------------------------------------------------
template <int RB, int NCT>
__global__ void __launch_bounds__(256) k_spmm_tile(int64_t nrows, int64_t nblk, const int32_t *__restrict__ blk_gptr,
                                                   const int32_t *__restrict__ ucol, const double *__restrict__ coef,
                                                   const double *__restrict__ x, IoPtr<const double> bin_, int64_t ldb,
                                                   int nrhs, const int32_t *__restrict__ p, const double *__restrict__ s,
                                                   int64_t roff, double *__restrict__ out) {
  // RB row tiles of 16 rows per block (host.hpp SpmmTiles::rb): a block's distinct columns are gathered ONCE for all of
  // them -- the product runs at the fabric's gather rate, and 32-row blocks gather a quarter less than 16-row blocks.
  // NCT: 16-column tiles of the batch in use (round 4: a batch of <= 16 columns gathers 128 of a source row's 512 bytes
  // and multiplies one tile per group; per column the arithmetic is the same whatever NCT: same bits at every width)
  const double *__restrict__ bin = bin_.get();
  const int lane = threadIdx.x & 63;
  const int kq = lane >> 4, jc = lane & 15;
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wave; b < nblk; b += nwaves) {
    const int32_t g0 = rfl(blk_gptr[b]), g1 = rfl(blk_gptr[b + 1]);
    v4f64 acc[RB][NCT];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[rb][ct] = v4f64{0.0, 0.0, 0.0, 0.0};
    // Software pipeline over the groups, all loads unconditional (indices clamped to the block's last group, whose
    // coefficient is then zeroed): source-row indices run 7 groups ahead, coefficient + B fragments 3 groups ahead of the
    // MFMAs -- the compiler counts its waits exactly, nothing but the oldest group is ever waited for.
    if (g0 < g1) {
      const int32_t gl = g1 - 1;
      int32_t src[8];
      double av[4][RB], bv[4][NCT];
#define HIFAMD_TL_SRC(slot, g) src[slot] = ucol[4 * (int64_t)min((g), gl) + kq];
#define HIFAMD_TL_LOAD(slot, sslot, g)                                                                               \
  _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) av[slot][rb] = coef[64 * (RB * (int64_t)min((g), gl) + rb) + lane]; \
  _Pragma("unroll") for (int ct = 0; ct < NCT; ++ct) bv[slot][ct] = x[((int64_t)src[sslot] << 6) + 16 * ct + jc];
#pragma unroll
      for (int q = 0; q < 7; ++q) { HIFAMD_TL_SRC(q, g0 + q) }
#pragma unroll
      for (int q = 0; q < 3; ++q) { HIFAMD_TL_LOAD(q, q, g0 + q) }
      for (int32_t g = g0; g < g1; g += 8) {
#pragma unroll
        for (int dd = 0; dd < 8; ++dd) {
          const int32_t cur = g + dd;
          HIFAMD_TL_LOAD((dd + 3) & 3, (dd + 3) & 7, cur + 3)
          HIFAMD_TL_SRC((dd + 7) & 7, cur + 7)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            const double am = cur < g1 ? av[dd & 3][rb] : 0.0;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[rb][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bv[dd & 3][ct], acc[rb][ct], 0, 0, 0);
          }
        }
      }
#undef HIFAMD_TL_SRC
#undef HIFAMD_TL_LOAD
    }
    // epilogue: C layout col = l & 15, row = (l >> 4) + 4 * reg
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t i = 16 * (RB * b + rb) + kq + 4 * r;
        if (i < nrows) {
          const int32_t srow = p[roff + i];
          const double sc = s[srow];
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const int c = 16 * ct + jc;
            const double rhs = c < nrhs ? sc * bin[(int64_t)srow * ldb + c] : 0.0;
            out[(i << 6) + c] = rhs - acc[rb][ct][r];
          }
        }
      }
  }
}

// The same product with ONE block per workgroup, its column groups dealt round-robin to the four waves (wave w
// takes groups g0 + w, g0 + w + 4, ...).  A block is a sequential chain of groups (each waits for its gathers), and the
// blocks of the reference's coupling matrices are long and uneven (level 2 of the 1M-row default hierarchy: 86 groups
// on average, 188 at most, 1,987 blocks of 16 rows -- fewer waves than the chip holds, and the launch lasts as long as
// the longest chain): four waves per block cut every chain to a quarter and quadruple the gathers in flight.  Partial
// tiles meet in LDS; wave w finishes column tile w of the block (fixed order: deterministic).
template <int RB, int NCT>
__global__ void __launch_bounds__(256) k_spmm_tile4(int64_t nrows, int64_t nblk, const int32_t *__restrict__ blk_gptr,
                                                    const int32_t *__restrict__ ucol, const double *__restrict__ coef,
                                                    const double *__restrict__ x, IoPtr<const double> bin_, int64_t ldb,
                                                    int nrhs, const int32_t *__restrict__ p, const double *__restrict__ s,
                                                    int64_t roff, double *__restrict__ out) {
  __shared__ double red[4][NCT][4][64];  // [wave][tile][reg][lane], 8 KB per column tile (one row tile at a time)
  const double *__restrict__ bin = bin_.get();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int kq = lane >> 4, jc = lane & 15;
  const int64_t b = (int64_t)xcd_block();
  if (b >= nblk) return;
  const int32_t g0 = rfl(blk_gptr[b]), g1 = rfl(blk_gptr[b + 1]);
  v4f64 acc[RB][NCT];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[rb][ct] = v4f64{0.0, 0.0, 0.0, 0.0};
  // this wave's groups: g0 + wave + 4 k, k = 0 .. ng - 1; same software pipeline as k_spmm_tile (loads unconditional,
  // indices clamped to the wave's last group, whose coefficient is then zeroed)
  const int32_t ng = (g1 - g0 > wave) ? (g1 - g0 - wave + 3) / 4 : 0;
  if (ng > 0) {
    const int32_t kl = ng - 1;
    int32_t src[8];
    double av[4][RB], bv[4][NCT];
#define HIFAMD_TL_GID(k) (g0 + wave + 4 * min((k), kl))
#define HIFAMD_TL_SRC(slot, k) src[slot] = ucol[4 * (int64_t)HIFAMD_TL_GID(k) + kq];
#define HIFAMD_TL_LOAD(slot, sslot, k)                                                                                \
  _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) av[slot][rb] = coef[64 * (RB * (int64_t)HIFAMD_TL_GID(k) + rb) + lane]; \
  _Pragma("unroll") for (int ct = 0; ct < NCT; ++ct) bv[slot][ct] = x[((int64_t)src[sslot] << 6) + 16 * ct + jc];
#pragma unroll
    for (int q = 0; q < 7; ++q) { HIFAMD_TL_SRC(q, q) }
#pragma unroll
    for (int q = 0; q < 3; ++q) { HIFAMD_TL_LOAD(q, q, q) }
    for (int32_t k = 0; k < ng; k += 8) {
#pragma unroll
      for (int dd = 0; dd < 8; ++dd) {
        const int32_t cur = k + dd;
        HIFAMD_TL_LOAD((dd + 3) & 3, (dd + 3) & 7, cur + 3)
        HIFAMD_TL_SRC((dd + 7) & 7, cur + 7)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const double am = cur < ng ? av[dd & 3][rb] : 0.0;
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) acc[rb][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bv[dd & 3][ct], acc[rb][ct], 0, 0, 0);
        }
      }
    }
#undef HIFAMD_TL_GID
#undef HIFAMD_TL_SRC
#undef HIFAMD_TL_LOAD
  }
  const int c = 16 * wave + jc;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    if (rb) __syncthreads();  // (the previous row tile's partials have been read)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][ct][r][lane] = acc[rb][ct][r];
    __syncthreads();
    // epilogue of column tile `wave`: C layout col = l & 15, row = (l >> 4) + 4 * reg
    if (wave >= NCT) continue;  // (a narrow batch: fewer column tiles than waves)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = 16 * (RB * b + rb) + kq + 4 * r;
      if (i < nrows) {
        const double tot = ((red[0][wave][r][lane] + red[1][wave][r][lane]) + red[2][wave][r][lane]) + red[3][wave][r][lane];
        const int32_t srow = p[roff + i];
        const double rhs = c < nrhs ? s[srow] * bin[(int64_t)srow * ldb + c] : 0.0;
        out[(i << 6) + c] = rhs - tot;
      }
    }
  }
}

// The tiled product for COMPLEX coupling blocks (round 4; host.hpp build_spmm_tiles(Csr<zdouble>)): one wave per 16-row
// block and 16-column slice of the (complex) batch -- blockIdx.y = slice --; a group's coefficients are two real tiles
// [re | im], a lane gathers the 16 bytes (re, im) of ITS source row and column, and the complex tile product is four real
// matrix instructions (rr, ii, ri, ir; out = (rr - ii) + i (ri + ir)).  Same software pipeline as k_spmm_tile: source-row
// indices 7 groups ahead, coefficient tiles and gathered rows 3 groups ahead, every load unconditional.
__global__ void __launch_bounds__(256) k_spmm_tile_z(int64_t nrows, int64_t nblk, const int32_t *__restrict__ blk_gptr,
                                                     const int32_t *__restrict__ ucol, const double *__restrict__ coef,
                                                     const cplx *__restrict__ x, IoPtr<const cplx> bin_, int64_t ldb, int nrhs,
                                                     const int32_t *__restrict__ p, const double *__restrict__ s, int64_t roff,
                                                     cplx *__restrict__ out) {
  const cplx *__restrict__ bin = bin_.get();
  const int lane = threadIdx.x & 63;
  const int kq = lane >> 4, jc = lane & 15;
  const int c = 16 * (int)blockIdx.y + jc;  // this lane's column of the 64-column (complex) arena
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wave; b < nblk; b += nwaves) {
    const int32_t g0 = rfl(blk_gptr[b]), g1 = rfl(blk_gptr[b + 1]);
    v4f64 a_rr = v4f64{0.0, 0.0, 0.0, 0.0}, a_ii = a_rr, a_ri = a_rr, a_ir = a_rr;
    if (g0 < g1) {
      const int32_t gl = g1 - 1;
      int32_t src[8];
      double ar[4], ai[4];
      cplx bv[4];
#define HIFAMD_TLZ_SRC(slot, g) src[slot] = ucol[4 * (int64_t)min((g), gl) + kq];
#define HIFAMD_TLZ_LOAD(slot, sslot, g)                              \
  ar[slot] = coef[128 * (int64_t)min((g), gl) + lane];               \
  ai[slot] = coef[128 * (int64_t)min((g), gl) + 64 + lane];          \
  bv[slot] = x[((int64_t)src[sslot] << 6) + c];
#pragma unroll
      for (int q = 0; q < 7; ++q) { HIFAMD_TLZ_SRC(q, g0 + q) }
#pragma unroll
      for (int q = 0; q < 3; ++q) { HIFAMD_TLZ_LOAD(q, q, g0 + q) }
      for (int32_t g = g0; g < g1; g += 8) {
#pragma unroll
        for (int dd = 0; dd < 8; ++dd) {
          const int32_t cur = g + dd;
          HIFAMD_TLZ_LOAD((dd + 3) & 3, (dd + 3) & 7, cur + 3)
          HIFAMD_TLZ_SRC((dd + 7) & 7, cur + 7)
          const double mr = cur < g1 ? ar[dd & 3] : 0.0, mi = cur < g1 ? ai[dd & 3] : 0.0;
          a_rr = __builtin_amdgcn_mfma_f64_16x16x4f64(mr, bv[dd & 3].x, a_rr, 0, 0, 0);
          a_ii = __builtin_amdgcn_mfma_f64_16x16x4f64(mi, bv[dd & 3].y, a_ii, 0, 0, 0);
          a_ri = __builtin_amdgcn_mfma_f64_16x16x4f64(mr, bv[dd & 3].y, a_ri, 0, 0, 0);
          a_ir = __builtin_amdgcn_mfma_f64_16x16x4f64(mi, bv[dd & 3].x, a_ir, 0, 0, 0);
        }
      }
#undef HIFAMD_TLZ_SRC
#undef HIFAMD_TLZ_LOAD
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = 16 * b + kq + 4 * r;
      if (i < nrows) {
        const int32_t srow = p[roff + i];
        cplx rhs = cplx{0.0, 0.0};
        if (c < nrhs) rhs = vscale(s[srow], bin[(int64_t)srow * ldb + c]);
        out[(i << 6) + c] = cplx{rhs.x - (a_rr[r] - a_ii[r]), rhs.y - (a_ri[r] + a_ir[r])};
      }
    }
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
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
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
// prec_prod (y = M b, alg/prec_prod.hpp:55-147): the inverse direction of the apply.  No dependent steps
// except the one LDU solve of the Schur coupling term, so these are plain row-gather kernels.
// ---------------------------------------------------------------------------------------------
// g[i] = b[q[i]] / t[q[i]], rows [0, cnt)   (:76, :97; with (p, s) for the transposed product)
template <class T>
__global__ void __launch_bounds__(256) k_gather_div(IoPtr<const T> bin_, int64_t ldb, int nrhs,
                                                    const int32_t *__restrict__ q, const double *__restrict__ t,
                                                    int64_t cnt, T *__restrict__ g, int logR) {
  const T *__restrict__ bin = bin_.get();
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < cnt; i += nwaves * lm.G) {
    const int32_t src = q[i];
    T val = vzero(T());
    if (lm.c < nrhs) val = vdivr(bin[(int64_t)src * ldb + lm.c], t[src]);
    g[(i << logR) + lm.c] = val;
  }
}

// y[i] = r[p_inv[i]] / s[i], rows [0, n)   (:132)
template <class T>
__global__ void __launch_bounds__(256) k_scatter_div(const T *__restrict__ r, const int32_t *__restrict__ pinv,
                                                     const double *__restrict__ s, int64_t n, IoPtr<T> yout_,
                                                     int64_t ldy, int nrhs, int logR) {
  T *__restrict__ yout = yout_.get();
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < n; i += nwaves * lm.G)
    if (lm.c < nrhs) yout[i * ldy + lm.c] = vdivr(r[((int64_t)pinv[i] << logR) + lm.c], s[i]);
}

// rows of a unit triangle times a vector, slot-ordered CSR (rowid): out[i] = (sum_k A(i,k) x[k] + x[i]) [* d[i]]
// (:101-103 with d, :106-108 without)
template <class T, bool SCALE>
__global__ void __launch_bounds__(256) k_prod_rows(int64_t nrows, const int32_t *__restrict__ ptr,
                                                   const int32_t *__restrict__ col, const T *__restrict__ val,
                                                   const int32_t *__restrict__ rowid, const T *__restrict__ x,
                                                   const T *__restrict__ d, T *__restrict__ out, int logR) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t sidx = wave * lm.G + lm.g; sidx < nrows; sidx += nwaves * lm.G) {
    const int64_t i = rowid[sidx];
    T acc = vzero(T());
    for (int32_t k = ptr[sidx]; k < ptr[sidx + 1]; ++k) acc = vadd(acc, vmul(x[((int64_t)col[k] << logR) + lm.c], val[k]));
    acc = vadd(acc, x[(i << logR) + lm.c]);
    if (SCALE) acc = vmul(acc, d[i]);
    out[(i << logR) + lm.c] = acc;
  }
}

// t = sum_k A(i,k) x[k];  MODE 0: out[i] = t, acc[i] += t  (F term, :113-114)
//                         MODE 1: out[i] = t + add[i]      (E term + child product, :125-127)
template <class T, int MODE>
__global__ void __launch_bounds__(256) k_spmm_prod(int64_t nrows, const int32_t *__restrict__ ptr,
                                                   const int32_t *__restrict__ col, const T *__restrict__ val,
                                                   const T *__restrict__ x, T *__restrict__ out, T *accio,
                                                   const T *__restrict__ add, int logR) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < nrows; i += nwaves * lm.G) {
    T t = vzero(T());
    for (int32_t k = ptr[i]; k < ptr[i + 1]; ++k) t = vadd(t, vmul(x[((int64_t)col[k] << logR) + lm.c], val[k]));
    if (MODE == 0) {
      out[(i << logR) + lm.c] = t;
      accio[(i << logR) + lm.c] = vadd(accio[(i << logR) + lm.c], t);
    } else {
      out[(i << logR) + lm.c] = vadd(t, add[(i << logR) + lm.c]);
    }
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
// Constant-mode null-space filter (NspFilter::_const_filter, NspFilter.hpp:161-175): every column loses
// the mean of its rows [r0, r1).  Two launches, no host round trip: per-block partial column sums, then
// every block re-adds the partials (fixed order: deterministic) and subtracts.
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_colsum_partial(int64_t r0, int64_t r1, int nrhs, const T *x, int64_t ldx,
                                                        T *partial /* [gridDim.x][nrhs] */) {
  __shared__ T sm[256];
  const int cpad = nrhs;  // nrhs <= 64 here
  const int rows_per_pass = 256 / cpad;
  const int c = threadIdx.x % cpad, rloc = threadIdx.x / cpad;
  T acc = vzero(T());
  if (rloc < rows_per_pass)
    for (int64_t i = r0 + (int64_t)blockIdx.x * rows_per_pass + rloc; i < r1; i += (int64_t)gridDim.x * rows_per_pass)
      acc = vadd(acc, x[i * ldx + c]);
  sm[threadIdx.x] = (rloc < rows_per_pass) ? acc : vzero(T());
  __syncthreads();
  if (threadIdx.x < cpad) {
    T tot = vzero(T());
    for (int r = 0; r < rows_per_pass; ++r) tot = vadd(tot, sm[r * cpad + threadIdx.x]);
    partial[(int64_t)blockIdx.x * nrhs + threadIdx.x] = tot;
  }
}

template <class T>
__global__ void __launch_bounds__(256) k_sub_colmean(int64_t r0, int64_t r1, int nrhs, T *x, int64_t ldx,
                                                     const T *__restrict__ partial, int nblk) {
  __shared__ T mean[64];
  if (threadIdx.x < nrhs) {
    T tot = vzero(T());
    for (int b = 0; b < nblk; ++b) tot = vadd(tot, partial[(int64_t)b * nrhs + threadIdx.x]);
    mean[threadIdx.x] = vdivr(tot, (double)(r1 - r0));
  }
  __syncthreads();
  const int64_t total = (r1 - r0) * nrhs;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = r0 + e / nrhs;
    const int c = (int)(e % nrhs);
    x[i * ldx + c] = vsub(x[i * ldx + c], mean[c]);
  }
}

// ---------------------------------------------------------------------------------------------
// Device-resident Arnoldi process of the batched GMRES driver (examples/advanced/gmres.hpp:57-103 for up to
// 64 columns in lock step): the Hessenberg column, the rotations, the residuals and the per-column state
// (iteration count, flag, active mask) live in HBM; the host sees three integers per inner step.
// The inner product is the Hermitian one, h = sum conj(q_i) v_i (for real data this IS hif::inner,
// utils/math.hpp:83; for complex data the example's sum conj(v_i) q_i would not orthogonalize).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double vconj(double a) { return a; }
__device__ __forceinline__ cplx vconj(cplx a) { return cplx{a.x, -a.y}; }
__device__ __forceinline__ double vreal(double a) { return a; }
__device__ __forceinline__ double vreal(cplx a) { return a.x; }
__device__ __forceinline__ double vfromreal(double r, double) { return r; }
__device__ __forceinline__ cplx vfromreal(double r, cplx) { return cplx{r, 0.0}; }
__device__ __forceinline__ double vabs1(double a) { return fabs(a); }
__device__ __forceinline__ double vabs1(cplx a) { return sqrt(a.x * a.x + a.y * a.y); }
__device__ __forceinline__ bool viszero(double a) { return a == 0.0; }
__device__ __forceinline__ bool viszero(cplx a) { return a.x == 0.0 && a.y == 0.0; }

template <class T>
struct GmState {
  T *w2;          // [nc][restart]           Hessenberg column of the running step (gmres.hpp:64,73-78)
  T *Jc;          // [nc][restart]           J(:,0)
  double *Js;     // [nc][restart]           J(:,1) (real)
  T *y;           // [nc][restart + 1]
  T *R;           // [nc][restart][restart]  column j of column c at R + (c * restart + j) * restart
  double *resid, *beta0;
  int *iter, *flag, *active, *jfin, *done, *sweeps;
  T *alpha;       // [nc] coefficient of the next column operation
  int *ctl;       // [0] columns still active after the step  [1] max jfin  [2] columns not done
  int restart, maxit;
  double rtol;
};

// One modified-Gram-Schmidt step, fused: first v -= h_prev q_prev (the axpy of the PREVIOUS step, :65), then the
// block's share of sum conj(q_i) v_i (:64) -- or of sum |v_i|^2 (:67) when q == nullptr.  v, q: [n][nc] contiguous.
template <class T>
__global__ void __launch_bounds__(256) k_gm_step(int64_t n, int nc, T *__restrict__ v, const T *__restrict__ qp,
                                                 const T *__restrict__ hp, const T *__restrict__ q,
                                                 T *__restrict__ partial /* [gridDim.x][nc] */) {
  __shared__ T sm[256];
  const int rpp = 256 / nc;
  const int c = threadIdx.x % nc, rloc = threadIdx.x / nc;
  T acc = vzero(T());
  if (rloc < rpp) {
    const T h = qp ? hp[c] : vzero(T());
    const int64_t step = (int64_t)gridDim.x * rpp;
    for (int64_t i = (int64_t)blockIdx.x * rpp + rloc; i < n; i += step) {
      T x = v[i * nc + c];
      if (qp) {
        x = vsub(x, vmul(h, qp[i * nc + c]));
        v[i * nc + c] = x;
      }
      acc = vadd(acc, q ? vmul(vconj(q[i * nc + c]), x) : vfromreal(vabs2(x), T()));
    }
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < nc) {
    T tot = vzero(T());
    for (int r = 0; r < rpp; ++r) tot = vadd(tot, sm[r * nc + threadIdx.x]);
    partial[(int64_t)blockIdx.x * nc + threadIdx.x] = tot;
  }
}

// Modified Gram-Schmidt against a BLOCK of MB basis vectors in one pass over v (the orthogonalization is bound by
// the bytes of v and Q it moves: 4 vector passes per basis vector one at a time, (2 MB + 2) / MB in blocks).  In exact
// arithmetic the coefficients of modified Gram-Schmidt satisfy
//     h_{k+a} = q_{k+a}^H v^(k+a) = q_{k+a}^H v^(k) - sum_{l<a} h_{k+l} (q_{k+a}^H q_{k+l}),
// so one pass delivers b_a = q_{k+a}^H v^(k) and the block's Gram entries g_{a,l} = q_{k+a}^H q_{k+l} (l < a) -- the
// basis vectors are being read anyway -- and a finishing kernel runs the recursion (k_gm_hblock).  The same pass first
// applies the PREVIOUS block's update v -= sum_l h_l q_l (gmres.hpp:65).  Value list of a block: b_0..b_{mn-1}, then
// g_{1,0}, g_{2,0}, g_{2,1}, ...; mn == 0: |v|^2 only (the pass behind the last block, :67).
constexpr int kGmBlock = 4;
constexpr int kGmVals = kGmBlock + kGmBlock * (kGmBlock - 1) / 2;
template <class T>
__global__ void __launch_bounds__(256) k_gm_block(int64_t n, int nc, T *__restrict__ v, const T *__restrict__ Qp, int mp,
                                                  const T *__restrict__ hp /* [kGmBlock][64] */, const T *__restrict__ Qn,
                                                  int mn, T *__restrict__ partial /* [gridDim.x][nv][nc] */) {
  __shared__ T sm[256];
  const int rpp = 256 / nc;
  const int c = threadIdx.x % nc, rloc = threadIdx.x / nc;
  const int64_t vec = n * nc;
  T acc[kGmVals];
#pragma unroll
  for (int a = 0; a < kGmVals; ++a) acc[a] = vzero(T());
  if (rloc < rpp) {
    T h[kGmBlock];
#pragma unroll
    for (int l = 0; l < kGmBlock; ++l) h[l] = (l < mp) ? hp[l * 64 + c] : vzero(T());
    const int64_t step = (int64_t)gridDim.x * rpp;
    for (int64_t i = (int64_t)blockIdx.x * rpp + rloc; i < n; i += step) {
      const int64_t o = i * nc + c;
      T x = v[o];
      if (mp) {
#pragma unroll
        for (int l = 0; l < kGmBlock; ++l)
          if (l < mp) x = vsub(x, vmul(h[l], Qp[(int64_t)l * vec + o]));
        v[o] = x;
      }
      if (mn == 0) {
        acc[0] = vadd(acc[0], vfromreal(vabs2(x), T()));
      } else {
        T q[kGmBlock];
#pragma unroll
        for (int a = 0; a < kGmBlock; ++a) q[a] = (a < mn) ? Qn[(int64_t)a * vec + o] : vzero(T());
#pragma unroll
        for (int a = 0; a < kGmBlock; ++a) {
          const T qc = vconj(q[a]);
          acc[a] = vadd(acc[a], vmul(qc, x));
#pragma unroll
          for (int l = 0; l < a; ++l) acc[kGmBlock + a * (a - 1) / 2 + l] = vadd(acc[kGmBlock + a * (a - 1) / 2 + l], vmul(qc, q[l]));
        }
      }
    }
  }
  // value a of the block lives at slot (a < kGmBlock ? a : mn + (a - kGmBlock)) of the launch's value list
  const int nv = mn ? mn + mn * (mn - 1) / 2 : 1;
#pragma unroll
  for (int a = 0; a < kGmVals; ++a) {
    int slot;
    if (a < kGmBlock) {
      slot = (a < mn || (mn == 0 && a == 0)) ? a : -1;
    } else {
      // (a - kGmBlock) enumerates pairs (row, l) with row = 1.., l < row in the order above
      int row = 1, rem = a - kGmBlock;
      while (rem >= row) rem -= row, ++row;
      slot = row < mn ? mn + row * (row - 1) / 2 + rem : -1;
    }
    if (slot < 0) continue;  // (block-uniform)
    __syncthreads();
    sm[threadIdx.x] = acc[a];
    __syncthreads();
    if (threadIdx.x < nc) {
      T tot = vzero(T());
      for (int r = 0; r < rpp; ++r) tot = vadd(tot, sm[r * nc + threadIdx.x]);
      partial[((int64_t)blockIdx.x * nv + slot) * nc + threadIdx.x] = tot;
    }
  }
}

// red[val][c] = sum over the blocks' partials, fixed order; one workgroup per value of the list
template <class T>
__global__ void __launch_bounds__(256) k_gm_reduce(const T *__restrict__ partial, int nblk, int nv, int nc, T *__restrict__ red) {
  __shared__ T sm[256];
  const int val = blockIdx.x;
  const int rpp = 256 / nc;
  const int c = threadIdx.x % nc, g = threadIdx.x / nc;
  T acc = vzero(T());
  if (g < rpp)
    for (int b = g; b < nblk; b += rpp) acc = vadd(acc, partial[((int64_t)b * nv + val) * nc + c]);
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < nc) {
    T tot = vzero(T());
    for (int r = 0; r < rpp; ++r) tot = vadd(tot, sm[r * nc + c]);
    red[val * 64 + c] = tot;
  }
}

// the recursion of a block (see k_gm_block): h_{k+a} into the Hessenberg column and into hb[a][c] for the next pass
template <class T>
__global__ void __launch_bounds__(64) k_gm_hblock(const T *__restrict__ red, int nc, int k, int mn, GmState<T> S,
                                                  T *__restrict__ hb /* [kGmBlock][64] */) {
  const int c = threadIdx.x;
  if (c >= nc) return;
  T h[kGmBlock];
#pragma unroll
  for (int a = 0; a < kGmBlock; ++a) {
    if (a >= mn) break;
    T t = red[a * 64 + c];
    for (int l = 0; l < a; ++l) t = vsub(t, vmul(h[l], red[(mn + a * (a - 1) / 2 + l) * 64 + c]));
    h[a] = t;
    S.w2[(size_t)c * S.restart + k + a] = t;
    hb[a * 64 + c] = t;
  }
}

// Finishes a reduction (fixed order: deterministic) and does the per-column scalar work of the driver.
//   mode 0: h_k of step k -> w2[k], alpha                                       (:64)
//   mode 1: |v|^2 of step j: rotations, residual, stopping rules                 (:67-103); alpha = |v| for :69-70
//   mode 2: start of an outer cycle: beta, y[0], active mask                    (:53-55)
//   mode 3: start of the solve: beta0, quick return                              (:30-36)
template <class T>
__global__ void __launch_bounds__(256) k_gm_finish(const T *__restrict__ partial, int nblk, int nc, int mode, int k,
                                                   int nirs, GmState<T> S) {
  __shared__ T sm[256];
  __shared__ int cnt[2];
  const int rpp = 256 / nc;
  const int c = threadIdx.x % nc, g = threadIdx.x / nc;
  T acc = vzero(T());
  if (g < rpp)
    for (int b = g; b < nblk; b += rpp) acc = vadd(acc, partial[(int64_t)b * nc + c]);
  sm[threadIdx.x] = acc;
  if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < nc) {
    T tot = vzero(T());
    for (int r = 0; r < rpp; ++r) tot = vadd(tot, sm[r * nc + c]);
    const int rs = S.restart;
    if (mode == 0) {
      S.w2[(size_t)c * rs + k] = tot;
      S.alpha[c] = tot;
    } else if (mode == 3) {
      const double b0 = sqrt(vreal(tot));
      S.beta0[c] = b0;
      S.done[c] = (b0 == 0.0);
      S.resid[c] = 1.0;
      S.iter[c] = 0;
      S.flag[c] = 0;
      S.sweeps[c] = 0;
      S.active[c] = 0;
      S.jfin[c] = -1;
      if (b0 != 0.0) atomicAdd(&cnt[1], 1);
    } else if (mode == 2) {
      const double beta = sqrt(vreal(tot));
      int dn = S.done[c];
      if (!dn && beta == 0.0) dn = 1;  // exact solution reached
      S.done[c] = dn;
      S.y[(size_t)c * (rs + 1)] = vfromreal(beta, T());
      S.alpha[c] = vfromreal(dn ? 0.0 : beta, T());
      S.active[c] = !dn;
      S.jfin[c] = -1;
      if (!dn) atomicAdd(&cnt[0], 1);
    } else {
      const int j = k;
      const double v_norm2 = vreal(tot), v_norm = sqrt(v_norm2);
      const int act = S.active[c];
      S.alpha[c] = vfromreal(act ? v_norm : 0.0, T());
      if (act) {
        S.sweeps[c] += nirs;
        T *wc = S.w2 + (size_t)c * rs, *yc = S.y + (size_t)c * (rs + 1), *Jc = S.Jc + (size_t)c * rs;
        double *Js = S.Js + (size_t)c * rs;
        T *Rc = S.R + (size_t)c * rs * rs;
        for (int cj = 0; cj + 1 <= j; ++cj) {  // :73-78
          const T t0 = wc[cj], t1 = wc[cj + 1];
          wc[cj] = vadd(vmul(vconj(Jc[cj]), t0), vscale(Js[cj], t1));
          wc[cj + 1] = vadd(vscale(-Js[cj], t0), vmul(Jc[cj], t1));
        }
        const double rho = sqrt(vabs2(wc[j]) + v_norm2);  // :79
        Jc[j] = vdivr(wc[j], rho);
        Js[j] = v_norm / rho;
        yc[j + 1] = vscale(-Js[j], yc[j]);
        yc[j] = vmul(vconj(Jc[j]), yc[j]);
        wc[j] = vfromreal(rho, T());
        for (int i = 0; i <= j; ++i) Rc[(size_t)j * rs + i] = wc[i];  // :85
        const double resid_prev = S.resid[c];
        const double resid = vabs1(yc[j + 1]) / S.beta0[c];  // :89
        S.resid[c] = resid;
        bool brk = false;
        if (resid >= resid_prev * (1.0 - 1e-8)) {  // :90-93
          S.flag[c] = 1;
          brk = true;
        } else if (S.iter[c] >= S.maxit) {  // :94-97
          S.flag[c] = 2;
          brk = true;
        } else {
          S.iter[c] += 1;
          if (resid <= S.rtol || j + 1 >= rs) brk = true;  // :102
        }
        if (brk) {
          S.jfin[c] = j;
          S.active[c] = 0;
        } else {
          atomicAdd(&cnt[0], 1);
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (mode == 1 || mode == 2) S.ctl[0] = cnt[0];
    if (mode == 3) S.ctl[2] = cnt[1];
  }
}

// End of an outer cycle: back substitution R y = g per column (:106-110), alpha = 1 for the columns that took part,
// done flags (:120); ctl[1] = max jfin, ctl[2] = columns not done.
template <class T>
__global__ void __launch_bounds__(64) k_gm_backsolve(int nc, GmState<T> S) {
  __shared__ int red[2];
  if (threadIdx.x < 2) red[threadIdx.x] = threadIdx.x ? 0 : -1;
  __syncthreads();
  const int c = threadIdx.x;
  if (c < nc) {
    const int rs = S.restart, jf = S.jfin[c];
    if (jf >= 0) {
      T *yc = S.y + (size_t)c * (rs + 1);
      const T *Rc = S.R + (size_t)c * rs * rs;
      for (int k = jf; k > -1; --k) {
        yc[k] = vdiv(yc[k], Rc[(size_t)k * rs + k]);
        const T t0 = yc[k];
        for (int i = k - 1; i > -1; --i) yc[i] = vsub(yc[i], vmul(t0, Rc[(size_t)k * rs + i]));
      }
      atomicMax(&red[0], jf);
      if (S.resid[c] <= S.rtol || S.flag[c] != 0) S.done[c] = 1;
    }
    S.alpha[c] = vfromreal(jf >= 0 ? 1.0 : 0.0, T());
    if (!S.done[c]) atomicAdd(&red[1], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    S.ctl[1] = red[0];
    S.ctl[2] = red[1];
  }
}

// out = (accumulate ? out : 0) + sum_{k <= jmax} coef_k Q_k, coef_k[c] = y[c][k] for the columns with jfin >= k, in
// the order of :112-116 / :214-218; Q_k = Q + k * n * nc, [n][nc] contiguous
template <class T>
__global__ void __launch_bounds__(256) k_gm_combine(int64_t n, int nc, T *__restrict__ out, int64_t ldo, int accumulate,
                                                    const T *__restrict__ Q, int jmax, GmState<T> S) {
  const int64_t total = n * nc;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / nc;
    const int c = (int)(e - i * nc);
    const int jf = S.jfin[c];
    const T *yc = S.y + (size_t)c * (S.restart + 1);
    T r = accumulate ? out[i * ldo + c] : vzero(T());
    for (int k = 0; k <= jmax; ++k) {
      const T a = (jf >= k) ? yc[k] : vzero(T());
      r = vadd(r, vmul(a, Q[(size_t)k * (size_t)total + (size_t)e]));
    }
    out[i * ldo + c] = r;
  }
}

// op 0: y[:,c] += alpha[c] x[:,c] | 1: y[:,c] = x[:,c] / real(alpha[c]) (0 where alpha[c] == 0: a column that has
// left the iteration keeps a zero basis vector); alpha on the device
template <class T>
__global__ void __launch_bounds__(256) k_gm_colop(int op, int64_t n, int nc, T *__restrict__ y, int64_t ldy,
                                                  const T *__restrict__ x, int64_t ldx, const T *__restrict__ alpha) {
  const int64_t total = n * nc;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / nc;
    const int c = (int)(e - i * nc);
    const T a = alpha[c];
    if (op == 0)
      y[i * ldy + c] = vadd(y[i * ldy + c], vmul(a, x[i * ldx + c]));
    else
      y[i * ldy + c] = viszero(a) ? vzero(T()) : vdivr(x[i * ldx + c], vreal(a));
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
// tri: 0 full, 1 upper (k >= row tile start), 2 lower (k < row tile end).  dscale/Out2 (optional):
// Out2[orow] = result / dscale[orow]  (the fused y /= d of the L solve, prec_solve.hpp:219).
// A is stored STRIP-MAJOR (host.hpp to_strip_layout): strip s = rows [16 s, 16 s + 16), element
// (row, k) at ((s * lda + k) * 16 + row % 16), lda = number of columns -- a strip streams through
// HBM sequentially, 128 B per k.
// Grid: (16-row strips) x (16-column tiles).  One 256-thread workgroup per 16x16 output tile: its 4
// waves split the K range (interleaved blocks of 32 k), each keeps two sets of 8 k-steps of operands
// (one in flight, one being multiplied; the loop is latency-bound at these sizes), and the partial
// tiles are summed through LDS in a fixed order.
template <int NW>
__global__ void __launch_bounds__(NW * 64) k_dense_gemm_d(int mrows_total, int mrows_valid, int kend, int tri,
                                                      const double *__restrict__ A, int lda,
                                                      const double *__restrict__ X, int logR,
                                                      const int32_t *__restrict__ rowmap,
                                                      double *__restrict__ Out,
                                                      const double *__restrict__ dscale,
                                                      double *__restrict__ Out2) {
  __shared__ double red[NW - 1][4][64];  // partial tiles of waves 1..NW-1: [wave-1][reg][lane]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i0 = blockIdx.x * 16;
  const int R = 1 << logR;
  const int colx = blockIdx.y * 16 + (lane & 15);
  const bool col_ok = colx < R;
  const int arow = i0 + (lane & 15);
  const int kq = lane >> 4;
  const bool arow_ok = arow < mrows_valid;
  v4f64 acc = v4f64{0.0, 0.0, 0.0, 0.0};
  const int kbeg = (tri == 1) ? i0 : 0;  // i0 is a multiple of 16, hence of 4
  if (tri == 2) kend = min(kend, i0 + 16);
  constexpr int KU = 8;  // k-steps (of 4) per operand set
  const int kstride = NW * (4 * KU);
  double a0[KU], b0[KU], a1[KU], b1[KU];
#define HIFAMD_LOAD_SET(aa, bb, kb_)                                                          \
  _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                            \
    const int kk = (kb_) + 4 * u + kq;                                                        \
    const bool kok = kk < kend;                                                               \
    aa[u] = (arow_ok && kok) ? A[((int64_t)blockIdx.x * lda + kk) * 16 + (lane & 15)] : 0.0; \
    bb[u] = (col_ok && kok) ? X[((int64_t)kk << logR) + colx] : 0.0;                          \
  }
#define HIFAMD_MFMA_SET(aa, bb) \
  _Pragma("unroll") for (int u = 0; u < KU; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u], bb[u], acc, 0, 0, 0);
  int kb = kbeg + wave * (4 * KU);
  if (kb < kend) { HIFAMD_LOAD_SET(a0, b0, kb) }
  while (kb < kend) {
    const int kb1 = kb + kstride;
    if (kb1 < kend) { HIFAMD_LOAD_SET(a1, b1, kb1) }
    HIFAMD_MFMA_SET(a0, b0)
    if (kb1 >= kend) break;
    const int kb2 = kb1 + kstride;
    if (kb2 < kend) { HIFAMD_LOAD_SET(a0, b0, kb2) }
    HIFAMD_MFMA_SET(a1, b1)
    kb = kb2;
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double val = acc[r];
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) val += red[w][r][lane];
    const int row = i0 + kq + 4 * r;
    if (col_ok && row < mrows_total) {
      const int orow = rowmap ? rowmap[row] : row;
      Out[((int64_t)orow << logR) + colx] = val;
      if (Out2) Out2[((int64_t)orow << logR) + colx] = val / dscale[orow];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Full dense operator on a 64-column block (the combined top operator of a level, host.hpp build_top_operator):
// Out[rowmap[r]] = sum_k A(r,k) X[k],  A strip-major (lda = kend = columns rounded up to 32, zero padded), X = [kend][64]
// (rows beyond the operator's columns must be readable and finite).
// One workgroup per 16-row strip: wave = (column tile 0..3) x (K split 0..KS-1), so the strip's A panel is fetched
// from HBM ONCE (the four column-tile waves of a K split read the same fragments, they meet in the CU's L1) -- with a
// workgroup per (strip, tile), k_dense_gemm_d, every tile refetched the panel: 3.3x the operator's bytes in FETCH_SIZE.
// Operand sets of 8 k-steps double buffered, partial tiles of the K splits summed through LDS in a fixed order.
// ---------------------------------------------------------------------------------------------
template <int KS>
__global__ void __launch_bounds__(256 * KS) k_strip_gemm_d(int nrows, int kend, const double *__restrict__ A, int lda,
                                                           const double *__restrict__ X,
                                                           const int32_t *__restrict__ rowmap,
                                                           double *__restrict__ Out) {
  __shared__ double red[KS > 1 ? KS - 1 : 1][4][4][64];  // partial tiles of K splits 1..KS-1: [split-1][tile][reg][lane]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ct = wave & 3, ks = wave >> 2;
  const int i0 = blockIdx.x * 16;
  const int colx = ct * 16 + (lane & 15);
  const int kq = lane >> 4;
  v4f64 acc0 = v4f64{0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
  constexpr int KU = 8;
  const int kstride = KS * (4 * KU);
  double a0[KU], b0[KU], a1[KU], b1[KU];
  const double *Ap = A + ((int64_t)blockIdx.x * lda) * 16 + (lane & 15);
  // (unconditional loads -- kend is a multiple of 32, A zero-padded in k, X readable (finite) there: the compiler then
  //  counts its waits exactly and the next set stays in flight behind the MFMAs of the current one)
#define HIFAMD_SG_LOAD(aa, bb, kb_)                                  \
  _Pragma("unroll") for (int u = 0; u < KU; ++u) {                   \
    const int kk = (kb_) + 4 * u + kq;                               \
    aa[u] = Ap[(int64_t)kk * 16];                                    \
    bb[u] = X[((int64_t)kk << 6) + colx];                            \
  }
#define HIFAMD_SG_MFMA(aa, bb)                                                            \
  _Pragma("unroll") for (int u = 0; u < KU; u += 2) {                                     \
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u], bb[u], acc0, 0, 0, 0);             \
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u + 1], bb[u + 1], acc1, 0, 0, 0);     \
  }
  int kb = ks * (4 * KU);
  if (kb < kend) { HIFAMD_SG_LOAD(a0, b0, kb) }
  while (kb < kend) {
    const int kb1 = kb + kstride;
    if (kb1 < kend) { HIFAMD_SG_LOAD(a1, b1, kb1) }
    HIFAMD_SG_MFMA(a0, b0)
    if (kb1 >= kend) break;
    const int kb2 = kb1 + kstride;
    if (kb2 < kend) { HIFAMD_SG_LOAD(a0, b0, kb2) }
    HIFAMD_SG_MFMA(a1, b1)
    kb = kb2;
  }
#undef HIFAMD_SG_LOAD
#undef HIFAMD_SG_MFMA
  const v4f64 acc = acc0 + acc1;
  if (ks > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[ks - 1][ct][r][lane] = acc[r];
  }
  __syncthreads();
  if (ks != 0) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double val = acc[r];
#pragma unroll
    for (int q = 0; q < KS - 1; ++q) val += red[q][ct][r][lane];
    const int row = i0 + kq + 4 * r;
    if (row < nrows) Out[((int64_t)(rowmap ? rowmap[row] : row) << 6) + colx] = val;
  }
}

// The same product with ONE A-fragment load per four MFMAs: a wave owns a K split (16 of them) and ALL four column
// tiles of the strip, so that per 4-k step it issues 1 load of A (HBM stream, every byte used once) and 4 loads of X
// (2 MB, L2-resident) for 4 MFMAs -- 1.25 wave-loads per MFMA instead of 2 (k_strip_gemm_d: the four column-tile waves
// of a K split load the same A fragment).  The product is bound by how many 512-byte wave-loads a compute unit gets
// issued (~180 per us), not by the matrix cores (tests/microbench/mfma_bench.hip).  Partial tiles are summed in two
// LDS rounds in a fixed order (deterministic).  kend: multiple of 8 * KU... of 32; A zero-padded in k; X finite there.
template <int KU>
__global__ void __launch_bounds__(1024) k_strip_gemm4_d(int nrows, int kend, const double *__restrict__ A, int lda,
                                                        const double *__restrict__ X, const int32_t *__restrict__ rowmap,
                                                        double *__restrict__ Out) {
  __shared__ double red[8][4][4][64];  // [wave][tile][reg][lane], 64 KB
  const int lane = threadIdx.x & 63;
  const int ks = threadIdx.x >> 6;  // 0..15
  const int cl = lane & 15, kq = lane >> 4;
  const int i0 = blockIdx.x * 16;
  v4f64 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
  const int kstride = 16 * (4 * KU);
  double a0[KU], b0[KU][4], a1[KU], b1[KU][4];
  const double *Ap = A + ((int64_t)blockIdx.x * lda) * 16 + cl;
  const double *Xp = X + cl;
#define HIFAMD_SG_LOAD(aa, bb, kb_)                                                   \
  _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                    \
    const int kk = (kb_) + 4 * u + kq;                                                \
    aa[u] = Ap[(int64_t)kk * 16];                                                     \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) bb[u][t] = Xp[((int64_t)kk << 6) + 16 * t]; \
  }
#define HIFAMD_SG_MFMA(aa, bb)                                                                    \
  _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                                \
    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                 \
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u], bb[u][t], acc[t], 0, 0, 0);          \
  }
  int kb = ks * (4 * KU);
  if (kb < kend) { HIFAMD_SG_LOAD(a0, b0, kb) }
  while (kb < kend) {
    const int kb1 = kb + kstride;
    if (kb1 < kend) { HIFAMD_SG_LOAD(a1, b1, kb1) }
    HIFAMD_SG_MFMA(a0, b0)
    if (kb1 >= kend) break;
    const int kb2 = kb1 + kstride;
    if (kb2 < kend) { HIFAMD_SG_LOAD(a0, b0, kb2) }
    HIFAMD_SG_MFMA(a1, b1)
    kb = kb2;
  }
#undef HIFAMD_SG_LOAD
#undef HIFAMD_SG_MFMA
  // round 1: waves 8..15 hand their tiles to waves 0..7; round 2: waves 1..7 to wave 0 (fixed order)
  if (ks >= 8) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[ks - 8][t][r][lane] = acc[t][r];
  }
  __syncthreads();
  if (ks < 8) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] += red[ks][t][r][lane];
  }
  __syncthreads();
  if (ks >= 1 && ks < 8) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[ks][t][r][lane] = acc[t][r];
  }
  __syncthreads();
  if (ks != 0) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = i0 + kq + 4 * r;
    if (row >= nrows) continue;
    const int64_t orow = (int64_t)(rowmap ? rowmap[row] : row) << 6;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      double val = acc[t][r];
#pragma unroll
      for (int q = 1; q < 8; ++q) val += red[q][t][r][lane];
      Out[orow + 16 * t + cl] = val;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Out[rowmap[r]] = sum_k G(r, k) X[k]: the combined top operator of a level and the tail operator of the hierarchy
// (round 3).  k_strip_gemm4_d gives a 16-row strip to a workgroup whose sixteen waves split K: every workgroup then pulls
// the WHOLE right-hand-side panel (nt x 512 B) through its compute unit's L1 beside its 16 x nt strip of G -- four times
// the bytes of the operator itself, served by the L2s: 48 us for 3,351 rows, 30 TFLOP/s (profiles/r02).  Here a
// workgroup owns 64 rows (four strips) x ONE K range of the grid's nks: its waves are (strip s, K quarter j); the
// panel's rows of the K range stream through LDS in chunks of 64 (double buffered, rows padded to 80 doubles: the two
// 16-lane halves of a ds_read_b64 then hit disjoint banks), so a panel row is fetched once per 64 output rows, and G
// streams from HBM once, each wave's fragments two chunks ahead in registers.  Partial tiles of the four K quarters are
// summed through LDS in a fixed order; nks > 1: the workgroup's 64 x 64 partial goes to part[ksplit][row][64] and
// k_top_reduce adds the splits in order (deterministic).  nct = 16-column tiles in use (a batch of <= 16 columns: one).
// Panel rows >= kvalid are read as zeros (the buffer behind them may hold anything, also non-finite values).
// G: strip-major MFMA operand (host.hpp mfma_operand), lda = K rounded up to 32.
// ---------------------------------------------------------------------------------------------
template <int NCT>
__global__ void __launch_bounds__(1024) k_top_gemm(int nrows, int kvalid, int kper /* K range per split, multiple of 64 */,
                                                   const double *__restrict__ A, int lda, const double *__restrict__ X,
                                                   const int32_t *__restrict__ rowmap, double *__restrict__ Out,
                                                   double *__restrict__ part, int nrows_pad, unsigned *tile_cnt) {
  // tile_cnt (round 4; NULL: the splits are added by k_top_reduce, a launch of its own): one counter per 64-row tile, zero
  // between launches.  The workgroup that finds its tile's other K splits already counted adds the gridDim.y partial tiles
  // in split order -- the order k_top_reduce uses: the same bits, replays stay bit-stable whichever split arrives last --
  // and writes the rows out; release / acquire fences at agent scope carry the partials across the XCDs' L2s.
  constexpr int nct = NCT;
  constexpr int XS = 80;               // LDS row stride of the panel (doubles)
  extern __shared__ double tg_lds[];   // two chunks of 64 panel rows: 2 x 64 x 80 doubles = 80 KB; reused for the reduction
  double (*xb)[64 * XS] = reinterpret_cast<double (*)[64 * XS]>(tg_lds);
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;     // 0..15
  const int sw = wv & 3, jq = wv >> 2; // strip within the tile, K quarter within a chunk
  const int cl = lane & 15, kq = lane >> 4;
  const int tile = blockIdx.x, ksp = blockIdx.y;
  const int strip = tile * 4 + sw;
  const int nstrips = (nrows + 15) >> 4;
  const int k0 = ksp * kper, k1 = min(k0 + kper, lda);
  const int nchunks = (k1 - k0 + 63) >> 6;
  v4f64 acc[NCT];
#pragma unroll
  for (int t = 0; t < NCT; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
  const bool live = strip < nstrips;
  // this wave's A fragments of chunk c: k = k0 + 64 c + 16 jq + 4 u + kq, u = 0..3 (512 contiguous bytes each).  Loads
  // are unconditional: a chunk index past the end re-reads the last chunk (never multiplied), a dead wave reads the
  // last strip, and k >= lda (the last chunk of the last split, at most 32 columns: the next strip's first ones, or the
  // padding behind the operand) meets panel rows >= kvalid, which are zero in LDS
  const double *Ap = A + ((int64_t)min(strip, nstrips - 1) * lda) * 16 + cl + (int64_t)(k0 + 16 * jq + kq) * 16;
  double a0[4], a1[4], a2[4];
#define HIFAMD_TG_A(aa, c_)                                                                   \
  {                                                                                           \
    const double *ap_ = Ap + (int64_t)min((c_), nchunks - 1) * (64 * 16);                     \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) aa[u] = ap_[u * 64];                        \
  }
  // panel chunk c into buffer b: 64 rows x 64 columns, 16 bytes per thread and pass (rows >= kvalid: zeros)
  typedef double v2f64 __attribute__((ext_vector_type(2)));
  const int xr = threadIdx.x >> 5, xc = (threadIdx.x & 31) * 2;  // row 0..31 (+32), columns xc, xc+1
  v2f64 xs0, xs1;
#define HIFAMD_TG_XLOAD(c_)                                                                   \
  {                                                                                           \
    const int r0_ = k0 + 64 * (c_) + xr, r1_ = r0_ + 32;                                      \
    xs0 = (r0_ < kvalid && xc < 16 * nct) ? *reinterpret_cast<const v2f64 *>(X + ((int64_t)r0_ << 6) + xc) : v2f64{0.0, 0.0}; \
    xs1 = (r1_ < kvalid && xc < 16 * nct) ? *reinterpret_cast<const v2f64 *>(X + ((int64_t)r1_ << 6) + xc) : v2f64{0.0, 0.0}; \
  }
#define HIFAMD_TG_XSTORE(b_)                                                   \
  {                                                                            \
    *reinterpret_cast<v2f64 *>(&xb[b_][xr * XS + xc]) = xs0;                   \
    *reinterpret_cast<v2f64 *>(&xb[b_][(xr + 32) * XS + xc]) = xs1;            \
  }
#define HIFAMD_TG_MFMA(aa, b_)                                                                     \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                  \
    const double *bp_ = &xb[b_][(16 * jq + 4 * u + kq) * XS + cl];                                 \
    _Pragma("unroll") for (int t = 0; t < NCT; ++t)                                                \
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u], bp_[16 * t], acc[t], 0, 0, 0);        \
  }
  HIFAMD_TG_A(a0, 0)
  HIFAMD_TG_A(a1, 1)
  HIFAMD_TG_XLOAD(0)
  HIFAMD_TG_XSTORE(0)
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {  // (A fragments two chunks ahead in registers, the panel one chunk ahead in LDS)
    HIFAMD_TG_A(a2, c + 2)
    if (c + 1 < nchunks) HIFAMD_TG_XLOAD(c + 1)
    HIFAMD_TG_MFMA(a0, (c & 1))
    if (c + 1 < nchunks) HIFAMD_TG_XSTORE(((c + 1) & 1))
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) a0[u] = a1[u], a1[u] = a2[u];
  }
#undef HIFAMD_TG_A
#undef HIFAMD_TG_XLOAD
#undef HIFAMD_TG_XSTORE
#undef HIFAMD_TG_MFMA
  // (the panel buffers become the reduction scratch: every wave is past the loop's last barrier)
  // K quarters 1..3 hand their tiles to quarter 0 of the same strip, summed in the order 0, 1, 2, 3
  double *red = &xb[0][0];  // [jq - 1][sw][t][r][lane]: 3 x 4 x 4 x 4 x 64 doubles = 96 KB > 80 KB: two rounds
  // round 1: quarters 2, 3 -> scratch; quarters 0, 1 add (0 += 2 is NOT the order wanted) -- keep it simple and exact:
  // quarter 1 first, then 2, then 3, one quarter per round (3 rounds x 32 KB)
  for (int q = 1; q < 4; ++q) {
    if (jq == q) {
#pragma unroll
      for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(((sw * 4 + t) * 4 + r) << 6) + lane] = acc[t][r];
    }
    __syncthreads();
    if (jq == 0) {
#pragma unroll
      for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += red[(((sw * 4 + t) * 4 + r) << 6) + lane];
    }
    __syncthreads();
  }
  if (jq == 0 && live) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * strip + kq + 4 * r;
      if (row >= nrows) continue;
      double *dst = (gridDim.y > 1) ? part + (((int64_t)ksp * nrows_pad + row) << 6) : Out + ((int64_t)(rowmap ? rowmap[row] : row) << 6);
#pragma unroll
      for (int t = 0; t < NCT; ++t) dst[16 * t + cl] = acc[t][r];
    }
  }
  if (gridDim.y == 1 || tile_cnt == nullptr) return;
  // ---- the last split of this tile to arrive adds the partial tiles (fixed order) and writes the rows out
  __shared__ int tg_last;
  __threadfence();  // (release: this workgroup's partial tile)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned seen = atomicAdd(&tile_cnt[tile], 1u);
    tg_last = (seen == gridDim.y - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!tg_last) return;
  __threadfence();  // (acquire: the other splits' partial tiles)
  {
    const int row = 64 * tile + (int)(threadIdx.x >> 4), c4 = (int)(threadIdx.x & 15) * 4;  // 64 rows x 16 groups of 4 columns
    if (row < nrows && c4 < 16 * nct) {
      typedef double v4d __attribute__((ext_vector_type(4)));
      const double *pp = part + ((int64_t)row << 6) + c4;
      v4d sacc = *reinterpret_cast<const v4d *>(pp);
      for (int q = 1; q < (int)gridDim.y; ++q) sacc += *reinterpret_cast<const v4d *>(pp + (((int64_t)q * nrows_pad) << 6));
      *reinterpret_cast<v4d *>(Out + ((int64_t)(rowmap ? rowmap[row] : row) << 6) + c4) = sacc;
    }
  }
  if (threadIdx.x == 0) tile_cnt[tile] = 0u;  // (ready for the next launch: kernel boundaries order it)
}

// the K splits of k_top_gemm, added in order
__global__ void __launch_bounds__(256) k_top_reduce(int nrows, int nks, const double *__restrict__ part, int nrows_pad,
                                                    const int32_t *__restrict__ rowmap, double *__restrict__ Out, int nct) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows || lane >= 16 * nct) return;
  double sacc = part[((int64_t)row << 6) + lane];
  for (int q = 1; q < nks; ++q) sacc += part[(((int64_t)q * nrows_pad + row) << 6) + lane];
  Out[((int64_t)(rowmap ? rowmap[row] : row) << 6) + lane] = sacc;
}

// ---------------------------------------------------------------------------------------------
// Block-dense thin bands, step (2): Out[rowmap[r]] = sum_{k<=r} Tinv(r,k) X[k] for one diagonal block
// (nb rows), Tinv = explicit inverse of the block's unit lower triangle, strip-major with
// lda = nb rounded up to 32 (zero padded; everything right of the diagonal is zero too).
// The product is MFMA-bound on gfx950 (v_mfma_f64_16x16x4_f64 occupies a SIMD for 64 cycles) and the
// triangle makes strip s cost s+1 units, so workgroup p owns the PAIR of strips (p, S-1-p): every
// workgroup then carries the same S+1 units and the grid is (pairs) x (16-column tiles) -- 256
// workgroups for a 2048-row block at 64 columns, one per compute unit.  The NW waves deal the
// pair's 32-k operand sets round-robin (split-K); a set is 8 A- and 8 B-fragments kept in registers,
// one set in flight while the previous one is multiplied, two accumulators per strip so that
// consecutive MFMAs do not wait for each other.  Partial tiles are summed through LDS in a fixed
// order (deterministic).  X must have nb rounded up to 32 readable rows (padding rows are masked).
// ---------------------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(NW * 64) k_tri_gemm_d(int nb, const double *__restrict__ A, int lda,
                                                        const double *__restrict__ X, int logR,
                                                        const int32_t *__restrict__ rowmap,
                                                        double *__restrict__ Out,
                                                        const double *__restrict__ dscale,
                                                        double *__restrict__ Out2) {
  __shared__ double red[NW - 1][2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = (nb + 15) >> 4;
  const int sA = blockIdx.x, sB = S - 1 - (int)blockIdx.x;
  const bool two = sB > sA;
  const int R = 1 << logR;
  const int colx = blockIdx.y * 16 + (lane & 15);
  const bool col_ok = colx < R;
  const int colc = col_ok ? colx : 0;
  const int kq = lane >> 4;
  const int nsA = (min(nb, 16 * (sA + 1)) + 31) >> 5;
  const int nsB = two ? (min(nb, 16 * (sB + 1)) + 31) >> 5 : 0;
  const int nsets = nsA + nsB;
  const double *Xl = X + colc;
  const double *A_a = A + ((int64_t)sA * lda) * 16 + (lane & 15);
  const double *A_b = A + ((int64_t)(two ? sB : sA) * lda) * 16 + (lane & 15);
  v4f64 accA0 = v4f64{0.0, 0.0, 0.0, 0.0}, accA1 = accA0, accB0 = accA0, accB1 = accA0;
  constexpr int KU = 8;
  double a0[KU], b0[KU], a1[KU], b1[KU];
#define HIFAMD_TG_LOAD(aa, bb, t_)                                                   \
  {                                                                                  \
    const bool isB_ = (t_) >= nsA;                                                   \
    const int kb_ = 32 * (isB_ ? (t_) - nsA : (t_)) + kq;                            \
    const double *ap_ = (isB_ ? A_b : A_a) + (int64_t)kb_ * 16;                      \
    const double *xp_ = Xl + ((int64_t)kb_ << logR);                                 \
    _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                 \
      aa[u] = ap_[u * 64];                                                           \
      const double xv_ = xp_[(int64_t)(4 * u) << logR];                              \
      bb[u] = (kb_ + 4 * u < nb) ? xv_ : 0.0;                                        \
    }                                                                                \
  }
#define HIFAMD_TG_MFMA(aa, bb, t_)                                                                 \
  if ((t_) >= nsA) {                                                                               \
    _Pragma("unroll") for (int u = 0; u < KU; u += 2) {                                            \
      accB0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u], bb[u], accB0, 0, 0, 0);                  \
      accB1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u + 1], bb[u + 1], accB1, 0, 0, 0);          \
    }                                                                                              \
  } else {                                                                                         \
    _Pragma("unroll") for (int u = 0; u < KU; u += 2) {                                            \
      accA0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u], bb[u], accA0, 0, 0, 0);                  \
      accA1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u + 1], bb[u + 1], accA1, 0, 0, 0);          \
    }                                                                                              \
  }
  int t = wave;
  if (t < nsets) HIFAMD_TG_LOAD(a0, b0, t)
  while (t < nsets) {
    const int t1 = t + NW;
    if (t1 < nsets) HIFAMD_TG_LOAD(a1, b1, t1)
    HIFAMD_TG_MFMA(a0, b0, t)
    if (t1 >= nsets) break;
    const int t2 = t1 + NW;
    if (t2 < nsets) HIFAMD_TG_LOAD(a0, b0, t2)
    HIFAMD_TG_MFMA(a1, b1, t1)
    t = t2;
  }
#undef HIFAMD_TG_LOAD
#undef HIFAMD_TG_MFMA
  const v4f64 accA = accA0 + accA1, accB = accB0 + accB1;
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[wave - 1][0][r][lane] = accA[r];
      red[wave - 1][1][r][lane] = accB[r];
    }
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h == 1 && !two) break;
    const int i0 = 16 * (h ? sB : sA);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double val = h ? accB[r] : accA[r];
#pragma unroll
      for (int w = 0; w < NW - 1; ++w) val += red[w][h][r][lane];
      const int row = i0 + kq + 4 * r;
      if (col_ok && row < nb) {
        const int orow = rowmap ? rowmap[row] : row;
        Out[((int64_t)orow << logR) + colx] = val;
        if (Out2) Out2[((int64_t)orow << logR) + colx] = val / dscale[orow];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Component-dense band (host.hpp plan_bands_cd), R = 64, real data.  Workgroup wg0 + blockIdx.x owns the components
// (groups) [wg_grp_ptr[g], wg_grp_ptr[g+1]); a component is a set of <= cd_rows rows that depend only on each other and
// on rows finished by EARLIER launches.  Per component (descriptor: host.hpp build_cd_streams):
//   phase 1  t[r] = rhs[r] - (the row's nonzeros whose sources lie outside the component and were not carried by the
//            previous launch).  Those nonzeros of ALL rows form one packed stream (column, value, local row); every wave
//            owns a contiguous chunk of rows and of the stream: it puts its rows' right-hand sides into LDS (all loads in
//            flight at once), then walks its entries as items of 64 with eight gathers per batch -- across row
//            boundaries, the running row's sum in a register, LDS touched once per row change.  No flags, no polling;
//   phase 2  x = Tinv_c * t on the f64 matrix cores: Tinv_c is the explicit inverse of the component's own unit
//            triangle (strip-major, built on the host), the B operand comes from LDS.  A unit is one 16-row strip x two
//            16-column tiles (an A fragment feeds two MFMAs); units go to the waves heaviest first in snake order over
//            the four SIMDs; operand sets of 8 k-steps are double buffered.
// No dependent step survives inside a launch, whatever the depth of the component.  Workgroups beyond n_band run the
// carried prefix of the NEXT band, exactly as in k_trsv_band_p.  The summation order differs from the reference's
// (tolerance-level, like every block-dense band); exact mode never plans such bands.
// ---------------------------------------------------------------------------------------------
// Development probe (make CSPROBE=1): every workgroup of k_band_cs / k_band_cd records wall-clock stamps (100 MHz) of its
// phases into a global buffer (engine.hip dumps it to HIFIR_AMD_CSPROBE_OUT); compiled out by default.
#ifdef HIFAMD_CSPROBE
__device__ unsigned long long *g_csprobe = nullptr;
__device__ unsigned g_csprobe_cap = 0, g_csprobe_cnt = 0;
#define HIFAMD_CSP_DECL unsigned long long csp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define HIFAMD_CSP(k) \
  if (threadIdx.x == 0) csp_[k] = wall_clock64();
#define HIFAMD_CSP_FLUSH(kind, a, b)                                                   \
  if (threadIdx.x == 0 && g_csprobe) {                                                 \
    const unsigned slot_ = atomicAdd(&g_csprobe_cnt, 1u);                              \
    if (slot_ < g_csprobe_cap) {                                                       \
      unsigned long long *r_ = g_csprobe + (size_t)slot_ * 12;                         \
      for (int k_ = 0; k_ < 8; ++k_) r_[k_] = csp_[k_];                                \
      r_[8] = ((unsigned long long)(kind) << 32) | blockIdx.x;                         \
      r_[9] = ((unsigned long long)gridDim.x << 32) | (unsigned)(a);                   \
      r_[10] = (unsigned long long)(b);                                                \
      r_[11] = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20) /* HW_REG_XCC_ID */; \
    }                                                                                  \
  }
#else
#define HIFAMD_CSP_DECL
#define HIFAMD_CSP(k)
#define HIFAMD_CSP_FLUSH(kind, a, b)
#endif

template <bool LOWER, bool SPARSE>
__global__ void __launch_bounds__(1024) k_band_cd(int32_t wg0, const int32_t *__restrict__ wg_grp_ptr,
                                                  const int32_t *__restrict__ cd_desc, const int32_t *__restrict__ ptr,
                                                  const int32_t *__restrict__ split, const int32_t *__restrict__ col,
                                                  const double *__restrict__ val, const int32_t *__restrict__ rowid,
                                                  const double *__restrict__ d, double *w, double *v,
                                                  const double *__restrict__ tinv, const int32_t *__restrict__ mid_col,
                                                  const double *__restrict__ mid_val, const uint8_t *__restrict__ mid_lrow,
                                                  int first_u, int32_t n_band, int32_t ps0, int32_t ps1, int32_t single_c0,
                                                  int32_t lds_rows, int32_t own_cap, int dbg, FirstL<double> fl,
                                                  const double *__restrict__ own_val, const uint8_t *__restrict__ own_lsrc,
                                                  const uint16_t *__restrict__ own_rptr, const uint8_t *__restrict__ own_lvl,
                                                  LastU<double> lu, RowSkip rs) {
  extern __shared__ double cd_tbuf[];  // [lds_rows][64] right-hand sides, then lds_rows row ids
  HIFAMD_CSP_DECL
  HIFAMD_CSP(0)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  if ((int32_t)blockIdx.x >= n_band) {  // carried prefix of the next band over the sources older than this band
    const int32_t pw = __builtin_amdgcn_readfirstlane(((int32_t)blockIdx.x - n_band) * nw + wave);
    if (dbg & 4) return;
    const double *pf_bin = (LOWER && fl.on()) ? fl.bin.get() : nullptr;
    trsv_stream_r64<double, 0, LOWER, true>(ps0 + pw, ps1, ((int32_t)gridDim.x - n_band) * nw, ptr, split, col, val, nullptr,
                                            rowid, d, LOWER ? w : v, w, lane, nullptr, 0, nullptr, true, nullptr, 0, 0, pf_bin, &fl);
    HIFAMD_CSP(6)
    HIFAMD_CSP_FLUSH(2 + (LOWER ? 0 : 4) + 16, n_band, 0)
    return;
  }
  double *x = LOWER ? w : v;
  const bool div_u = !LOWER && first_u;
  // S1 fused (FirstL): a band that touches its L rows first takes rhs[i] = s[p[i]] * b[p[i]] from the level's input
  const bool first_l = LOWER && first_u && fl.on();
  const double *rhs = div_u ? (const double *)w : (first_l ? fl.bin.get() : (const double *)x);
  const int64_t rstride = first_l ? fl.ldb : 64;
  const int rlane = first_l ? min(lane, fl.nrhs - 1) : lane;
  // single_c0 >= 0: every workgroup of this band owns exactly ONE component, number single_c0 + blockIdx.x (saves the
  // dependent load of the workgroup's group range)
  int32_t c_first, c_last;
  if (single_c0 >= 0) {
    c_first = single_c0 + (int32_t)blockIdx.x;
    c_last = c_first + 1;
  } else {
    const int g = wg0 + (int)blockIdx.x;
    c_first = wg_grp_ptr[g];
    c_last = wg_grp_ptr[g + 1];
  }
  const int kq = lane >> 4;
  int32_t *cd_rowid = reinterpret_cast<int32_t *>(cd_tbuf + (size_t)lds_rows * 64);  // the component's row ids, for phase 2
  // SPARSE: the component's own nonzeros (value, local source row), their row offsets and the depth levels
  double *ow_val = reinterpret_cast<double *>(cd_rowid + ((lds_rows + 1) & ~1));
  uint8_t *ow_src = reinterpret_cast<uint8_t *>(ow_val + own_cap);  // (own_cap: multiple of 64, host: the plan's maximum)
  uint16_t *ow_rptr = reinterpret_cast<uint16_t *>(ow_src + own_cap);
  uint8_t *ow_lvl = reinterpret_cast<uint8_t *>(ow_rptr + 260);
  // fused S7 (LastU): output row and scale of every row of the component, behind everything else
  const bool last_u = !LOWER && lu.on();
  double *cd_ot = SPARSE ? reinterpret_cast<double *>(ow_lvl + 264) : reinterpret_cast<double *>(cd_rowid + ((lds_rows + 1) & ~1));
  int32_t *cd_oi = reinterpret_cast<int32_t *>(cd_ot + lds_rows);
  double *yout = last_u ? lu.out.get() : nullptr;
  for (int32_t c = c_first; c < c_last; ++c) {
    const int32_t *dsc = cd_desc + (int64_t)c * 28;
    const int32_t s0 = dsc[0], nb = dsc[1], mid0 = dsc[2];
    const int64_t inv_off = ((int64_t)(uint32_t)dsc[5] << 32) | (uint32_t)dsc[4];
    const uint8_t *wrow = reinterpret_cast<const uint8_t *>(dsc + 6);
    const uint16_t *wmid = reinterpret_cast<const uint16_t *>(dsc + 11);
    const int r0 = wrow[wave], nr = (int)wrow[wave + 1] - r0;
    const int32_t e0 = mid0 + (int32_t)wmid[wave], e1 = mid0 + (int32_t)wmid[wave + 1];
    // phase 2's first operand set is requested NOW: it does not depend on phase 1 and its latency hides behind it
    const int lda = (nb + 31) & ~31;
    const double *Ac = tinv + inv_off;
    const int S = (nb + 15) >> 4, nunits = S * 2;
    const int wrow4 = wave >> 2, wcol4 = wave & 3;
    const int q_first = wrow4 * 4 + ((wrow4 & 1) ? 3 - wcol4 : wcol4);  // snake order over the SIMDs (wave % 4)
    constexpr int KU = 8;
    double a0[KU], a1[KU];
    if (SPARSE) {  // own nonzeros, row offsets and levels into LDS (coalesced, in flight during phase 1)
      const int32_t own0 = dsc[20], nown = dsc[21], orp0 = dsc[22], lvl0 = dsc[23], nlvl = dsc[24];
      for (int32_t t = (int32_t)threadIdx.x; t < nown; t += (int32_t)blockDim.x) {
        ow_val[t] = own_val[own0 + t];
        ow_src[t] = own_lsrc[own0 + t];
      }
      for (int32_t t = (int32_t)threadIdx.x; t <= nb; t += (int32_t)blockDim.x) ow_rptr[t] = own_rptr[orp0 + t];
      for (int32_t t = (int32_t)threadIdx.x; t <= nlvl; t += (int32_t)blockDim.x) ow_lvl[t] = own_lvl[lvl0 + t];
    }
    if (!SPARSE && q_first < nunits) {
      const double *ap_ = Ac + ((int64_t)(S - 1 - (q_first >> 1)) * lda) * 16 + (lane & 15) + (int64_t)kq * 16;
#pragma unroll
      for (int u = 0; u < KU; ++u) a0[u] = ap_[u * 64];
    }
    // ---- phase 1a: right-hand sides of this wave's rows into LDS (row ids one per lane, eight loads in flight)
    int32_t h_i = 0, h_p = LOWER ? 0 : -1;
    double h_d = 1.0;  // (U: the pivot; fused S1: the row's scale)
    double h_s = 0.0;  // (U, rowflag bit 1: the row's scale)
    const bool from_b = !LOWER && SPARSE && div_u && rs.flag != nullptr;
    const double *bsrc = from_b ? fl.bin.get() : nullptr;
    if (lane < nr) {
      h_i = rowid[s0 + r0 + lane];
      const int hf = (SPARSE && rs.flag) ? (int)rs.flag[s0 + r0 + lane] : 0;
      cd_rowid[r0 + lane] = (hf & 1) ? ~h_i : h_i;  // (negative: not stored)
      if (div_u) h_d = d[h_i];
      if (from_b && (hf & 2)) {
        h_p = fl.p[h_i];
        h_s = fl.s[h_p];
      }
      if (first_l) {
        h_p = fl.p[h_i];
        h_d = fl.s[h_p];
      }
      if (last_u) {
        const int32_t oi = lu.q[h_i];
        cd_oi[r0 + lane] = oi;
        cd_ot[r0 + lane] = lu.t[oi];
      }
    }
    // first item of the wave's entry stream: requested before the right-hand sides are waited for
    int32_t colv = 0, lrv = 0;
    double valv = 0.0;
    if (e0 + lane < e1) {
      colv = mid_col[e0 + lane];
      valv = mid_val[e0 + lane];
      lrv = mid_lrow[e0 + lane];
    }
    for (int j = 0; j < nr; j += 8) {
      double t_[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int32_t i = rl32(first_l ? h_p : h_i, min(j + q, 63));
        const double *src = rhs + (int64_t)i * rstride + rlane;
        if (from_b) {
          const int32_t pb = rl32(h_p, min(j + q, 63));  // (wave-uniform)
          if (pb >= 0) src = bsrc + (int64_t)pb * fl.ldb + min(lane, fl.nrhs - 1);
        }
        t_[q] = (j + q < nr) ? *src : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (j + q < nr) {
          const double hd = rl64(h_d, min(j + q, 63));
          double tq = t_[q];
          if (from_b && rl32(h_p, min(j + q, 63)) >= 0) tq = lane < fl.nrhs ? rl64(h_s, min(j + q, 63)) * tq : 0.0;
          cd_tbuf[((r0 + j + q) << 6) + lane] = div_u ? tq / hd : (first_l ? (lane < fl.nrhs ? hd * tq : 0.0) : tq);
        }
    }
    HIFAMD_CSP(3)
    // ---- phase 1b: the wave's entries, items of 64, eight gathers per batch; the running row's sum stays in a register
    int cur_r = -1;
    double acc = 0.0;
    for (int32_t e = e0; e < ((dbg & 1) ? e0 : e1); e += 64) {
      const int cnt = min(64, e1 - e);
      int32_t colv2 = 0, lrv2 = 0;
      double valv2 = 0.0;
      if (e + 64 + lane < e1) {
        colv2 = mid_col[e + 64 + lane];
        valv2 = mid_val[e + 64 + lane];
        lrv2 = mid_lrow[e + 64 + lane];
      }
      for (int t = 0; t < cnt; t += 8) {
        int32_t j_[8], r_[8];
        double a_[8], xv_[8];
#pragma unroll
        for (int b2 = 0; b2 < 8; ++b2) {
          const int idx = min(t + b2, 63);
          j_[b2] = rl32(colv, idx);
          a_[b2] = rl64(valv, idx);
          r_[b2] = rl32(lrv, idx);
        }
#pragma unroll
        for (int b2 = 0; b2 < 8; ++b2)
          if (t + b2 < cnt) xv_[b2] = x[((int64_t)j_[b2] << 6) + lane];
#pragma unroll
        for (int b2 = 0; b2 < 8; ++b2)
          if (t + b2 < cnt) {
            if (r_[b2] != cur_r) {  // (wave-uniform)
              if (cur_r >= 0) cd_tbuf[(cur_r << 6) + lane] = acc;
              cur_r = r_[b2];
              acc = cd_tbuf[(cur_r << 6) + lane];
            }
            acc = acc - a_[b2] * xv_[b2];
          }
      }
      colv = colv2;
      valv = valv2;
      lrv = lrv2;
    }
    if (cur_r >= 0) cd_tbuf[(cur_r << 6) + lane] = acc;
    // (the inverse product reads whole operand sets of 32 rows: rows nb .. lda - 1 are zero; lds_rows is a multiple of 32)
    if (!SPARSE)
      for (int t = nb * 64 + (int)threadIdx.x; t < lda * 64; t += (int)blockDim.x) cd_tbuf[t] = 0.0;
    HIFAMD_CSP(4)
    __syncthreads();
    HIFAMD_CSP(5)
    if (SPARSE) {
      // ---- phase 2, sparse: substitution inside LDS, depth level by depth level.  The rows of a level are independent;
      // every own source sits in an earlier level (host.hpp build_cd_streams).  A wave takes a row at a time: the row's
      // (value, source) pairs come from LDS, so do the source rows -- no global access on the dependent path.
      const int nlvl = dsc[24];
      if (dbg & 512) {  // (timing experiments: rows stored as they are, no substitution -- WRONG results)
        for (int r = wave; r < nb; r += nw) {
          const double a2 = cd_tbuf[(r << 6) + lane];
          if (last_u) {
            if (lane < lu.nrhs) yout[(int64_t)cd_oi[r] * lu.ldy + lane] = cd_ot[r] * a2;
          } else if (cd_rowid[r] >= 0) {
            x[((int64_t)cd_rowid[r] << 6) + lane] = a2;
          }
        }
        __syncthreads();
        continue;
      }
      for (int lv = 0; lv < nlvl; ++lv) {
        const int r_lo = ow_lvl[lv], r_hi = ow_lvl[lv + 1];
        for (int r = r_lo + wave; r < r_hi; r += nw) {
          double a2 = cd_tbuf[(r << 6) + lane];
          const int eb = ow_rptr[r], ee = ow_rptr[r + 1];
          for (int e = eb; e < ee; ++e) a2 = a2 - ow_val[e] * cd_tbuf[((int)ow_src[e] << 6) + lane];
          if (ee > eb) cd_tbuf[(r << 6) + lane] = a2;
          if (last_u) {
            if (lane < lu.nrhs) yout[(int64_t)cd_oi[r] * lu.ldy + lane] = cd_ot[r] * a2;
          } else {
            const int32_t rid = cd_rowid[r];
            if (rid >= 0) x[((int64_t)rid << 6) + lane] = a2;
          }
        }
        __syncthreads();
      }
      continue;
    }
    // ---- phase 2: x = Tinv * t (units are handed out heaviest first)
    for (int q = q_first; q < ((dbg & 2) ? 0 : nunits); q += nw) {
      const int strip = S - 1 - (q >> 1), ch = q & 1;
      const int kend = min(nb, 16 * (strip + 1));
      const int nsets = (kend + 31) >> 5;
      const double *Ap = Ac + ((int64_t)strip * lda) * 16 + (lane & 15);
      const double *Bp = cd_tbuf + ch * 32 + (lane & 15);
      v4f64 acc0 = v4f64{0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#define HIFAMD_CD_LOAD(aa, t_)                                          \
  {                                                                     \
    const double *ap_ = Ap + (int64_t)(32 * (t_) + kq) * 16;            \
    _Pragma("unroll") for (int u = 0; u < KU; ++u) aa[u] = ap_[u * 64]; \
  }
#define HIFAMD_CD_MFMA(aa, t_)                                                    \
  {                                                                               \
    const int kb_ = 32 * (t_) + kq;                                               \
    _Pragma("unroll") for (int u = 0; u < KU; ++u) {                              \
      const double b0_ = Bp[(kb_ + 4 * u) << 6];                                  \
      const double b1_ = Bp[((kb_ + 4 * u) << 6) + 16];                           \
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u], b0_, acc0, 0, 0, 0);     \
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u], b1_, acc1, 0, 0, 0);     \
    }                                                                             \
  }
      int t = 0;
      if (q != q_first) HIFAMD_CD_LOAD(a0, 0)  // (the first unit's first set was requested before phase 1)
      while (t < nsets) {
        if (t + 1 < nsets) HIFAMD_CD_LOAD(a1, t + 1)
        HIFAMD_CD_MFMA(a0, t)
        if (t + 1 >= nsets) break;
        if (t + 2 < nsets) HIFAMD_CD_LOAD(a0, t + 2)
        HIFAMD_CD_MFMA(a1, t + 1)
        t += 2;
      }
#undef HIFAMD_CD_LOAD
#undef HIFAMD_CD_MFMA
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * strip + kq + 4 * r;
        if (row < nb) {
          if (last_u) {
            const int cc = ch * 32 + (lane & 15);
            double *yo = yout + (int64_t)cd_oi[row] * lu.ldy + cc;
            const double tt = cd_ot[row];
            if (cc < lu.nrhs) yo[0] = tt * acc0[r];
            if (cc + 16 < lu.nrhs) yo[16] = tt * acc1[r];
          } else {
            double *xo = x + ((int64_t)cd_rowid[row] << 6) + ch * 32 + (lane & 15);
            xo[0] = acc0[r];
            xo[16] = acc1[r];
          }
        }
      }
    }
    __syncthreads();  // (the next component overwrites the LDS block)
  }
  HIFAMD_CSP(6)
  HIFAMD_CSP_FLUSH(1 + (LOWER ? 0 : 4) + (SPARSE ? 8 : 0) + 16, n_band, 0)
}

// ---------------------------------------------------------------------------------------------
// Sparse-own U band with streamed sinks (round 4; host.hpp build_us_plan).  One component per workgroup (1,024 threads), R = 64.
//   phase 0  descriptors, the component's own entries / row offsets / black levels into LDS; output rows of a fused S7
//   phase A  the BLACK rows' right-hand sides (w / d, or -- RowSkip -- s b[p] / d) minus their outside entries -> LDS;
//            the first batch of sink right-hand sides is requested here too and arrives during phase B
//   phase B  the black rows level by level in LDS, a wave per row (k_band_cd<false, sparse>'s loop); results stored
//   phase C  the sinks, a wave per row, eight rows of a wave in flight: right-hand side, outside entries, own entries
//            gathered from the black rows in LDS, result stored -- a sink never touches LDS
// Per row: the operations of k_band_cd in the same order -- the same bits (width independence is tested against the slice
// kernel k_band_cs, which narrow batches keep).  LDS: black rows x 512 B + the own entries: two workgroups per compute unit
// on the reference's hierarchies (<= 64 VGPRs), the occupancy the level-0 bands are bound by.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
    k_band_us(int32_t c0, const int32_t *__restrict__ desc, const int32_t *__restrict__ rowid, const int32_t *__restrict__ oslot,
              const int32_t *__restrict__ mptr, const int32_t *__restrict__ mcol, const double *__restrict__ mval,
              const double *__restrict__ own_val, const uint8_t *__restrict__ own_src, const uint16_t *__restrict__ own_rptr,
              const uint8_t *__restrict__ own_lvl, const double *__restrict__ d, const double *__restrict__ w, double *v,
              int32_t lds_black, int32_t own_cap, FirstL<double> fl, LastU<double> lu, RowSkip rs) {
  extern __shared__ double us_tb[];  // [lds_black][64] black rows; then own values, row ids, S7 rows, own sources, offsets, levels
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  constexpr int NW = 16;
  const int32_t *dsc = desc + (int64_t)(c0 + (int32_t)blockIdx.x) * 8;
  const int32_t s0 = dsc[0], nb = dsc[1], nbk = dsc[2], own0 = dsc[3], orp0 = dsc[4], lvl0 = dsc[5], nlvl = dsc[6];
  double *ow_val = us_tb + (size_t)lds_black * 64;
  double *s_ot = ow_val + own_cap;
  int32_t *s_rowid = reinterpret_cast<int32_t *>(s_ot + 256);
  int32_t *s_oi = s_rowid + 256;
  uint16_t *ow_rptr = reinterpret_cast<uint16_t *>(s_oi + 256);
  uint8_t *ow_src = reinterpret_cast<uint8_t *>(ow_rptr + 260);
  uint8_t *ow_lvl = ow_src + own_cap;
  const bool last_u = lu.on();
  const bool from_b = rs.flag != nullptr;
  const double *bsrc = from_b ? fl.bin.get() : nullptr;
  double *yout = last_u ? lu.out.get() : nullptr;
  // ---- phase 0
  const int32_t nown = own_rptr[orp0 + nb];
  for (int32_t t = (int32_t)threadIdx.x; t < nown; t += 1024) {
    ow_val[t] = own_val[own0 + t];
    ow_src[t] = own_src[own0 + t];
  }
  for (int32_t t = (int32_t)threadIdx.x; t <= nb; t += 1024) ow_rptr[t] = own_rptr[orp0 + t];
  for (int32_t t = (int32_t)threadIdx.x; t <= nlvl; t += 1024) ow_lvl[t] = own_lvl[lvl0 + t];
  for (int32_t t = (int32_t)threadIdx.x; t < nb; t += 1024) {
    const int32_t i = rowid[s0 + t];
    const int hf = from_b ? (int)rs.flag[oslot[s0 + t]] : 0;
    s_rowid[t] = (hf & 1) ? ~i : i;  // (negative: not stored)
    if (last_u) {
      const int32_t oi = lu.q[i];
      s_oi[t] = oi;
      s_ot[t] = lu.t[oi];
    }
  }
  // one row's right-hand side as k_band_cd forms it: w[i] / d[i], or (RowSkip bit 1) (s[p[i]] * b[p[i]]) / d[i]
  // lane k of a wave holds the scalars of the wave's k-th row of the batch
  auto row_scalars = [&](int32_t r, bool valid, int32_t &h_i, int32_t &h_p, double &h_d, double &h_s, int32_t &h_m0, int32_t &h_m1) {
    h_i = 0, h_p = -1, h_d = 1.0, h_s = 0.0, h_m0 = 0, h_m1 = 0;
    if (valid) {
      h_i = rowid[s0 + r];
      h_m0 = mptr[s0 + r];
      h_m1 = mptr[s0 + r + 1];
      h_d = d[h_i];
      if (from_b && (rs.flag[oslot[s0 + r]] & 2)) {
        h_p = fl.p[h_i];
        h_s = fl.s[h_p];
      }
    }
  };
  // ---- phase A: black rows w, w + 16, ... (at most 16 per wave: 256 rows / 16 waves), eight at a time
  for (int32_t kb = 0; wave + NW * kb < nbk; kb += 8) {
    int32_t h_i, h_p, h_m0, h_m1;
    double h_d, h_s;
    const int32_t myr = wave + NW * (kb + lane);
    row_scalars(myr, lane < 8 && myr < nbk, h_i, h_p, h_d, h_s, h_m0, h_m1);
    double t_[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int32_t i = rl32(h_i, q), pb = rl32(h_p, q);
      const double *src = (pb >= 0) ? bsrc + (int64_t)pb * fl.ldb + min(lane, fl.nrhs - 1) : w + ((int64_t)i << 6) + lane;
      t_[q] = (wave + NW * (kb + q) < nbk) ? *src : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int32_t r = wave + NW * (kb + q);
      if (r < nbk) {
        double tq = t_[q];
        if (rl32(h_p, q) >= 0) tq = lane < fl.nrhs ? rl64(h_s, q) * tq : 0.0;
        double acc = tq / rl64(h_d, q);
        const int32_t m0 = rl32(h_m0, q), m1 = rl32(h_m1, q);  // (outside entries: few rows of level 0 have any)
        for (int32_t e = m0; e < m1; ++e) acc = acc - mval[e] * v[((int64_t)mcol[e] << 6) + lane];
        us_tb[(r << 6) + lane] = acc;
      }
    }
  }
  // the first batch of sinks: rows nbk + w, nbk + w + 16, ... requested NOW (nothing of the black solve feeds their loads)
  int32_t g_i, g_p, g_m0, g_m1;
  double g_d, g_s;
  double st_[8];
  {
    const int32_t myr = nbk + wave + NW * lane;
    row_scalars(myr, lane < 8 && myr < nb, g_i, g_p, g_d, g_s, g_m0, g_m1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int32_t i = rl32(g_i, q), pb = rl32(g_p, q);
      const double *src = (pb >= 0) ? bsrc + (int64_t)pb * fl.ldb + min(lane, fl.nrhs - 1) : w + ((int64_t)i << 6) + lane;
      st_[q] = (nbk + wave + NW * q < nb) ? *src : 0.0;
    }
  }
  __syncthreads();
  // ---- phase B: the black rows, depth level by depth level
  for (int lv = 0; lv < nlvl; ++lv) {
    const int r_lo = ow_lvl[lv], r_hi = ow_lvl[lv + 1];
    for (int r = r_lo + wave; r < r_hi; r += NW) {
      double a2 = us_tb[(r << 6) + lane];
      const int eb = ow_rptr[r], ee = ow_rptr[r + 1];
      for (int e = eb; e < ee; ++e) a2 = a2 - ow_val[e] * us_tb[((int)ow_src[e] << 6) + lane];
      if (ee > eb) us_tb[(r << 6) + lane] = a2;
      if (last_u) {
        if (lane < lu.nrhs) yout[(int64_t)s_oi[r] * lu.ldy + lane] = s_ot[r] * a2;
      } else {
        const int32_t rid = s_rowid[r];
        if (rid >= 0) v[((int64_t)rid << 6) + lane] = a2;
      }
    }
    __syncthreads();
  }
  // ---- phase C: the sinks (eight rows of this wave per batch; the first batch is already in registers)
  for (int32_t kb = 0; nbk + wave + NW * kb < nb; kb += 8) {
    if (kb) {
      const int32_t myr = nbk + wave + NW * (kb + lane);
      row_scalars(myr, lane < 8 && myr < nb, g_i, g_p, g_d, g_s, g_m0, g_m1);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int32_t i = rl32(g_i, q), pb = rl32(g_p, q);
        const double *src = (pb >= 0) ? bsrc + (int64_t)pb * fl.ldb + min(lane, fl.nrhs - 1) : w + ((int64_t)i << 6) + lane;
        st_[q] = (nbk + wave + NW * (kb + q) < nb) ? *src : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int32_t r = nbk + wave + NW * (kb + q);
      if (r < nb) {
        double tq = st_[q];
        if (rl32(g_p, q) >= 0) tq = lane < fl.nrhs ? rl64(g_s, q) * tq : 0.0;
        double a2 = tq / rl64(g_d, q);
        const int32_t m0 = rl32(g_m0, q), m1 = rl32(g_m1, q);
        for (int32_t e = m0; e < m1; ++e) a2 = a2 - mval[e] * v[((int64_t)mcol[e] << 6) + lane];
        const int eb = ow_rptr[r], ee = ow_rptr[r + 1];
        for (int e = eb; e < ee; ++e) a2 = a2 - ow_val[e] * us_tb[((int)ow_src[e] << 6) + lane];
        if (last_u) {
          if (lane < lu.nrhs) yout[(int64_t)s_oi[r] * lu.ldy + lane] = s_ot[r] * a2;
        } else {
          const int32_t rid = s_rowid[r];
          if (rid >= 0) v[((int64_t)rid << 6) + lane] = a2;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Column-sliced component band (round 3).  The same plan, descriptors and packed streams as k_band_cd, but a workgroup
// solves ONE component for a SLICE of 16 right-hand-side columns: blockIdx.x = component-workgroup * nsl + slice, and
// every step of the solve is column-separable, so the nsl slices of a component never talk to each other.  Why:
//   * a narrow band (tens to a few hundred components: everything above the leaves of a triangle) lasts as long as ONE
//     compute unit needs for its heaviest component -- sliced four ways, that component's gathers, its inverse product
//     and its loads / stores run on four units at once (the critical path shrinks, the launch count does not change);
//   * a batch of fewer than 49 columns launches only the slices it has: nrhs <= 16 moves a quarter of the vector bytes
//     and issues a quarter of the gathers of the 64-column kernel (BASELINE configs[1] and [4]).
// Mapping: 4 waves; a wave is four ROWS of 16 lanes (lane = 16 g + c: column cbase + c).  Lane group q = 4 wave + g
// takes chunk q of the descriptor's sixteen row / entry chunks (k_band_cd gives chunk q to wave q): it loads its rows'
// right-hand sides into LDS, then walks ITS entries -- (column, value, local row) fetched 16 at a time with one
// coalesced load per lane group and broadcast inside the 16 lanes with DPP row_newbcast (a VALU move: no LDS, no
// scalar round trip), eight 128-byte gathers per lane group in flight, the four lane groups of a wave in lock step.
// Arithmetic per row and column is EXACTLY k_band_cd's (rhs, then the entries subtracted one by one in stream order;
// the inverse product accumulated over k in steps of four from k = 0): both kernels give the same bits, whatever
// the slicing -- a column's result does not depend on the width of the batch it travels in.
// Workgroups beyond n_band * nsl run the carried prefix of the next band (full 64-column rows, as in k_band_cd).
// ---------------------------------------------------------------------------------------------
template <int IDX>
__device__ __forceinline__ int32_t bc16(int32_t v) {  // lane IDX of every 16-lane row, to all lanes of that row
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + IDX, 0xf, 0xf, false);
}
template <int IDX>
__device__ __forceinline__ double bc16(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)bc16<IDX>((int32_t)(b & 0xffffffffLL));
  const unsigned hi = (unsigned)bc16<IDX>((int32_t)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ int wave_max4(int v) {  // max over the four 16-lane rows of a wave (v uniform inside a row)
  return max(max(rl32(v, 0), rl32(v, 16)), max(rl32(v, 32), rl32(v, 48)));
}

// eight entries of every lane group (positions T0 .. T0+7 of the current item), see k_band_cs phase 1b
template <int T0>
__device__ __forceinline__ void cs_batch(int32_t colv, double valv, int32_t lrv, int32_t left, const double *__restrict__ x,
                                         int cc, int l16, double *tb, int &cur, double &acc) {
  int32_t j_[8], r_[8];
  double a_[8], xv_[8], rv_[8];
  bool nw_[8];
  static_for<0, 8>([&](auto ic) {
    constexpr int u = decltype(ic)::value;
    j_[u] = bc16<T0 + u>(colv);
    a_[u] = bc16<T0 + u>(valv);
    r_[u] = bc16<T0 + u>(lrv);
  });
#pragma unroll
  for (int u = 0; u < 8; ++u) xv_[u] = (T0 + u < left) ? x[((int64_t)j_[u] << 6) + cc] : 0.0;
  // right-hand sides of the rows that START in this batch: requested beside the gathers (a lane group meets its rows
  // once, in order, and nobody else touches them in this phase)
  int prev = cur;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    nw_[u] = (T0 + u < left) && r_[u] != prev;
    rv_[u] = nw_[u] ? tb[(r_[u] << 4) + l16] : 0.0;
    if (T0 + u < left) prev = r_[u];
  }
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (T0 + u < left) {
      if (nw_[u]) {
        if (cur >= 0) tb[(cur << 4) + l16] = acc;
        cur = r_[u];
        acc = rv_[u];
      }
      acc = acc - a_[u] * xv_[u];
    }
}

template <bool LOWER, bool SPARSE>
__global__ void __launch_bounds__(256) k_band_cs(int32_t wg0, const int32_t *__restrict__ wg_grp_ptr,
                                                 const int32_t *__restrict__ cd_desc, const int32_t *__restrict__ ptr,
                                                 const int32_t *__restrict__ split, const int32_t *__restrict__ col,
                                                 const double *__restrict__ val, const int32_t *__restrict__ rowid,
                                                 const double *__restrict__ d, double *w, double *v,
                                                 const double *__restrict__ tinv, const int32_t *__restrict__ mid_col,
                                                 const double *__restrict__ mid_val, const uint8_t *__restrict__ mid_lrow,
                                                 int first_u, int32_t n_band, int32_t nsl, int32_t ps0, int32_t ps1,
                                                 int32_t single_c0, int32_t lds_rows, int32_t own_cap, int dbg, FirstL<double> fl,
                                                 const double *__restrict__ own_val, const uint8_t *__restrict__ own_lsrc,
                                                 const uint16_t *__restrict__ own_rptr, const uint8_t *__restrict__ own_lvl,
                                                 LastU<double> lu, RowSkip rs) {
  extern __shared__ double cs_buf[];
  HIFAMD_CSP_DECL
  HIFAMD_CSP(0)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int32_t nbw = n_band * nsl;
  if (dbg & 8) return;  // (timing experiments: the bare launch)
  if ((int32_t)blockIdx.x >= nbw) {  // carried prefix of the next band over the sources older than this band
    const int32_t pw = __builtin_amdgcn_readfirstlane(((int32_t)blockIdx.x - nbw) * 4 + wave);
    if (dbg & 4) return;
    const double *pf_bin = (LOWER && fl.on()) ? fl.bin.get() : nullptr;
    trsv_stream_r64<double, 0, LOWER, true>(ps0 + pw, ps1, ((int32_t)gridDim.x - nbw) * 4, ptr, split, col, val, nullptr,
                                            rowid, d, LOWER ? w : v, w, lane, nullptr, 0, nullptr, true, nullptr, 0, 0, pf_bin, &fl);
    HIFAMD_CSP(6)
    HIFAMD_CSP_FLUSH(2 + (LOWER ? 0 : 4), n_band, nsl)
    return;
  }
  const int32_t bw = (int32_t)blockIdx.x / nsl, slice = (int32_t)blockIdx.x - bw * nsl;
  const int grp = lane >> 4, l16 = lane & 15, gq = wave * 4 + grp;  // lane group gq of 16 owns chunk gq
  const int cc = slice * 16 + l16;                                   // this lane's column of the 64-column arena
  double *x = LOWER ? w : v;
  const bool div_u = !LOWER && first_u;
  const bool first_l = LOWER && first_u && fl.on();
  const double *rhs = div_u ? (const double *)w : (first_l ? fl.bin.get() : (const double *)x);
  const int64_t rstride = first_l ? fl.ldb : 64;
  const int rcol = first_l ? min(cc, fl.nrhs - 1) : cc;
  int32_t c_first, c_last;
  if (single_c0 >= 0) {
    c_first = single_c0 + bw;
    c_last = c_first + 1;
  } else {
    c_first = wg_grp_ptr[wg0 + bw];
    c_last = wg_grp_ptr[wg0 + bw + 1];
  }
  // LDS: right-hand sides [lds_rows][16], then per row: pivot / scale, output scale, row id, input row, output row;
  // sparse-own plans: the component's own nonzeros (value, local source), row offsets, depth levels
  double *tb = cs_buf;
  double *s_hd = tb + (size_t)lds_rows * 16;
  double *s_ot = s_hd + lds_rows;
  double *ow_val = s_ot + lds_rows;
  int32_t *s_rowid = reinterpret_cast<int32_t *>(ow_val + (SPARSE ? own_cap : 0));
  int32_t *s_hp = s_rowid + lds_rows;
  int32_t *s_oi = s_hp + lds_rows;
  uint16_t *ow_rptr = reinterpret_cast<uint16_t *>(s_oi + lds_rows);
  uint8_t *ow_src = reinterpret_cast<uint8_t *>(ow_rptr + 260);
  uint8_t *ow_lvl = ow_src + own_cap;
  const bool last_u = !LOWER && lu.on();
  double *yout = last_u ? lu.out.get() : nullptr;
  const int kq = grp;
  // (RowSkip: a row of the U solve whose right-hand side comes from the level's input keeps p[i] in s_hp -- -1 otherwise --
  // and its scale in s_ot; the fused S7 never meets the flags: they belong to a level's first solve)
  const bool from_b = !LOWER && SPARSE && div_u && rs.flag != nullptr;
  const double *bsrc = from_b ? fl.bin.get() : nullptr;
  for (int32_t c = c_first; c < c_last; ++c) {
    const int32_t *dsc = cd_desc + (int64_t)c * 28;
    const int32_t s0 = dsc[0], nb = dsc[1], mid0 = dsc[2];
    const int64_t inv_off = ((int64_t)(uint32_t)dsc[5] << 32) | (uint32_t)dsc[4];
    const uint8_t *wrow = reinterpret_cast<const uint8_t *>(dsc + 6);
    const uint16_t *wmid = reinterpret_cast<const uint16_t *>(dsc + 11);
    const int r0 = wrow[gq], nr = (int)wrow[gq + 1] - r0;
    const int32_t e0 = mid0 + (int32_t)wmid[gq], ne = (int32_t)wmid[gq + 1] - (int32_t)wmid[gq];
    const int lda = (nb + 31) & ~31;
    const double *Ac = tinv + inv_off;
    const int S = (nb + 15) >> 4;
    constexpr int KU = 8;
    double a0[KU], a1[KU];
    // ---- phase 0: the component's row ids and per-row scalars (coalesced), sparse: its own nonzeros
    for (int32_t t = (int32_t)threadIdx.x; t < nb; t += 256) {
      const int32_t i = rowid[s0 + t];
      const int hf = (SPARSE && rs.flag) ? (int)rs.flag[s0 + t] : 0;
      s_rowid[t] = (hf & 1) ? ~i : i;  // (negative: not stored)
      if (div_u) s_hd[t] = d[i];
      if (from_b) {
        int32_t pp = -1;
        if (hf & 2) {
          pp = fl.p[i];
          s_ot[t] = fl.s[pp];
        }
        s_hp[t] = pp;
      }
      if (first_l) {
        const int32_t pp = fl.p[i];
        s_hp[t] = pp;
        s_hd[t] = fl.s[pp];
      }
      if (last_u) {
        const int32_t oi = lu.q[i];
        s_oi[t] = oi;
        s_ot[t] = lu.t[oi];
      }
    }
    if (SPARSE) {
      const int32_t own0 = dsc[20], nown = dsc[21], orp0 = dsc[22], lvl0 = dsc[23], nlvl = dsc[24];
      for (int32_t t = (int32_t)threadIdx.x; t < nown; t += 256) {
        ow_val[t] = own_val[own0 + t];
        ow_src[t] = own_lsrc[own0 + t];
      }
      for (int32_t t = (int32_t)threadIdx.x; t <= nb; t += 256) ow_rptr[t] = own_rptr[orp0 + t];
      for (int32_t t = (int32_t)threadIdx.x; t <= nlvl; t += 256) ow_lvl[t] = own_lvl[lvl0 + t];
    }
    // phase 2's first operand set and phase 1b's first item do not depend on anything computed here: requested now
    if (!SPARSE && wave < S) {
      const double *ap_ = Ac + ((int64_t)(S - 1 - wave) * lda) * 16 + l16 + (int64_t)kq * 16;
#pragma unroll
      for (int u = 0; u < KU; ++u) a0[u] = ap_[u * 64];
    }
    int32_t colv = 0, lrv = 0;
    double valv = 0.0;
    if (l16 < ne) {
      colv = mid_col[e0 + l16];
      valv = mid_val[e0 + l16];
      lrv = mid_lrow[e0 + l16];
    }
    HIFAMD_CSP(1)
    __syncthreads();
    HIFAMD_CSP(2)
    if (dbg & 16) return;  // (timing experiments: descriptor + row ids only)
    // ---- phase 1a: right-hand sides of this lane group's rows into LDS, eight rows in flight
    const int nrmax = wave_max4(nr);
    for (int j = 0; j < nrmax; j += 8) {
      double t_[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = r0 + min(j + q, max(nr - 1, 0));
        const int32_t rid = s_rowid[r];
        const int32_t i = first_l ? s_hp[r] : (rid < 0 ? ~rid : rid);
        const double *src = rhs + (int64_t)i * rstride + rcol;
        if (from_b) {
          const int32_t pb = s_hp[r];
          if (pb >= 0) src = bsrc + (int64_t)pb * fl.ldb + min(cc, fl.nrhs - 1);
        }
        t_[q] = (j + q < nr) ? *src : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (j + q < nr) {
          const double hd = s_hd[r0 + j + q];
          double tq = t_[q];
          if (from_b && s_hp[r0 + j + q] >= 0) tq = cc < fl.nrhs ? s_ot[r0 + j + q] * tq : 0.0;
          tb[((r0 + j + q) << 4) + l16] = div_u ? tq / hd : (first_l ? (cc < fl.nrhs ? hd * tq : 0.0) : tq);
        }
    }
    HIFAMD_CSP(3)
    if (dbg & 32) return;  // (timing experiments: ... + right-hand sides)
    // ---- phase 1b: this lane group's entries, items of 16, eight gathers per batch, the running row's value in a register
    int cur = -1;
    double acc = 0.0;
    const int32_t nemax = (dbg & 1) ? 0 : wave_max4(ne);
    for (int32_t base = 0; base < nemax; base += 16) {
      int32_t colv2 = 0, lrv2 = 0;
      double valv2 = 0.0;
      if (base + 16 + l16 < ne) {
        colv2 = mid_col[e0 + base + 16 + l16];
        valv2 = mid_val[e0 + base + 16 + l16];
        lrv2 = mid_lrow[e0 + base + 16 + l16];
      }
      const int32_t left = ne - base;  // entries of this lane group from `base` on (may be <= 0)
      cs_batch<0>(colv, valv, lrv, left, x, cc, l16, tb, cur, acc);
      if (base + 8 < nemax) cs_batch<8>(colv, valv, lrv, left, x, cc, l16, tb, cur, acc);
      colv = colv2;
      valv = valv2;
      lrv = lrv2;
    }
    if (cur >= 0) tb[(cur << 4) + l16] = acc;
    if (!SPARSE)  // (rows nb .. lda - 1 are zero for the inverse product; lds_rows is a multiple of 32)
      for (int t = nb * 16 + (int)threadIdx.x; t < lda * 16; t += 256) tb[t] = 0.0;
    HIFAMD_CSP(4)
    __syncthreads();
    HIFAMD_CSP(5)
    if (SPARSE) {
      // ---- phase 2, sparse: substitution inside LDS, depth level by depth level; a lane group takes a row at a time
      const int nlvl = dsc[24];
      for (int lv = 0; lv < nlvl; ++lv) {
        const int r_lo = ow_lvl[lv], r_hi = ow_lvl[lv + 1];
        for (int r = r_lo + gq; r < r_hi; r += 16) {
          double a2 = tb[(r << 4) + l16];
          const int eb = ow_rptr[r], ee = ow_rptr[r + 1];
          for (int e = eb; e < ee; ++e) a2 = a2 - ow_val[e] * tb[((int)ow_src[e] << 4) + l16];
          if (ee > eb) tb[(r << 4) + l16] = a2;
          if (last_u) {
            if (cc < lu.nrhs) yout[(int64_t)s_oi[r] * lu.ldy + cc] = s_ot[r] * a2;
          } else {
            const int32_t rid = s_rowid[r];
            if (rid >= 0) x[((int64_t)rid << 6) + cc] = a2;
          }
        }
        __syncthreads();
      }
      continue;
    }
    // ---- phase 2: x = Tinv * t on the matrix cores, one 16-row strip x this slice's 16 columns per step; strips are
    // dealt heaviest first in snake order over the four waves
    for (int rnd = 0;; ++rnd) {
      const int q = 4 * rnd + ((rnd & 1) ? 3 - wave : wave);
      if (q >= ((dbg & 2) ? 0 : S)) {
        if (4 * rnd >= S) break;
        continue;
      }
      const int strip = S - 1 - q;
      const int kend = min(nb, 16 * (strip + 1));
      const int nsets = (kend + 31) >> 5;
      const double *Ap = Ac + ((int64_t)strip * lda) * 16 + l16;
      const double *Bp = tb + l16;
      v4f64 acc0 = v4f64{0.0, 0.0, 0.0, 0.0};
#define HIFAMD_CS_LOAD(aa, t_)                                          \
  {                                                                     \
    const double *ap_ = Ap + (int64_t)(32 * (t_) + kq) * 16;            \
    _Pragma("unroll") for (int u = 0; u < KU; ++u) aa[u] = ap_[u * 64]; \
  }
#define HIFAMD_CS_MFMA(aa, t_)                                                \
  {                                                                           \
    const int kb_ = 32 * (t_) + kq;                                           \
    _Pragma("unroll") for (int u = 0; u < KU; ++u) {                          \
      const double b0_ = Bp[(kb_ + 4 * u) << 4];                              \
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u], b0_, acc0, 0, 0, 0); \
    }                                                                         \
  }
      int t = 0;
      if (rnd != 0) HIFAMD_CS_LOAD(a0, 0)  // (the first strip's first set was requested before phase 1)
      while (t < nsets) {
        if (t + 1 < nsets) HIFAMD_CS_LOAD(a1, t + 1)
        HIFAMD_CS_MFMA(a0, t)
        if (t + 1 >= nsets) break;
        if (t + 2 < nsets) HIFAMD_CS_LOAD(a0, t + 2)
        HIFAMD_CS_MFMA(a1, t + 1)
        t += 2;
      }
#undef HIFAMD_CS_LOAD
#undef HIFAMD_CS_MFMA
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * strip + kq + 4 * r;
        if (row < nb) {
          if (last_u) {
            if (cc < lu.nrhs) yout[(int64_t)s_oi[row] * lu.ldy + cc] = s_ot[row] * acc0[r];
          } else {
            x[((int64_t)s_rowid[row] << 6) + cc] = acc0[r];
          }
        }
      }
    }
    __syncthreads();  // (the next component overwrites the LDS block)
  }
  HIFAMD_CSP(6)
  HIFAMD_CSP_FLUSH(1 + (LOWER ? 0 : 4) + (SPARSE ? 8 : 0), n_band, nsl)
}

// ---------------------------------------------------------------------------------------------
// Component band on coefficient TILES (round 4; host.hpp build_ct_tiles).  Same plan, same launch shape as k_band_cs
// (blockIdx.x = component-workgroup * nsl + slice of 16 columns, four waves, carried-prefix workgroups behind), but the
// entries a component reads from rows finished by earlier launches are not walked one by one: per 16-row strip of the
// component the host has cut the strip's distinct sources into groups of four and written the 16 x 4 coefficient tile of
// every group; a wave owns whole strips (descriptor words 22 / 23) and spends ONE v_mfma_f64_16x16x4 per tile,
//     acc[16 rows][16 columns] += coef[16 x 4] * x[4 gathered source rows][16 columns],
// -- a dozen entries per matrix instruction on the reference's 1M-row hierarchies instead of a dozen instructions per
// entry -- with the tile's four source rows gathered once (a lane loads the 8 bytes of ITS source row k = lane / 16,
// column lane % 16: the B operand needs no shuffle, no LDS).  The wave then forms t = rhs - acc for its strips (S1 /
// the pivot division fused as in k_band_cd) straight into LDS; phase 2, the product with the component's explicit
// inverse, is k_band_cs's.  Column-separable like every kernel here: a column's bits do not depend on the batch width,
// so this kernel serves narrow batches (nsl < 4) and full ones alike.  Summation order: tile by tile (tolerance-level).
// ---------------------------------------------------------------------------------------------
template <bool LOWER, int NCT>
__global__ void __launch_bounds__(256) k_band_ct(int32_t wg0, const int32_t *__restrict__ wg_grp_ptr,
                                                 const int32_t *__restrict__ ct_desc, const int32_t *__restrict__ ptr,
                                                 const int32_t *__restrict__ split, const int32_t *__restrict__ col,
                                                 const double *__restrict__ val, const int32_t *__restrict__ rowid,
                                                 const double *__restrict__ d, double *w, double *v,
                                                 const double *__restrict__ tinv, const int32_t *__restrict__ ct_sptr,
                                                 const int32_t *__restrict__ ct_src, const double *__restrict__ ct_coef,
                                                 int first_u, int32_t n_band, int32_t nsl, int32_t ps0, int32_t ps1,
                                                 int32_t single_c0, int32_t lds_rows, int dbg, FirstL<double> fl,
                                                 LastU<double> lu) {
  extern __shared__ double cs_buf[];
  HIFAMD_CSP_DECL
  HIFAMD_CSP(0)
  constexpr int W = 16 * NCT;           // columns of this workgroup's slice
  constexpr int BU = NCT == 1 ? 8 : (NCT == 2 ? 4 : 2);  // tiles per batch (one batch in flight ahead of the one being multiplied)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // blocks [0, nbp): component-workgroup x slice, laid out so that the slices of a component sit 8 blocks apart (blocks
  // are dealt round-robin over the eight XCDs: the slices then share one L2 -- the component's tiles and inverse are
  // fetched from HBM once; placement is a speed matter only)
  const int32_t nbp = ((n_band + 7) >> 3) * 8 * nsl;
  // (dbg & 128: the carried workgroups take the FIRST block numbers instead of the last)
  const int32_t ncar = (int32_t)gridDim.x - nbp;
  const bool car_first = (dbg & 128) != 0;
  const int32_t bx = car_first ? (int32_t)blockIdx.x - ncar : (int32_t)blockIdx.x;
  if (car_first ? bx < 0 : bx >= nbp) {  // carried prefix of the next band over the sources older than this band
    const int32_t pw = __builtin_amdgcn_readfirstlane((car_first ? (int32_t)blockIdx.x : bx - nbp) * 4 + wave);
    if (dbg & 4) return;
    const double *pf_bin = (LOWER && fl.on()) ? fl.bin.get() : nullptr;
    trsv_stream_r64<double, 0, LOWER, true>(ps0 + pw, ps1, ncar * 4, ptr, split, col, val, nullptr,
                                            rowid, d, LOWER ? w : v, w, lane, nullptr, 0, nullptr, true, nullptr, 0, 0, pf_bin, &fl);
    HIFAMD_CSP(6)
    HIFAMD_CSP_FLUSH(2 + (LOWER ? 0 : 4) + 32, n_band, nsl)
    return;
  }
  const int32_t bgrp = bx / (8 * nsl), brem = bx - bgrp * 8 * nsl;
  const int32_t slice = brem >> 3, bw = bgrp * 8 + (brem & 7);
  if (bw >= n_band) return;
  const int kq = lane >> 4, l16 = lane & 15;
  const int cc = slice * W + l16;  // this lane's column of the 64-column arena (first column tile)
  double *x = LOWER ? w : v;
  const bool div_u = !LOWER && first_u;
  const bool first_l = LOWER && first_u && fl.on();
  const double *rhs = div_u ? (const double *)w : (first_l ? fl.bin.get() : (const double *)x);
  const int64_t rstride = first_l ? fl.ldb : 64;
  int32_t c_first, c_last;
  if (single_c0 >= 0) {
    c_first = single_c0 + bw;
    c_last = c_first + 1;
  } else {
    c_first = wg_grp_ptr[wg0 + bw];
    c_last = wg_grp_ptr[wg0 + bw + 1];
  }
  // LDS: right-hand sides [lds_rows][W], per row: pivot / scale, output scale, row id, input row, output row; the
  // component's strip -> tile offsets
  double *tb = cs_buf;
  double *s_hd = tb + (size_t)lds_rows * W;
  double *s_ot = s_hd + lds_rows;
  int32_t *s_rowid = reinterpret_cast<int32_t *>(s_ot + lds_rows);
  int32_t *s_hp = s_rowid + lds_rows;
  int32_t *s_oi = s_hp + lds_rows;
  int32_t *s_sptr = s_oi + lds_rows;  // 17 entries (a component has at most 16 strips)
  const bool last_u = !LOWER && lu.on();
  double *yout = last_u ? lu.out.get() : nullptr;
  constexpr int KU = 8;
  for (int32_t c = c_first; c < c_last; ++c) {
    const int32_t *dsc = ct_desc + (int64_t)c * 28;
    const int32_t s0 = dsc[0], nb = dsc[1], sp0 = dsc[20];
    const int64_t inv_off = ((int64_t)(uint32_t)dsc[5] << 32) | (uint32_t)dsc[4];
    const uint32_t mword = (uint32_t)dsc[22 + (wave >> 1)];
    const uint32_t mymask = (wave & 1) ? (mword >> 16) : (mword & 0xffffu);  // the strips this wave owns
    const int lda = (nb + 31) & ~31;
    const double *Ac = tinv + inv_off;
    const int S = (nb + 15) >> 4;
    double a0[KU], a1[KU];
    // ---- phase 0: the component's row ids, per-row scalars and strip offsets (coalesced)
    for (int32_t t = (int32_t)threadIdx.x; t < nb; t += 256) {
      const int32_t i = rowid[s0 + t];
      s_rowid[t] = i;
      if (div_u) s_hd[t] = d[i];
      if (first_l) {
        const int32_t pp = fl.p[i];
        s_hp[t] = pp;
        s_hd[t] = fl.s[pp];
      }
      if (last_u) {
        const int32_t oi = lu.q[i];
        s_oi[t] = oi;
        s_ot[t] = lu.t[oi];
      }
    }
    if ((int32_t)threadIdx.x <= S) s_sptr[threadIdx.x] = ct_sptr[sp0 + (int32_t)threadIdx.x];
    // phase 2's first operand set does not depend on anything computed here: a narrow band (one 16-column slice per
    // workgroup) requests it NOW -- its launch is a chain of round trips, and this one then hides behind phase 1
    if (NCT == 1 && !(dbg & 256) && wave < S) {
      const double *ap_ = Ac + ((int64_t)(S - 1 - wave) * lda) * 16 + l16 + (int64_t)kq * 16;
#pragma unroll
      for (int u = 0; u < KU; ++u) a0[u] = ap_[u * 64];
    }
    HIFAMD_CSP(1)
    __syncthreads();
    HIFAMD_CSP(2)
    // ---- phase 1: this wave's strips -- right-hand sides requested first, then the strip's tiles BU at a time (the
    // next BU source ids and coefficient tiles are in flight while the current ones are gathered and multiplied)
    for (int s = 0; s < S; ++s) {
      if (!((mymask >> s) & 1u)) continue;  // (wave-uniform)
      double tr[NCT][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = min(16 * s + kq + 4 * j, nb - 1);
        const int32_t i = first_l ? s_hp[row] : s_rowid[row];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int cx = cc + 16 * ct;
          tr[ct][j] = rhs[(int64_t)i * rstride + (first_l ? min(cx, fl.nrhs - 1) : cx)];
        }
      }
      const int32_t t0 = s_sptr[s], t1 = (dbg & 1) ? t0 : s_sptr[s + 1];
      v4f64 acc[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[ct] = v4f64{0.0, 0.0, 0.0, 0.0};
      if (t0 < t1) {
        int32_t sv[BU];
        double cv[BU];
#pragma unroll
        for (int u = 0; u < BU; ++u) {
          const int32_t tt = min(t0 + u, t1 - 1);
          sv[u] = ct_src[4 * (int64_t)tt + kq];
          cv[u] = ct_coef[64 * (int64_t)tt + lane];
          if (t0 + u >= t1) cv[u] = 0.0;
        }
        for (int32_t t = t0; t < t1; t += BU) {
          double bv[NCT][BU];
#pragma unroll
          for (int u = 0; u < BU; ++u)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) bv[ct][u] = x[((int64_t)sv[u] << 6) + cc + 16 * ct];
          int32_t sn[BU];
          double cn[BU];
#pragma unroll
          for (int u = 0; u < BU; ++u) {  // (clamped: the loads past the strip's last tile read that tile again, weight zero)
            const int32_t tt = min(t + BU + u, t1 - 1);
            sn[u] = ct_src[4 * (int64_t)tt + kq];
            cn[u] = ct_coef[64 * (int64_t)tt + lane];
            if (t + BU + u >= t1) cn[u] = 0.0;
          }
#pragma unroll
          for (int u = 0; u < BU; ++u)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv[u], bv[ct][u], acc[ct], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < BU; ++u) sv[u] = sn[u], cv[u] = cn[u];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 16 * s + kq + 4 * j;
        if (row < nb) {
          const double hd = (div_u || first_l) ? s_hd[row] : 1.0;
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const double t_ = div_u ? tr[ct][j] / hd : (first_l ? (cc + 16 * ct < fl.nrhs ? hd * tr[ct][j] : 0.0) : tr[ct][j]);
            tb[row * W + 16 * ct + l16] = t_ - acc[ct][j];
          }
        }
      }
    }
    // (a wide band: requested behind phase 1, before the barrier -- there the registers buy more than the latency)
    if ((NCT != 1 || (dbg & 256)) && wave < S) {
      const double *ap_ = Ac + ((int64_t)(S - 1 - wave) * lda) * 16 + l16 + (int64_t)kq * 16;
#pragma unroll
      for (int u = 0; u < KU; ++u) a0[u] = ap_[u * 64];
    }
    // (rows nb .. lda - 1 are zero for the inverse product; lds_rows is a multiple of 32)
    for (int t = nb * W + (int)threadIdx.x; t < lda * W; t += 256) tb[t] = 0.0;
    HIFAMD_CSP(4)
    __syncthreads();
    HIFAMD_CSP(5)
    // ---- phase 2: x = Tinv * t on the matrix cores, one 16-row strip x this slice's columns per step; strips are dealt
    // heaviest first in snake order over the four waves (k_band_cs)
    for (int rnd = 0;; ++rnd) {
      const int q = 4 * rnd + ((rnd & 1) ? 3 - wave : wave);
      if (q >= ((dbg & 2) ? 0 : S)) {
        if (4 * rnd >= S) break;
        continue;
      }
      const int strip = S - 1 - q;
      const int kend = min(nb, 16 * (strip + 1));
      const int nsets = (kend + 31) >> 5;
      const double *Ap = Ac + ((int64_t)strip * lda) * 16 + l16;
      const double *Bp = tb + l16;
      v4f64 acc0[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc0[ct] = v4f64{0.0, 0.0, 0.0, 0.0};
#define HIFAMD_CT_LOAD(aa, t_)                                          \
  {                                                                     \
    const double *ap_ = Ap + (int64_t)(32 * (t_) + kq) * 16;            \
    _Pragma("unroll") for (int u = 0; u < KU; ++u) aa[u] = ap_[u * 64]; \
  }
#define HIFAMD_CT_MFMA(aa, t_)                                                              \
  {                                                                                         \
    const int kb_ = 32 * (t_) + kq;                                                         \
    _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                        \
      _Pragma("unroll") for (int ct = 0; ct < NCT; ++ct) {                                  \
        const double b0_ = Bp[(kb_ + 4 * u) * W + 16 * ct];                                 \
        acc0[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[u], b0_, acc0[ct], 0, 0, 0);     \
      }                                                                                     \
    }                                                                                       \
  }
      int t = 0;
      if (rnd != 0) HIFAMD_CT_LOAD(a0, 0)  // (the first strip's first set was requested before the barrier)
      while (t < nsets) {
        if (t + 1 < nsets) HIFAMD_CT_LOAD(a1, t + 1)
        HIFAMD_CT_MFMA(a0, t)
        if (t + 1 >= nsets) break;
        if (t + 2 < nsets) HIFAMD_CT_LOAD(a0, t + 2)
        HIFAMD_CT_MFMA(a1, t + 1)
        t += 2;
      }
#undef HIFAMD_CT_LOAD
#undef HIFAMD_CT_MFMA
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * strip + kq + 4 * r;
        if (row < nb) {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const int cx = cc + 16 * ct;
            if (last_u) {
              if (cx < lu.nrhs) yout[(int64_t)s_oi[row] * lu.ldy + cx] = s_ot[row] * acc0[ct][r];
            } else {
              x[((int64_t)s_rowid[row] << 6) + cx] = acc0[ct][r];
            }
          }
        }
      }
    }
    __syncthreads();  // (the next component overwrites the LDS block)
  }
  HIFAMD_CSP(6)
  HIFAMD_CSP_FLUSH(1 + (LOWER ? 0 : 4) + 32, n_band, nsl)
}

// ---------------------------------------------------------------------------------------------
// Column-sliced component band for COMPLEX data (round 3): k_band_cd_z for ONE slice of 16 complex columns per
// workgroup (blockIdx.x = component-workgroup * nsl + slice), the complex twin of k_band_cs.  A batch of at most 48
// columns launches only the slices it has: BASELINE config 5 (nrhs = 16) moves a quarter of the vector bytes and
// issues a quarter of the gathers of the 64-column kernel.  4 waves; lane group q = 4 wave + g (16 lanes = 16 complex
// columns) takes chunk q of the descriptor: its rows' right-hand sides go into two real LDS planes [rows][16], its
// entries are fetched 16 at a time per lane group and broadcast with DPP row_newbcast, four 256-byte gathers per lane
// group in flight; the inverse product is four real MFMA streams per 16-row strip (x_re = P_re t_re - P_im t_im,
// x_im = P_re t_im + P_im t_re), strips dealt heaviest first in snake order.  Per row and column the arithmetic and
// its order are k_band_cd_z's: the same bits.  No carried prefixes, no sparse-own variant (complex plans have neither).
// ---------------------------------------------------------------------------------------------
// SPARSE (round 4): sparse-own components of a complex triangle (level 0 of a PDE hierarchy: ~2 nonzeros per row) -- the
// component's own nonzeros (two real LDS arrays), row offsets and depth levels are staged in LDS and the component is
// solved level by level, a lane group per row, exactly as k_band_cs<*, true> does it for real data; no inverse exists.
template <bool LOWER, bool SPARSE>
__global__ void __launch_bounds__(256) k_band_cs_z(int32_t wg0, const int32_t *__restrict__ wg_grp_ptr,
                                                   const int32_t *__restrict__ cd_desc, const int32_t *__restrict__ rowid,
                                                   const cplx *__restrict__ d, cplx *w, cplx *v,
                                                   const double *__restrict__ tinv, const int32_t *__restrict__ mid_col,
                                                   const cplx *__restrict__ mid_val, const uint8_t *__restrict__ mid_lrow,
                                                   int first_u, int32_t nsl, int32_t lds_rows, FirstL<cplx> fl,
                                                   int32_t own_cap, const cplx *__restrict__ own_val,
                                                   const uint8_t *__restrict__ own_lsrc, const uint16_t *__restrict__ own_rptr,
                                                   const uint8_t *__restrict__ own_lvl, LastU<cplx> lu) {
  extern __shared__ double cs_buf[];
  double *t_re = cs_buf, *t_im = cs_buf + (size_t)lds_rows * 16;
  double *s_hdx = t_im + (size_t)lds_rows * 16, *s_hdy = s_hdx + lds_rows;
  int32_t *s_rowid = reinterpret_cast<int32_t *>(s_hdy + lds_rows);
  int32_t *s_hp = s_rowid + lds_rows;
  // SPARSE: own nonzeros (re, im), their local sources, row offsets, depth levels
  double *ow_re = reinterpret_cast<double *>(s_hp + ((lds_rows + 1) & ~1));
  double *ow_im = ow_re + (SPARSE ? own_cap : 0);
  uint16_t *ow_rptr = reinterpret_cast<uint16_t *>(ow_im + (SPARSE ? own_cap : 0));
  uint8_t *ow_src = reinterpret_cast<uint8_t *>(ow_rptr + 260);
  uint8_t *ow_lvl = ow_src + (SPARSE ? own_cap : 0);
  // fused S7 (LastU, round 4 for complex data): output row and scale of every row, behind everything else
  double *s_ot = reinterpret_cast<double *>(ow_lvl + (SPARSE ? 264 : 0) + 8);
  s_ot = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(s_ot) + 7) & ~(uintptr_t)7);
  int32_t *s_oi = reinterpret_cast<int32_t *>(s_ot + lds_rows);
  const bool last_u = !LOWER && lu.on();
  cplx *yout = last_u ? lu.out.get() : nullptr;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int32_t bw = (int32_t)blockIdx.x / nsl, slice = (int32_t)blockIdx.x - bw * nsl;
  const int grp = lane >> 4, l16 = lane & 15, gq = wave * 4 + grp;
  const int cc = slice * 16 + l16;
  cplx *x = LOWER ? w : v;
  const bool div_u = !LOWER && first_u;
  const bool first_l = LOWER && first_u && fl.on();
  const cplx *rhs = div_u ? (const cplx *)w : (first_l ? fl.bin.get() : (const cplx *)x);
  const int64_t rstride = first_l ? fl.ldb : 64;
  const int rcol = first_l ? min(cc, fl.nrhs - 1) : cc;
  const int32_t c_first = wg_grp_ptr[wg0 + bw], c_last = wg_grp_ptr[wg0 + bw + 1];
  const int kq = grp;
  for (int32_t c = c_first; c < c_last; ++c) {
    const int32_t *dsc = cd_desc + (int64_t)c * 28;
    const int32_t s0 = dsc[0], nb = dsc[1], mid0 = dsc[2];
    const int64_t inv_off = ((int64_t)(uint32_t)dsc[5] << 32) | (uint32_t)dsc[4];
    const uint8_t *wrow = reinterpret_cast<const uint8_t *>(dsc + 6);
    const uint16_t *wmid = reinterpret_cast<const uint16_t *>(dsc + 11);
    const int r0 = wrow[gq], nr = (int)wrow[gq + 1] - r0;
    const int32_t e0 = mid0 + (int32_t)wmid[gq], ne = (int32_t)wmid[gq + 1] - (int32_t)wmid[gq];
    const int lda = (nb + 31) & ~31;
    // ---- phase 0: row ids and per-row scalars
    for (int32_t t = (int32_t)threadIdx.x; t < nb; t += 256) {
      const int32_t i = rowid[s0 + t];
      s_rowid[t] = i;
      if (div_u) {
        const cplx dd = d[i];
        s_hdx[t] = dd.x, s_hdy[t] = dd.y;
      }
      if (first_l) {
        const int32_t pp = fl.p[i];
        s_hp[t] = pp;
        s_hdx[t] = fl.s[pp];
      }
      if (last_u) {
        const int32_t oi = lu.q[i];
        s_oi[t] = oi;
        s_ot[t] = lu.t[oi];
      }
    }
    if (SPARSE) {
      const int32_t own0 = dsc[20], nown = dsc[21], orp0 = dsc[22], lvl0 = dsc[23], nlvl = dsc[24];
      for (int32_t t = (int32_t)threadIdx.x; t < nown; t += 256) {
        const cplx ov = own_val[own0 + t];
        ow_re[t] = ov.x, ow_im[t] = ov.y;
        ow_src[t] = own_lsrc[own0 + t];
      }
      for (int32_t t = (int32_t)threadIdx.x; t <= nb; t += 256) ow_rptr[t] = own_rptr[orp0 + t];
      for (int32_t t = (int32_t)threadIdx.x; t <= nlvl; t += 256) ow_lvl[t] = own_lvl[lvl0 + t];
    }
    int32_t colv = 0, lrv = 0;
    cplx valv = cplx{0.0, 0.0};
    if (l16 < ne) {
      colv = mid_col[e0 + l16];
      valv = mid_val[e0 + l16];
      lrv = mid_lrow[e0 + l16];
    }
    __syncthreads();
    // ---- phase 1a: right-hand sides of this lane group's rows into the two LDS planes, four rows in flight
    const int nrmax = wave_max4(nr);
    for (int j = 0; j < nrmax; j += 4) {
      cplx t_[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r0 + min(j + q, max(nr - 1, 0));
        const int32_t i = first_l ? s_hp[r] : s_rowid[r];
        t_[q] = (j + q < nr) ? rhs[(int64_t)i * rstride + rcol] : cplx{0.0, 0.0};
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j + q < nr) {
          const int r = r0 + j + q;
          cplx val = t_[q];
          if (div_u)
            val = vdiv(val, cplx{s_hdx[r], s_hdy[r]});
          else if (first_l)
            val = cc < fl.nrhs ? vscale(s_hdx[r], val) : cplx{0.0, 0.0};
          t_re[(r << 4) + l16] = val.x;
          t_im[(r << 4) + l16] = val.y;
        }
    }
    // ---- phase 1b: this lane group's entries, items of 16, four gathers per batch
    int cur = -1;
    cplx acc = cplx{0.0, 0.0};
    const int32_t nemax = wave_max4(ne);
    for (int32_t base = 0; base < nemax; base += 16) {
      int32_t colv2 = 0, lrv2 = 0;
      cplx valv2 = cplx{0.0, 0.0};
      if (base + 16 + l16 < ne) {
        colv2 = mid_col[e0 + base + 16 + l16];
        valv2 = mid_val[e0 + base + 16 + l16];
        lrv2 = mid_lrow[e0 + base + 16 + l16];
      }
      const int32_t left = ne - base;
      static_for<0, 4>([&](auto ib) {
        constexpr int T0 = 4 * decltype(ib)::value;
        if (T0 < nemax - base) {
          int32_t j_[4], r_[4];
          cplx a_[4], xv_[4];
          static_for<0, 4>([&](auto iu) {
            constexpr int u = decltype(iu)::value;
            j_[u] = bc16<T0 + u>(colv);
            a_[u] = cplx{bc16<T0 + u>(valv.x), bc16<T0 + u>(valv.y)};
            r_[u] = bc16<T0 + u>(lrv);
          });
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (T0 + u < left) xv_[u] = x[((int64_t)j_[u] << 6) + cc];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (T0 + u < left) {
              if (r_[u] != cur) {
                if (cur >= 0) {
                  t_re[(cur << 4) + l16] = acc.x;
                  t_im[(cur << 4) + l16] = acc.y;
                }
                cur = r_[u];
                acc = cplx{t_re[(cur << 4) + l16], t_im[(cur << 4) + l16]};
              }
              acc = vsub(acc, vmul(a_[u], xv_[u]));
            }
        }
      });
      colv = colv2;
      valv = valv2;
      lrv = lrv2;
    }
    if (cur >= 0) {
      t_re[(cur << 4) + l16] = acc.x;
      t_im[(cur << 4) + l16] = acc.y;
    }
    if (SPARSE) {
      __syncthreads();
      // ---- phase 2, sparse: substitution inside LDS, depth level by depth level; a lane group takes a row at a time
      const int nlvl = dsc[24];
      for (int lv = 0; lv < nlvl; ++lv) {
        const int r_lo = ow_lvl[lv], r_hi = ow_lvl[lv + 1];
        for (int r = r_lo + gq; r < r_hi; r += 16) {
          cplx a2 = cplx{t_re[(r << 4) + l16], t_im[(r << 4) + l16]};
          const int eb = ow_rptr[r], ee = ow_rptr[r + 1];
          for (int e = eb; e < ee; ++e) {
            const int sr = (int)ow_src[e];
            a2 = vsub(a2, vmul(cplx{ow_re[e], ow_im[e]}, cplx{t_re[(sr << 4) + l16], t_im[(sr << 4) + l16]}));
          }
          if (ee > eb) t_re[(r << 4) + l16] = a2.x, t_im[(r << 4) + l16] = a2.y;
          if (last_u) {
            if (cc < lu.nrhs) yout[(int64_t)s_oi[r] * lu.ldy + cc] = vscale(s_ot[r], a2);
          } else {
            x[((int64_t)s_rowid[r] << 6) + cc] = a2;
          }
        }
        __syncthreads();
      }
      continue;
    }
    // (rows nb .. lda - 1 are zero for the inverse product; lds_rows is a multiple of 32)
    for (int t = nb * 16 + (int)threadIdx.x; t < lda * 16; t += 256) t_re[t] = 0.0, t_im[t] = 0.0;
    __syncthreads();
    // ---- phase 2: x = Tinv * t on the real matrix cores, four products per strip
    const int S = (nb + 15) >> 4;
    const double *Are = tinv + inv_off, *Aim = Are + ((int64_t)((nb + 15) & ~15) * lda);  // plane_elems(nb, lda)
    for (int rnd = 0; 4 * rnd < S; ++rnd) {
      const int q = 4 * rnd + ((rnd & 1) ? 3 - wave : wave);
      if (q >= S) continue;
      const int strip = S - 1 - q;
      const int kend = min(nb, 16 * (strip + 1));
      const int nk = (kend + 3) >> 2;  // k-steps of four columns
      const double *Apr = Are + ((int64_t)strip * lda) * 16 + l16 + (int64_t)kq * 16;
      const double *Api = Aim + ((int64_t)strip * lda) * 16 + l16 + (int64_t)kq * 16;
      const double *Bre = t_re + l16, *Bim = t_im + l16;
      v4f64 a_rr = v4f64{0.0, 0.0, 0.0, 0.0}, a_ii = a_rr, a_ri = a_rr, a_ir = a_rr;
      constexpr int KU = 4;
      double pr0[KU], pi0[KU], pr1[KU], pi1[KU];
#define HIFAMD_CSZ_LOAD(pr, pi, t_)                                                       \
  _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                        \
    pr[u] = Apr[(int64_t)(KU * (t_) + u) * 64];                                           \
    pi[u] = Api[(int64_t)(KU * (t_) + u) * 64];                                           \
  }
#define HIFAMD_CSZ_MFMA(pr, pi, t_)                                                       \
  _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                        \
    const int kb_ = 4 * (KU * (t_) + u) + kq;                                             \
    const double br_ = Bre[kb_ << 4], bi_ = Bim[kb_ << 4];                                \
    a_rr = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[u], br_, a_rr, 0, 0, 0);               \
    a_ii = __builtin_amdgcn_mfma_f64_16x16x4f64(pi[u], bi_, a_ii, 0, 0, 0);               \
    a_ri = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[u], bi_, a_ri, 0, 0, 0);               \
    a_ir = __builtin_amdgcn_mfma_f64_16x16x4f64(pi[u], br_, a_ir, 0, 0, 0);               \
  }
      // (sets of KU = 4 k-steps = 16 columns: the operand planes and the LDS rows are zero padded up to lda, a multiple
      //  of 32, so a set never leaves the strip; k-steps past nk add exact zeros)
      const int nsets = (nk + KU - 1) / KU;
      int t = 0;
      HIFAMD_CSZ_LOAD(pr0, pi0, 0)
      while (t < nsets) {
        if (t + 1 < nsets) HIFAMD_CSZ_LOAD(pr1, pi1, t + 1)
        HIFAMD_CSZ_MFMA(pr0, pi0, t)
        if (t + 1 >= nsets) break;
        if (t + 2 < nsets) HIFAMD_CSZ_LOAD(pr0, pi0, t + 2)
        HIFAMD_CSZ_MFMA(pr1, pi1, t + 1)
        t += 2;
      }
#undef HIFAMD_CSZ_LOAD
#undef HIFAMD_CSZ_MFMA
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * strip + kq + 4 * r;
        if (row < nb) {
          const cplx res = cplx{a_rr[r] - a_ii[r], a_ri[r] + a_ir[r]};
          if (last_u) {
            if (cc < lu.nrhs) yout[(int64_t)s_oi[row] * lu.ldy + cc] = vscale(s_ot[row], res);
          } else {
            x[((int64_t)s_rowid[row] << 6) + cc] = res;
          }
        }
      }
    }
    __syncthreads();  // (the next component overwrites the LDS planes)
  }
}

// ---------------------------------------------------------------------------------------------
// Component band on coefficient tiles for COMPLEX data (round 4): k_band_ct's phase 1 (a wave owns whole strips; per tile
// the four source rows are gathered once -- 16 bytes (re, im) per lane -- and multiplied with the tile's two real
// coefficient planes by four real matrix instructions, rr / ii / ri / ir) in front of k_band_cs_z's phase 2 (the
// component's explicit inverse as two real planes).  One workgroup per (component, slice of 16 complex columns) at every
// batch width; column-separable: a column's bits do not depend on the batch it travels in.  LastU (fused S7) as in
// k_band_cs_z.
// ---------------------------------------------------------------------------------------------
template <bool LOWER>
__global__ void __launch_bounds__(256) k_band_ct_z(int32_t wg0, const int32_t *__restrict__ wg_grp_ptr,
                                                   const int32_t *__restrict__ ct_desc, const int32_t *__restrict__ rowid,
                                                   const cplx *__restrict__ d, cplx *w, cplx *v,
                                                   const double *__restrict__ tinv, const int32_t *__restrict__ ct_sptr,
                                                   const int32_t *__restrict__ ct_src, const double *__restrict__ ct_coef,
                                                   int first_u, int32_t nsl, int32_t lds_rows, int dbg, FirstL<cplx> fl,
                                                   LastU<cplx> lu) {
  extern __shared__ double cs_buf[];
  double *t_re = cs_buf, *t_im = cs_buf + (size_t)lds_rows * 16;
  double *s_hdx = t_im + (size_t)lds_rows * 16, *s_hdy = s_hdx + lds_rows;
  double *s_ot = s_hdy + lds_rows;
  int32_t *s_rowid = reinterpret_cast<int32_t *>(s_ot + lds_rows);
  int32_t *s_hp = s_rowid + lds_rows;
  int32_t *s_oi = s_hp + lds_rows;
  int32_t *s_sptr = s_oi + lds_rows;  // 17 entries
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int32_t bw = (int32_t)blockIdx.x / nsl, slice = (int32_t)blockIdx.x - bw * nsl;
  const int kq = lane >> 4, l16 = lane & 15;
  const int cc = slice * 16 + l16;
  cplx *x = LOWER ? w : v;
  const bool div_u = !LOWER && first_u;
  const bool first_l = LOWER && first_u && fl.on();
  const bool last_u = !LOWER && lu.on();
  cplx *yout = last_u ? lu.out.get() : nullptr;
  const cplx *rhs = div_u ? (const cplx *)w : (first_l ? fl.bin.get() : (const cplx *)x);
  const int64_t rstride = first_l ? fl.ldb : 64;
  const int rcol = first_l ? min(cc, fl.nrhs - 1) : cc;
  const int32_t c_first = wg_grp_ptr[wg0 + bw], c_last = wg_grp_ptr[wg0 + bw + 1];
  for (int32_t c = c_first; c < c_last; ++c) {
    const int32_t *dsc = ct_desc + (int64_t)c * 28;
    const int32_t s0 = dsc[0], nb = dsc[1], sp0 = dsc[20];
    const int64_t inv_off = ((int64_t)(uint32_t)dsc[5] << 32) | (uint32_t)dsc[4];
    const uint32_t mword = (uint32_t)dsc[22 + (wave >> 1)];
    const uint32_t mymask = (wave & 1) ? (mword >> 16) : (mword & 0xffffu);  // the strips this wave owns
    const int lda = (nb + 31) & ~31;
    const int S = (nb + 15) >> 4;
    // ---- phase 0: row ids, per-row scalars, strip offsets
    for (int32_t t = (int32_t)threadIdx.x; t < nb; t += 256) {
      const int32_t i = rowid[s0 + t];
      s_rowid[t] = i;
      if (div_u) {
        const cplx dd = d[i];
        s_hdx[t] = dd.x, s_hdy[t] = dd.y;
      }
      if (first_l) {
        const int32_t pp = fl.p[i];
        s_hp[t] = pp;
        s_hdx[t] = fl.s[pp];
      }
      if (last_u) {
        const int32_t oi = lu.q[i];
        s_oi[t] = oi;
        s_ot[t] = lu.t[oi];
      }
    }
    if ((int32_t)threadIdx.x <= S) s_sptr[threadIdx.x] = ct_sptr[sp0 + (int32_t)threadIdx.x];
    __syncthreads();
    // ---- phase 1: this wave's strips -- right-hand sides requested first, then the strip's tiles four at a time
    constexpr int BU = 4;
    for (int s = 0; s < S; ++s) {
      if (!((mymask >> s) & 1u)) continue;  // (wave-uniform)
      cplx tr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = min(16 * s + kq + 4 * j, nb - 1);
        const int32_t i = first_l ? s_hp[row] : s_rowid[row];
        tr[j] = rhs[(int64_t)i * rstride + rcol];
      }
      const int32_t t0 = s_sptr[s], t1 = (dbg & 1) ? t0 : s_sptr[s + 1];
      v4f64 a_rr = v4f64{0.0, 0.0, 0.0, 0.0}, a_ii = a_rr, a_ri = a_rr, a_ir = a_rr;
      if (t0 < t1) {
        int32_t sv[BU];
        double cr[BU], ci[BU];
#pragma unroll
        for (int u = 0; u < BU; ++u) {
          const int32_t tt = min(t0 + u, t1 - 1);
          sv[u] = ct_src[4 * (int64_t)tt + kq];
          cr[u] = ct_coef[128 * (int64_t)tt + lane];
          ci[u] = ct_coef[128 * (int64_t)tt + 64 + lane];
          if (t0 + u >= t1) cr[u] = 0.0, ci[u] = 0.0;
        }
        for (int32_t t = t0; t < t1; t += BU) {
          cplx bv[BU];
#pragma unroll
          for (int u = 0; u < BU; ++u) bv[u] = x[((int64_t)sv[u] << 6) + cc];
          int32_t sn[BU];
          double crn[BU], cin[BU];
#pragma unroll
          for (int u = 0; u < BU; ++u) {  // (clamped: the loads past the strip's last tile read that tile again, weight zero)
            const int32_t tt = min(t + BU + u, t1 - 1);
            sn[u] = ct_src[4 * (int64_t)tt + kq];
            crn[u] = ct_coef[128 * (int64_t)tt + lane];
            cin[u] = ct_coef[128 * (int64_t)tt + 64 + lane];
            if (t + BU + u >= t1) crn[u] = 0.0, cin[u] = 0.0;
          }
#pragma unroll
          for (int u = 0; u < BU; ++u) {
            a_rr = __builtin_amdgcn_mfma_f64_16x16x4f64(cr[u], bv[u].x, a_rr, 0, 0, 0);
            a_ii = __builtin_amdgcn_mfma_f64_16x16x4f64(ci[u], bv[u].y, a_ii, 0, 0, 0);
            a_ri = __builtin_amdgcn_mfma_f64_16x16x4f64(cr[u], bv[u].y, a_ri, 0, 0, 0);
            a_ir = __builtin_amdgcn_mfma_f64_16x16x4f64(ci[u], bv[u].x, a_ir, 0, 0, 0);
          }
#pragma unroll
          for (int u = 0; u < BU; ++u) sv[u] = sn[u], cr[u] = crn[u], ci[u] = cin[u];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 16 * s + kq + 4 * j;
        if (row < nb) {
          cplx val = tr[j];
          if (div_u)
            val = vdiv(val, cplx{s_hdx[row], s_hdy[row]});
          else if (first_l)
            val = cc < fl.nrhs ? vscale(s_hdx[row], val) : cplx{0.0, 0.0};
          t_re[(row << 4) + l16] = val.x - (a_rr[j] - a_ii[j]);
          t_im[(row << 4) + l16] = val.y - (a_ri[j] + a_ir[j]);
        }
      }
    }
    // (rows nb .. lda - 1 are zero for the inverse product; lds_rows is a multiple of 32)
    for (int t = nb * 16 + (int)threadIdx.x; t < lda * 16; t += 256) t_re[t] = 0.0, t_im[t] = 0.0;
    __syncthreads();
    // ---- phase 2: x = Tinv * t on the real matrix cores, four products per strip (k_band_cs_z)
    const double *Are = tinv + inv_off, *Aim = Are + ((int64_t)((nb + 15) & ~15) * lda);  // plane_elems(nb, lda)
    for (int rnd = 0; 4 * rnd < S; ++rnd) {
      const int q = 4 * rnd + ((rnd & 1) ? 3 - wave : wave);
      if (q >= S) continue;
      const int strip = S - 1 - q;
      const int kend = min(nb, 16 * (strip + 1));
      const int nk = (kend + 3) >> 2;  // k-steps of four columns
      const double *Apr = Are + ((int64_t)strip * lda) * 16 + l16 + (int64_t)kq * 16;
      const double *Api = Aim + ((int64_t)strip * lda) * 16 + l16 + (int64_t)kq * 16;
      const double *Bre = t_re + l16, *Bim = t_im + l16;
      v4f64 a_rr = v4f64{0.0, 0.0, 0.0, 0.0}, a_ii = a_rr, a_ri = a_rr, a_ir = a_rr;
      constexpr int KU = 4;
      double pr0[KU], pi0[KU], pr1[KU], pi1[KU];
#define HIFAMD_CTZ_LOAD(pr, pi, t_)                                                       \
  _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                        \
    pr[u] = Apr[(int64_t)(KU * (t_) + u) * 64];                                           \
    pi[u] = Api[(int64_t)(KU * (t_) + u) * 64];                                           \
  }
#define HIFAMD_CTZ_MFMA(pr, pi, t_)                                                       \
  _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                        \
    const int kb_ = 4 * (KU * (t_) + u) + kq;                                             \
    const double br_ = Bre[kb_ << 4], bi_ = Bim[kb_ << 4];                                \
    a_rr = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[u], br_, a_rr, 0, 0, 0);               \
    a_ii = __builtin_amdgcn_mfma_f64_16x16x4f64(pi[u], bi_, a_ii, 0, 0, 0);               \
    a_ri = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[u], bi_, a_ri, 0, 0, 0);               \
    a_ir = __builtin_amdgcn_mfma_f64_16x16x4f64(pi[u], br_, a_ir, 0, 0, 0);               \
  }
      const int nsets = (nk + KU - 1) / KU;
      int t = 0;
      HIFAMD_CTZ_LOAD(pr0, pi0, 0)
      while (t < nsets) {
        if (t + 1 < nsets) HIFAMD_CTZ_LOAD(pr1, pi1, t + 1)
        HIFAMD_CTZ_MFMA(pr0, pi0, t)
        if (t + 1 >= nsets) break;
        if (t + 2 < nsets) HIFAMD_CTZ_LOAD(pr0, pi0, t + 2)
        HIFAMD_CTZ_MFMA(pr1, pi1, t + 1)
        t += 2;
      }
#undef HIFAMD_CTZ_LOAD
#undef HIFAMD_CTZ_MFMA
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * strip + kq + 4 * r;
        if (row < nb) {
          const cplx res = cplx{a_rr[r] - a_ii[r], a_ri[r] + a_ir[r]};
          if (last_u) {
            if (cc < lu.nrhs) yout[(int64_t)s_oi[row] * lu.ldy + cc] = vscale(s_ot[row], res);
          } else {
            x[((int64_t)s_rowid[row] << 6) + cc] = res;
          }
        }
      }
    }
    __syncthreads();  // (the next component overwrites the LDS planes)
  }
}

// ---------------------------------------------------------------------------------------------
// Component-dense band for COMPLEX data (gfx950 has no complex MFMA): the same scheme as k_band_cd<LOWER, false> --
// a dependency component per workgroup, LDS-resident, x_c = Tinv_c (rhs_c - older-source sums) -- with the component's
// right-hand sides kept as TWO real planes in LDS (re[rows][64], im[rows][64]) and the explicit inverse as two real
// planes in HBM (host.hpp build_dense_block: P_re, then P_im), so that the product is four real MFMA streams per
// output tile: x_re = P_re t_re - P_im t_im, x_im = P_re t_im + P_im t_re.  Rows are 1 KB (64 complex columns):
// cd_rows is capped at 96 for complex handles (96 KB of LDS).  No carried prefixes (complex plans do not fuse bands),
// no sparse-own variant, no staged top.  Units are (strip, 16-column tile): 4 per strip, heaviest strip first.
// ---------------------------------------------------------------------------------------------
template <bool LOWER>
__global__ void __launch_bounds__(1024) k_band_cd_z(int32_t wg0, const int32_t *__restrict__ wg_grp_ptr,
                                                    const int32_t *__restrict__ cd_desc, const int32_t *__restrict__ rowid,
                                                    const cplx *__restrict__ d, cplx *w, cplx *v,
                                                    const double *__restrict__ tinv, const int32_t *__restrict__ mid_col,
                                                    const cplx *__restrict__ mid_val, const uint8_t *__restrict__ mid_lrow,
                                                    int first_u, int32_t lds_rows, FirstL<cplx> fl) {
  extern __shared__ double cd_tbuf[];  // re[lds_rows][64], im[lds_rows][64], then lds_rows row ids
  double *t_re = cd_tbuf, *t_im = cd_tbuf + (size_t)lds_rows * 64;
  int32_t *cd_rowid = reinterpret_cast<int32_t *>(cd_tbuf + (size_t)lds_rows * 128);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  cplx *x = LOWER ? w : v;
  const bool div_u = !LOWER && first_u;
  const bool first_l = LOWER && first_u && fl.on();
  const cplx *rhs = div_u ? (const cplx *)w : (first_l ? fl.bin.get() : (const cplx *)x);
  const int64_t rstride = first_l ? fl.ldb : 64;
  const int rlane = first_l ? min(lane, fl.nrhs - 1) : lane;
  const int g = wg0 + (int)blockIdx.x;
  const int32_t c_first = wg_grp_ptr[g], c_last = wg_grp_ptr[g + 1];
  const int kq = lane >> 4;
  for (int32_t c = c_first; c < c_last; ++c) {
    const int32_t *dsc = cd_desc + (int64_t)c * 28;
    const int32_t s0 = dsc[0], nb = dsc[1], mid0 = dsc[2];
    const int64_t inv_off = ((int64_t)(uint32_t)dsc[5] << 32) | (uint32_t)dsc[4];
    const uint8_t *wrow = reinterpret_cast<const uint8_t *>(dsc + 6);
    const uint16_t *wmid = reinterpret_cast<const uint16_t *>(dsc + 11);
    const int r0 = wrow[wave], nr = (int)wrow[wave + 1] - r0;
    const int32_t e0 = mid0 + (int32_t)wmid[wave], e1 = mid0 + (int32_t)wmid[wave + 1];
    // ---- phase 1a: right-hand sides of this wave's rows into the two LDS planes
    int32_t h_i = 0, h_p = 0;
    cplx h_d = cplx{1.0, 0.0};  // (U: the pivot; fused S1: the row's real scale in .x)
    if (lane < nr) {
      h_i = rowid[s0 + r0 + lane];
      cd_rowid[r0 + lane] = h_i;
      if (div_u) h_d = d[h_i];
      if (first_l) {
        h_p = fl.p[h_i];
        h_d = cplx{fl.s[h_p], 0.0};
      }
    }
    int32_t colv = 0, lrv = 0;
    cplx valv = cplx{0.0, 0.0};
    if (e0 + lane < e1) {  // first item of the wave's entry stream
      colv = mid_col[e0 + lane];
      valv = mid_val[e0 + lane];
      lrv = mid_lrow[e0 + lane];
    }
    for (int j = 0; j < nr; j += 4) {
      cplx t_[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int32_t i = rl32(first_l ? h_p : h_i, min(j + q, 63));
        t_[q] = (j + q < nr) ? rhs[(int64_t)i * rstride + rlane] : cplx{0.0, 0.0};
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j + q < nr) {
          const cplx hd = cplx{rl64(h_d.x, min(j + q, 63)), rl64(h_d.y, min(j + q, 63))};
          cplx val = t_[q];
          if (div_u)
            val = vdiv(val, hd);
          else if (first_l)
            val = lane < fl.nrhs ? vscale(hd.x, val) : cplx{0.0, 0.0};
          t_re[((r0 + j + q) << 6) + lane] = val.x;
          t_im[((r0 + j + q) << 6) + lane] = val.y;
        }
    }
    // ---- phase 1b: the wave's entries (sources finished by earlier launches), items of 64, four gathers per batch
    int cur_r = -1;
    cplx acc = cplx{0.0, 0.0};
    for (int32_t e = e0; e < e1; e += 64) {
      const int cnt = min(64, e1 - e);
      int32_t colv2 = 0, lrv2 = 0;
      cplx valv2 = cplx{0.0, 0.0};
      if (e + 64 + lane < e1) {
        colv2 = mid_col[e + 64 + lane];
        valv2 = mid_val[e + 64 + lane];
        lrv2 = mid_lrow[e + 64 + lane];
      }
      for (int t = 0; t < cnt; t += 4) {
        int32_t j_[4], r_[4];
        cplx a_[4], xv_[4];
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) {
          const int idx = min(t + b2, 63);
          j_[b2] = rl32(colv, idx);
          a_[b2] = cplx{rl64(valv.x, idx), rl64(valv.y, idx)};
          r_[b2] = rl32(lrv, idx);
        }
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2)
          if (t + b2 < cnt) xv_[b2] = x[((int64_t)j_[b2] << 6) + lane];
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2)
          if (t + b2 < cnt) {
            if (r_[b2] != cur_r) {  // (wave-uniform)
              if (cur_r >= 0) {
                t_re[(cur_r << 6) + lane] = acc.x;
                t_im[(cur_r << 6) + lane] = acc.y;
              }
              cur_r = r_[b2];
              acc = cplx{t_re[(cur_r << 6) + lane], t_im[(cur_r << 6) + lane]};
            }
            acc = vsub(acc, vmul(a_[b2], xv_[b2]));
          }
      }
      colv = colv2;
      valv = valv2;
      lrv = lrv2;
    }
    if (cur_r >= 0) {
      t_re[(cur_r << 6) + lane] = acc.x;
      t_im[(cur_r << 6) + lane] = acc.y;
    }
    __syncthreads();
    // ---- phase 2: x = Tinv * t on the real matrix cores, four products per output tile
    const int lda = (nb + 31) & ~31;
    const int S = (nb + 15) >> 4, nunits = S * 4;
    const double *Are = tinv + inv_off, *Aim = Are + ((int64_t)((nb + 15) & ~15) * lda);  // plane_elems(nb, lda)
    for (int q = wave; q < nunits; q += nw) {
      const int strip = S - 1 - (q >> 2), ct = q & 3;
      const int kend = min(nb, 16 * (strip + 1));
      const int nk = (kend + 3) >> 2;  // k-steps of four columns
      const double *Apr = Are + ((int64_t)strip * lda) * 16 + (lane & 15) + (int64_t)kq * 16;
      const double *Api = Aim + ((int64_t)strip * lda) * 16 + (lane & 15) + (int64_t)kq * 16;
      const double *Bre = t_re + ct * 16 + (lane & 15), *Bim = t_im + ct * 16 + (lane & 15);
      v4f64 a_rr = v4f64{0.0, 0.0, 0.0, 0.0}, a_ii = a_rr, a_ri = a_rr, a_ir = a_rr;
      constexpr int KU = 2;
      double pr0[KU], pi0[KU], pr1[KU], pi1[KU];
#define HIFAMD_CDZ_LOAD(pr, pi, t_)                                                       \
  _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                        \
    pr[u] = Apr[(int64_t)(KU * (t_) + u) * 64];                                           \
    pi[u] = Api[(int64_t)(KU * (t_) + u) * 64];                                           \
  }
#define HIFAMD_CDZ_MFMA(pr, pi, t_)                                                       \
  _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                        \
    const int kb_ = 4 * (KU * (t_) + u) + kq;                                             \
    const bool ok_ = kb_ < nb;                                                            \
    const double br_ = ok_ ? Bre[kb_ << 6] : 0.0, bi_ = ok_ ? Bim[kb_ << 6] : 0.0;        \
    a_rr = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[u], br_, a_rr, 0, 0, 0);               \
    a_ii = __builtin_amdgcn_mfma_f64_16x16x4f64(pi[u], bi_, a_ii, 0, 0, 0);               \
    a_ri = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[u], bi_, a_ri, 0, 0, 0);               \
    a_ir = __builtin_amdgcn_mfma_f64_16x16x4f64(pi[u], br_, a_ir, 0, 0, 0);               \
  }
      // (the operand planes are zero padded in k up to lda, a multiple of 32: sets of KU k-steps never leave the strip)
      const int nsets = (nk + KU - 1) / KU;
      int t = 0;
      HIFAMD_CDZ_LOAD(pr0, pi0, 0)
      while (t < nsets) {
        if (t + 1 < nsets) HIFAMD_CDZ_LOAD(pr1, pi1, t + 1)
        HIFAMD_CDZ_MFMA(pr0, pi0, t)
        if (t + 1 >= nsets) break;
        if (t + 2 < nsets) HIFAMD_CDZ_LOAD(pr0, pi0, t + 2)
        HIFAMD_CDZ_MFMA(pr1, pi1, t + 1)
        t += 2;
      }
#undef HIFAMD_CDZ_LOAD
#undef HIFAMD_CDZ_MFMA
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * strip + kq + 4 * r;
        if (row < nb) x[((int64_t)cd_rowid[row] << 6) + ct * 16 + (lane & 15)] = cplx{a_rr[r] - a_ii[r], a_ri[r] + a_ir[r]};
      }
    }
    __syncthreads();  // (the next component overwrites the LDS planes)
  }
}

// Complex products on the real matrix cores: with X viewed as a real [rows][2R] block (re, im interleaved),
// T1 = A_re * X and T2 = A_im * X are two real MFMA products (k_dense_gemm_d / k_tri_gemm_d at logR + 1);
// this kernel recombines  out = (T1_re - T2_im) + i (T1_im + T2_re),  applies the output row permutation
// and the optional fused division (same epilogue as the real kernels).
__global__ void __launch_bounds__(256) k_zcombine(int64_t nrows, const double *__restrict__ T1,
                                                  const double *__restrict__ T2, int logR,
                                                  const int32_t *__restrict__ rowmap, cplx *__restrict__ Out,
                                                  const cplx *__restrict__ dscale, cplx *__restrict__ Out2) {
  const LaneMap lm = lane_map(logR);
  const int64_t wave = ((int64_t)xcd_block() * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave * lm.G + lm.g; i < nrows; i += nwaves * lm.G) {
    const int64_t o = (i << (logR + 1)) + 2 * lm.c;
    const cplx val{T1[o] - T2[o + 1], T1[o + 1] + T2[o]};
    const int64_t orow = rowmap ? rowmap[i] : i;
    Out[(orow << logR) + lm.c] = val;
    if (Out2) Out2[(orow << logR) + lm.c] = vdiv(val, dscale[orow]);
  }
}

// ---------------------------------------------------------------------------------------------
// Set-up (hifamd_finalize): the explicit inverses of the diagonal blocks of the block-dense and component-dense bands,
// formed ON THE DEVICE straight in operand layout -- host.hpp build_dense_block restated (forward substitution on the
// identity; the host version computes 16 columns side by side, here a lane owns a column): the same operations per
// column in the same order, the same bits (tests compare the two through hifamd_debug_checksums).  30 GB of operators
// for the 16.8 M-row hierarchy took the host threads 21 s and the PCIe link; this takes the factors that are on the
// device anyway.  Block q = rows [slot0, slot1) of the slot-ordered triangle; element (r, c) of its inverse lives at
// ((r >> 4) * ldk + c) * 16 + (r & 15) (complex: imaginary parts one plane behind); a lane reads back what it wrote
// (y_c of an earlier row c), everything above the diagonal stays the zero it was set to.  growth[q] = largest |entry|
// (NaN when a nonzero of the block is not finite: the host arithmetic would have produced NaN from inf * 0).
// grid (blocks, ceil(max rows / 256)), 256 threads.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double vabs(double a) { return fabs(a); }
__device__ __forceinline__ double vabs(cplx a) { return hypot(a.x, a.y); }
__device__ __forceinline__ bool vfinite(double a) { return isfinite(a); }
__device__ __forceinline__ bool vfinite(cplx a) { return isfinite(a.x) && isfinite(a.y); }
__device__ __forceinline__ double op_get(const double *ops, int64_t, int64_t idx, double) { return ops[idx]; }
__device__ __forceinline__ cplx op_get(const double *ops, int64_t plane, int64_t idx, cplx) { return cplx{ops[idx], ops[plane + idx]}; }
__device__ __forceinline__ void op_put(double *ops, int64_t, int64_t idx, double v) { ops[idx] = v; }
__device__ __forceinline__ void op_put(double *ops, int64_t plane, int64_t idx, cplx v) { ops[idx] = v.x, ops[plane + idx] = v.y; }
template <class T>
__global__ void __launch_bounds__(256) k_block_inverse(const int32_t *__restrict__ blk_slot0, const int32_t *__restrict__ blk_slot1,
                                                       const int64_t *__restrict__ blk_inv_off, const int32_t *__restrict__ ptr,
                                                       const int32_t *__restrict__ srcslot, const T *__restrict__ val, double *tinv,
                                                       unsigned long long *growth_bits) {
  const int q = (int)blockIdx.x;
  const int32_t r0 = blk_slot0[q], nb = blk_slot1[q] - r0;
  if ((int32_t)blockIdx.y * 256 >= nb) return;
  const int32_t j = (int32_t)blockIdx.y * 256 + (int32_t)threadIdx.x;  // this lane's column of the inverse
  const bool live = j < nb;
  if (__builtin_amdgcn_readfirstlane(j) >= nb) return;  // (a wave without a column)
  const int32_t c0 = j & ~15;  // (the host works on chunks of 16 columns: a column's sum starts at its chunk's first row)
  const int64_t ldk = ((int64_t)nb + 31) & ~(int64_t)31, plane = (((int64_t)nb + 15) / 16) * 16 * ldk;
  double *ops = tinv + blk_inv_off[q];
  double g = 1.0;
  if (live) op_put(ops, plane, (((int64_t)(j >> 4)) * ldk + j) * 16 + (j & 15), vfromreal(1.0, T()));
  const int32_t rbeg = __builtin_amdgcn_readfirstlane(c0) + 1;  // (the wave's first chunk; later chunks skip rows <= their c0)
  for (int32_t r = rbeg; r < nb; ++r) {
    const int32_t s = r0 + r;
    T acc = vfromreal(r == j ? 1.0 : 0.0, T());
    const bool mine = live && r > c0;
    for (int32_t k = ptr[s]; k < ptr[s + 1]; ++k) {
      const int32_t c = srcslot[k] - r0;
      if (c < 0) continue;  // (a source outside the block: the band kernel's business)
      const T a = val[k];
      if (!vfinite(a)) g = __builtin_nan("");
      if (mine && c >= c0) acc = vsub(acc, vmul(a, op_get(ops, plane, (((int64_t)(c >> 4)) * ldk + j) * 16 + (c & 15), T())));
    }
    if (mine && j <= r) {
      if (j < r) op_put(ops, plane, (((int64_t)(r >> 4)) * ldk + j) * 16 + (r & 15), acc);
      g = vfinite(acc) ? fmax(g, vabs(acc)) + (isnan(g) ? g : 0.0) : __builtin_nan("");
    }
  }
  // largest entry of the block: positive doubles order like their bit patterns; NaN (all ones in the exponent) wins
  if (live) atomicMax(&growth_bits[q], (unsigned long long)__double_as_longlong(isnan(g) ? __builtin_nan("") : g));
}

}  // namespace hifamd
