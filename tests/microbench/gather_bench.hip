// Microbenchmark (development aid, not part of the product): how fast can ONE wave / ONE compute unit of
// an MI355X gather random 512-byte rows (64 lanes x 8 B, the access of the triangular-solve kernels) as
// a function of the number of gathers a wave keeps in flight (D) and of the waves per workgroup?
// Prints ns per gather and wave, and gathers per microsecond and compute unit.
//   hipcc -O3 --offload-arch=gfx950 gather_bench.hip -o gather_bench.bin && ./gather_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_fill(double *x, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = 1.0 + 1e-9 * (double)(i & 1023);
}

// every wave performs `iters` rounds of D gathers; row index = hash(wave, round, slot) mod nrows (scalar unit)
template <int D>
__global__ void __launch_bounds__(1024) k_gather(const double *__restrict__ x, unsigned nrows, int iters, double *out,
                                                 unsigned long long *ticks) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  unsigned state = wave * 2654435761u + 12345u;
  double acc = 0.0;
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    double v[D];
#pragma unroll
    for (int b = 0; b < D; ++b) {
      state = state * 1664525u + 1013904223u;
      const unsigned j = (unsigned)(((unsigned long long)(state >> 4) * nrows) >> 28);
      v[b] = x[((size_t)j << 6) + lane];
    }
#pragma unroll
    for (int b = 0; b < D; ++b) acc = acc - 0.5 * v[b];
  }
  const unsigned long long t1 = wall_clock64();
  out[((size_t)wave << 6) + lane] = acc;
  if (lane == 0) ticks[wave] = t1 - t0;
}

// two rows per load instruction: lanes 0-31 fetch row j0, lanes 32-63 row j1, 16 bytes (two columns) per lane
template <int D>
__global__ void __launch_bounds__(1024) k_gather2(const double *__restrict__ x, unsigned nrows, int iters, double *out,
                                                  unsigned long long *ticks) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  unsigned state = wave * 2654435761u + 12345u;
  double acc0 = 0.0, acc1 = 0.0;
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    double2 v[D];
#pragma unroll
    for (int b = 0; b < D; ++b) {
      state = state * 1664525u + 1013904223u;
      const unsigned j0 = (unsigned)(((unsigned long long)(state >> 4) * nrows) >> 28);
      state = state * 1664525u + 1013904223u;
      const unsigned j1 = (unsigned)(((unsigned long long)(state >> 4) * nrows) >> 28);
      const unsigned j = lane < 32 ? j0 : j1;
      v[b] = *(const double2 *)(x + ((size_t)j << 6) + 2 * (lane & 31));
    }
#pragma unroll
    for (int b = 0; b < D; ++b) {
      acc0 = acc0 - 0.5 * v[b].x;
      acc1 = acc1 - 0.5 * v[b].y;
    }
  }
  const unsigned long long t1 = wall_clock64();
  out[((size_t)wave << 6) + lane] = acc0 + acc1;
  if (lane == 0) ticks[wave] = t1 - t0;
}

template <int D>
static void run2(const double *x, unsigned nrows, int wgs, int waves, double *out, unsigned long long *ticks, bool refill, double *xw) {
  const int iters = 256 / D;  // 256 load instructions = 512 rows per wave
  if (refill) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, xw, (size_t)nrows * 64);
  hipLaunchKernelGGL(k_gather2<D>, dim3(wgs), dim3(waves * 64), 0, 0, x, nrows, iters, out, ticks);
  OK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)wgs * waves);
  OK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (auto t : h) sum += (double)t;
  const double ns_per_row = sum / h.size() * 10.0 / 512.0;
  printf("PAIRED rows %8u %s wgs %4d waves/wg %2d in-flight %2d instr (2 rows each): %7.1f ns per ROW and wave, %7.1f rows/us per CU, %6.1f GB/s per CU\n",
         nrows, refill ? "fresh " : "steady", wgs, waves, D, ns_per_row, waves * 1000.0 / ns_per_row, waves * 512.0 / ns_per_row);
}

template <int D>
static void run(const double *x, unsigned nrows, int wgs, int waves, double *out, unsigned long long *ticks, bool refill, double *xw) {
  const int iters = 512 / D;  // 512 gathers per wave
  if (refill) {  // the rows were just written by another kernel (as in the solver): not resident in the reader's L2
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, xw, (size_t)nrows * 64);
  }
  hipLaunchKernelGGL(k_gather<D>, dim3(wgs), dim3(waves * 64), 0, 0, x, nrows, iters, out, ticks);
  OK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)wgs * waves);
  OK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0, mx = 0;
  for (auto t : h) { sum += (double)t; mx = mx > (double)t ? mx : (double)t; }
  const double ns_per_gather = sum / h.size() * 10.0 / 512.0;  // 100 MHz counter
  const double per_cu_per_us = waves * 1000.0 / ns_per_gather;
  printf("rows %8u %s wgs %4d waves/wg %2d in-flight %2d : %7.1f ns per gather and wave (slowest wave %7.1f), %7.1f gathers/us per CU, %6.1f GB/s per CU\n",
         nrows, refill ? "fresh " : "steady", wgs, waves, D, ns_per_gather, mx * 10.0 / 512.0, per_cu_per_us, per_cu_per_us * 512e-3);
}

int main() {
  const unsigned sizes[2] = {100000u, 1000000u};
  double *out;
  unsigned long long *ticks;
  OK(hipMalloc(&out, (size_t)1024 * 16 * 64 * 8));
  OK(hipMalloc(&ticks, (size_t)1024 * 16 * 8));
  for (unsigned nrows : sizes) {
    double *x;
    OK(hipMalloc(&x, (size_t)nrows * 64 * 8));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, x, (size_t)nrows * 64);
    OK(hipDeviceSynchronize());
    for (int refill = 0; refill < 2; ++refill)
      for (int wgs : {64, 256})
        for (int waves : {1, 4, 16}) {
          run<1>(x, nrows, wgs, waves, out, ticks, refill, x);
          run<2>(x, nrows, wgs, waves, out, ticks, refill, x);
          run<4>(x, nrows, wgs, waves, out, ticks, refill, x);
          run<8>(x, nrows, wgs, waves, out, ticks, refill, x);
          run<16>(x, nrows, wgs, waves, out, ticks, refill, x);
          run<32>(x, nrows, wgs, waves, out, ticks, refill, x);
          run2<2>(x, nrows, wgs, waves, out, ticks, refill, x);
          run2<4>(x, nrows, wgs, waves, out, ticks, refill, x);
          run2<8>(x, nrows, wgs, waves, out, ticks, refill, x);
          run2<16>(x, nrows, wgs, waves, out, ticks, refill, x);
        }
    OK(hipFree(x));
  }
  return 0;
}
