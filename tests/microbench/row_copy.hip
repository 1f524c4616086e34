// Microbenchmark (development aid): what the memory system gives a band kernel's traffic shape -- every workgroup (1024
// threads, one per compute unit resident, 96 KB of LDS claimed) reads 192 rows of 512 B through an index list and writes
// 192 rows through another, as a function of WHERE those rows lie: one contiguous block, runs of 2 rows inside a window
// of 2,500 rows (level 0 of the 1M-row Poisson hierarchy: PLAN2 id_runs / id_span_sum), or anywhere.
//   hipcc --offload-arch=gfx950 -O3 tests/microbench/row_copy.hip -o /tmp/row_copy && /tmp/row_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// rows per workgroup 192, wave w takes rows w, w + 16, ...: twelve loads in flight per lane, then twelve stores
__global__ void __launch_bounds__(1024) k_rows(const double *__restrict__ in, double *__restrict__ out, const int *__restrict__ src,
                                               const int *__restrict__ dst, int ncomp) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = blockIdx.x; c < ncomp; c += gridDim.x) {
    double r[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) r[j] = in[(size_t)src[c * 192 + wave + 16 * j] * 64 + lane];
#pragma unroll
    for (int j = 0; j < 12; ++j) out[(size_t)dst[c * 192 + wave + 16 * j] * 64 + lane] = r[j] * 1.0000001;
  }
  if (lds[0] == 123.0) out[0] = 0;  // (keeps the LDS claim)
}

// the same with 16 bytes per lane: a wave instruction moves two rows (lanes 0-31 one row, lanes 32-63 the next); NT: nontemporal
template <bool NT>
__global__ void __launch_bounds__(1024) k_rows16(const double *__restrict__ in, double *__restrict__ out, const int *__restrict__ src,
                                                 const int *__restrict__ dst, int ncomp) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
  typedef double v2 __attribute__((ext_vector_type(2)));
  for (int c = blockIdx.x; c < ncomp; c += gridDim.x) {
    v2 r[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const v2 *ptr = reinterpret_cast<const v2 *>(in + (size_t)src[c * 192 + 2 * (wave + 16 * j) + half] * 64) + l32;
      r[j] = NT ? __builtin_nontemporal_load(ptr) : *ptr;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      v2 *ptr = reinterpret_cast<v2 *>(out + (size_t)dst[c * 192 + 2 * (wave + 16 * j) + half] * 64) + l32;
      const v2 val = r[j] * 1.0000001;
      if (NT) __builtin_nontemporal_store(val, ptr); else *ptr = val;
    }
  }
  if (lds[0] == 123.0) out[0] = 0;
}

int main() {
  const int ncomp = 4096, rows = ncomp * 192;
  double *in, *out;
  int *src, *dst;
  OK(hipMalloc(&in, (size_t)rows * 512));
  OK(hipMalloc(&out, (size_t)rows * 512));
  OK(hipMalloc(&src, rows * 4));
  OK(hipMalloc(&dst, rows * 4));
  OK(hipMemset(in, 0, (size_t)rows * 512));
  OK(hipFuncSetAttribute((const void *)k_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  OK(hipFuncSetAttribute((const void *)k_rows16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  OK(hipFuncSetAttribute((const void *)k_rows16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  std::mt19937 g(5);
  auto pattern = [&](int kind) {  // the row list of every component
    std::vector<int> v(rows);
    std::iota(v.begin(), v.end(), 0);
    if (kind == 1) {  // runs of 2 rows, shuffled inside windows of 2,496 rows (13 components)
      for (int w0 = 0; w0 < rows; w0 += 2496) {
        const int n2 = std::min(2496, rows - w0) / 2;
        std::vector<int> pr(n2);
        std::iota(pr.begin(), pr.end(), 0);
        std::shuffle(pr.begin(), pr.end(), g);
        for (int k = 0; k < n2; ++k) v[w0 + 2 * k] = w0 + 2 * pr[k], v[w0 + 2 * k + 1] = w0 + 2 * pr[k] + 1;
      }
    } else if (kind == 2) {  // single rows shuffled inside the windows
      for (int w0 = 0; w0 < rows; w0 += 2496) std::shuffle(v.begin() + w0, v.begin() + std::min(rows, w0 + 2496), g);
    } else if (kind == 3) {
      std::shuffle(v.begin(), v.end(), g);
    }
    return v;
  };
  const char *names[] = {"contiguous", "runs of 2 in a 2,496-row window", "single rows in a 2,496-row window", "anywhere"};
  hipEvent_t e0, e1;
  OK(hipEventCreate(&e0));
  OK(hipEventCreate(&e1));
  printf("%-36s %-36s %8s %8s\n", "read rows", "written rows", "us", "TB/s");
  for (int ks = 0; ks < 4; ++ks)
    for (int kd = 0; kd < 4; ++kd) {
      if (!(ks == kd)) continue;
      auto s = pattern(ks), d = pattern(kd);
      OK(hipMemcpy(src, s.data(), rows * 4, hipMemcpyHostToDevice));
      OK(hipMemcpy(dst, d.data(), rows * 4, hipMemcpyHostToDevice));
      for (int var : {0, 1, 2}) {
        const int grid = 256;
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
          OK(hipEventRecord(e0, 0));
          if (var == 0) hipLaunchKernelGGL(k_rows, dim3(grid), dim3(1024), 97 * 1024, 0, in, out, src, dst, ncomp);
          if (var == 1) hipLaunchKernelGGL(k_rows16<false>, dim3(grid), dim3(1024), 97 * 1024, 0, in, out, src, dst, ncomp);
          if (var == 2) hipLaunchKernelGGL(k_rows16<true>, dim3(grid), dim3(1024), 97 * 1024, 0, in, out, src, dst, ncomp);
          OK(hipEventRecord(e1, 0));
          OK(hipEventSynchronize(e1));
          float ms;
          OK(hipEventElapsedTime(&ms, e0, e1));
          best = std::min(best, ms);
        }
        printf("%-36s %-36s %s %8.1f %8.2f\n", names[ks], names[kd], var == 0 ? " 8 B/lane   " : (var == 1 ? "16 B/lane   " : "16 B/lane nt"), best * 1e3, 2.0 * rows * 512 / (best * 1e-3) / 1e12);
      }
    }
  return 0;
}
