// Microbenchmark (development aid): latency of ONE dependent global load as a function of the footprint it jumps
// around in (L2 / Infinity Cache / HBM, translation-cache reach), one lane per workgroup chasing pointers, few or many
// workgroups.    hipcc --offload-arch=gfx950 -O3 tests/microbench/hop_latency.hip -o /tmp/hop && /tmp/hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_chase(const unsigned *__restrict__ next, unsigned start_stride, int hops, unsigned *out, unsigned long long *cyc) {
  unsigned idx = blockIdx.x * start_stride;
  const unsigned long long t0 = wall_clock64();
  for (int h = 0; h < hops; ++h) idx = next[idx];
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[blockIdx.x] = idx;
    cyc[blockIdx.x] = t1 - t0;
  }
}

int main() {
  hipStream_t st;
  OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  printf("%12s %8s %6s | ns per dependent load (median over workgroups; 100 MHz clock => 10 ns resolution x hops)\n", "footprint", "stride", "wgs");
  for (size_t mb : {1, 16, 128, 1024, 8192}) {
    for (size_t stride_b : {128, 4096, 65536, 2097152 + 128}) {  // distance between consecutive chain elements (bytes)
      const size_t n = mb * 1024 * 1024 / 4;
      const size_t step = stride_b / 4;
      const size_t nelem = n / step;
      if (nelem < 64) continue;
      // a random cycle over the elements {0, step, 2 step, ...}
      std::vector<unsigned> perm(nelem);
      std::iota(perm.begin(), perm.end(), 0u);
      std::mt19937 g(7);
      std::shuffle(perm.begin(), perm.end(), g);
      unsigned *d;
      OK(hipMalloc(&d, n * 4));
      std::vector<unsigned> h(nelem);
      // write only the chain elements (sparse writes through a staging kernel would be nicer; memset + scatter copy)
      OK(hipMemset(d, 0, n * 4));
      std::vector<unsigned> nxt(nelem);
      for (size_t i = 0; i < nelem; ++i) nxt[perm[i]] = (unsigned)(perm[(i + 1) % nelem] * step);
      for (size_t i = 0; i < nelem; ++i) OK(hipMemcpyAsync(d + i * step, &nxt[i], 4, hipMemcpyHostToDevice, st));
      OK(hipStreamSynchronize(st));
      for (int wgs : {1, 256}) {
        unsigned *out;
        unsigned long long *cyc;
        OK(hipMalloc(&out, 4096 * 4));
        OK(hipMalloc(&cyc, 4096 * 8));
        const int hops = (int)std::min<size_t>(64, nelem / std::max(1, wgs) > 8 ? 64 : 8);
        // flush caches between runs by touching another big buffer? here: first run cold-ish, report second too
        for (int rep = 0; rep < 2; ++rep) {
          hipLaunchKernelGGL(k_chase, dim3(wgs), dim3(64), 0, st, d, (unsigned)(step * (nelem / (size_t)wgs > 0 ? nelem / (size_t)wgs : 1)), hops, out, cyc);
          OK(hipStreamSynchronize(st));
          std::vector<unsigned long long> c(wgs);
          OK(hipMemcpy(c.data(), cyc, wgs * 8, hipMemcpyDeviceToHost));
          std::sort(c.begin(), c.end());
          printf("%9zu MB %8zu %6d | run %d: %7.0f\n", mb, stride_b, wgs, rep, (double)c[wgs / 2] * 10.0 / hops);
        }
        OK(hipFree(out));
        OK(hipFree(cyc));
      }
      OK(hipFree(d));
    }
  }
  return 0;
}
