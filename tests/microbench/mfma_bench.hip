// How fast does gfx950 issue v_mfma_f64_16x16x4_f64?  Register-only loop, NACC independent accumulators per wave,
// W waves per workgroup (one workgroup per CU).  Build: hipcc --offload-arch=gfx950 -O3 mfma_bench.hip -o mfma_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double *out, int iters, double a0, double b0) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4f64{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int waves, int blocks) {
  double *d;
  hipMalloc(&d, sizeof(double) * blocks * waves * 64);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<NACC><<<blocks, waves * 64>>>(d, 10, 1.0, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, waves * 64>>>(d, iters, 1.0, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)blocks * waves * iters * NACC;
  printf("NACC=%d waves/WG=%2d blocks=%4d: %.3f ms, %.1f ns per MFMA per wave, %.2f TFLOP/s\n", NACC, waves, blocks, ms,
         ms * 1e6 / (iters * NACC), nm * 2048 / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() {
  for (int w : {1, 4, 8, 16}) {
    run<1>(w, 256);
    run<2>(w, 256);
    run<4>(w, 256);
    run<8>(w, 256);
  }
  run<4>(16, 512);
  return 0;
}
