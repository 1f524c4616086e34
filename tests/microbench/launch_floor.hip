// Microbenchmark (development aid): what does ONE dependent launch cost inside a replayed hipGraph on this machine, as a
// function of (a) grid size / workgroup size and (b) the number of dependent cold memory round trips inside the kernel?
//   hipcc --offload-arch=gfx950 -O3 tests/microbench/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// every workgroup: `hops` dependent loads through a permutation table (each lane its own chain, 512-B strided), then a store
__global__ void k_chain(const int *__restrict__ next, int hops, int *out, int stride_elems, int salt) {
  int idx = (blockIdx.x * blockDim.x + threadIdx.x + salt) * (stride_elems ? 1 : 0) + (stride_elems ? 0 : 0);
  idx = ((blockIdx.x * 977 + salt * 131) % 65536) * 128 + (threadIdx.x & 63);
  for (int h = 0; h < hops; ++h) idx = next[idx];
  if (hops >= 0) out[blockIdx.x * blockDim.x + threadIdx.x] = idx;
}

int main() {
  const int N = 65536 * 128;  // 32 MB table of ints (fits the Infinity Cache, not the L2s)
  std::vector<int> h(N);
  std::mt19937 g(1);
  // next[row*128 + l] = (random row)*128 + l : every hop lands on a different 512-byte row
  std::vector<int> rows(65536);
  std::iota(rows.begin(), rows.end(), 0);
  std::shuffle(rows.begin(), rows.end(), g);
  for (int r = 0; r < 65536; ++r)
    for (int l = 0; l < 128; ++l) h[(size_t)r * 128 + l] = rows[r] * 128 + l;
  int *d_next, *d_out;
  OK(hipMalloc(&d_next, (size_t)N * 4));
  OK(hipMalloc(&d_out, (size_t)4096 * 1024 * 4));
  OK(hipMemcpy(d_next, h.data(), (size_t)N * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int NL = 100;  // launches per graph
  printf("%6s %6s %5s | us per dependent launch (graph replay, %d launches per graph)\n", "grid", "block", "hops", NL);
  for (int block : {256, 1024})
    for (int grid : {64, 256, 1024})
      for (int hops : {-1, 0, 1, 2, 4, 8}) {
        hipGraph_t gr;
        hipGraphExec_t ex;
        OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NL; ++i) hipLaunchKernelGGL(k_chain, dim3(grid), dim3(block), 0, st, d_next, hops, d_out, 1, i);
        OK(hipStreamEndCapture(st, &gr));
        OK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
        hipEvent_t e0, e1;
        OK(hipEventCreate(&e0));
        OK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) OK(hipGraphLaunch(ex, st));
        OK(hipEventRecord(e0, st));
        const int reps = 10;
        for (int r = 0; r < reps; ++r) OK(hipGraphLaunch(ex, st));
        OK(hipEventRecord(e1, st));
        OK(hipEventSynchronize(e1));
        float ms = 0;
        OK(hipEventElapsedTime(&ms, e0, e1));
        printf("%6d %6d %5d | %7.2f\n", grid, block, hops, ms * 1e3 / (reps * NL));
        OK(hipGraphExecDestroy(ex));
        OK(hipGraphDestroy(gr));
      }
  return 0;
}
