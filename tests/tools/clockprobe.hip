// Test tool: measures the shader clock a lone, latency-bound wave runs at, and the
// cost of simple dependent instruction chains (clock64 = shader cycles,
// wall_clock64 = constant 100 MHz counter).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint64_t* out, int iters, int waves_busy) {
  uint32_t v = threadIdx.x;
  uint64_t c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    // 8 dependent VALU + 1 ballot/ffs per iteration
    v = v * 1664525u + 1013904223u;
    v ^= v >> 7;
    v += i;
    v = v * 22695477u + 1u;
    v ^= v << 3;
    uint64_t b = __ballot(v & 1);
    v += (uint32_t)__builtin_ctzll(b | 1);
    v = (uint32_t)__builtin_amdgcn_readlane((int)v, 5) + v;
  }
  uint64_t c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = v; }
}
int main() {
  uint64_t* d; hipMalloc(&d, 64);
  for (int rep = 0; rep < 6; ++rep) {
    int blocks = rep < 3 ? 1 : 1024;
    int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, d, iters, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    uint64_t h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("blocks=%d iters=%d: %.1f us, shader cycles %llu (%.1f/iter), wall ticks %llu => %.0f MHz shader clock, %.1f ns/iter\n",
           blocks, iters, ms * 1e3, (unsigned long long)h[0], (double)h[0] / iters, (unsigned long long)h[1],
           h[0] / (h[1] / 100.0), ms * 1e6 / iters);
  }
  return 0;
}
