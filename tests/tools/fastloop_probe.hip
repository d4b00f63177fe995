// Developer probe: cycles per request of match_fast_loop (yadcc_amd/csrc/match_kernel.h)
// on one wave with synthetic class lists. hipcc --offload-arch=gfx950 -O3 -I yadcc_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels.h"
using namespace ydc;

__global__ __launch_bounds__(64) void probe(uint32_t n_classes, uint32_t blocks, uint64_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x;
  const uint32_t rshift = 5, R = 32;
  // ring: class c entry e has rank e * n_classes + c (round robin), all resident.
  for (uint32_t e = 0; e < R; ++e) {
    if (lane < n_classes) {
      lds[(lane << rshift) + e] = ~(e * n_classes + lane);
      lds[2048 + (lane << rshift) + e] = 1000 + e * n_classes + lane;
    }
  }
  __syncthreads();
  uint32_t total_i = 0;
  uint64_t t_asm = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (uint32_t b = 0; b < blocks; ++b) {
    // inverted ranks: class c entry e has rank e * n_classes + c
    uint32_t hq = lane < n_classes ? ~lane : 0u, nq = lane < n_classes ? ~(n_classes + lane) : 0u;
    const uint32_t base = (uint32_t)(uintptr_t)lds + ((lane << rshift) << 2);
    uint32_t an = base + 4, res = kIdxTimeout;  // address of `next` (entry 1)
    uint32_t i = 0;
    const uint32_t mlo = n_classes >= 32 ? 0xFFFFFFFFu : ((1u << n_classes) - 1);
    const uint32_t mhi = n_classes > 32 ? (n_classes >= 64 ? 0xFFFFFFFFu : (1u << (n_classes - 32)) - 1) : 0;
    uint32_t steps = 1;
    while ((1u << steps) < n_classes) ++steps;
    const uint64_t a = __builtin_readcyclecounter();
    // 20 requests per block: at most 20 picks per class, the ring (32) never wraps.
    uint32_t st = match_fast_loop(i, 20, mlo, mhi, kNone, kNone, 0ull, 0ull, 0ull, res, hq, nq, an,
                                  2048u << 2, R * 4 - 1, steps, 0u);
    t_asm += __builtin_readcyclecounter() - a;
    total_i += i + st + (res & 1);
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) {
    out[0] = t1 - t0;
    out[1] = t_asm;
    out[2] = total_i;
  }
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 64);
  for (uint32_t nc : {2u, 16u, 30u, 64u}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 16384, 0, nc, 2000u, d);
      hipDeviceSynchronize();
    }
    uint64_t h[3];
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("classes %u: %.1f cycles/request in asm (%.1f incl. glue), s_memtime ticks; i-sum %llu\n", nc,
           (double)h[1] / (2000.0 * 20), (double)h[0] / (2000.0 * 20), (unsigned long long)h[2]);
  }
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("device clock rate attr: %d kHz\n", clk);
  return 0;
}
