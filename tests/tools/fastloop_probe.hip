// Developer probe: time per request of match_fast_loop (yadcc_amd/csrc/match_kernel.h) with
// synthetic class lists — one wave alone on the chip, and 1 / 2 / 3 waves per SIMD on every CU,
// single step and pairs. Answers "what does an instruction more or less in the loop cost".
//   hipcc --offload-arch=gfx950 -O3 -I yadcc_amd/csrc -I include tests/tools/fastloop_probe.hip -o /tmp/flp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels.h"
using namespace ydc;

// Every wave: `blocks` blocks of 20 requests. Class c's entry e has rank e * n_classes + c, so
// consecutive requests (all classes eligible) win on consecutive lanes: every pair succeeds.
__global__ __launch_bounds__(64) void probe(uint32_t n_classes, uint32_t blocks, uint32_t pair,
                                            uint64_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x;
  const uint32_t rshift = 5, R = 32;
  for (uint32_t e = 0; e < R; ++e) {
    if (lane < n_classes) {
      lds[(lane << rshift) + e] = ~(e * n_classes + lane);
      lds[2048 + (lane << rshift) + e] = 1000 + e * n_classes + lane;
    }
  }
  __syncthreads();
  uint32_t total_i = 0;
  uint64_t t_asm = 0;
  const uint64_t w0 = wall_clock64();
  for (uint32_t b = 0; b < blocks; ++b) {
    uint32_t hq = lane < n_classes ? ~lane : 0u, nq = lane < n_classes ? ~(n_classes + lane) : 0u;
    const uint32_t base = (uint32_t)(uintptr_t)lds + ((lane << rshift) << 2);
    uint32_t an = base + 4, raw = 0, left = 0x7FFFFFFFu;  // address of `next` (entry 1)
    uint32_t i = 0;
    const uint32_t mlo = n_classes >= 32 ? 0xFFFFFFFFu : ((1u << n_classes) - 1);
    const uint32_t mhi = n_classes > 32 ? (n_classes >= 64 ? 0xFFFFFFFFu : (1u << (n_classes - 32)) - 1) : 0;
    uint32_t steps = 1;
    while ((1u << steps) < n_classes) ++steps;
    const BlockMasks bm = block_masks(mlo, mhi, steps > 5);
    const uint64_t a = wall_clock64();
    // 20 requests per block: at most 20 picks per class, the ring (32) never wraps.
    uint32_t st = match_fast_loop<false>(i, 20, bm, kNone, kNone, 0ull, 0ull, 0ull, raw, hq, nq, an,
                                  2048u << 2, R * 4 - 1, steps, pair && steps >= 3 ? 1u : 0u, left);
    t_asm += wall_clock64() - a;
    total_i += i + st + (raw & 1);
  }
  const uint64_t w1 = wall_clock64();
  if (lane == 0 && blockIdx.x == 0) {
    out[0] = w1 - w0;
    out[1] = t_asm;
    out[2] = total_i;
  }
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const uint32_t blocks = 2000;
  for (uint32_t nc : {2u, 4u, 16u, 30u, 64u}) {
    for (uint32_t pair : {0u, 1u}) {
      if (pair && nc < 5) continue;
      for (uint32_t waves_per_simd : {0u, 1u, 2u, 3u}) {  // 0: one wave on the whole chip
        const uint32_t grid = waves_per_simd == 0 ? 1 : 256 * 4 * waves_per_simd;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0, 0);
          hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 16384, 0, nc, blocks, pair, d);
          hipEventRecord(e1, 0);
          hipDeviceSynchronize();
          hipEventElapsedTime(&ms, e0, e1);
        }
        uint64_t h[3];
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        // wall_clock64: 100 MHz. Per-SIMD rate: the launch's time over the requests of one SIMD.
        const double per_req_wave = (double)h[1] * 10.0 / (blocks * 20.0);
        const double per_req_simd =
            waves_per_simd ? ms * 1e6 / (blocks * 20.0 * waves_per_simd) : ms * 1e6 / (blocks * 20.0);
        printf("classes %2u pair %u waves/SIMD %u: %.1f ns/request in a wave's loop, %.1f ns/request "
               "per SIMD (launch %.3f ms)\n", nc, pair, waves_per_simd, per_req_wave, per_req_simd, ms);
      }
    }
  }
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("device clock rate attr: %d kHz\n", clk);
  return 0;
}
