// Developer probe: one pass of the radix sort of kernels.h (k_radix_hist -> k_radix_scan ->
// k_radix_scatter) on M random 8-byte records, timed with HIP events, with the scatter's parts
// switched off one at a time (SortIn::dbg: 1 = stores go to the element's own place (coalesced),
// 2 = no gather of the tile's histogram column, 4 = no stores at all). What bounds the pass?
//   hipcc --offload-arch=gfx950 -O3 -DYDC_SCATTER_PROBE -I yadcc_amd/csrc -I include tests/tools/scatter_probe.hip -o /tmp/scp && /tmp/scp [M=5000000] [bits=8] [shift=8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "kernels.h"
using namespace ydc;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const uint32_t M = argc > 1 ? (uint32_t)atol(argv[1]) : 5000000u;
  const uint32_t bits = argc > 2 ? (uint32_t)atoi(argv[2]) : 8u;
  const uint32_t shift = argc > 3 ? (uint32_t)atoi(argv[3]) : 8u;
  const uint32_t items = 8, n_tiles = (M + kSortThreads * items - 1) / (kSortThreads * items), radix = 1u << bits;
  std::vector<uint2> h(M);
  std::mt19937 rng(1);
  // keys as the previous pass leaves them: sorted by the bits below `shift`, random above
  for (uint32_t i = 0; i < M; ++i) h[i] = make_uint2((rng() << shift) | (uint32_t)(((uint64_t)i << shift) / M), i);
  uint2 *d_in, *d_out;
  uint32_t *d_hist, *d_rt;
  DeviceParams* d_prm;
  CK(hipMalloc(&d_in, (size_t)M * 8));
  CK(hipMalloc(&d_out, (size_t)M * 8));
  CK(hipMalloc(&d_hist, (size_t)radix * n_tiles * 4));
  CK(hipMalloc(&d_rt, radix * 4));
  CK(hipMalloc(&d_prm, sizeof(DeviceParams)));
  CK(hipMemcpy(d_in, h.data(), (size_t)M * 8, hipMemcpyHostToDevice));
  DeviceParams prm{};
  prm.n_slots = M;
  CK(hipMemcpy(d_prm, &prm, sizeof prm, hipMemcpyHostToDevice));
  hipEvent_t e0, e1, e2, e3;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
  printf("M %u, %u tiles of %u, digit: %u bits at %u\n", M, n_tiles, kSortThreads * items, bits, shift);
  for (uint32_t xcd = 0; xcd < 2; ++xcd)
    for (uint32_t dbg : {0u, 1u, 2u, 4u, 6u}) {
      SortIn<uint32_t> in{(const uint32_t*)d_in, nullptr, nullptr, shift, bits, items, 0u, 0u, 0xFFFFFFFFu, 1u, 0u};
      in.xcd_hist = xcd;
      in.xcd_scatter = xcd;
      in.dbg = dbg;
      float th = 0, ts = 0, tc = 0;
      const int reps = 20;
      for (int r = -3; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_radix_hist<uint32_t>, dim3(xcd_grid(n_tiles)), dim3(kSortThreads), radix * 4, 0, in, d_prm,
                           n_tiles, d_hist, PrefixArgs{});
        hipEventRecord(e1);
        hipLaunchKernelGGL(k_radix_scan, dim3(radix), dim3(256), 0, 0, n_tiles, d_hist, d_rt);
        hipEventRecord(e2);
        hipLaunchKernelGGL((k_radix_scatter<uint32_t, uint32_t>), dim3(xcd_grid(n_tiles)), dim3(kSortThreads),
                           (size_t)(kSortWaves + 1) * radix * 4, 0, in, d_prm, n_tiles, d_hist, d_rt, (uint32_t*)d_out,
                           nullptr);
        hipEventRecord(e3);
        CK(hipEventSynchronize(e3));
        float a, b, c;
        hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2); hipEventElapsedTime(&c, e2, e3);
        if (r >= 0) { th += a; ts += b; tc += c; }
      }
      printf("xcd %u dbg %u: hist %.1f us  scan %.1f us  scatter %.1f us\n", xcd, dbg, 1e3 * th / reps, 1e3 * ts / reps,
             1e3 * tc / reps);
    }
  return 0;
}
