// What do the next pass's tile histograms cost when every element of a radix scatter adds itself
// to them with a device-scope atomic? n elements, 256 threads x 8 elements
// per workgroup like k_radix_scatter; element e of tile t "lands" at a pseudo-random output
// position that is contiguous for runs of 8 elements (a tile's elements of one digit), and adds 1
// to table[(position / 2048) * 256 + next_digit]. Compared with: the same loop without the atomic,
// and k_radix_hist's way (LDS histogram per tile, 256 plain stores).
// build: hipcc --offload-arch=gfx950 -O2 -o tests/tools/atomic_probe tests/tools/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>  // 0: nothing, 1: scattered atomics, 2: returning atomics (for comparison)
__global__ __launch_bounds__(256) void k_scatter_like(const uint2* in, uint32_t n, uint32_t* table, uint32_t* sink) {
  uint32_t acc = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t e = blockIdx.x * 2048 + j * 256 + threadIdx.x;
    if (e < n) {
      const uint2 r = in[e];
      const uint32_t run = mix((e >> 3) * 2654435761u) % (n >> 3);  // where this run of 8 lands
      const uint32_t pos = run * 8 + (e & 7);
      const uint32_t next_digit = (r.x >> 8) & 255u;
      if (MODE == 1) atomicAdd(&table[(pos >> 11) * 256 + next_digit], 1u);
      if (MODE == 2) acc += atomicAdd(&table[(pos >> 11) * 256 + next_digit], 1u);
      acc += pos ^ r.y;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_hist_like(const uint2* in, uint32_t n, uint32_t* table, uint32_t n_tiles) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t e = blockIdx.x * 2048 + j * 256 + threadIdx.x;
    if (e < n) atomicAdd(&h[(in[e].x >> 8) & 255u], 1u);
  }
  __syncthreads();
  table[threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 4966419u;
  const int reps = 50;
  const uint32_t tiles = (n + 2047) / 2048;
  uint2* in; uint32_t *table, *sink;
  CK(hipMalloc(&in, (size_t)n * 8)); CK(hipMalloc(&table, (size_t)tiles * 256 * 4)); CK(hipMalloc(&sink, 4));
  uint2* h = (uint2*)malloc((size_t)n * 8);
  for (uint32_t i = 0; i < n; ++i) h[i] = make_uint2((uint32_t)rand() * 2654435761u, i);
  CK(hipMemcpy(in, h, (size_t)n * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* what, auto launch) {
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("%-64s %8.2f us per launch\n", what, ms * 1e3 / reps);
  };
  printf("%u elements, %u tiles, table %u KB\n", n, tiles, tiles);
  timeit("memset of the table", [&] { hipMemsetAsync(table, 0, (size_t)tiles * 1024, 0); });
  timeit("scatter-like pass, no atomics (reads only)", [&] { hipLaunchKernelGGL(k_scatter_like<0>, dim3(tiles), dim3(256), 0, 0, in, n, table, sink); });
  timeit("... + one scattered device-scope atomicAdd per element", [&] { hipLaunchKernelGGL(k_scatter_like<1>, dim3(tiles), dim3(256), 0, 0, in, n, table, sink); });
  timeit("... + one RETURNING atomicAdd per element", [&] { hipLaunchKernelGGL(k_scatter_like<2>, dim3(tiles), dim3(256), 0, 0, in, n, table, sink); });
  timeit("histogram launch: LDS histogram per tile, 256 stores", [&] { hipLaunchKernelGGL(k_hist_like, dim3(tiles), dim3(256), 0, 0, in, n, table, tiles); });
  return 0;
}
