// tools/hbm_calib.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the
// access widths this repository's kernels use (MI355X_MICROARCH.md, HBM section: only 16 B/lane
// streaming reads are documented — "FETCH_SIZE reports exactly half" —, every other width is
// "uncalibrated: calibrate on a known byte count"). Each kernel streams a buffer far larger
// than the 256 MiB Infinity Cache exactly once with one access width, so the bytes are known:
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/hbm_calib      (and --pmc WRITE_SIZE)
// tools/rocprof_summary.py calib turns the two passes into factors (known bytes / counter).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(e)                                                              \
  do {                                                                        \
    hipError_t err__ = (e);                                                   \
    if (err__ != hipSuccess) {                                                \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err__));         \
      std::exit(1);                                                           \
    }                                                                         \
  } while (0)

template <typename T>
__global__ __launch_bounds__(256) void k_calib_read(const T* __restrict__ in, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T v = in[i];
    const unsigned char* b = (const unsigned char*)&v;
    acc += b[0];
  }
  if (acc == 0x12345678u) *sink = acc;  // (never: keeps the loads alive)
}

template <typename T>
__global__ __launch_bounds__(256) void k_calib_write(T* __restrict__ out, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    T v;
    unsigned char* b = (unsigned char*)&v;
    for (size_t k = 0; k < sizeof(T); ++k) b[k] = (unsigned char)(seed + i + k);
    out[i] = v;
  }
}

// Scattered 4-byte writes, one per 64-byte line touched (the pattern of a digit-major
// histogram table or of per-request result scatters).
__global__ __launch_bounds__(256) void k_calib_write_strided(uint32_t* out, size_t n_lines, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x)
    out[i * 16] = seed + (uint32_t)i;
}
// Gather of 4 bytes per 64-byte line (rank -> slot -> owner table lookups).
__global__ __launch_bounds__(256) void k_calib_read_strided(const uint32_t* in, size_t n_lines, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x)
    acc += in[i * 16];
  if (acc == 0x12345678u) *sink = acc;
}

struct alignas(8) U2 { uint32_t a, b; };
struct alignas(16) U4 { uint32_t a, b, c, d; };

int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB >> 256 MiB Infinity Cache
  void* buf;
  uint32_t* sink;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc((void**)&sink, 4));
  CHECK(hipMemset(buf, 1, bytes));
  const dim3 grid(256 * 16), block(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_calib_read<unsigned char>, grid, block, 0, 0, (const unsigned char*)buf, bytes, sink);
    hipLaunchKernelGGL(k_calib_read<uint32_t>, grid, block, 0, 0, (const uint32_t*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(k_calib_read<U2>, grid, block, 0, 0, (const U2*)buf, bytes / 8, sink);
    hipLaunchKernelGGL(k_calib_read<U4>, grid, block, 0, 0, (const U4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(k_calib_read_strided, grid, block, 0, 0, (const uint32_t*)buf, bytes / 64, sink);
    hipLaunchKernelGGL(k_calib_write<unsigned char>, grid, block, 0, 0, (unsigned char*)buf, bytes, rep);
    hipLaunchKernelGGL(k_calib_write<uint32_t>, grid, block, 0, 0, (uint32_t*)buf, bytes / 4, rep);
    hipLaunchKernelGGL(k_calib_write<U2>, grid, block, 0, 0, (U2*)buf, bytes / 8, rep);
    hipLaunchKernelGGL(k_calib_write<U4>, grid, block, 0, 0, (U4*)buf, bytes / 16, rep);
    hipLaunchKernelGGL(k_calib_write_strided, grid, block, 0, 0, (uint32_t*)buf, bytes / 64, rep);
  }
  CHECK(hipDeviceSynchronize());
  std::printf("hbm_calib: %zu bytes per streaming kernel, %zu lines (x 4 B) per strided kernel\n", bytes,
              bytes / 64);
  return 0;
}
