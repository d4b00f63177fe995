// Test tool: what ONE small dispatch costs between "the host has a request" and "the host has
// the answer" on this box — the budget the small-batch path (ydc_dispatch_tick) lives in.
//   a  empty kernel + hipStreamSynchronize
//   b  kernel stores a stamp to coherent page-locked host memory, the host spins on it
//   c  the same with the work of a 2k / 16k-servant pick in front (1024 threads read K x 28 B each,
//      reduce, one lane stores)
//   d  b with 1 KB of kernel arguments (requests and registry deltas travel as arguments)
//   e  a resident workgroup polled through a page-locked mailbox (no launch at all)
// Prints p50 / p99 / mean in microseconds over `reps` round trips each.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
using Clk = std::chrono::steady_clock;

struct Args1k { uint32_t w[256]; };

__global__ void k_empty() {}
__global__ void k_stamp(volatile uint32_t* host, uint32_t seq) {
  if (threadIdx.x == 0) __hip_atomic_store((uint32_t*)host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_stamp_args(volatile uint32_t* host, uint32_t seq, Args1k a) {
  if (threadIdx.x == 0) __hip_atomic_store((uint32_t*)host, seq + (a.w[seq & 255] & 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int K>
__global__ void __launch_bounds__(1024) k_pick(const uint32_t* cols, uint32_t S, volatile uint32_t* host, uint32_t seq) {
  __shared__ uint32_t part[16];
  uint32_t best = 0xFFFFFFFFu;
  const uint32_t t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t s = t * K + k;
    if (s < S) {
      uint32_t v = 0;
#pragma unroll
      for (int c = 0; c < 7; ++c) v += cols[(size_t)c * S + s];
      v = (v << 16) | s;
      best = v < best ? v : best;
    }
  }
  for (int o = 32; o; o >>= 1) { uint32_t x = __shfl_xor(best, o); best = x < best ? x : best; }
  if ((t & 63) == 0) part[t >> 6] = best;
  __syncthreads();
  if (t < 16) {
    best = part[t];
    for (int o = 8; o; o >>= 1) { uint32_t x = __shfl_xor(best, o); best = x < best ? x : best; }
    if (t == 0) {
      host[1] = best;
      __hip_atomic_store((uint32_t*)host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// Resident workgroup: waits for mailbox[0] == seq, answers in mailbox[16]; quits on seq == ~0 or
// after `max_polls` empty polls (never hangs the box).
__global__ void k_resident(uint32_t* box, unsigned long long max_polls) {
  uint32_t want = 1;
  unsigned long long polls = 0;
  if (threadIdx.x != 0) return;
  for (;;) {
    const uint32_t got = __hip_atomic_load(box, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (got == 0xFFFFFFFFu) return;
    if (got == want) {
      __hip_atomic_store(box + 16, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      ++want;
      polls = 0;
    } else if (++polls > max_polls) {
      __hip_atomic_store(box + 17, 0xDEADu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}

static void report(const char* name, std::vector<double>& us) {
  std::sort(us.begin(), us.end());
  double sum = 0;
  for (double v : us) sum += v;
  std::printf("%-44s p50 %7.2f  p99 %7.2f  mean %7.2f  min %7.2f us (%zu)\n", name, us[us.size() / 2],
              us[us.size() * 99 / 100], sum / us.size(), us[0], us.size());
}

static bool spin(volatile uint32_t* p, uint32_t want, double limit_s = 2.0) {
  auto t0 = Clk::now();
  for (unsigned i = 0;; ++i) {
    if (__atomic_load_n((uint32_t*)p, __ATOMIC_ACQUIRE) == want) return true;
    if ((i & 1023) == 1023 && std::chrono::duration<double>(Clk::now() - t0).count() > limit_s) return false;
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? std::atoi(argv[1]) : 2000;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint32_t* host = nullptr;
  CK(hipHostMalloc((void**)&host, 4096, hipHostMallocCoherent | hipHostMallocMapped));
  uint32_t* host_dev = nullptr;
  CK(hipHostGetDevicePointer((void**)&host_dev, host, 0));
  for (int i = 0; i < 1024; ++i) host[i] = 0;
  uint32_t* cols = nullptr;
  const uint32_t Smax = 16384;
  CK(hipMalloc((void**)&cols, (size_t)7 * Smax * 4));
  CK(hipMemset(cols, 1, (size_t)7 * Smax * 4));
  CK(hipDeviceSynchronize());
  std::vector<double> us;
  auto run = [&](const char* name, auto&& body) {
    us.clear();
    for (int r = -50; r < reps; ++r) {
      auto t0 = Clk::now();
      body((uint32_t)(r + 51));
      auto t1 = Clk::now();
      if (r >= 0) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    CK(hipStreamSynchronize(st));
    report(name, us);
  };
  run("a empty kernel + hipStreamSynchronize", [&](uint32_t) {
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
    CK(hipStreamSynchronize(st));
  });
  run("a' 4 empty kernels + hipStreamSynchronize", [&](uint32_t) {
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
    CK(hipStreamSynchronize(st));
  });
  run("b stamp to pinned host + host spin", [&](uint32_t seq) {
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, st, (volatile uint32_t*)host_dev, seq);
    if (!spin(host, seq)) { std::fprintf(stderr, "b: no stamp\n"); std::exit(1); }
  });
  run("b' stamp + spin, then hipStreamSynchronize", [&](uint32_t seq) {
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, st, (volatile uint32_t*)host_dev, seq + 100000);
    if (!spin(host, seq + 100000)) { std::fprintf(stderr, "b': no stamp\n"); std::exit(1); }
    CK(hipStreamSynchronize(st));
  });
  run("c pick over 2048 servants (K=2) + spin", [&](uint32_t seq) {
    hipLaunchKernelGGL(k_pick<2>, dim3(1), dim3(1024), 0, st, cols, 2048u, (volatile uint32_t*)host_dev, seq + 200000);
    if (!spin(host, seq + 200000)) { std::fprintf(stderr, "c: no stamp\n"); std::exit(1); }
  });
  run("c pick over 16384 servants (K=16) + spin", [&](uint32_t seq) {
    hipLaunchKernelGGL(k_pick<16>, dim3(1), dim3(1024), 0, st, cols, 16384u, (volatile uint32_t*)host_dev, seq + 300000);
    if (!spin(host, seq + 300000)) { std::fprintf(stderr, "c16: no stamp\n"); std::exit(1); }
  });
  Args1k a{};
  run("d stamp with 1 KB of arguments + spin", [&](uint32_t seq) {
    a.w[seq & 255] = seq;
    hipLaunchKernelGGL(k_stamp_args, dim3(1), dim3(64), 0, st, (volatile uint32_t*)host_dev, seq + 400000, a);
    if (!spin(host, seq + 400000)) { std::fprintf(stderr, "d: no stamp\n"); std::exit(1); }
  });
  run("b'' stamp + spin behind a hipMemcpyAsync H2D 256 B", [&](uint32_t seq) {
    CK(hipMemcpyAsync(cols, host + 512, 256, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, st, (volatile uint32_t*)host_dev, seq + 500000);
    if (!spin(host, seq + 500000)) { std::fprintf(stderr, "b'': no stamp\n"); std::exit(1); }
  });
  // e: resident workgroup on a second stream.
  {
    hipStream_t st2;
    CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    uint32_t* box = host + 256;
    uint32_t* box_dev = host_dev + 256;
    for (int i = 0; i < 32; ++i) box[i] = 0;
    hipLaunchKernelGGL(k_resident, dim3(1), dim3(64), 0, st2, box_dev, 400000000ull);
    us.clear();
    bool ok = true;
    for (int r = -50; r < reps && ok; ++r) {
      const uint32_t seq = (uint32_t)(r + 51);
      auto t0 = Clk::now();
      __atomic_store_n(box, seq, __ATOMIC_RELEASE);
      ok = spin(box + 16, seq);
      auto t1 = Clk::now();
      if (r >= 0) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    __atomic_store_n(box, 0xFFFFFFFFu, __ATOMIC_RELEASE);
    CK(hipStreamSynchronize(st2));
    if (ok) report("e resident workgroup, mailbox round trip", us);
    else std::printf("e resident workgroup: no answer (flag %x)\n", box[17]);
    // e': the same while the first stream runs kernels (does the resident wave starve / get starved?)
    for (int i = 0; i < 32; ++i) box[i] = 0;
    hipLaunchKernelGGL(k_resident, dim3(1), dim3(64), 0, st2, box_dev, 400000000ull);
    us.clear();
    ok = true;
    for (int r = -50; r < reps && ok; ++r) {
      const uint32_t seq = (uint32_t)(r + 51);
      hipLaunchKernelGGL(k_pick<16>, dim3(1), dim3(1024), 0, st, cols, 16384u, (volatile uint32_t*)host_dev, 7u);
      auto t0 = Clk::now();
      __atomic_store_n(box, seq, __ATOMIC_RELEASE);
      ok = spin(box + 16, seq);
      auto t1 = Clk::now();
      if (r >= 0) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    __atomic_store_n(box, 0xFFFFFFFFu, __ATOMIC_RELEASE);
    CK(hipStreamSynchronize(st2));
    CK(hipStreamSynchronize(st));
    if (ok) report("e' the same with kernels on another stream", us);
  }
  return 0;
}
