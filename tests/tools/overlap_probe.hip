// Does a one-wave kernel on a second stream start while a chip-filling kernel of the first stream
// is still running — and how long after the event it waits for? (The question behind running a
// sequential walk BESIDE the parallel matching pass: DESIGN.md 9.3.)
// build: hipcc --offload-arch=gfx950 -O2 -o tests/tools/overlap_probe tests/tools/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t wall() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ uint64_t wall100() { uint64_t t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

// `front`: a short kernel (the sort's last pass, say). `big`: n workgroups of one wave, each busy for `us`.
__global__ void k_front(uint64_t* stamps) { if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = wall100(); }
__global__ void k_big(uint64_t* stamps, uint32_t ticks, volatile uint32_t* flag, uint32_t seq, uint32_t* seen) {
  const uint64_t t0 = wall100();
  if (blockIdx.x == 0 && threadIdx.x == 0) stamps[1] = t0;
  // workgroup 7 waits for the side kernel's word (bounded), the others just stay busy
  if (blockIdx.x == 7 && threadIdx.x == 0) {
    uint32_t ok = 0;
    while (wall100() - t0 < ticks) {
      if (__atomic_load_n((uint32_t*)flag, __ATOMIC_RELAXED) == seq) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    stamps[4] = wall100();
    *seen = ok;
  }
  while (wall100() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) stamps[2] = wall100();
}
__global__ void k_side(uint64_t* stamps, uint32_t* flag, uint32_t seq) {
  if (threadIdx.x == 0) {
    stamps[3] = wall100();
    __atomic_store_n(flag, seq, __ATOMIC_RELAXED);
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  const uint32_t big_us = argc > 2 ? atoi(argv[2]) : 150;
  const uint32_t waves = argc > 3 ? atoi(argv[3]) : 1954;
  const int mode = argc > 4 ? atoi(argv[4]) : 3;  // bit 0: the side kernel and its events, bit 1: the join behind the big kernel
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev_front, ev_side;
  CK(hipEventCreateWithFlags(&ev_front, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
  uint64_t* stamps; uint32_t *flag, *seen;
  CK(hipHostMalloc(&stamps, 64)); CK(hipMalloc(&flag, 4)); CK(hipHostMalloc(&seen, 4));
  CK(hipMemset(flag, 0, 4));
  std::vector<double> side_after_front, side_after_big, big_len, seen_at, big_after_front, next_after_big;
  int seen_n = 0;
  for (int r = 0; r < reps + 10; ++r) {
    hipLaunchKernelGGL(k_front, dim3(256), dim3(256), 0, s1, stamps);
    if (mode & 1) {
      CK(hipEventRecord(ev_front, s1));
      CK(hipStreamWaitEvent(s2, ev_front, 0));
      hipLaunchKernelGGL(k_side, dim3(1), dim3(64), 16384, s2, stamps, flag, (uint32_t)r + 1);
      CK(hipEventRecord(ev_side, s2));
    }
    hipLaunchKernelGGL(k_big, dim3(waves), dim3(64), 16384, s1, stamps, big_us * 100, flag, (uint32_t)r + 1, seen);
    if ((mode & 3) == 3) CK(hipStreamWaitEvent(s1, ev_side, 0));
    hipLaunchKernelGGL(k_front, dim3(1), dim3(64), 0, s1, stamps + 5);
    CK(hipStreamSynchronize(s1));
    CK(hipStreamSynchronize(s2));
    if (r >= 10) {
      side_after_front.push_back(((double)stamps[3] - (double)stamps[0]) / 100.0);
      side_after_big.push_back(((double)stamps[3] - (double)stamps[1]) / 100.0);
      big_len.push_back(((double)stamps[2] - (double)stamps[1]) / 100.0);
      seen_at.push_back(((double)stamps[4] - (double)stamps[1]) / 100.0);
      seen_n += *seen;
      big_after_front.push_back(((double)stamps[1] - (double)stamps[0]) / 100.0);
      next_after_big.push_back(((double)stamps[5] - (double)stamps[2]) / 100.0);
    }
  }
  auto pr = [](const char* what, std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    printf("%-64s p50 %8.2f  p90 %8.2f  p99 %8.2f  max %8.2f us\n", what, v[v.size() / 2], v[v.size() * 9 / 10],
           v[v.size() * 99 / 100], v.back());
  };
  printf("%d rounds: a front kernel, then %u one-wave workgroups busy for %u us on stream 1; a one-wave kernel on stream 2 "
         "behind an event recorded after the front\n", reps, waves, big_us);
  pr("side kernel's start after the front kernel's start", side_after_front);
  pr("side kernel's start after the big kernel's start (< 0: before it)", side_after_big);
  pr("the big kernel: first wave in -> last wave out", big_len);
  pr("workgroup 7 of the big kernel sees the side kernel's word after", seen_at);
  pr("the big kernel starts after the front kernel has started", big_after_front);
  pr("the kernel behind the big one starts after its last wave", next_after_big);
  printf("mode %d; seen in %d of %d rounds\n", mode, seen_n, reps);
  return 0;
}
