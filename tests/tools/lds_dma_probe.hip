// Microbenchmark / semantics check (not part of the library): global -> LDS loads that bypass the
// registers (global_load_lds_dword), as k_walk_groups uses them for the fetch of a class's
// entry-after-next: which LDS word does lane l of wave w write for a given M0, are inactive lanes
// left alone, do offsets above 64 KB work?   hipcc --offload-arch=gfx950 -O2 lds_dma_probe.hip -o lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const uint32_t* p, uint32_t* out, uint32_t stage_words_at) {
  extern __shared__ uint32_t sm[];
  const uint32_t tid = threadIdx.x, wave = tid >> 6;
  for (uint32_t i = tid; i < 40000; i += blockDim.x) sm[i] = 0xDEAD0000u + i;
  __syncthreads();
  uint32_t* stage = sm + stage_words_at;
  const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(stage + wave * 64));
  const uint32_t* a = p + tid * 3;
  if (tid & 1) {  // odd lanes only
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" :: "s"(m0v), "v"(a) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[tid] = stage[tid];
}

int main() {
  const int n = 256;
  std::vector<uint32_t> h(n * 3), o(n);
  for (int i = 0; i < n * 3; ++i) h[i] = 1000 + i;
  uint32_t *dp, *dout;
  hipMalloc(&dp, h.size() * 4);
  hipMalloc(&dout, n * 4);
  hipMemcpy(dp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
  for (uint32_t at : {256u, 20000u, 39000u}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(n), 160000, 0, dp, dout, at);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int ok = 0, untouched = 0, bad = 0;
    for (int i = 0; i < n; ++i) {
      if (i & 1) { if (o[i] == 1000u + i * 3) ++ok; else ++bad; }
      else { if (o[i] == 0xDEAD0000u + at + i) ++untouched; else ++bad; }
    }
    printf("stage at word %u (byte %u): %d odd lanes loaded right, %d even lanes untouched, %d wrong; e.g. out[1]=%u out[65]=%u out[2]=0x%x\n",
           at, at * 4, ok, untouched, bad, o[1], o[65], o[2]);
  }
  return 0;
}
