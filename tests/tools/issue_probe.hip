// Developer probe: what one instruction of the matching loop costs on gfx950 — the patterns the
// loop of match_kernel.h is made of (dependent VALU, DPP chains with their wait states, the
// VALU -> SGPR -> SALU -> branch hops, LDS round trips), for one wave alone on the chip and for
// 1 / 2 / 3 waves on every SIMD. Prints ns and shader cycles (at the measured clock) per pattern.
//   hipcc --offload-arch=gfx950 -O3 tests/tools/issue_probe.hip -o /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__device__ __forceinline__ uint64_t wall() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz

#define KERNEL(name, body, per_rep)                                                        \
  __global__ __launch_bounds__(64) void name(uint32_t iters, uint64_t* out, uint32_t* sink) { \
    extern __shared__ uint32_t lds[];                                                      \
    lds[threadIdx.x] = threadIdx.x << 2;                                                        \
    __syncthreads();                                                                       \
    uint32_t v = threadIdx.x + 1, a = threadIdx.x << 2;                                    \
    const uint64_t t0 = wall();                                                            \
    for (uint32_t it = 0; it < iters; ++it) {                                              \
      asm volatile(REP64(body)                                                             \
                   : "+v"(v), "+v"(a)                                                      \
                   :                                                                       \
                   : "v40", "v41", "v42", "v43", "v44", "v45", "s40", "s41", "s42", "s43", "s44",   \
                     "s45", "s46", "s47", "vcc", "scc", "memory");                         \
    }                                                                                      \
    const uint64_t t1 = wall();                                                            \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                             \
    if (v == 0xdeadbeef) sink[0] = v + a;                                                  \
  }                                                                                        \
  static const int name##_per_rep = per_rep;

// (instructions per repetition in the last argument)
KERNEL(k_valu_dep, "v_add_u32 %0, 1, %0\n", 1)
KERNEL(k_valu_indep, "v_add_u32 v40, 1, %0\nv_add_u32 v41, 1, %0\nv_add_u32 v42, 1, %0\nv_add_u32 v43, 1, %0\n", 4)
KERNEL(k_salu_dep, "s_add_u32 s40, s40, 1\n", 1)
KERNEL(k_salu_indep, "s_add_u32 s40, s44, 1\ns_add_u32 s41, s44, 1\ns_add_u32 s42, s44, 1\ns_add_u32 s43, s44, 1\n", 4)
KERNEL(k_nop0, "s_nop 0\n", 1)
KERNEL(k_nop1, "s_nop 1\n", 1)
KERNEL(k_valu_salu_mix, "v_add_u32 v40, 1, %0\ns_add_u32 s40, s44, 1\n", 2)
// One DPP chain with its wait states (the single step's reduction, 5 steps).
KERNEL(k_dpp_chain,
       "v_max_u32_dpp v40, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\ns_nop 1\n"
       "v_max_u32_dpp v40, v40, v40 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\ns_nop 1\n"
       "v_max_u32_dpp v40, v40, v40 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\ns_nop 1\n"
       "v_max_u32_dpp v40, v40, v40 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\ns_nop 1\n"
       "v_max_u32_dpp v40, v40, v40 row_bcast:15 row_mask:0xa bank_mask:0xf bound_ctrl:0\ns_nop 1\n", 10)
// Two interleaved chains (the pair's reduction).
KERNEL(k_dpp_pair,
       "v_max_u32_dpp v40, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\ns_nop 0\n"
       "v_max_u32_dpp v41, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
       "v_max_u32_dpp v40, v40, v40 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\ns_nop 0\n"
       "v_max_u32_dpp v41, v41, v41 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
       "v_max_u32_dpp v40, v40, v40 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\ns_nop 0\n"
       "v_max_u32_dpp v41, v41, v41 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
       "v_max_u32_dpp v40, v40, v40 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\ns_nop 0\n"
       "v_max_u32_dpp v41, v41, v41 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
       "v_max_u32_dpp v40, v40, v40 row_bcast:15 row_mask:0xa bank_mask:0xf bound_ctrl:0\ns_nop 0\n"
       "v_max_u32_dpp v41, v41, v41 row_bcast:15 row_mask:0xa bank_mask:0xf bound_ctrl:0\n", 15)
// VALU -> SGPR -> SALU compare -> branch (not taken).
KERNEL(k_readlane_cmp_branch,
       "v_readlane_b32 s40, %0, 31\ns_cmp_eq_u32 s40, 0\ns_cbranch_scc1 1f\n1:\n", 3)
// The tail of a step: winner compare, exec = winner, advance from registers + LDS read, exec back.
KERNEL(k_tail,
       "v_readlane_b32 s40, %0, 31\n"
       "s_cmp_eq_u32 s40, 0\ns_cbranch_scc1 2f\n"
       "v_cmp_eq_u32 vcc, s40, %0\n"
       "s_waitcnt lgkmcnt(0)\n"
       "s_mov_b64 exec, vcc\n"
       "v_mov_b32 v41, v42\n"
       "v_add_u32 v43, 4, %1\n"
       "v_bfi_b32 v44, s44, v43, %1\n"
       "ds_read_b32 v42, %1\n"
       "s_mov_b64 exec, -1\n"
       "v_writelane_b32 v45, s40, 3\n2:\n", 12)
// SGPR written by v_readlane, used as a lane mask two wait states later.
KERNEL(k_readlane_cndmask,
       "v_readlane_b32 s40, %0, 5\ns_mov_b32 s41, 0\ns_nop 0\nv_cndmask_b32 v40, 0, %0, s[40:41]\n", 4)
KERNEL(k_lds_roundtrip, "ds_read_b32 %1, %1\ns_waitcnt lgkmcnt(0)\n", 2)
KERNEL(k_bitcmp_branch, "s_bitcmp1_b64 s[44:45], 3\ns_cbranch_scc1 3f\n3:\n", 2)
KERNEL(k_writelane, "v_writelane_b32 v45, s44, 3\n", 1)

template <typename K>
static void run(const char* name, K kernel, int per_rep, uint64_t* d, uint32_t* sink, double mhz) {
  const uint32_t iters = 400;
  printf("%-24s", name);
  for (uint32_t wps : {0u, 1u, 2u, 3u}) {
    const uint32_t grid = wps == 0 ? 1 : 256 * 4 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 1024, 0, iters, d, sink);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
    }
    uint64_t h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double insts = (double)iters * 64 * per_rep;
    const double ns_wave = (double)h * 10.0 / insts;  // per instruction as the wave sees it
    const double ns_simd = wps ? ns_wave / wps : ns_wave;  // per instruction of the SIMD's stream
    printf(" | %u w/SIMD: %6.2f ns/inst (%5.1f cyc), SIMD %5.2f ns", wps, ns_wave, ns_wave * mhz / 1e3,
           ns_simd);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  printf("\n");
}

__global__ void k_clock(uint64_t* out) {
  const uint64_t w0 = wall();
  const uint64_t c0 = __builtin_readcyclecounter();
  while (wall() - w0 < 100000) {}  // 1 ms
  const uint64_t c1 = __builtin_readcyclecounter();
  const uint64_t w1 = wall();
  out[0] = c1 - c0;
  out[1] = w1 - w0;
}

int main() {
  uint64_t* d;
  uint32_t* sink;
  hipMalloc(&d, 64);
  hipMalloc(&sink, 64);
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(1), 0, 0, d);
  uint64_t h[2];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  // s_memtime against the 100 MHz wall clock: the shader clock while one thread spins.
  const double mhz = (double)h[0] / ((double)h[1] / 100.0);
  printf("s_memtime / wall clock while one wave spins: %.0f MHz\n", mhz);
  printf("(0 w/SIMD = one wave on the whole chip; 'cyc' at that clock)\n");
#define RUN(k) run(#k, k, k##_per_rep, d, sink, mhz)
  RUN(k_valu_dep);
  RUN(k_valu_indep);
  RUN(k_salu_dep);
  RUN(k_salu_indep);
  RUN(k_nop0);
  RUN(k_nop1);
  RUN(k_valu_salu_mix);
  RUN(k_dpp_chain);
  RUN(k_dpp_pair);
  RUN(k_readlane_cmp_branch);
  RUN(k_tail);
  RUN(k_readlane_cndmask);
  RUN(k_lds_roundtrip);
  RUN(k_bitcmp_branch);
  RUN(k_writelane);
  return 0;
}
