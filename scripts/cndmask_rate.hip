// What does a v_cndmask_b32 cost on gfx950?  (scripts/inst_rate.hip measured 12.6 cycles per wave-instruction at four
// waves per SIMD against 3.3 for a float64 add: a stream of them on eight independent registers, vcc never written.)
// Variants: e32 with vcc, e64 with an SGPR pair, alternating with a compare that writes the mask, and the arithmetic
// stand-ins for "mask ? K : 0".   hipcc -O3 --offload-arch=gfx950 scripts/cndmask_rate.hip -o scratch/cndmask_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
constexpr int kIters = 256;
#define REP8(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#define KERNEL(NAME, PRE, BODY)                                                                         \
  __global__ void NAME(long long* out, int seed, double fseed) {                                         \
    int a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    int b = seed * 3; double f = fseed, g = fseed * 1.5;                                                 \
    PRE                                                                                                  \
    __syncthreads();                                                                                     \
    const long long t0 = __builtin_amdgcn_s_memtime();                                                   \
    for (int it = 0; it < kIters; ++it) { REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) }                  \
    const long long t1 = __builtin_amdgcn_s_memtime();                                                   \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123456789) out[0] = 1;                                  \
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0; \
  }
#define B_E32(A) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(A) : "v"(b) : );
#define B_E64(A) asm volatile("v_cndmask_b32_e64 %0, %1, %0, s[20:21]" : "+v"(A) : "v"(b) : );
#define B_ADD(A) asm volatile("v_add_u32 %0, %1, %0" : "+v"(A) : "v"(b) : );
#define B_CMPSEL(A) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_cndmask_b32 %0, 0, %3, vcc" : "+v"(A) : "v"(f), "v"(g), "v"(b) : "vcc");
#define B_CMPSEL64(A) asm volatile("v_cmp_gt_f64 s[20:21], %1, %2\n v_cndmask_b32_e64 %0, 0, %3, s[20:21]" : "+v"(A) : "v"(f), "v"(g), "v"(b) : "s20", "s21");
#define B_CMP(A) asm volatile("v_cmp_gt_f64 vcc, %1, %2" : "+v"(A) : "v"(f), "v"(g) : "vcc");
#define B_SIGN(A) asm volatile("v_add_f64 %1, %1, -%2\n v_ashrrev_i32 %0, 31, %0\n v_and_b32 %0, %3, %0" : "+v"(A), "+v"(f) : "v"(g), "v"(b) : );
KERNEL(k_e32, asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(seed), "v"(b) : "vcc");, B_E32)
KERNEL(k_e64, asm volatile("v_cmp_gt_i32 s[20:21], %0, %1" :: "v"(seed), "v"(b) : "s20", "s21");, B_E64)
KERNEL(k_add, , B_ADD)
KERNEL(k_cmp, , B_CMP)
KERNEL(k_cmpsel, , B_CMPSEL)
KERNEL(k_cmpsel64, , B_CMPSEL64)
template <class K>
void run(const char* name, K kernel, int per_body, long long* dout) {
  printf("%-44s", name);
  for (int wps : {1, 2, 4}) {
    const int blocks = 256, threads = 64 * 4 * wps;
    std::vector<long long> h(1 + blocks * threads / 64);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, dout, 12345, 1.000001);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), dout, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<long long> t(h.begin() + 1, h.end());
    std::sort(t.begin(), t.end());
    printf("  %dw/SIMD %6.2f", wps, (double)t[t.size() / 2] / (32.0 * kIters * wps * per_body));
  }
  printf("   cycles of one SIMD per wave64 instruction\n");
}
int main() {
  long long* dout;
  hipMalloc(&dout, (1 + 256 * 16) * sizeof(long long));
  hipMemset(dout, 0, (1 + 256 * 16) * sizeof(long long));
  run("v_add_u32", k_add, 1, dout);
  run("v_cndmask_b32 (e32, vcc set once)", k_e32, 1, dout);
  run("v_cndmask_b32_e64 (SGPR pair set once)", k_e64, 1, dout);
  run("v_cmp_gt_f64 vcc", k_cmp, 1, dout);
  run("v_cmp_gt_f64 vcc + v_cndmask (per instr)", k_cmpsel, 2, dout);
  run("v_cmp_gt_f64 s[] + v_cndmask_e64 (per instr)", k_cmpsel64, 2, dout);
  return 0;
}
