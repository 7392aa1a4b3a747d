// Dependent-issue latency of the instructions on the pair loops' critical path (gfx950): one wave per SIMD, ONE
// dependency chain, s_memtime around 4096 instructions.   hipcc -O3 --offload-arch=gfx950 scripts/inst_latency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define KERNEL(NAME, T, ASM)                                                                    \
  __global__ void NAME(long long* out, T seed) {                                                \
    T a = seed, b = seed, c = seed;                                                             \
    const long long t0 = __builtin_amdgcn_s_memtime();                                          \
    for (int it = 0; it < 256; ++it) {                                                          \
      asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
      asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
      asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
      asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
      asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
      asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
      asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
      asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
    }                                                                                           \
    const long long t1 = __builtin_amdgcn_s_memtime();                                          \
    if (a == (T)12345) out[0] = 1;                                                              \
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0; \
  }
KERNEL(k_fma_f64, double, "v_fma_f64 %0, %0, %1, %2")
KERNEL(k_mul_f64, double, "v_mul_f64 %0, %0, %1")
KERNEL(k_add_f64, double, "v_add_f64 %0, %0, %1")
KERNEL(k_rsq_f64, double, "v_rsq_f64 %0, %0")
KERNEL(k_fma_f32, float, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_mul_lo, int, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_xor, int, "v_xor_b32 %0, %0, %1")
KERNEL(k_add3, int, "v_add3_u32 %0, %0, %1, %2")
__global__ void k_cvt(long long* out, int seed) {
  int a = seed; double d = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 2048; ++it) {
    asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d) : "v"(a));
    asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a) : "v"(d));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (a == 12345) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <class K, class A> void run(const char* name, K k, A seed, long long* d) {
  printf("%-14s", name);
  for (int wps : {1, 2, 4}) {
    std::vector<long long> h(1 + 256 * 4 * wps);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(256 * wps), 0, 0, d, seed);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<long long> t(h.begin() + 1, h.end());
    std::sort(t.begin(), t.end());
    printf("  %dw/SIMD: %6.2f per instr (one wave's chain)", wps, (double)t[t.size() / 2] / 4096.0);
  }
  printf("\n");
}
int main() {
  long long* d; hipMalloc(&d, (1 + 256 * 16) * sizeof(long long));
  run("v_fma_f64", k_fma_f64, 1.0000001, d); run("v_mul_f64", k_mul_f64, 1.0000001, d); run("v_add_f64", k_add_f64, 1.0000001, d);
  run("v_rsq_f64", k_rsq_f64, 1.0000001, d); run("v_fma_f32", k_fma_f32, 1.0000001f, d); run("v_mul_lo_u32", k_mul_lo, 12345, d);
  run("v_xor_b32", k_xor, 12345, d); run("v_add3_u32", k_add3, 12345, d); run("cvt f64<->u32", k_cvt, 12345, d);
  return 0;
}
