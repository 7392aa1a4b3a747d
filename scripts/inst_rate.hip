// Issue cost of the instructions the pair loops are made of, on gfx950: cycles of one SIMD per wave64
// instruction, measured with s_memtime around a long unrolled stream of independent instructions at 1, 2 and 4
// waves per SIMD.  Feeds the fp64-issue roofline of bench.py / DESIGN.md.
//   hipcc -O3 --offload-arch=gfx950 scripts/inst_rate.hip -o /tmp/inst_rate && /tmp/inst_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                      \
  do {                                                              \
    hipError_t e = (x);                                             \
    if (e != hipSuccess) {                                          \
      printf("%s: %s\n", #x, hipGetErrorString(e));                 \
      exit(1);                                                      \
    }                                                               \
  } while (0)

constexpr int kChains = 8, kUnroll = 4, kIters = 256;

#define OP8(ASM_D)                                                                                     \
  asm volatile(ASM_D : "+v"(a0) : "v"(b), "v"(c));                                                     \
  asm volatile(ASM_D : "+v"(a1) : "v"(b), "v"(c));                                                     \
  asm volatile(ASM_D : "+v"(a2) : "v"(b), "v"(c));                                                     \
  asm volatile(ASM_D : "+v"(a3) : "v"(b), "v"(c));                                                     \
  asm volatile(ASM_D : "+v"(a4) : "v"(b), "v"(c));                                                     \
  asm volatile(ASM_D : "+v"(a5) : "v"(b), "v"(c));                                                     \
  asm volatile(ASM_D : "+v"(a6) : "v"(b), "v"(c));                                                     \
  asm volatile(ASM_D : "+v"(a7) : "v"(b), "v"(c));

#define KERNEL(NAME, T, ASM_D)                                                                         \
  __global__ void NAME(long long* out, T seed) {                                                       \
    T a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed;          \
    T b = seed, c = seed;                                                                              \
    __syncthreads();                                                                                   \
    const long long t0 = __builtin_amdgcn_s_memtime();                                                 \
    for (int it = 0; it < kIters; ++it) {                                                              \
      OP8(ASM_D) OP8(ASM_D) OP8(ASM_D) OP8(ASM_D)                                                      \
    }                                                                                                  \
    const long long t1 = __builtin_amdgcn_s_memtime();                                                 \
    T s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                       \
    if (s == (T)12345.678) out[0] = 1;                                                                 \
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0; \
  }

KERNEL(k_fma_f64, double, "v_fma_f64 %0, %1, %2, %0")
KERNEL(k_mul_f64, double, "v_mul_f64 %0, %1, %0")
KERNEL(k_add_f64, double, "v_add_f64 %0, %1, %0")
KERNEL(k_min_f64, double, "v_min_f64 %0, %1, %0")
KERNEL(k_rsq_f64, double, "v_rsq_f64 %0, %0")
KERNEL(k_rcp_f64, double, "v_rcp_f64 %0, %0")
KERNEL(k_sqrt_f64, double, "v_sqrt_f64 %0, %0")
KERNEL(k_cmp_f64, double, "v_cmp_gt_f64 vcc, %1, %0")
KERNEL(k_fma_f32, float, "v_fma_f32 %0, %1, %2, %0")
KERNEL(k_rsq_f32, float, "v_rsq_f32 %0, %0")
KERNEL(k_mul_lo_u32, int, "v_mul_lo_u32 %0, %1, %0")
KERNEL(k_mul_hi_u32, int, "v_mul_hi_u32 %0, %1, %0")
KERNEL(k_mul_u24, int, "v_mul_u32_u24 %0, %1, %0")
KERNEL(k_mad_u24, int, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL(k_xor_b32, int, "v_xor_b32 %0, %1, %0")
KERNEL(k_add_u32, int, "v_add_u32 %0, %1, %0")
KERNEL(k_alignbit, int, "v_alignbit_b32 %0, %0, %0, 13")
KERNEL(k_cndmask, int, "v_cndmask_b32 %0, %1, %0, vcc")
KERNEL(k_add3, int, "v_add3_u32 %0, %1, %2, %0")
KERNEL(k_xad, int, "v_xad_u32 %0, %1, %2, %0")

// conversions and 64-bit integer forms need their own operand shapes
__global__ void k_cvt_f64_i32(long long* out, int seed) {
  int b = seed;
  double a0, a1, a2, a3, a4, a5, a6, a7;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters * kUnroll; ++it) {
    asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a0) : "v"(b));
    asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a1) : "v"(b));
    asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a2) : "v"(b));
    asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a3) : "v"(b));
    asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a4) : "v"(b));
    asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a5) : "v"(b));
    asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a6) : "v"(b));
    asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a7) : "v"(b));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s == 12345.678) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ void k_cvt_f32_f64(long long* out, double seed) {
  double b = seed;
  float a0, a1, a2, a3, a4, a5, a6, a7;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters * kUnroll; ++it) {
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a0) : "v"(b));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a1) : "v"(b));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a2) : "v"(b));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a3) : "v"(b));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a4) : "v"(b));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a5) : "v"(b));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a6) : "v"(b));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a7) : "v"(b));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s == 12345.678f) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ void k_mad_u64_u32(long long* out, unsigned seed) {
  unsigned b = seed, c = seed | 1u;
  unsigned long long a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters * kUnroll; ++it) {
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c) : "vcc");
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c) : "vcc");
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c) : "vcc");
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c) : "vcc");
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a4) : "v"(b), "v"(c) : "vcc");
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a5) : "v"(b), "v"(c) : "vcc");
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a6) : "v"(b), "v"(c) : "vcc");
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a7) : "v"(b), "v"(c) : "vcc");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s == 12345678ull) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// packed f32 (two lanes of work per instruction)
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_pk_fma_f32(long long* out, float seed) {
  f2 b = {seed, seed}, c = {seed, seed};
  f2 a0 = b, a1 = b, a2 = b, a3 = b, a4 = b, a5 = b, a6 = b, a7 = b;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters * kUnroll; ++it) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a4) : "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a5) : "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a6) : "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a7) : "v"(b), "v"(c));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s.x + s.y == 12345.678f) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// LDS reads: conflict-free, lane-consecutive addresses
template <int BYTES>
__global__ void k_ds_read(long long* out, int seed) {
  __shared__ double lds[4096];
  for (int k = threadIdx.x; k < 4096; k += blockDim.x) lds[k] = k + seed;
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)lds + (threadIdx.x & 255) * BYTES;
  double a0 = 0, a1 = 0;
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 q0 = {0, 0}, q1 = {0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters * kUnroll; ++it) {
    if (BYTES == 8) {
      asm volatile("ds_read_b64 %0, %1" : "=v"(a0) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(a1) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:4096" : "=v"(a0) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:6144" : "=v"(a1) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:8192" : "=v"(a0) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:10240" : "=v"(a1) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:12288" : "=v"(a0) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:14336" : "=v"(a1) : "v"(addr));
    } else {
      asm volatile("ds_read_b128 %0, %1" : "=v"(q0) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(q1) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(q0) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(q1) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(q0) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:20480" : "=v"(q1) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:24576" : "=v"(q0) : "v"(addr));
      asm volatile("ds_read_b128 %0, %1 offset:28672" : "=v"(q1) : "v"(addr));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (a0 + a1 + q0.x + q1.y == 12345.678) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <class K, class A>
void run(const char* name, K kernel, A seed, long long* dout) {
  printf("%-18s", name);
  for (int wps : {1, 2, 4}) {  // waves per SIMD: one workgroup of 4 * wps waves per CU
    const int threads = 256 * wps, blocks = 256;
    std::vector<long long> h(1 + blocks * threads / 64);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, dout, seed);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h.data(), dout, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    std::vector<long long> t(h.begin() + 1, h.end());
    std::sort(t.begin(), t.end());
    const double cyc = (double)t[t.size() / 2];
    const double per = cyc / ((double)kChains * kUnroll * kIters * wps);  // SIMD cycles per wave-instruction
    printf("  %dw/SIMD %6.2f", wps, per);
  }
  printf("   cycles of one SIMD per wave64 instruction\n");
}

int main() {
  long long* dout;
  CHK(hipMalloc(&dout, (1 + 256 * 16) * sizeof(long long)));
  CHK(hipMemset(dout, 0, (1 + 256 * 16) * sizeof(long long)));
  run("v_fma_f64", k_fma_f64, 1.000001, dout);
  run("v_mul_f64", k_mul_f64, 1.000001, dout);
  run("v_add_f64", k_add_f64, 1.000001, dout);
  run("v_min_f64", k_min_f64, 1.000001, dout);
  run("v_cmp_gt_f64", k_cmp_f64, 1.000001, dout);
  run("v_rsq_f64", k_rsq_f64, 1.000001, dout);
  run("v_rcp_f64", k_rcp_f64, 1.000001, dout);
  run("v_sqrt_f64", k_sqrt_f64, 1.000001, dout);
  run("v_cvt_f64_i32", k_cvt_f64_i32, 12345, dout);
  run("v_cvt_f32_f64", k_cvt_f32_f64, 1.000001, dout);
  run("v_fma_f32", k_fma_f32, 1.000001f, dout);
  run("v_pk_fma_f32", k_pk_fma_f32, 1.000001f, dout);
  run("v_rsq_f32", k_rsq_f32, 1.000001f, dout);
  run("v_mul_lo_u32", k_mul_lo_u32, 12345, dout);
  run("v_mul_hi_u32", k_mul_hi_u32, 12345, dout);
  run("v_mad_u64_u32", k_mad_u64_u32, 12345u, dout);
  run("v_mul_u32_u24", k_mul_u24, 12345, dout);
  run("v_mad_u32_u24", k_mad_u24, 12345, dout);
  run("v_xor_b32", k_xor_b32, 12345, dout);
  run("v_add_u32", k_add_u32, 12345, dout);
  run("v_add3_u32", k_add3, 12345, dout);
  run("v_xad_u32", k_xad, 12345, dout);
  run("v_alignbit_b32", k_alignbit, 12345, dout);
  run("v_cndmask_b32", k_cndmask, 12345, dout);
  run("ds_read_b64", k_ds_read<8>, 1, dout);
  run("ds_read_b128", k_ds_read<16>, 1, dout);
  return 0;
}
