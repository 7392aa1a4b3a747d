// Returning atomics on a handful of hot counters: does it matter whether the counters share a cache line?
//   hipcc -O3 --offload-arch=gfx950 scripts/atomic_lines.hip -o /tmp/atomic_lines && /tmp/atomic_lines
// Every wave's leader lanes (25 per wave, like a scrambled wave of the pile-up regime) add to one of K hot counters,
// spaced `stride` ints apart; the kernel ends when the last atomic has returned.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void hammer(int* __restrict__ ctr, int K, int stride, int per_wave, int* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int got = 0;
  if (lane < per_wave) got = atomicAdd(&ctr[((wave * 7 + lane) % K) * stride], 1);
  if (got == -12345) sink[0] = got;
}
int main() {
  int *ctr, *sink;
  hipMalloc(&ctr, 1 << 24);
  hipMalloc(&sink, 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int waves = 16384, per_wave = 25;
  for (int K : {1, 32, 512})
    for (int stride : {1, 32, 1024}) {
      hipMemset(ctr, 0, 1 << 24);
      hammer<<<waves / 4, 256>>>(ctr, K, stride, per_wave, sink);  // warm
      hipDeviceSynchronize();
      hipEventRecord(a);
      for (int r = 0; r < 10; ++r) hammer<<<waves / 4, 256>>>(ctr, K, stride, per_wave, sink);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("%4d hot counters, %5d B apart: %7.1f us per launch, %6.2f ns per atomic (%d atomics)\n", K, stride * 4, ms * 100, ms * 1e5 / (waves * per_wave), waves * per_wave);
    }
  return 0;
}
