// Wave-wide min / max through DPP row shifts and broadcasts (gfx9 family) against the shuffle tree, on random data.
//   hipcc -O3 --offload-arch=gfx950 -Isand_crate_amd/csrc -Iinclude scripts/dpp_reduce_check.hip -o /tmp/dpp_check && /tmp/dpp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sc_device.h"
__global__ void k(const int* in, int* out) {
  const int v = in[blockIdx.x * blockDim.x + threadIdx.x];
  int mn = v, mx = v;
  for (int o = 32; o > 0; o >>= 1) {
    mn = min(mn, __shfl_xor(mn, o, 64));
    mx = max(mx, __shfl_xor(mx, o, 64));
  }
  const int a = sc::wave_min_all(v), b = sc::wave_max_all(v);
  out[blockIdx.x * blockDim.x + threadIdx.x] = (a == mn && b == mx) ? 0 : 1;
}
int main() {
  const int n = 1 << 20;
  std::vector<int> h(n);
  srand(5);
  for (int i = 0; i < n; ++i) h[i] = (i % 7 == 0) ? (rand() % 3 == 0 ? INT_MAX : -rand()) : rand() - RAND_MAX / 2;
  int *din, *dout;
  hipMalloc(&din, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(din, dout);
  std::vector<int> r(n);
  hipMemcpy(r.data(), dout, n * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < n; ++i) bad += r[i];
  printf("%ld mismatching lanes of %d\n", bad, n);
  return bad != 0;
}
