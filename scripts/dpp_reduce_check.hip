// Wave-wide min / max and the xor-lane exchanges on the DPP path (gfx9 family) against the shuffles, on random data.
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
  bool ok = a == mn && b == mx;
  ok = ok && sc::xor_lane<1>(v) == __shfl_xor(v, 1, 64) && sc::xor_lane<2>(v) == __shfl_xor(v, 2, 64) && sc::xor_lane<4>(v) == __shfl_xor(v, 4, 64) &&
       sc::xor_lane<8>(v) == __shfl_xor(v, 8, 64) && sc::xor_lane<16>(v) == __shfl_xor(v, 16, 64) && sc::xor_lane<32>(v) == __shfl_xor(v, 32, 64);
  {
    const int small = v & 1023;
    int incl = small;
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if ((int)(threadIdx.x & 63) >= o) incl += u;
    }
    ok = ok && sc::wave_scan_add(small) == incl;
  }
  const double dv = (double)v * 1.0000001;
  ok = ok && sc::xor_lane<4>(dv) == __shfl_xor(dv, 4, 64) && sc::xor_lane<8>(dv) == __shfl_xor(dv, 8, 64);
  out[blockIdx.x * blockDim.x + threadIdx.x] = ok ? 0 : 1;
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
