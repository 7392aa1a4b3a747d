// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 in THIS code's access width:
// a float64 copy, 8 bytes per lane, coalesced, over buffers far larger than the 256 MiB Infinity
// Cache.  MI355X_MICROARCH.md (HBM section): FETCH_SIZE is known to read 1/2 of a 16-B-per-lane
// stream; other widths must be calibrated on a known byte count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void copy_f64(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
int main() {
  const size_t n = (size_t)1 << 27;  // 1 GiB read + 1 GiB written per launch
  double *a, *b;
  if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) return 1;
  hipMemset(a, 0, n * 8);
  for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(copy_f64, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n);
  hipDeviceSynchronize();
  printf("bytes_read_per_launch %zu bytes_written_per_launch %zu\n", n * 8, n * 8);
  return 0;
}
