// Does hipStreamWaitValue32 work here, on which kinds of memory, and how soon after the write does the stream go on?
// Result on the MI355X pool (ROCm 7.2): signal memory cannot be allocated (hipExtMallocWithFlags: invalid argument);
// on hipMalloc memory the wait NEVER returns -- the run had to be killed.  So by default only the signal-memory case
// is tried; `./wait_value_probe all` repeats the other two (run it under `timeout -k 5 60`).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin_then_write(int* flag, long long* stamp, long long cycles, int value) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  stamp[0] = wall_clock64();
  __threadfence_system();
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // keep running for a while: the waiter must get going while this kernel is still alive
  const long long t1 = wall_clock64();
  while (wall_clock64() - t1 < cycles) { }
  stamp[2] = wall_clock64();
}
__global__ void after(long long* stamp) { stamp[1] = wall_clock64(); }
int run(int kind) {
  int* flag = nullptr;
  if (kind == 0) CK(hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory));
  else if (kind == 1) CK(hipMalloc((void**)&flag, 64));
  else CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped));
  CK(hipMemset(flag, 0, 64));
  long long* stamp; CK(hipMalloc((void**)&stamp, 64)); CK(hipMemset(stamp, 0, 64));
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  hipError_t e = hipStreamWaitValue32(b, flag, 7, hipStreamWaitValueGte, 0xFFFFFFFFu);
  if (e != hipSuccess) { printf("kind %d: hipStreamWaitValue32 -> %s\n", kind, hipGetErrorString(e)); return 0; }
  hipLaunchKernelGGL(after, dim3(1), dim3(1), 0, b, stamp);
  hipLaunchKernelGGL(spin_then_write, dim3(1), dim3(1), 0, a, flag, stamp, 5000000LL /* 50 ms at 100 MHz */, 7);
  CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
  long long h[3]; CK(hipMemcpy(h, stamp, sizeof h, hipMemcpyDeviceToHost));
  printf("kind %d (%s): waiter ran %.1f us after the write, %.1f us before the writer ended\n", kind,
         kind == 0 ? "signal memory" : kind == 1 ? "hipMalloc" : "host mapped", (h[1] - h[0]) / 100.0, (h[2] - h[1]) / 100.0);
  return 0;
}
int main(int argc, char** argv) {
  const int last = argc > 1 ? 3 : 1;
  for (int k = 0; k < last; ++k)
    if (run(k)) return 1;
  return 0;
}
