// Does hipStreamWaitValue32 work here, and how soon after the write does the stream go on?
// Result on the MI355X pool (ROCm 7.2): signal memory -- the only kind of memory the call is specified for -- cannot be
// allocated (hipExtMallocWithFlags(hipMallocSignalMemory): invalid argument).  Round 2 also tried the call on hipMalloc
// and on host-mapped memory, with the wait enqueued BEFORE the writing kernel and the full mask: that wait never
// returned and the run had to be killed.  Those two cases are outside the call's contract (the command processor's
// wait packet polls memory through its own path, which a kernel's store to coarse-grained or host-mapped memory is
// not coherent with while kernels run), they wedge a queue for good, and they are no longer in this probe: it prints
// the device's own answer (hipDeviceAttributeCanUseStreamWaitValue) and tries signal memory only.
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
  CK(hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory));
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
  printf("signal memory: waiter ran %.1f us after the write, %.1f us before the writer ended\n", (h[1] - h[0]) / 100.0,
         (h[2] - h[1]) / 100.0);
  return 0;
}
int main() {
  int can = -1;
  (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  return run(0);
}
