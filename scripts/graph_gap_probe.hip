// Is the gap between two dependent kernels shorter when the chain is replayed from a hipGraph than when it is launched
// kernel by kernel on a stream?  A chain of chip-filling kernels of a known length each (a spin on the 100 MHz clock) is
// timed both ways; what exceeds links x spin is the boundary cost.
//   hipcc -O3 --offload-arch=gfx950 scripts/graph_gap_probe.hip -o scratch/graph_gap_probe && scratch/graph_gap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (ticks < 0) sink[0] = 1;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int* sink; CK(hipMalloc((void**)&sink, 64));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int links = 100, reps = 20;
  for (int blocks : {256, 4096}) {
    for (long long ticks : {500LL, 2000LL}) {  // 5 us, 20 us
      for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, ticks, sink);
      CK(hipStreamSynchronize(s));
      double t0 = now_us();
      for (int r = 0; r < reps; ++r)
        for (int i = 0; i < links; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, ticks, sink);
      CK(hipStreamSynchronize(s));
      const double per_stream = (now_us() - t0) / (links * reps);
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < links; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, ticks, sink);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      t0 = now_us();
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      const double per_graph = (now_us() - t0) / (links * reps);
      printf("%5d workgroups, spin %5.1f us: per kernel on a stream %6.2f us, from a graph %6.2f us\n", blocks, ticks * 0.01,
             per_stream, per_graph);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
