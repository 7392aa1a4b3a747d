// k_scan_cells on its own: random cell counts, some of them big buckets, many blocks; checks the bucket starts against a
// host prefix sum and the task list against the buckets.  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17
//   -Iinclude -Isand_crate_amd/csrc scripts/scan_check.hip -o scratch/scan_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include "sc_device.h"
#include "sc_kernels.h"
using namespace sc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  for (int n : {1000, 300000, 1200000, 5000000}) {
    for (int bigs : {0, 5, 3000}) {
      std::vector<int> cnt(n + 1, 0);
      srand(n + bigs);
      for (int i = 0; i < n; ++i) cnt[i] = rand() % 9;
      for (int b = 0; b < bigs; ++b) cnt[rand() % n] = 97 + rand() % 4000;
      const int nb = (n + 1 + kScanPerBlock - 1) / kScanPerBlock;
      int *in, *out, *counters; unsigned long long* desc; int2* tasks;
      CK(hipMalloc((void**)&in, (n + 1) * 4)); CK(hipMalloc((void**)&out, (n + 2) * 4)); CK(hipMalloc((void**)&counters, C_ALLOC * 4));
      CK(hipMalloc((void**)&desc, (nb + 4) * 8)); CK(hipMalloc((void**)&tasks, kMaxSortTasks * 8));
      CK(hipMemset(desc, 0, (nb + 4) * 8)); CK(hipMemset(tasks, 0xFF, kMaxSortTasks * 8));
      CK(hipMemcpy(in, cnt.data(), (n + 1) * 4, hipMemcpyHostToDevice));
      for (unsigned stamp = 1; stamp <= 3; ++stamp) {
        CK(hipMemset(counters, 0, C_ALLOC * 4));
        hipLaunchKernelGGL(k_scan_cells, dim3(nb), dim3(kBlock), 0, 0, in, out, n, desc, stamp, counters, tasks, kScanMaxPolls);
        CK(hipDeviceSynchronize());
      }
      std::vector<int> got(n + 1), ctr(C_ALLOC); std::vector<int2> tk(kMaxSortTasks);
      CK(hipMemcpy(got.data(), out, (n + 1) * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ctr.data(), counters, C_ALLOC * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(tk.data(), tasks, kMaxSortTasks * 8, hipMemcpyDeviceToHost));
      long long run = 0; int bad = 0, nbig = 0, ntasks = 0;
      std::map<long long, int> want;  // (cell, chunk) -> length
      for (int i = 0; i <= n; ++i) {
        if (got[i] != run && bad++ < 3) printf("  start[%d] = %d, expected %lld\n", i, got[i], run);
        if (i < n) {
          run += cnt[i];
          if (cnt[i] > kSortThreshold) { ++nbig; for (int j = 0; j * kSortChunk < cnt[i]; ++j) { want[(long long)i << 20 | j] = std::min(kSortChunk, cnt[i] - j * kSortChunk); ++ntasks; } }
        }
      }
      int tbad = 0;
      if (ctr[C_NBIG] != nbig || ctr[C_NTASKS] != ntasks || ctr[C_NT] != run) { printf("  counters: big %d (%d) tasks %d (%d) total %d (%lld) flags %d\n", ctr[C_NBIG], nbig, ctr[C_NTASKS], ntasks, ctr[C_NT], run, ctr[C_FLAGS]); ++tbad; }
      if (ntasks <= kMaxSortTasks) {
        for (int t = 0; t < ntasks; ++t) {
          const long long key = (long long)tk[t].x << 20 | (tk[t].y & 0xFFFFF);
          auto it = want.find(key);
          if (it == want.end() || it->second != (tk[t].y >> 20) + 1) { if (tbad++ < 3) printf("  task %d: cell %d chunk %d len %d unexpected\n", t, tk[t].x, tk[t].y & 0xFFFFF, (tk[t].y >> 20) + 1); }
          else want.erase(it);
        }
        if (!want.empty()) { printf("  %zu tasks missing\n", want.size()); ++tbad; }
      }
      printf("%8d cells, %5d big buckets, %d blocks: starts %s, tasks %s (%d)\n", n, nbig, nb, bad ? "WRONG" : "ok", tbad ? "WRONG" : "ok", ntasks);
      hipFree(in); hipFree(out); hipFree(counters); hipFree(desc); hipFree(tasks);
    }
  }
  return 0;
}
