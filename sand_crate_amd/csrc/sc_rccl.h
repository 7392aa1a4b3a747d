// RCCL transport of the halo exchange (send/recv pairs to the left and right
// neighbor inside one group, on the context's stream).  Included once by sandcrate_hip.hip.
//
// librccl is NOT a link-time dependency: the single-GPU path must load without it.  It is dlopen()ed
// on first use -- the copy already in the process if there is one (torch brings its own), else the given
// path, else the default search path -- and only the eight entry points below are looked up.  Types are
// restated from rccl.h (NCCL 2.x ABI): ncclComm_t is an opaque pointer, ncclUniqueId is 128 bytes passed
// by value, ncclDouble = 8, ncclSuccess = 0.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

namespace sc {

struct RcclUniqueId {
  char internal[128];
};
typedef void* RcclComm;
constexpr int kRcclDouble = 8;

struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string origin, error;
};

inline RcclApi& rccl_api() {
  static RcclApi api;
  return api;
}

// 0 on success; the reason stays in rccl_api().error
inline int rccl_load(const char* path) {
  RcclApi& a = rccl_api();
  if (a.handle) return 0;
  const char* tried[4] = {nullptr, nullptr, nullptr, nullptr};
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (h) a.origin = "librccl.so.1 (already loaded)";
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (h && a.origin.empty()) a.origin = "librccl.so (already loaded)";
  if (!h && path && *path) {
    h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (h) a.origin = path;
  }
  if (!h) {
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (h) a.origin = "librccl.so.1";
  }
  if (!h) {
    h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) a.origin = "librccl.so";
  }
  (void)tried;
  if (!h) {
    const char* e = dlerror();
    a.error = std::string("librccl not found: ") + (e ? e : "?");
    return -1;
  }
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(h, name);
    if (!p && a.error.empty()) a.error = std::string("librccl has no ") + name;
    return p;
  };
  a.error.clear();
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
  a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
  a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
  a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
  a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
  if (!a.error.empty()) return -1;
  a.handle = h;
  return 0;
}

inline const char* rccl_error(int rc) {
  RcclApi& a = rccl_api();
  return a.GetErrorString ? a.GetErrorString(rc) : "?";
}

}  // namespace sc
