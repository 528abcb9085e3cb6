// Private to the C-ABI translation units (gmf_api.cpp, gmf_api_train.cpp): the handle, the workspace arena and the
// status helpers.  Not installed; the public interface is include/gmf_hip.h.
#pragma once
#include "../../include/gmf_hip.h"

#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "launchers.hpp"

struct gmf_handle {
  int device = 0;
  std::string err;
  void* arena = nullptr;
  size_t arena_bytes = 0;
  size_t arena_used = 0;
  gmf::Tuning tune;   // per-handle knobs (gmf_set_tuning); no process-global state
  // optional in-situ timing of the dominant kernel (k_scattn): event pairs recorded on the caller's stream
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
};


constexpr int kC = 128;
constexpr size_t kTileFloats = 32 * kC;

inline int fail(gmf_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

inline int hip_fail(gmf_handle* h, hipError_t e, const char* where) {
  return fail(h, GMF_ERR_HIP, std::string(where) + ": " + hipGetErrorString(e));
}

#define GMF_HIP(call)                                        \
  do {                                                       \
    hipError_t _e = (call);                                  \
    if (_e != hipSuccess) return hip_fail(h, _e, #call);     \
  } while (0)

#define GMF_REQUIRE(cond, code, msg)                          \
  do {                                                        \
    if (!(cond)) return fail(h, code, std::string("gmf: ") + msg); \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over one device block.  Growing frees and reallocates (hipFree synchronises the
// device, so no kernel can still be using the old block).
inline int arena_reserve(gmf_handle* h, size_t bytes) {
  h->arena_used = 0;
  if (bytes <= h->arena_bytes) return GMF_OK;
  if (h->arena) {
    hipError_t e = hipFree(h->arena);
    h->arena = nullptr;
    h->arena_bytes = 0;
    if (e != hipSuccess) return hip_fail(h, e, "hipFree(workspace)");
  }
  const size_t want = align_up(bytes + bytes / 8, 1 << 20);
  hipError_t e = hipMalloc(&h->arena, want);
  if (e != hipSuccess) {
    h->arena = nullptr;
    return fail(h, GMF_ERR_OOM, std::string("hipMalloc(workspace ") + std::to_string(want) + " B): " + hipGetErrorString(e));
  }
  h->arena_bytes = want;
  return GMF_OK;
}

template <typename T>
T* arena_take(gmf_handle* h, size_t count) {
  const size_t off = align_up(h->arena_used, 256);
  h->arena_used = off + count * sizeof(T);
  return reinterpret_cast<T*>(static_cast<char*>(h->arena) + off);
}

inline size_t arena_need(size_t count, size_t elem) { return align_up(count * elem, 256) + 256; }

inline hipStream_t S(gmf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int tiles_of(int n) { return (n + 31) / 32; }

struct SetDevice {
  gmf_handle* h;
  explicit SetDevice(gmf_handle* hh) : h(hh) { (void)hipSetDevice(hh->device); }
};


