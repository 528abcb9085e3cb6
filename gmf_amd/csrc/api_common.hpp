// Private to the C-ABI translation units (gmf_api.cpp, gmf_api_train.cpp): the handle, the workspace arena and the
// status helpers.  Not installed; the public interface is include/gmf_hip.h.
#pragma once
#include "../../include/gmf_hip.h"

#include <hip/hip_runtime.h>

#include <mutex>
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
  // caller-provided workspace (gmf_set_workspace): when set, the library allocates nothing; a call that needs more returns
  // GMF_ERR_WORKSPACE and gmf_workspace_wanted() says how much
  bool arena_external = false;
  size_t arena_wanted = 0;
  // calls on one handle are serialised (the workspace of one call is reused by the next); a call on another stream than the
  // previous one first makes its stream wait for the previous stream's work (the two would otherwise share the workspace)
  std::mutex mu;
  hipStream_t last_stream = nullptr;
  bool have_last_stream = false;
  hipEvent_t xs_event = nullptr;
  // host copies of the per-pair tables of ragged calls: a ring of PINNED slots, each guarded by an event recorded behind its
  // upload - a slot is only rewritten once the copy that read it has run (a pageable std::vector handed to
  // hipMemcpyAsync is only safe if the runtime happens to stage it synchronously)
  struct PtabSlot { gmf::PairTab* host = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; };
  static constexpr int kPtabSlots = 8;
  PtabSlot ptab_ring[kPtabSlots];
  int ptab_next = 0, ptab_cur = 0;
  // sticky status word in host-mapped memory (gmf_status_read): kernels OR bits into it through status_dev
  int* status_host = nullptr;
  int* status_dev = nullptr;
  gmf::Tuning tune;   // per-handle knobs (gmf_set_tuning); no process-global state
  const float* sigma_dev = nullptr;   // [ABI 5] gmf_set_sigma_device: PointDSC's learnable sigma read on the device instead of by value
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

// Bump allocator over one device block.  Library-owned (default): growing frees and reallocates (hipFree synchronises the
// device, so no kernel can still be using the old block - and it must not happen inside a stream capture: run every shape
// once before capturing).  Caller-provided (gmf_set_workspace): never allocates; too small -> GMF_ERR_WORKSPACE.
inline int arena_reserve(gmf_handle* h, size_t bytes) {
  h->arena_used = 0;
  if (bytes <= h->arena_bytes) return GMF_OK;
  if (h->arena_external) {
    h->arena_wanted = align_up(bytes + bytes / 8, 1 << 20);
    return fail(h, GMF_ERR_WORKSPACE, "gmf: the caller-provided workspace holds " + std::to_string(h->arena_bytes) + " B, this call needs " +
                                          std::to_string(bytes) + " B (gmf_workspace_wanted, gmf_set_workspace)");
  }
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

// Entered by every entry point after its argument checks: takes the handle's lock, makes the handle's device current for the
// duration of the call and restores the caller's device afterwards, and orders this call's stream behind the previous
// call's when they differ (both use the same workspace).
struct SetDevice {
  gmf_handle* h;
  std::unique_lock<std::mutex> lock;
  int prev = -1;
  explicit SetDevice(gmf_handle* hh) : h(hh), lock(hh->mu) { enter(); }
  SetDevice(gmf_handle* hh, gmf_stream_t stream) : h(hh), lock(hh->mu) {
    enter();
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (h->have_last_stream && st != h->last_stream) {
      if (!h->xs_event) (void)hipEventCreateWithFlags(&h->xs_event, hipEventDisableTiming);
      if (h->xs_event && hipEventRecord(h->xs_event, h->last_stream) == hipSuccess) (void)hipStreamWaitEvent(st, h->xs_event, 0);
    }
    h->last_stream = st;
    h->have_last_stream = true;
  }
  ~SetDevice() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  SetDevice(const SetDevice&) = delete;
  SetDevice& operator=(const SetDevice&) = delete;

 private:
  void enter() {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != h->device) prev = cur;
    if (cur != h->device) (void)hipSetDevice(h->device);
  }
};

