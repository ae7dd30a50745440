// Device scratch registry: one buffer per (device, stream, key).
//
// Why: the kernels' internal scratch (arrival tickets of the fused attention merge, arg-max slots, the activation image
// and split-K partials of the wide mat-mul path, the dense path's staging buffer) used to be process-wide statics --
// two streams raced on the same tickets, a second device faulted on a buffer allocated on the first, and growing a
// buffer freed memory a captured hipGraph still pointed to.  Rules here:
//   * a buffer belongs to the stream that asked for it (the reference runs one model per process and stream; two models
//     on two streams now get two buffers);
//   * growing never frees: the old allocation is retired (kept until mi355_scratch_release_all), because a graph captured
//     earlier may replay launches that hold its address.  Sizes double, so the retired total stays below the live size;
//   * nothing is allocated while the stream is capturing (hipMalloc would break the capture): the caller gets
//     hipErrorStreamCaptureUnsupported and the host layer's eager warm-up step does the growth.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "scratch.h"

namespace {
struct Buf { void* p = nullptr; size_t bytes = 0; };
std::mutex g_mu;
std::map<std::tuple<int, hipStream_t, int>, Buf> g_live;
std::vector<std::pair<int, void*>> g_retired;          // (device, pointer)
}  // namespace

int mi355_scratch_get(void** out, int key, size_t bytes, hipStream_t st, bool zero_on_create) {
    if (!out) return (int)hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    std::lock_guard<std::mutex> lock(g_mu);
    Buf& b = g_live[std::make_tuple(dev, st, key)];
    if (b.p && b.bytes >= bytes) { *out = b.p; return 0; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return (int)hipErrorStreamCaptureUnsupported;
    const size_t want = b.p ? bytes * 2 : bytes;
    void* p = nullptr;
    e = hipMalloc(&p, want);
    if (e != hipSuccess) return (int)e;
    if (zero_on_create) {
        e = hipMemsetAsync(p, 0, want, st);                 // ordered before the first launch on `st` that uses it
        if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
    }
    if (b.p) g_retired.emplace_back(dev, b.p);
    b.p = p; b.bytes = want;
    *out = p;
    return 0;
}

// Sticky error of the `void` entry points (the reference's FFI symbols return nothing, src/backend/gptq.rs:115-194,
// cache.rs:127-162): the first failure since the last clear is kept until the host reads it.
namespace { std::atomic<int> g_last_error{0}; }
void mi355_note_error(int code) {
    int expect = 0;
    if (code) g_last_error.compare_exchange_strong(expect, code);
}
extern "C" int mi355_last_error(void) { return g_last_error.load(); }
extern "C" void mi355_clear_error(void) { g_last_error.store(0); }

extern "C" void mi355_scratch_release_all(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    (void)hipDeviceSynchronize();
    for (auto& kv : g_live) if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& r : g_retired) (void)hipFree(r.second);
    g_live.clear();
    g_retired.clear();
}
