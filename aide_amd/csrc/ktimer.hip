// Optional per-kernel timing (bench.py's `roofline` object: the MFMA convolution kernels; its `critical` and `streaming`
// blocks: every kernel family of the step with its launch stream).
//
// When a timer is armed for a kernel family, its launcher passes a start/stop event pair to hipExtLaunchKernelGGL:
// the pair then carries the DISPATCH's own begin / end timestamps (what rocprofv3 --kernel-trace reports for the
// kernel), with no extra marker packets in the queue -- so every launch of every timed step can be measured under the
// real two-stream schedule without perturbing it.  Events are created once, before the timed region.
#include "common.h"
#include <vector>

namespace {
struct KTimer {
    unsigned mask = 0;
    size_t used = 0;
    long dropped = 0;
    std::vector<hipEvent_t> ev;        // 2 per slot
    std::vector<int> fam;
    std::vector<double> flops;
    std::vector<hipStream_t> stream;
} T;
}  // namespace

extern "C" {

int aide_ktimer_slot(int family, double flops, hipStream_t stream, hipEvent_t* e0, hipEvent_t* e1) {
    if (!(T.mask >> family & 1u)) return 0;
    if (T.used >= T.fam.size()) { ++T.dropped; return 0; }
    const size_t s = T.used++;
    T.fam[s] = family; T.flops[s] = flops; T.stream[s] = stream;
    *e0 = T.ev[2 * s]; *e1 = T.ev[2 * s + 1];
    return 1;
}

// arm the timer for the families in `family_mask` (bit = AIDE_KT_* id) with room for `capacity` launches
int aide_ktimer_start(int family_mask, int capacity) {
    if (capacity < 0) return AIDE_ERR_ARG;
    T.mask = 0;
    while (T.ev.size() < 2 * (size_t)capacity) {
        hipEvent_t e;
        const hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) return (int)rc;
        T.ev.push_back(e);
    }
    T.fam.assign(capacity, 0); T.flops.assign(capacity, 0.0); T.stream.assign(capacity, (hipStream_t)0);
    T.used = 0; T.dropped = 0;
    T.mask = (unsigned)family_mask;
    return AIDE_OK;
}

int aide_ktimer_stop(void) { T.mask = 0; return AIDE_OK; }

// re-arm after aide_ktimer_stop without touching the recorded slots (start = create + reset + arm)
int aide_ktimer_arm(int family_mask) { T.mask = (unsigned)family_mask; return AIDE_OK; }

// after the device is idle: launches / total milliseconds / total algorithmic flop of one family since aide_ktimer_start;
// returns the number of launches that found no free slot (all families) or a negative / hip error code
int aide_ktimer_read(int family, int64_t* launches, double* ms, double* flops, double* max_ms) {
    if (!launches || !ms || !flops) return AIDE_ERR_ARG;
    int64_t n = 0; double t = 0.0, f = 0.0, mx = 0.0;
    for (size_t s = 0; s < T.used; ++s) {
        if (T.fam[s] != family) continue;
        float el = 0.f;
        const hipError_t rc = hipEventElapsedTime(&el, T.ev[2 * s], T.ev[2 * s + 1]);
        if (rc != hipSuccess) return -(int)rc - 1000;
        ++n; t += el; f += T.flops[s];
        if (el > mx) mx = el;
    }
    *launches = n; *ms = t; *flops = f;
    if (max_ms) *max_ms = mx;
    return (int)(T.dropped > 0x7fffffff ? 0x7fffffff : T.dropped);
}

// after the device is idle: every recorded launch in launch order -- family, work, the dispatch's begin / end in milliseconds
// after the begin of the first recorded launch (negative for a dispatch that started before it on another stream), launch
// stream.  Arrays of `capacity` entries; returns the number of launches written or a negative / hip error code.
int aide_ktimer_dump(int capacity, int* family, double* work, double* begin_ms, double* end_ms, uint64_t* stream) {
    if (capacity < 0 || !family || !work || !begin_ms || !end_ms || !stream) return AIDE_ERR_ARG;
    const size_t n = T.used < (size_t)capacity ? T.used : (size_t)capacity;
    for (size_t s = 0; s < n; ++s) {
        float b = 0.f, d = 0.f;
        hipError_t rc = s ? hipEventElapsedTime(&b, T.ev[0], T.ev[2 * s]) : hipSuccess;
        if (rc == hipSuccess) rc = hipEventElapsedTime(&d, T.ev[2 * s], T.ev[2 * s + 1]);
        if (rc != hipSuccess) return -(int)rc - 1000;
        family[s] = T.fam[s]; work[s] = T.flops[s]; begin_ms[s] = b; end_ms[s] = (double)b + d;
        stream[s] = (uint64_t)(uintptr_t)T.stream[s];
    }
    return (int)n;
}

// ---- stream ordering without torch objects (the engine's launch tapes re-issue these like any other call)
int aide_event_create(void** ev) {
    if (!ev) return AIDE_ERR_ARG;
    hipEvent_t e;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) return (int)rc;
    *ev = (void*)e;
    return AIDE_OK;
}

int aide_event_destroy(void* ev) {
    if (!ev) return AIDE_ERR_ARG;
    return (int)hipEventDestroy((hipEvent_t)ev);
}

// everything enqueued on `to` after this call runs after everything enqueued on `from` before it
int aide_stream_order(void* ev, hipStream_t from, hipStream_t to) {
    if (!ev) return AIDE_ERR_ARG;
    hipError_t rc = hipEventRecord((hipEvent_t)ev, from);
    if (rc == hipSuccess) rc = hipStreamWaitEvent(to, (hipEvent_t)ev, 0);
    return (int)rc;
}

// the two halves of aide_stream_order as separate calls: the record sits right behind the producer on `from`, the wait is
// enqueued on `to` only where the consumer is -- work enqueued on `to` in between does not wait (a wait on an event that
// was never recorded is a no-op in HIP, so the caller records first)
int aide_event_record(void* ev, hipStream_t from) {
    if (!ev) return AIDE_ERR_ARG;
    return (int)hipEventRecord((hipEvent_t)ev, from);
}

int aide_stream_wait_event(hipStream_t to, void* ev) {
    if (!ev) return AIDE_ERR_ARG;
    return (int)hipStreamWaitEvent(to, (hipEvent_t)ev, 0);
}

}  // extern "C"
