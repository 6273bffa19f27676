// What does a main -> side hand-over cost the MAIN queue?  A chain of N ~20 us kernels on one stream, timed from the first
// kernel's start to the last kernel's end, with between every two kernels: (a) nothing, (b) hipEventRecord + a second stream
// waiting on it (the engine's aide_stream_order), (c) the event attached to the producing kernel as hipExtLaunchKernelGGL's
// stop event + the wait, (d) record without any waiter, (e) hipStreamWriteValue32 on main + hipStreamWaitValue32 on side.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void spin(float* p, int iters, int* stamp, int value) {           // producer: stamps `value` when it is done
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    p[threadIdx.x] = v;
    __syncthreads();
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) { __threadfence(); atomicMax(stamp, value); }
}
__global__ void consume(float* p, int iters, const int* stamp, int expect, int* early) {   // consumer: was the producer done?
    if (blockIdx.x == 0 && threadIdx.x == 0 && __atomic_load_n(stamp, __ATOMIC_RELAXED) < expect) atomicAdd(early, 1);
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    p[threadIdx.x] = v;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int N = 40, ITERS = 1200;
    float *a, *b; CK(hipMalloc(&a, 1024)); CK(hipMalloc(&b, 1024)); CK(hipMemset(a, 0, 1024)); CK(hipMemset(b, 0, 1024));
    unsigned* flag; CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    int *stamp, *early; CK(hipMalloc(&stamp, 4)); CK(hipMalloc(&early, 4));
    hipStream_t m, s; CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const char* names[] = {"plain chain", "hipEventRecord + side waits", "stop event on the kernel + side waits", "hipEventRecord, nobody waits",
                           "write value on main + wait value on side"};
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(flag, 0, 4)); CK(hipMemset(stamp, 0, 4)); CK(hipMemset(early, 0, 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, m));
            for (int i = 0; i < N; ++i) {
                if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, m, nullptr, ev[i], 0, a, ITERS, stamp, i + 1);
                else hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, m, a, ITERS, stamp, i + 1);
                if (mode == 1 || mode == 3) CK(hipEventRecord(ev[i], m));
                if (mode == 1 || mode == 2) { CK(hipStreamWaitEvent(s, ev[i], 0)); hipLaunchKernelGGL(consume, dim3(64), dim3(256), 0, s, b, ITERS / 2, stamp, i + 1, early); }
                if (mode == 4) {
                    CK(hipStreamWriteValue32(m, flag, i + 1, 0));
                    CK(hipStreamWaitValue32(s, flag, i + 1, hipStreamWaitValueGte, 0xffffffffu));
                    hipLaunchKernelGGL(consume, dim3(64), dim3(256), 0, s, b, ITERS / 2, stamp, i + 1, early);
                }
            }
            CK(hipEventRecord(t1, m));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            int h_early = -1; CK(hipMemcpy(&h_early, early, 4, hipMemcpyDeviceToHost));
            if (rep == 2) printf("%-44s main chain of %d kernels: %8.1f us  (%.2f us per kernel)   consumers that started early: %d\n", names[mode], N, ms * 1e3, ms * 1e3 / N, h_early);
        }
    }
    return 0;
}
