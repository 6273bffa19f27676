// Microbenchmark: how many independent VALU / LDS instructions of the SAME wave hide under one
// v_mfma_f32_32x32x2_f32 (1 wave per SIMD)?  Prints cycles per MFMA slot for k = 0..16 fillers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int K, int KIND>
__global__ __launch_bounds__(512, 1) void kern(float* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) float sm[4096];
    f32x16 acc[4];
    for (int p = 0; p < 4; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f, x = 0.5f, y = 1.000001f, x1 = 0.25f, x2 = 0.125f, x3 = 0.0625f;
    int sx = iters; const float* gp = out + 65536 + threadIdx.x * 4; f32x4 w4 = {a, a, a, a}; f32x2 pk = {a, a}, pk1 = {1.f, 1.f};
    sm[threadIdx.x] = a; __syncthreads();
    const float* lp = sm + (threadIdx.x & 63);
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                else if (KIND == 3) {
                    if ((k & 3) == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                    else if ((k & 3) == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x1) : "v"(y));
                    else if ((k & 3) == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x2) : "v"(y));
                    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x3) : "v"(y));
                }
                else if (KIND == 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
                else if (KIND == 5) { f32x4 t; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(gp)); asm volatile("" :: "v"(t)); }
                else if (KIND == 6) asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)((threadIdx.x & 255) * 16)), "v"(w4));
                else if (KIND == 7) { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)((threadIdx.x & 63) * 16))); asm volatile("" :: "v"(t)); }
                else if (KIND == 8) { float t; asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(gp)); asm volatile("" :: "v"(t)); }
                else if (KIND == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk) : "v"(pk1));
                else if (KIND == 1) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)lp + 0u)); asm volatile("" :: "v"(t)); }
                else asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)((threadIdx.x & 255) * 4)), "v"(x));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    long long t1 = __builtin_readcyclecounter();
    float s = pk[0] + pk[1] + x + x1 + x2 + x3 + (float)sx;
    for (int p = 0; p < 4; ++p) s += acc[p][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int K, int KIND> void run(float* out, long long* cyc, int threads = 256) {
    const int iters = 20000;
    hipLaunchKernelGGL((kern<K, KIND>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<K, KIND>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = 256.0 * (threads / 64) * iters * 4;     // MFMAs issued chip-wide
    printf("thr %d kind %d  K=%2d  %.1f shader-clock ticks/MFMA   wall %.3f ms  -> %.1f TFLOP/s executed\n", threads, KIND, K,
           (double)c / (iters * 4.0), ms, mf * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4 * 4); hipMalloc(&cyc, 8);
    run<0, 0>(out, cyc);
    run<1, 5>(out, cyc); run<2, 5>(out, cyc); run<4, 5>(out, cyc);
    run<1, 8>(out, cyc); run<2, 8>(out, cyc); run<4, 8>(out, cyc);
    run<1, 6>(out, cyc); run<2, 6>(out, cyc); run<4, 6>(out, cyc);
    run<1, 7>(out, cyc); run<2, 7>(out, cyc); run<4, 7>(out, cyc);
    run<2, 9>(out, cyc); run<4, 9>(out, cyc);
    run<2, 5>(out, cyc, 512); run<2, 6>(out, cyc, 512);
    return 0;
}
