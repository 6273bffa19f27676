// Microbenchmark: fp32 MFMA shapes at one wave per SIMD -- 32x32x2 (64 cycles, 16 acc regs) vs 16x16x4 (32 cycles, 4 acc
// regs): sustained rate with independent accumulators, and with K filler VALU / ds_read_b128 instructions per 64 MFMA cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int K, int KIND>
__global__ __launch_bounds__(256, 1) void kern(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[4096];
    f32x16 a32[4]; f32x4 a16[16];
    for (int p = 0; p < 4; ++p) for (int r = 0; r < 16; ++r) a32[p][r] = 0.f;
    for (int p = 0; p < 16; ++p) for (int r = 0; r < 4; ++r) a16[p][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f, x = 0.5f, y = 1.000001f;
    sm[threadIdx.x] = a; __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {                      // one "slot" = 64 MFMA cycles
            if (SHAPE == 32) a32[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a32[p], 0, 0, 0);
            else {
                a16[2 * p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a16[2 * p], 0, 0, 0);
                a16[2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a16[2 * p + 1], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                else { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)((threadIdx.x & 63) * 16))); asm volatile("" :: "v"(t)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    float s = x;
    for (int p = 0; p < 4; ++p) s += a32[p][0];
    for (int p = 0; p < 16; ++p) s += a16[p][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int K, int KIND> void run(float* out) {
    const int iters = 20000;
    hipLaunchKernelGGL((kern<SHAPE, K, KIND>), dim3(256), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<SHAPE, K, KIND>), dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 4 * iters * 4 * 4096.0;
    printf("shape %2d  filler kind %d x %2d per 64 cycles: wall %.3f ms -> %.1f TFLOP/s executed\n", SHAPE, KIND, K, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    run<32, 0, 0>(out); run<16, 0, 0>(out);
    run<32, 1, 0>(out); run<16, 1, 0>(out);
    run<32, 2, 0>(out); run<16, 2, 0>(out);
    run<32, 4, 0>(out); run<16, 4, 0>(out);
    run<32, 1, 1>(out); run<16, 1, 1>(out);
    run<32, 2, 1>(out); run<16, 2, 1>(out);
    return 0;
}
