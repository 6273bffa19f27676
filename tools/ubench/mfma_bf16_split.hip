// Facts for an fp32-by-split-bf16 F(4x4) kernel (HISTORY §8): sustained rate of the bf16 MFMA shapes at one wave per SIMD with
// independent accumulators, alone and with K filler VALU instructions per MFMA -- is a VALU instruction paid in bf16-MFMA
// time as it is in fp32-MFMA time?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int K>
__global__ __launch_bounds__(256, 1) void kern(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[8192];
    if (threadIdx.x == 9999) out[0] = sm[0];
    f32x16 a32[8]; f32x4 a16[8];
    for (int p = 0; p < 8; ++p) { for (int r = 0; r < 16; ++r) a32[p][r] = 0.f; for (int r = 0; r < 4; ++r) a16[p][r] = 0.f; }
    bf16x8 a, b; s16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)1.0f; }
    for (int i = 0; i < 4; ++i) { a4[i] = (short)threadIdx.x; b4[i] = 0x3f80; }
    float x = 0.5f, y = 1.000001f; double dd = (double)threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (SHAPE == 3216) a32[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a32[p], 0, 0, 0);
            else if (SHAPE == 328) a32[p] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, a32[p], 0, 0, 0);
            else if (SHAPE == 1632) a16[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a16[p], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < (K % 100); ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
            if (K >= 100) {                                // + LDS reads / a store per MFMA (the staging traffic of a fused stage)
                f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)((threadIdx.x & 63) * 16))); asm volatile("" :: "v"(t));
                if (K >= 200) { asm volatile("ds_read_b32 %0, %1" : "=v"(t[0]) : "v"((unsigned)((threadIdx.x & 63) * 4 + 4096))); asm volatile("" :: "v"(t[0]));
                                asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)((threadIdx.x & 255) * 8 + 8192)), "v"(dd)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = x;
    for (int p = 0; p < 8; ++p) s += a32[p][0] + a16[p][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int K> void run(float* out, double flops_per_mfma) {
    const int iters = 20000;
    hipLaunchKernelGGL((kern<SHAPE, K>), dim3(256), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<SHAPE, K>), dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = 256.0 * 4 * iters * 8;
    printf("shape %4d  %d VALU per MFMA: %.3f ms, %.1f ns per MFMA per SIMD -> %.0f TFLOP/s\n", SHAPE, K, ms, ms * 1e6 / (iters * 8.0),
           n * flops_per_mfma / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    run<3216, 0>(out, 32768.0); run<3216, 1>(out, 32768.0); run<3216, 2>(out, 32768.0); run<3216, 4>(out, 32768.0);
    run<328, 0>(out, 16384.0); run<328, 1>(out, 16384.0); run<328, 2>(out, 16384.0);
    run<328, 4>(out, 16384.0); run<328, 6>(out, 16384.0); run<328, 8>(out, 16384.0);
    run<328, 104>(out, 16384.0); run<328, 204>(out, 16384.0); run<328, 206>(out, 16384.0); run<328, 208>(out, 16384.0);
    run<3216, 6>(out, 32768.0); run<3216, 8>(out, 32768.0); run<3216, 208>(out, 32768.0);
    run<1632, 0>(out, 16384.0); run<1632, 1>(out, 16384.0); run<1632, 2>(out, 16384.0);
    return 0;
}
