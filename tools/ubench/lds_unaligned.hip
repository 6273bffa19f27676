// Does gfx950 execute ds_write_b128 / ds_read_b128 at addresses that are only 4-byte aligned (and how fast)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int iters, int misalign, long long* cyc) {
    __shared__ __attribute__((aligned(16))) float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = -1.f;
    __syncthreads();
    const unsigned addr = (unsigned)((threadIdx.x * 4 + misalign) * 4);        // bytes
    f32x4 v = {threadIdx.x + 0.f, threadIdx.x + 0.25f, threadIdx.x + 0.5f, threadIdx.x + 0.75f};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)");
    long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    for (int i = threadIdx.x; i < 1040; i += 256) out[i] = sm[i];
    out[2048 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* o; long long* c; hipMalloc(&o, 4096 * 4); hipMalloc(&c, 8);
    for (int mis = 0; mis < 4; ++mis) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, o, 1000, mis, c);
        float h[4096]; long long cy; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 256; ++t) for (int j = 0; j < 4; ++j) if (h[t * 4 + mis + j] != t + 0.25f * j) ++bad;
        int badr = 0;
        for (int t = 0; t < 256; ++t) if (h[2048 + t] != 4.f * t + 1.5f) ++badr;
        printf("misalign %d dwords: %d wrong stored values, %d wrong read-backs, first words %.2f %.2f %.2f %.2f %.2f %.2f, %.1f cycles per ds_write_b128 (4 waves)\n",
               mis, bad, badr, h[0], h[1], h[2], h[3], h[4], h[5], cy / 1000.0);
    }
    return 0;
}
