// Semantics of v_dot2c_f32_bf16 as a "v - float(bf16 half)" primitive for the three-term split (conv3x3_wino4s.hip):
// which half does an inline -1.0 select, is the result exact, are tiny remainders flushed?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* o, const float* a, unsigned s_lo, unsigned s_hi) {
    const int t = threadIdx.x;
    float va = a[2 * t], vb = a[2 * t + 1];
    const bf16x2 H = __builtin_convertvector(f32x2{va, vb}, bf16x2);
    const unsigned Hu = __builtin_bit_cast(unsigned, H);
    float r0 = va, r1 = vb, r2 = va, r3 = vb;
    asm volatile("v_dot2c_f32_bf16 %0, -1.0, %1" : "+v"(r0) : "v"(Hu));           // inline constant
    asm volatile("v_dot2c_f32_bf16 %0, 0xbf800000, %1" : "+v"(r1) : "v"(Hu));     // literal, high half
    asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(r2) : "v"(Hu), "s"(s_lo));  // SGPR 0x0000bf80
    asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(r3) : "v"(Hu), "s"(s_hi));  // SGPR 0xbf800000
    o[8 * t + 0] = r0; o[8 * t + 1] = r1; o[8 * t + 2] = r2; o[8 * t + 3] = r3;
    o[8 * t + 4] = va - __builtin_bit_cast(float, Hu << 16);
    o[8 * t + 5] = vb - __builtin_bit_cast(float, Hu & 0xffff0000u);
    o[8 * t + 6] = __builtin_bit_cast(float, Hu);
}
int main() {
    float h[128], *d, *o, ho[512];
    srand(1);
    for (int i = 0; i < 128; ++i) h[i] = ldexpf((float)rand() / RAND_MAX - 0.5f, (i % 40) - 30);
    h[0] = 1.00390625f; h[1] = 3.0f; h[2] = 1e-38f; h[3] = 1.17549435e-38f * 1.001f;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d, 0x0000bf80u, 0xbf800000u);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int bad[4] = {0, 0, 0, 0};
    for (int t = 0; t < 64; ++t) {
        const float e_lo = ho[8 * t + 4], e_hi = ho[8 * t + 5];
        if (memcmp(&ho[8 * t + 0], &e_lo, 4)) bad[0]++;
        if (memcmp(&ho[8 * t + 1], &e_hi, 4)) bad[1]++;
        if (memcmp(&ho[8 * t + 2], &e_lo, 4)) bad[2]++;
        if (memcmp(&ho[8 * t + 3], &e_hi, 4)) bad[3]++;
        if (t < 4) printf("va %.9g vb %.9g | inline(-1.0) %.9g lit_hi %.9g sgpr_lo %.9g sgpr_hi %.9g | expect lo %.9g hi %.9g\n", h[2 * t], h[2 * t + 1],
                          ho[8 * t], ho[8 * t + 1], ho[8 * t + 2], ho[8 * t + 3], e_lo, e_hi);
    }
    printf("mismatches of 64: inline -1.0 (expects low half) %d, literal high %d, sgpr low %d, sgpr high %d\n", bad[0], bad[1], bad[2], bad[3]);
    return 0;
}
