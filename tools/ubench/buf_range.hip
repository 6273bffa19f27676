// What does the raw-buffer range check of gfx950 cover?  (1) is soffset part of the checked offset, (2) is a dwordx4 load
// that straddles num_records checked per dword or as a whole.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, int nrec_bytes, unsigned voff, unsigned soff, float* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, nrec_bytes, 0x00020000);
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    for (int i = 0; i < 4; ++i) out[i] = v[i];
}
int main() {
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
    float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    struct { int nrec; unsigned voff, soff; const char* what; } cs[] = {
        {64, 48, 0, "in range (last 16 bytes)"},
        {64, 56, 0, "voffset straddles the end by 8 bytes"},
        {64, 64, 0, "voffset at the end"},
        {64, 0, 56, "soffset straddles the end by 8 bytes"},
        {64, 0, 64, "soffset at the end"},
        {64, 0, 128, "soffset far past the end"},
        {64, 32, 24, "voffset + soffset straddle by 8"},
        {64, 60, 0, "voffset straddles by 12"},
    };
    for (auto& c : cs) {
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, c.nrec, c.voff, c.soff, o);
        float r[4]; hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
        printf("%-44s nrec %3d voff %3u soff %3u -> %6.1f %6.1f %6.1f %6.1f\n", c.what, c.nrec, c.voff, c.soff, r[0], r[1], r[2], r[3]);
    }
    return 0;
}
