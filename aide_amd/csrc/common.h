// Shared helpers for the AIDE hot-path HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define AIDE_OK 0
#define AIDE_ERR_ARG (-1)

// ---- SRSRC buffer loads (cdna_hip_programming.md T8/T20): 32-bit byte offsets instead of
// 64-bit addresses (half the address VGPRs) and hardware zero-fill for out-of-range offsets.
// Every descriptor spans 2 GiB; an element that must read as zero (halo outside the image,
// channel padding) gets voffset = BUF_OOB, which is >= num_records under either reading of
// the soffset range-check rule. Hosts assert tensor extents < 2 GiB.
#define BUF_OOB 0x80000000u
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, BUF_OOB, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 buf_load_f32x2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 buf_load_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

static inline int aide_launch_status() { return (int)hipGetLastError(); }

// XCD-aware bijective remap of a 1-D block id: the hardware dispatches block b to XCD b % 8;
// give every XCD a contiguous range of logical tiles so neighbouring tiles (shared halo rows,
// shared filter blocks) hit the same private L2 (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, k = b >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum of NV doubles per thread (blockDim.x multiple of 64, <= 1024). Result valid in
// thread 0. `sm` must hold NV * (blockDim.x/64) doubles.
template <int NV>
__device__ __forceinline__ void block_sum_d(double (&v)[NV], double* sm) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum_d(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) sm[i * nw + wid] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = 0.0;
            for (int w = 0; w < nw; ++w) s += sm[i * nw + w];
            v[i] = s;
        }
    }
}
