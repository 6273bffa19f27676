// Shared helpers for the AIDE hot-path HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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

// ---- bf16-stored activations (precision='bf16'): typed vector loads / stores.  Widening is exact, narrowing is
// round-to-nearest-even (v_cvt_pk_bf16_f32).  T = float or bf16_store_t.
typedef uint16_t bf16_store_t;
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf16_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const bf16_store_t* p) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    return f32x4{bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1])};
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_store_t* p, f32x4 v) {
    *reinterpret_cast<u32x2*>(p) = u32x2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
}
__device__ __forceinline__ f32x2 ld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
__device__ __forceinline__ f32x2 ld2(const bf16_store_t* p) {
    const unsigned v = *reinterpret_cast<const unsigned*>(p);
    return f32x2{bf16_lo(v), bf16_hi(v)};
}
__device__ __forceinline__ void st2(float* p, f32x2 v) { *reinterpret_cast<f32x2*>(p) = v; }
__device__ __forceinline__ void st2(bf16_store_t* p, f32x2 v) { *reinterpret_cast<unsigned*>(p) = cvt_pk_bf16(v[0], v[1]); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_store_t* p) { return __builtin_bit_cast(float, (unsigned)*p << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_store_t* p, float v) { *p = (bf16_store_t)(cvt_pk_bf16(v, 0.f) & 0xffffu); }

static inline int aide_launch_status() { return (int)hipGetLastError(); }

// Dynamic-LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) of a launcher's kernels: once per DEVICE -- the attribute
// is set for the function on the device that is current at the time, so a process that drives several devices needs it on
// each -- and with its status returned to the caller (a refused opt-in would otherwise surface as a failed launch later).
//   static AideLdsOptIn lds;  if (int rc = lds.ensure([] { return hipFuncSetAttribute(...); })) return rc;
#include <atomic>
struct AideLdsOptIn {
    std::atomic<unsigned long long> done{0ull};
    template <class F>
    int ensure(F&& set_all) {
        int dev = 0;
        const hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return 0;
        const hipError_t rc = set_all();           // (idempotent: two threads racing here both set the same values)
        if (rc != hipSuccess) return (int)rc;
        done.fetch_or(bit, std::memory_order_release);
        return 0;
    }
};

// ---- optional per-kernel timing (ktimer.hip; include/aide_hip.h "kernel timer"): a launch of an armed family carries
// a start / stop event pair that receives the dispatch's own begin / end timestamps.  Families 0-9: the MFMA convolution
// kernels (work = algorithmic flop); 10-15: the streaming kernels (work = algorithmic bytes; 0 where nobody prices it)
enum { AIDE_KT_IGEMM = 0, AIDE_KT_WINO2 = 1, AIDE_KT_WINO4 = 2, AIDE_KT_WGRAD = 3, AIDE_KT_WGRAD_WINO2 = 4,
       AIDE_KT_WGRAD4 = 5, AIDE_KT_WGRAD_STEM = 6, AIDE_KT_BF16 = 7, AIDE_KT_WGRAD_BF16 = 8, AIDE_KT_CONVT = 9,
       AIDE_KT_BN_FWD = 10, AIDE_KT_BN_BWD = 11, AIDE_KT_POOL = 12, AIDE_KT_UPSAMPLE = 13, AIDE_KT_REDUCE = 14,
       AIDE_KT_OTHER = 15 };
extern "C" int aide_ktimer_slot(int family, double work, hipStream_t stream, hipEvent_t* e0, hipEvent_t* e1);
#define AIDE_LAUNCH_TIMED(FAM, FLOPS, kernel, grid, block, lds, stream, ...)                                  \
    do {                                                                                                      \
        hipEvent_t kt_e0_, kt_e1_;                                                                            \
        if (aide_ktimer_slot(FAM, FLOPS, stream, &kt_e0_, &kt_e1_))                                           \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, kt_e0_, kt_e1_, 0, __VA_ARGS__);          \
        else                                                                                                  \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                \
    } while (0)
// launch with an optional completion event (aide_event_create) attached to the DISPATCH itself: the hand-over to another
// stream then costs the launching queue ~1.4 us instead of the ~5 us hole of a separate record packet
// (tools/ubench/handover_cost.hip)
#define AIDE_LAUNCH_DONE(DONE, kernel, grid, block, lds, stream, ...)                                                 \
    do {                                                                                                              \
        if (DONE) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, (hipEvent_t)(DONE), 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                      \
    } while (0)
// ... of a timed family: while the timer is armed the dispatch's stop event is the timer's, and the hand-over event is a
// record packet of its own behind it (instrumented steps only)
#define AIDE_LAUNCH_DONE_TIMED(FAM, WORK, DONE, kernel, grid, block, lds, stream, ...)                                \
    do {                                                                                                              \
        hipEvent_t kt_e0_, kt_e1_;                                                                                    \
        if (aide_ktimer_slot(FAM, WORK, stream, &kt_e0_, &kt_e1_)) {                                                  \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, kt_e0_, kt_e1_, 0, __VA_ARGS__);                  \
            if (DONE) (void)hipEventRecord((hipEvent_t)(DONE), stream);                                               \
        } else if (DONE) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, (hipEvent_t)(DONE), 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                      \
    } while (0)
#define AIDE_CONV_FLOPS(N, H, W, Co, Ci) (2.0 * (double)(N) * (double)(H) * (double)(W) * (double)(Co) * (double)(Ci) * 9.0)

// XCD-aware bijective remap of a 1-D block id: the hardware dispatches block b to XCD b % 8;
// give every XCD a contiguous range of logical tiles so neighbouring tiles (shared halo rows,
// shared filter blocks) hit the same private L2 (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, k = b >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

// Wave-wide sums on the DPP path: an inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8; lanes shifted in from
// outside the row, or inactive ones, read 0), then the row totals travel on (row_bcast:15 into rows 1 and 3, row_bcast:31 into
// rows 2 and 3) -- six dependent vector adds, total in lane 63, broadcast by v_readlane.  (As `__shfl_xor` butterflies every
// step was a ds_bpermute_b32 LDS round trip behind its own s_waitcnt lgkmcnt(0) -- two per step for a double: 12 / 24 round
// trips per value at the head or tail of kernels that run for 15-25 us.)  Lane 63 must be active: every caller runs whole waves.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <typename T>
__device__ __forceinline__ T wave_scan_total(T v) {        // total in lane 63
    v += dpp_mov<0x111, 0xf>(v);
    v += dpp_mov<0x112, 0xf>(v);
    v += dpp_mov<0x114, 0xf>(v);
    v += dpp_mov<0x118, 0xf>(v);
    v += dpp_mov<0x142, 0xa>(v);
    v += dpp_mov<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = wave_scan_total(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v = wave_scan_total(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
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
