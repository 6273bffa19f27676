// Weight gradient of the STEM convolutions (Ci <= 3: the two 3->32 input layers of FuseUNet, 3->64 of UNet) for gfx950.
//
// Replaces (reference): autograd's weight gradient of nn.Conv2d(3, C, 3, padding=1) at models_twomodalinputs/netblocks.py:24
// (modal1_downblock1 / modal2_downblock1, fuseunet.py:12,24) and models_singlemodalinput/UNet.py:19 (down_block1).
//
// The general weight-gradient kernels tile (co, ci) and give every tap its own accumulator: with 3 of 32 / 64 input
// channel columns in use they run at the speed of a full tile (66 us at 256x256 x4, 195 us at 512x512 x8 -- these two
// launches are the tail of every backward pass, after the main stream has nothing left to overlap them).  Here the nine
// taps are folded into the GEMM's N dimension instead:
//     dW[co][(ci, tap)] = sum_pixels dz[co][p] * x[ci][p + tap]          M = 32 co, N = 27 (of 32), K = pixels
// on v_mfma_f32_32x32x2_f32, so a pixel pair costs ONE MFMA per wave and the kernel is bound by reading dz once:
// workgroup = 4 waves, tile = 4 rows x 64 columns of one image (wave = one row), operands staged through LDS
// (dz [32][4 x 64], co stride 258 = 2 mod 64 -> the 64 A lanes (co, k) hit 64 banks; x [3][6][68] halo tile, the B lane
// (ci, kh, kw, k) reads x[ci][row + kh][col + kw + k]), next tile's global loads in flight under the MFMAs, four
// workgroups per CU.  Slabs [split][9][Co][Ci] in the layout of the other weight-gradient kernels, summed by their
// fixed-order reduce (bit-reproducible).  dz may be bf16-stored (precision='bf16'); x is the fp32 input image.
// Measured: 3->32 at 256x256 x4 66 -> 25 us (fp32), at 512x512 x8 195 -> 88 us (bf16-stored dz).
// Two things that mattered: (1) the staged registers stay RAW until put() -- converting them right after the load made
// every iteration wait for its loads before the MFMAs (239 -> 116 us at 256 workgroups); (2) no inline asm on those
// registers: a non-volatile `v_cvt_pk_bf16_f32` asm consuming registers loaded in the PREVIOUS loop iteration ran without
// the s_waitcnt and produced garbage -- the rounding is spelled out in integer arithmetic (rne_bf16) instead.
#include "common.h"

int aide_wgrad_reduce_launch(const float* ws, int splits, int Co, int Ci, float* dw, void* queue, hipStream_t stream);   // conv3x3_wgrad.hip

namespace {

struct StemArgs {
    const void* dz;
    const float* x;
    float* slabs;
    long dz_bs, x_bs;
    int N, Co, Ci, H, W;
    int n_co_tiles, splits, tiles_w, tiles_h, tiles_total;
};

constexpr int ST_R = 4, ST_C = 64;              // tile rows / columns
constexpr int ST_DCS = ST_R * ST_C + 2;         // dz co stride (floats): 2 mod 64
constexpr int ST_XRS = 68;                      // x row stride: columns -1 .. 64 (+ 2 pad), 4 mod 64
constexpr int ST_XCS = (ST_R + 2) * ST_XRS;     // x channel stride: 408 = 24 mod 64
constexpr int ST_XN = 3 * (ST_R + 2) * (ST_C + 2);   // halo elements per tile (3 channels)
constexpr int ST_NX = (ST_XN + 255) / 256;

// fp32 bits -> the fp32 bits of the nearest bf16 (ties to even; finite inputs), as v_cvt_pk_bf16_f32 rounds
__device__ __forceinline__ unsigned rne_bf16(unsigned u) { return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; }

// ROUND: the precision='bf16' contract (operands rounded to bf16, round-to-nearest-even, fp32 accumulation) -- the fp32
// image and an fp32 dz are rounded when staged, so the result equals that of the bf16-MFMA kernels on the same tensors
template <bool DZ_BF16, bool ROUND>
__global__ __launch_bounds__(256, 4) void wgrad_stem_kernel(const StemArgs g) {
    __shared__ __attribute__((aligned(16))) float dzs[32 * ST_DCS];
    __shared__ float xs[3 * ST_XCS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int co_tile = blockIdx.x % g.n_co_tiles, split = blockIdx.x / g.n_co_tiles;
    const int co0 = co_tile * 32;
    const int HW = g.H * g.W;

    // ---- staging units (identical for every tile) ----
    // dz: 32 co x 4 rows x 64 columns; one unit = 4 (fp32) or 8 (bf16) consecutive columns = 16 bytes
    constexpr int DPU = DZ_BF16 ? 8 : 4;                    // pixels per unit
    constexpr int ND = 32 * ST_R * ST_C / DPU / 256;        // units per thread: 8 / 4
    int d_off[ND], d_lds[ND];
    bool d_ok[ND];
#pragma unroll
    for (int e = 0; e < ND; ++e) {
        const int u = tid + e * 256;
        const int upr = ST_C / DPU;                         // units per row
        const int co = u / (ST_R * upr), rem = u - co * (ST_R * upr);
        const int r = rem / upr, c = (rem - r * upr) * DPU;
        d_ok[e] = co0 + co < g.Co;
        d_off[e] = (co0 + co) * HW + r * g.W + c;
        d_lds[e] = co * ST_DCS + r * ST_C + c;
    }
    // x halo: element q -> (ci, row -1 .. 4, column -1 .. 64)
    int x_ci[ST_NX], x_r[ST_NX], x_c[ST_NX];
#pragma unroll
    for (int e = 0; e < ST_NX; ++e) {
        const int q = tid + e * 256;
        const int ci = q / ((ST_R + 2) * (ST_C + 2)), rem = q - ci * ((ST_R + 2) * (ST_C + 2));
        x_ci[e] = q < ST_XN ? ci : -1;
        x_r[e] = rem / (ST_C + 2);
        x_c[e] = rem - x_r[e] * (ST_C + 2);
    }

    // ---- MFMA operand lanes: A lane = (co = j, k = half); B lane = (n = (ci, tap) = j, k = half) ----
    const int n = min(j, g.Ci * 9 - 1);                     // columns >= Ci * 9 duplicate the last one (never stored)
    const int ci_n = n / 9, tap = n - ci_n * 9, kh = tap / 3, kw = tap - kh * 3;
    const float* la = dzs + j * ST_DCS + wid * ST_C + half;
    const float* lb = xs + ci_n * ST_XCS + (wid + kh) * ST_XRS + kw + half;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    // raw staging registers: conversion / rounding happens in put(), so nothing waits for the loads before the MFMAs
    u32x4 dreg[ND];
    float xreg[ST_NX];
    auto fetch = [&](int tile) {
        const int tw = tile % g.tiles_w, t2 = tile / g.tiles_w;
        const int th = t2 % g.tiles_h, img = t2 / g.tiles_h;
        const int h0 = th * ST_R, w0 = tw * ST_C;
#pragma unroll
        for (int e = 0; e < ND; ++e) {
            dreg[e] = u32x4{0u, 0u, 0u, 0u};
            if (d_ok[e]) {
                if constexpr (DZ_BF16)
                    dreg[e] = *reinterpret_cast<const u32x4*>((const bf16_store_t*)g.dz + (long)img * g.dz_bs + d_off[e] + h0 * g.W + w0);
                else
                    dreg[e] = *reinterpret_cast<const u32x4*>((const float*)g.dz + (long)img * g.dz_bs + d_off[e] + h0 * g.W + w0);
            }
        }
#pragma unroll
        for (int e = 0; e < ST_NX; ++e) {
            const int ih = h0 - 1 + x_r[e], iw = w0 - 1 + x_c[e];
            const bool ok = x_ci[e] >= 0 && x_ci[e] < g.Ci && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
            xreg[e] = ok ? g.x[(long)img * g.x_bs + (long)x_ci[e] * HW + ih * g.W + iw] : 0.0f;
        }
    };
    auto put = [&]() {
#pragma unroll
        for (int e = 0; e < ND; ++e) {
            float* p = dzs + d_lds[e];                      // co stride 258: 8-byte aligned pieces
            const u32x4 v = dreg[e];
            if constexpr (DZ_BF16) {
                *reinterpret_cast<f32x2*>(p) = f32x2{bf16_lo(v[0]), bf16_hi(v[0])};
                *reinterpret_cast<f32x2*>(p + 2) = f32x2{bf16_lo(v[1]), bf16_hi(v[1])};
                *reinterpret_cast<f32x2*>(p + 4) = f32x2{bf16_lo(v[2]), bf16_hi(v[2])};
                *reinterpret_cast<f32x2*>(p + 6) = f32x2{bf16_lo(v[3]), bf16_hi(v[3])};
            } else if constexpr (ROUND) {
                *reinterpret_cast<u32x2*>(p) = u32x2{rne_bf16(v[0]), rne_bf16(v[1])};
                *reinterpret_cast<u32x2*>(p + 2) = u32x2{rne_bf16(v[2]), rne_bf16(v[3])};
            } else {
                *reinterpret_cast<u32x2*>(p) = u32x2{v[0], v[1]};
                *reinterpret_cast<u32x2*>(p + 2) = u32x2{v[2], v[3]};
            }
        }
#pragma unroll
        for (int e = 0; e < ST_NX; ++e)
            if (x_ci[e] >= 0)
                xs[x_ci[e] * ST_XCS + x_r[e] * ST_XRS + x_c[e]] =
                    ROUND ? __builtin_bit_cast(float, rne_bf16(__builtin_bit_cast(unsigned, xreg[e]))) : xreg[e];
    };

    int tile = split;
    if (tile < g.tiles_total) fetch(tile);
    while (tile < g.tiles_total) {
        put();
        __syncthreads();
        const int next = tile + g.splits;
        if (next < g.tiles_total) fetch(next);              // in flight under the MFMAs
#pragma unroll 8
        for (int s = 0; s < ST_C / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(la[2 * s], lb[2 * s], acc, 0, 0, 0);
        __syncthreads();
        tile = next;
    }

    // ---- sum the four waves' partial [co][n] tiles through LDS, write this split's slab [9][Co][Ci] ----
    float* red = dzs;                                       // 4 x 32 x 33 floats <= 32 x 258
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * half;   // D layout: row (r, half), column j
        red[(wid * 32 + co) * 33 + j] = acc[r];
    }
    __syncthreads();
    const int ncol = g.Ci * 9;
    for (int q = tid; q < 32 * ncol; q += 256) {
        const int co = q / ncol, nn = q - co * ncol;
        if (co0 + co >= g.Co) continue;
        const float v = (red[co * 33 + nn] + red[(32 + co) * 33 + nn]) + (red[(64 + co) * 33 + nn] + red[(96 + co) * 33 + nn]);
        const int ci = nn / 9, tp = nn - ci * 9;
        g.slabs[(((long)split * 9 + tp) * g.Co + co0 + co) * g.Ci + ci] = v;
    }
}

}  // namespace

extern "C" {

// slabs (= workgroups per co tile) for an [N][.][H][W] problem: 4-row x 64-column tiles dealt round-robin
int aide_conv3x3_wgrad_stem_splits(int N, int H, int W) {
    const long target = 512;                     // (512x512 x8: 256 -> 116 us, 512 -> 88, 1024 -> 93, 2048 -> 118)
    const long tiles = (long)N * (H / ST_R) * (W / ST_C);
    long s = tiles / 4;                          // >= 4 tiles per workgroup: the next tile's loads fly under the MFMAs
    if (s < 256) s = 256;
    if (s > target) s = target;
    if (s > tiles) s = tiles;
    return (int)(s < 1 ? 1 : s);
}

int aide_conv3x3_wgrad_stem_supported(int Co, int Ci, int H, int W) {
    return (Ci >= 1 && Ci * 9 <= 32 && Co >= 1 && H % ST_R == 0 && W % ST_C == 0) ? 1 : 0;
}

// dw [Co][Ci][3][3] = weight gradient of a 3x3 convolution with Ci <= 3 input channels.
//   dz : [N][Co][H][W] fp32 (dz_bf16 = 0) or bf16 (1), batch stride dz_bs elements;  x : [N][Ci][H][W] fp32, stride x_bs
//   ws : splits * 9 * Co * Ci floats (the slab layout / split count of aide_conv3x3_wgrad or aide_conv3x3_wgrad_bf16)
//   round_bf16 : operands rounded to bf16 when staged (the precision='bf16' contract); implied by dz_bf16
int aide_conv3x3_wgrad_stem(const void* dz, int dz_bf16, int64_t dz_bs, const float* x, int64_t x_bs, float* dw,
                            int N, int Co, int Ci, int H, int W, float* ws, int splits, int round_bf16,
                            void* queue, hipStream_t stream) {
    if (!dz || !x || !dw || !ws || N <= 0 || splits < 1 || !aide_conv3x3_wgrad_stem_supported(Co, Ci, H, W))
        return AIDE_ERR_ARG;
    if (dz_bs % (dz_bf16 ? 8 : 4)) return AIDE_ERR_ARG;
    StemArgs g;
    g.dz = dz; g.x = x; g.slabs = ws; g.dz_bs = dz_bs; g.x_bs = x_bs;
    g.N = N; g.Co = Co; g.Ci = Ci; g.H = H; g.W = W;
    g.n_co_tiles = (Co + 31) / 32;
    g.tiles_w = W / ST_C; g.tiles_h = H / ST_R; g.tiles_total = N * g.tiles_w * g.tiles_h;
    g.splits = splits < g.tiles_total ? splits : g.tiles_total;
    const unsigned nb = (unsigned)(g.n_co_tiles * g.splits);
    const double fl = AIDE_CONV_FLOPS(N, H, W, Co, Ci);
    if (dz_bf16) AIDE_LAUNCH_TIMED(AIDE_KT_WGRAD_STEM, fl, (wgrad_stem_kernel<true, true>), dim3(nb), dim3(256), 0, stream, g);     // bf16 storage: bf16 mode
    else if (round_bf16) AIDE_LAUNCH_TIMED(AIDE_KT_WGRAD_STEM, fl, (wgrad_stem_kernel<false, true>), dim3(nb), dim3(256), 0, stream, g);
    else AIDE_LAUNCH_TIMED(AIDE_KT_WGRAD_STEM, fl, (wgrad_stem_kernel<false, false>), dim3(nb), dim3(256), 0, stream, g);
    const int rc = aide_launch_status();
    if (rc != 0) return rc;
    return aide_wgrad_reduce_launch(ws, g.splits, Co, Ci, dw, queue, stream);
}

}  // extern "C"
