// Weight gradient of the 3x3 / pad 1 convolution as an fp32 MFMA implicit GEMM (gfx950).
//
// Replaces (reference): autograd's convolution_backward weight/bias gradient for the nn.Conv2d
// at models_twomodalinputs/netblocks.py:17,24,26 (SURVEY.md §2.3 "Conv2d wgrad").
//
//   dW[co][ci][kh][kw] = sum_{n,h,w} dz[n][co][h][w] * a[n][ci][h+kh-1][w+kw-1]
//
// GEMM view: M = Co (MFMA rows), N = Ci (MFMA columns, one accumulator per filter tap), K = pixels.
// One v_mfma_f32_32x32x2_f32 consumes two horizontally adjacent pixels.  A workgroup owns a
// (32*WAVES_CO) x (32*WAVES_CI) block of (co, ci) pairs for all 9 taps and a contiguous range of
// 4x16-pixel spatial tiles ("split" of the K dimension); WAVES_PX waves share one (co,ci) block and
// interleave the pixel pairs of each tile, their accumulators are summed through LDS at the end.
// Every split writes a partial slab [9][Co][Ci]; aide_conv3x3_wgrad sums the slabs in a fixed
// order into dW[Co][Ci][3][3] (deterministic, no atomics).
#include "common.h"
#include <stdlib.h>

extern "C" int aide_conv3x3_wgrad_stem_splits(int N, int H, int W);
extern "C" int aide_conv3x3_wgrad_stem_supported(int Co, int Ci, int H, int W);                  // conv3x3_wgrad_stem.hip
extern "C" int aide_conv3x3_wgrad_stem(const void* dz, int dz_bf16, int64_t dz_bs, const float* x, int64_t x_bs, float* dw,
                                       int N, int Co, int Ci, int H, int W, float* ws, int splits, int round_bf16,
                                       void* queue, hipStream_t stream);

namespace {

struct WgradArgs {
    const float* dz;
    const float* a;
    float* slabs;
    long dz_bs, a_bs;
    int N, Co, Ci, H, W;
    int tiles_w, tiles_h, n_co_tiles, n_ci_tiles, splits, tiles_total;
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int WAVES_CO, int WAVES_CI, int WAVES_PX, bool RAGGED>
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_kernel(const WgradArgs g) {
    static_assert(WAVES_CO * WAVES_CI * WAVES_PX == 4, "4 waves");
    constexpr int TCO = WAVES_CO * 32, TCI = WAVES_CI * 32;
    constexpr int PT_H = 4, PT_W = 16, NPX = PT_H * PT_W;
    // dz tile [TCO][4][16]: channel stride 68 keeps 16-B alignment for ds_write_b128 (the A-fragment
    // read is then 4-way conflicted, one read per 9 MFMAs: irrelevant).  Halo tile [TCI][6][18]:
    // odd channel stride 109 -> the nine B-fragment reads per k-step are conflict-free.
    constexpr int DS = NPX + 4;
    constexpr int RS = PT_W + 2, CSR = (PT_H + 2) * RS, CS = CSR + 1;
    constexpr int DL = TCO * DS, AL = TCI * CS, BUF = DL + AL;
    constexpr int RED = (WAVES_PX > 1) ? (WAVES_CO * WAVES_CI) * 9 * 16 * 64 : 0;
    constexpr int KSTEPS = NPX / 2 / WAVES_PX;         // pixel pairs per wave per tile
    constexpr int HALF = KSTEPS / 2;
    // Staged units per thread and tile.  Dense sizes (H % 4 == 0, W % 16 == 0) use 16-byte loads:
    //   A: dz rows            TCO*4*4 float4        B: halo interior cols 0..15   TCI*6*4 float4
    //   C: halo edge columns -1 and 16              TCI*6*2 dwords
    // RAGGED sizes (deep UNet-320 levels) fall back to one dword per unit with full bounds tests.
    constexpr int NA = RAGGED ? (TCO * NPX + 255) / 256 : (TCO * 16 + 255) / 256;
    constexpr int NB = RAGGED ? (TCI * CSR + 255) / 256 : (TCI * 24 + 255) / 256;
    constexpr int NC = RAGGED ? 0 : (TCI * 12 + 255) / 256;
    constexpr int NL = NA + NB + NC;                   // global loads per thread per tile
    constexpr int NW = RAGGED ? NL : NA + 4 * NB + NC; // LDS stores per thread per tile
    constexpr int PERL = (NL + HALF - 1) / HALF;
    constexpr int PERW = (NW + HALF - 1) / HALF;
    static_assert(PERL <= 9 && PERW <= 9, "one staging op per MFMA slot");

    // Two LDS tile buffers: while the MFMAs chew on buffer `cur`, the next tile is fetched to
    // registers during the first half of the k-steps and written to the other buffer during the
    // second half, so the matrix pipe never waits for a staging burst (one wave per SIMD here).
    __shared__ __attribute__((aligned(16))) float lds[cmax(2 * BUF, RED)];
    static_assert(BUF % 4 == 0 && DL % 4 == 0, "16-byte aligned tile buffers");

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int wave_px = wid % WAVES_PX;
    const int wave_ci = (wid / WAVES_PX) % WAVES_CI;
    const int wave_co = wid / (WAVES_PX * WAVES_CI);

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int ci_tile = b % g.n_ci_tiles; b /= g.n_ci_tiles;
    const int co_tile = b % g.n_co_tiles;
    const int split = b / g.n_co_tiles;
    const int co0 = co_tile * TCO, ci0 = ci_tile * TCI;
    const int HW = g.H * g.W;

    const int tps = (g.tiles_total + g.splits - 1) / g.splits;
    const int t_begin = split * tps, t_end = min(t_begin + tps, g.tiles_total);

    // SRSRC buffer loads: 32-bit offsets; units that must read as zero carry the BUF_OOB offset
    // (hardware returns 0, no branch, no memory traffic).
    const __amdgpu_buffer_rsrc_t drs = make_rsrc(g.dz + (long)co0 * HW);
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(g.a + (long)ci0 * HW - (g.W + 1));

    // ---- tile-invariant per-thread unit descriptors (VGPRs; 512-register budget at 1 wave/SIMD) ----
    unsigned offA[NA], ldsA[NA];
    unsigned offB[NB], ldsB[NB], brdB[NB];
    unsigned offC[NC > 0 ? NC : 1], ldsC[NC > 0 ? NC : 1], brdC[NC > 0 ? NC : 1];
#pragma unroll
    for (int e = 0; e < NA; ++e) {
        const int q = tid + e * 256;
        if (RAGGED) {
            const int c = q / NPX, p = q - c * NPX, ph = p / PT_W, pw = p - ph * PT_W;
            const bool ok = q < TCO * NPX && (co0 + c) < g.Co;
            offA[e] = ok ? (unsigned)(c * HW + ph * g.W + pw) * 4u : BUF_OOB;
            ldsA[e] = q < TCO * NPX ? (unsigned)(c * DS + p) : 0xffffffffu;
        } else {
            const int c = q / 16, rem = q - c * 16, ph = rem / 4, s4 = rem - ph * 4;
            const bool ok = q < TCO * 16 && (co0 + c) < g.Co;
            offA[e] = ok ? (unsigned)(c * HW + ph * g.W + 4 * s4) * 4u : BUF_OOB;
            ldsA[e] = q < TCO * 16 ? (unsigned)(c * DS + ph * 16 + 4 * s4) : 0xffffffffu;
        }
    }
#pragma unroll
    for (int e = 0; e < NB; ++e) {
        const int q = tid + e * 256;
        if (RAGGED) {
            const int c = q / CSR, rem = q - c * CSR;
            const bool ok = q < TCI * CSR && (ci0 + c) < g.Ci;
            offB[e] = ok ? (unsigned)(c * HW + (rem / RS) * g.W + rem % RS) * 4u : BUF_OOB;
            ldsB[e] = q < TCI * CSR ? (unsigned)(DL + c * CS + rem) : 0xffffffffu;
            brdB[e] = 0;
        } else {
            const int c = q / 24, rem = q - c * 24, r = rem / 4, s4 = rem - r * 4;
            const bool ok = q < TCI * 24 && (ci0 + c) < g.Ci;
            offB[e] = ok ? (unsigned)(c * HW + r * g.W + 1 + 4 * s4) * 4u : BUF_OOB;
            ldsB[e] = q < TCI * 24 ? (unsigned)(DL + c * CS + r * RS + 1 + 4 * s4) : 0xffffffffu;
            brdB[e] = (r == 0 ? 1u : 0u) | (r == PT_H + 1 ? 2u : 0u);
        }
    }
#pragma unroll
    for (int e = 0; e < NC; ++e) {
        const int q = tid + e * 256;
        const int c = q / 12, rem = q - c * 12, r = rem / 2, side = rem - r * 2;
        const int col = side ? PT_W + 1 : 0;
        const bool ok = q < TCI * 12 && (ci0 + c) < g.Ci;
        offC[e] = ok ? (unsigned)(c * HW + r * g.W + col) * 4u : BUF_OOB;
        ldsC[e] = q < TCI * 12 ? (unsigned)(DL + c * CS + r * RS + col) : 0xffffffffu;
        brdC[e] = (r == 0 ? 1u : 0u) | (r == PT_H + 1 ? 2u : 0u) | (side ? 8u : 4u);
    }

    int th0 = 0, tw0 = 0;                  // origin of the tile being fetched
    unsigned dso = 0, aso = 0, tcode = 0;
    auto set_tile = [&](int t) {
        t = min(t, t_end - 1);             // past the end: refetch the last tile (never consumed)
        const int tw = t % g.tiles_w;
        const int r2 = t / g.tiles_w;
        const int th = r2 % g.tiles_h;
        const int tn = r2 / g.tiles_h;
        th0 = th * PT_H; tw0 = tw * PT_W;
        dso = (unsigned)((long)tn * g.dz_bs + th0 * g.W + tw0) * 4u;
        aso = (unsigned)((long)tn * g.a_bs + th0 * g.W + tw0) * 4u;
        // which image borders the tile touches: {top, bottom, left, right}
        tcode = (th0 == 0 ? 1u : 0u) | (th0 + PT_H >= g.H ? 2u : 0u) | (tw0 == 0 ? 4u : 0u) |
                (tw0 + PT_W >= g.W ? 8u : 0u);
    };

    f32x4 stA[NA], stB[NB];      // RAGGED uses only component 0
    float stC[NC > 0 ? NC : 1];
    // global load number l (compile-time) of the current set_tile()
    auto fetch = [&](int l) {
        if (l < NA) {
            unsigned off = offA[l];
            if (RAGGED) {
                const int p = (tid + l * 256) % NPX;
                if ((th0 + p / PT_W) >= g.H || (tw0 + p % PT_W) >= g.W) off = BUF_OOB;
                stA[l][0] = buf_load_f32(drs, off, dso);
            } else {
                stA[l] = buf_load_f32x4(drs, off, dso);
            }
        } else if (l < NA + NB) {
            const int e = l - NA;
            unsigned off = offB[e];
            if (RAGGED) {
                const int rem = (tid + e * 256) % CSR;
                const int ih = th0 - 1 + rem / RS, iw = tw0 - 1 + rem % RS;
                if (ih < 0 || ih >= g.H || iw < 0 || iw >= g.W) off = BUF_OOB;
                stB[e][0] = buf_load_f32(ars, off, aso);
            } else {
                if ((brdB[e] & tcode) != 0u) off = BUF_OOB;
                stB[e] = buf_load_f32x4(ars, off, aso);
            }
        } else {
            const int e = l - NA - NB;
            unsigned off = offC[e];
            if ((brdC[e] & tcode) != 0u) off = BUF_OOB;
            stC[e] = buf_load_f32(ars, off, aso);
        }
    };
    // LDS store number w (compile-time) into tile buffer `buf`
    auto put = [&](int w, float* buf) {
        if (w < NA) {
            if (ldsA[w] != 0xffffffffu) {
                if (RAGGED) buf[ldsA[w]] = stA[w][0];
                else *reinterpret_cast<f32x4*>(buf + ldsA[w]) = stA[w];
            }
        } else if (RAGGED) {
            const int e = w - NA;
            if (ldsB[e] != 0xffffffffu) buf[ldsB[e]] = stB[e][0];
        } else if (w < NA + 4 * NB) {
            const int e = (w - NA) / 4, i = (w - NA) % 4;
            if (ldsB[e] != 0xffffffffu) buf[ldsB[e] + i] = stB[e][i];
        } else {
            const int e = w - NA - 4 * NB;
            if (ldsC[e] != 0xffffffffu) buf[ldsC[e]] = stC[e];
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // this wave handles pixel pairs s = wave_px + WAVES_PX*u ; first pixel 2*s + half
    const int la_off = (wave_co * 32 + j) * DS + 2 * wave_px + half;
    const int lb_off = DL + (wave_ci * 32 + j) * CS + 2 * wave_px + half;
    const bool active = __builtin_amdgcn_readfirstlane(
        (int)((co0 + wave_co * 32) < g.Co && (ci0 + wave_ci * 32) < g.Ci)) != 0;   // wave-uniform

    // prologue: first tile straight into buffer 0
    set_tile(t_begin);
#pragma unroll
    for (int l = 0; l < NL; ++l) fetch(l);
#pragma unroll
    for (int w = 0; w < NW; ++w) put(w, lds);
    __syncthreads();

    int cur = 0;
    for (int t = t_begin; t < t_end; ++t) {
        const float* la = lds + cur * BUF + la_off;
        const float* lb = lds + cur * BUF + lb_off;
        float* nxt = lds + (cur ^ 1) * BUF;
        set_tile(t + 1);
        constexpr int STEP = 2 * WAVES_PX;                // pixels advanced per u (divides PT_W)
        // Issue order is pinned slot by slot: ONE MFMA, then a small slice of the other work (one
        // fragment read for the next k-step, at most one global fetch or one LDS store), so every
        // non-matrix instruction issues in the shadow of a 64-cycle MFMA.
        // (two separately named fragment sets: an array indexed by u&1 gets demoted to memory)
        float afA, afB, bfA[9], bfB[9];
        afA = la[0];
#pragma unroll
        for (int k = 0; k < 9; ++k) bfA[k] = lb[(k / 3) * RS + (k % 3)];
        auto kstep = [&](int u, float& afc, float (&bfc)[9], float& afn, float (&bfn)[9]) {
            const int pn = (u + 1) * STEP;                // first pixel of the next k-step
            const int phn = pn / PT_W, pwn = pn % PT_W;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(afc, bfc[k], acc[k], 0, 0, 0);
                if (u + 1 < KSTEPS) {
                    if (k == 0) afn = la[phn * 16 + pwn];
                    bfn[k] = lb[(phn + k / 3) * RS + pwn + (k % 3)];
                }
                if (u < HALF) {
                    if (k < PERL && u * PERL + k < NL) fetch(u * PERL + k);
                } else {
                    if (k < PERW && (u - HALF) * PERW + k < NW) put((u - HALF) * PERW + k, nxt);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll
        for (int u = 0; u < KSTEPS; u += 2) {
            kstep(u, afA, bfA, afB, bfB);
            kstep(u + 1, afB, bfB, afA, bfA);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- sum the WAVES_PX partial accumulators of each (co,ci) wave block through LDS ----
    if (WAVES_PX > 1) {
        float* red = lds + (wave_co * WAVES_CI + wave_ci) * (9 * 16 * 64);
        for (int gsrc = 1; gsrc < WAVES_PX; ++gsrc) {
            __syncthreads();
            if (wave_px == gsrc) {
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(k * 16 + r) * 64 + lane] = acc[k][r];
            }
            __syncthreads();
            if (wave_px == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[k][r] += red[(k * 16 + r) * 64 + lane];
            }
        }
    }

    if (wave_px == 0 && active) {   // partial (co,ci) blocks: idle waves multiplied zeros
        float* slab = g.slabs + (long)split * 9 * g.Co * g.Ci;
        const int ci = ci0 + wave_ci * 32 + j;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wave_co * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < g.Co && ci < g.Ci) slab[((long)k * g.Co + co) * g.Ci + ci] = acc[k][r];
            }
        }
    }
}

// dW[co][ci][t] = sum_s slab[s][t][co][ci].  Block = R consecutive (co, ci) pairs x 9 taps x YG split
// groups: thread (x, y) sums splits y, y + YG, ... for its 9 taps (nine independent coalesced row loads
// in flight per iteration), the YG partial sums are combined in a fixed order through LDS
// (deterministic), and the 576 results leave as one contiguous run of the [co][ci][9] layout.
template <int R, int YG>
__global__ __launch_bounds__(R * YG) void wgrad_reduce_kernel(const float* __restrict__ slabs, int splits, int Co,
                                                              int Ci, float* __restrict__ dw) {
    __shared__ float sm[YG][9][R];
    const long cc = (long)Co * Ci, total = 9 * cc;
    const int x = threadIdx.x % R, y = threadIdx.x / R;
    const long rem0 = (long)blockIdx.x * R;
    const bool ok = rem0 + x < cc;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    if (ok) {
        const float* p = slabs + rem0 + x;
#pragma unroll 2
        for (int s = y; s < splits; s += YG) {
            const float* q = p + (long)s * total;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] += q[t * cc];
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) sm[y][t][x] = acc[t];
    __syncthreads();
    const int n_out = (int)min((long)R, cc - rem0) * 9;
    for (int e = threadIdx.x; e < n_out; e += R * YG) {
        const int xx = e / 9, t = e - xx * 9;
        float r = sm[0][t][xx];
#pragma unroll 8
        for (int g = 1; g < YG; ++g) r += sm[g][t][xx];
        dw[rem0 * 9 + e] = r;
    }
}

// R (co, ci) pairs per block: 64 when that still gives >= 256 blocks, else 16 with more split groups
static int launch_wgrad_reduce_vec(const float* ws, int splits, int Co, int Ci, float* dw, hipStream_t stream);   // below

static int launch_wgrad_reduce(const float* ws, int splits, int Co, int Ci, float* dw, hipStream_t stream) {
    const long cc = (long)Co * Ci;
    // 16-byte form (one descriptor of the batched kernel): the dword kernels below keep 4 bytes per lane in flight and ran the
    // bf16 mode's per-layer reduces -- 1.3 ms per C5 step on the weight-gradient stream -- at a fraction of the bandwidth
    if (splits >= 2 && cc % 4 == 0 && ((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(dw)) & 15) == 0)
        return launch_wgrad_reduce_vec(ws, splits, Co, Ci, dw, stream);
    const bool narrow = cc / 64 < 256 && splits >= 16;
    const unsigned nb = (unsigned)((cc + (narrow ? 15 : 63)) / (narrow ? 16 : 64));
    if (narrow)
        AIDE_LAUNCH_TIMED(AIDE_KT_REDUCE, 0.0, (wgrad_reduce_kernel<16, 64>), dim3(nb), dim3(1024), 0, stream, ws, splits, Co, Ci, dw);
    else if (splits >= 32)
        AIDE_LAUNCH_TIMED(AIDE_KT_REDUCE, 0.0, (wgrad_reduce_kernel<64, 16>), dim3(nb), dim3(1024), 0, stream, ws, splits, Co, Ci, dw);
    else if (splits >= 3)
        AIDE_LAUNCH_TIMED(AIDE_KT_REDUCE, 0.0, (wgrad_reduce_kernel<64, 4>), dim3(nb), dim3(256), 0, stream, ws, splits, Co, Ci, dw);
    else
        AIDE_LAUNCH_TIMED(AIDE_KT_REDUCE, 0.0, (wgrad_reduce_kernel<64, 1>), dim3(nb), dim3(64), 0, stream, ws, splits, Co, Ci, dw);
    return aide_launch_status();
}


template <int WAVES_CO, int WAVES_CI, int WAVES_PX>
int launch_wgrad(WgradArgs g, hipStream_t stream) {
    g.n_co_tiles = (g.Co + WAVES_CO * 32 - 1) / (WAVES_CO * 32);
    g.n_ci_tiles = (g.Ci + WAVES_CI * 32 - 1) / (WAVES_CI * 32);
    const long nb = (long)g.n_co_tiles * g.n_ci_tiles * g.splits;
    if (g.H % 4 != 0 || g.W % 16 != 0)
        AIDE_LAUNCH_TIMED(AIDE_KT_WGRAD, AIDE_CONV_FLOPS(g.N, g.H, g.W, g.Co, g.Ci),
                          (conv3x3_wgrad_kernel<WAVES_CO, WAVES_CI, WAVES_PX, true>), dim3((unsigned)nb),
                          dim3(256), 0, stream, g);
    else
        AIDE_LAUNCH_TIMED(AIDE_KT_WGRAD, AIDE_CONV_FLOPS(g.N, g.H, g.W, g.Co, g.Ci),
                          (conv3x3_wgrad_kernel<WAVES_CO, WAVES_CI, WAVES_PX, false>), dim3((unsigned)nb),
                          dim3(256), 0, stream, g);
    return aide_launch_status();
}

int wgrad_variant(int Co, int Ci) {
    if (Co >= 64 && Ci >= 64) return 0;      // 64co x 64ci
    if (Ci >= 64) return 1;                  // 32co x 64ci, 2-way pixel split
    if (Co >= 64) return 2;                  // 64co x 32ci, 2-way pixel split
    return 3;                                // 32co x 32ci, 4-way pixel split
}

void wgrad_tiles(int variant, int Co, int Ci, int* nco, int* nci) {
    const int tco = (variant == 0 || variant == 2) ? 64 : 32;
    const int tci = (variant == 0 || variant == 1) ? 64 : 32;
    *nco = (Co + tco - 1) / tco;
    *nci = (Ci + tci - 1) / tci;
}

// ---- deferred, batched slab reduce.  Every weight-gradient kernel of a backward pass leaves its per-split slabs in its
// own workspace region; ONE launch then reduces the slabs of many layers (descriptors travel in the kernel-argument
// segment).  Replaces ~32 small, latency-bound reduce launches that serialised the weight-gradient stream.
struct RDesc {
    const float* ws; float* dw;
    int splits, Co, Ci, narrow;
    long block_start;
};
constexpr int RB_MAX = 48;
struct RBatch { int n, pad; RDesc d[RB_MAX]; };

template <int R, int YG>
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ slabs, int splits, int Co, int Ci,
                                                  float* __restrict__ dw, long blk, float* sm /*[YG][9][R]*/) {
    const long cc = (long)Co * Ci, total = 9 * cc;
    const int x = threadIdx.x % R, y = threadIdx.x / R;
    const long rem0 = blk * R;
    const bool ok = rem0 + x < cc;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    if (ok) {
        const float* p = slabs + rem0 + x;
#pragma unroll 2
        for (int s = y; s < splits; s += YG) {
            const float* q = p + (long)s * total;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] += q[t * cc];
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) sm[(y * 9 + t) * R + x] = acc[t];
    __syncthreads();
    const int n_out = (int)min((long)R, cc - rem0) * 9;
    for (int e = threadIdx.x; e < n_out; e += R * YG) {
        const int xx = e / 9, t = e - xx * 9;
        float r = sm[t * R + xx];
#pragma unroll 8
        for (int g = 1; g < YG; ++g) r += sm[(g * 9 + t) * R + xx];        // fixed order: deterministic
        dw[rem0 * 9 + e] = r;
    }
}

// 16 bytes per load: a workgroup = (256 / YG lanes x 4 consecutive (co, ci) pairs) x YG split groups; a thread sums the slabs
// s = y, y + YG, ... of its 4 pairs x 9 taps (36 accumulators, 9 independent 16-byte loads per slab), the YG partial sums are
// added in LDS in the fixed order y = 0 .. YG-1, and the [9][R] -> [R][9] transposition makes the dw stores contiguous.
// (The dword form above kept 4 bytes per lane in flight and ran the batched reduce at 1.3 TB/s.)  Co * Ci % 4 == 0.
constexpr int RM_PAD = 4;
template <int YG>
__device__ __forceinline__ void wgrad_reduce_body4(const float* __restrict__ slabs, int splits, int Co, int Ci,
                                                   float* __restrict__ dw, long blk, float* sm /*[YG][9][R + RM_PAD]*/) {
    constexpr int XT = 256 / YG, R = 4 * XT, RP = R + RM_PAD;
    const long cc = (long)Co * Ci, total = 9 * cc;
    const int x = threadIdx.x % XT, y = threadIdx.x / XT;
    const long rem0 = blk * R;
    const bool ok = rem0 + 4 * x < cc;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok) {
        const float* p = slabs + rem0 + 4 * x;
#pragma unroll 2
        for (int s = y; s < splits; s += YG) {
            const float* q = p + (long)s * total;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] += *reinterpret_cast<const f32x4*>(q + t * cc);
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4*>(sm + (y * 9 + t) * RP + 4 * x) = acc[t];
    __syncthreads();
    const int n_out = (int)min((long)R, cc - rem0) * 9;     // a multiple of 4 (cc, R are)
    for (int e4 = threadIdx.x * 4; e4 < n_out; e4 += 1024) {
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = e4 + k, xx = e / 9, t = e - xx * 9;
            float r = sm[t * RP + xx];
#pragma unroll
            for (int g = 1; g < YG; ++g) r += sm[(g * 9 + t) * RP + xx];    // fixed order: deterministic
            o[k] = r;
        }
        *reinterpret_cast<f32x4*>(dw + rem0 * 9 + e4) = o;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const RBatch b) {
    __shared__ __attribute__((aligned(16))) float sm[9 * (1024 + 64 * RM_PAD)];     // YG x 9 x (1024 / YG + pad), YG <= 64
    const long blk = blockIdx.x;
    int lo = 0, hi = b.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.d[mid].block_start <= blk) lo = mid; else hi = mid - 1;
    }
    const RDesc d = b.d[lo];
    if (d.narrow == 64) wgrad_reduce_body4<64>(d.ws, d.splits, d.Co, d.Ci, d.dw, blk - d.block_start, sm);
    else if (d.narrow == 16) wgrad_reduce_body4<16>(d.ws, d.splits, d.Co, d.Ci, d.dw, blk - d.block_start, sm);
    else if (d.narrow == 4) wgrad_reduce_body4<4>(d.ws, d.splits, d.Co, d.Ci, d.dw, blk - d.block_start, sm);
    else wgrad_reduce_body4<1>(d.ws, d.splits, d.Co, d.Ci, d.dw, blk - d.block_start, sm);
}

// a caller-owned queue of slab reduces that wait for ONE batched launch (aide_wgrad_queue_*): host memory only
struct RQueue { int n = 0; RDesc d[512]; };

// split groups per workgroup (RDesc::narrow): enough of them to keep every thread busy, few enough to leave
// >= 64 workgroups per layer
// (a layer with MANY splits takes 16 groups whatever its size: with 4, the stems -- 96 (co, ci) pairs x 256 slabs -- were ONE
// workgroup walking 64 slabs per thread, and 32->64 x 128 slabs eight workgroups walking 32: the last batched reduce of
// a backward pass, the launch everything else waits for, took 51 us for 40 MB)
static int reduce_groups(int splits, long cc) {
    // (64 groups for small layers with >= 64 slabs measured +-0 and is not built in)
    return splits >= 16 ? 16 : (splits >= 4 ? 4 : (splits >= 2 && cc <= 256 * 256 ? 4 : 1));
}

static int launch_wgrad_reduce_vec(const float* ws, int splits, int Co, int Ci, float* dw, hipStream_t stream) {
    RBatch b;
    b.n = 1; b.pad = 0;
    RDesc& r = b.d[0];
    r.ws = ws; r.dw = dw; r.splits = splits; r.Co = Co; r.Ci = Ci;
    const long cc = (long)Co * Ci;
    r.narrow = reduce_groups(splits, cc);
    r.block_start = 0;
    const int R = 1024 / r.narrow;
    AIDE_LAUNCH_TIMED(AIDE_KT_REDUCE, 0.0, wgrad_reduce_multi_kernel, dim3((unsigned)((cc + R - 1) / R)), dim3(256), 0, stream, b);
    return aide_launch_status();
}

}  // namespace

// shared with conv3x3_wgrad4.hip, conv3x3_wgrad_stem.hip, conv3x3_bf16.hip: reduce now (queue == nullptr, or the slabs do
// not fit the batched kernel), or remember the slabs in the caller's queue for its batched launch (aide_wgrad_queue_flush)
int aide_wgrad_reduce_launch(const float* ws, int splits, int Co, int Ci, float* dw, void* queue, hipStream_t stream) {
    RQueue* q = static_cast<RQueue*>(queue);
    if (q != nullptr && q->n < 512 && ((long)Co * Ci) % 4 == 0 && ((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(dw)) & 15) == 0) {
        RDesc& r = q->d[q->n++];
        r.ws = ws; r.dw = dw; r.splits = splits; r.Co = Co; r.Ci = Ci;
        r.narrow = reduce_groups(splits, (long)Co * Ci);
        r.block_start = 0;
        return AIDE_OK;
    }
    return launch_wgrad_reduce(ws, splits, Co, Ci, dw, stream);
}

extern "C" int aide_wgrad_queue_create(void** queue) {
    if (!queue) return AIDE_ERR_ARG;
    *queue = new (std::nothrow) RQueue();
    return *queue ? AIDE_OK : AIDE_ERR_ARG;
}

extern "C" int aide_wgrad_queue_destroy(void* queue) {
    delete static_cast<RQueue*>(queue);
    return AIDE_OK;
}

// error path of a backward pass: forget every pending descriptor (they point into a pass that did not finish)
extern "C" int aide_wgrad_queue_discard(void* queue) {
    if (!queue) return AIDE_ERR_ARG;
    RQueue* q = static_cast<RQueue*>(queue);
    const int n = q->n;
    q->n = 0;
    return n;
}

extern "C" int aide_wgrad_queue_pending(const void* queue) { return queue ? static_cast<const RQueue*>(queue)->n : AIDE_ERR_ARG; }

// one launch per RB_MAX pending layers, in the order they were queued
extern "C" int aide_wgrad_queue_flush(void* queue, hipStream_t stream) {
    if (!queue) return AIDE_ERR_ARG;
    RQueue& g_red = *static_cast<RQueue*>(queue);
    int done = 0, rc = AIDE_OK;
    while (done < g_red.n && rc == AIDE_OK) {
        RBatch b;
        b.n = 0; b.pad = 0;
        long blocks = 0;
        while (done < g_red.n && b.n < RB_MAX) {
            RDesc r = g_red.d[done++];
            const long cc = (long)r.Co * r.Ci;
            const int R = 1024 / r.narrow;
            r.block_start = blocks;
            blocks += (cc + R - 1) / R;
            b.d[b.n++] = r;
        }
        AIDE_LAUNCH_TIMED(AIDE_KT_REDUCE, 0.0, wgrad_reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, b);
        rc = aide_launch_status();
    }
    g_red.n = 0;
    return rc;
}


extern "C" {

// Number of pixel-range splits (= slabs) used for this problem.
int aide_conv3x3_wgrad_splits(int N, int Co, int Ci, int H, int W) {
    if (aide_conv3x3_wgrad_stem_supported(Co, Ci, H, W))        // folded-tap kernel: 4 x 64-pixel tiles, 4 workgroups per CU
        return aide_conv3x3_wgrad_stem_splits(N, H, W);
    int nco, nci;
    wgrad_tiles(wgrad_variant(Co, Ci), Co, Ci, &nco, &nci);
    const long tiles = (long)N * ((H + 3) / 4) * ((W + 15) / 16);
    // one workgroup per CU (two 45 KB tile buffers + 144 accumulator registers per lane): aim for
    // one full round of 256 workgroups, each paying the tile-pipeline prologue only once
    const long blocks = (long)nco * nci;
    const long target = 256;
    long s = (target + blocks - 1) / blocks;
    if (s > tiles) s = tiles;
    if (s < 1) s = 1;
    return (int)s;
}

size_t aide_conv3x3_wgrad_ws_bytes(int N, int Co, int Ci, int H, int W) {
    return (size_t)aide_conv3x3_wgrad_splits(N, Co, Ci, H, W) * 9 * Co * Ci * sizeof(float);
}

//   dz : [N][Co][H][W] (batch stride dz_bs)   a : [N][Ci][H][W] (batch stride a_bs)
//   dw : [Co][Ci][3][3]                       ws : aide_conv3x3_wgrad_ws_bytes() bytes
int aide_conv3x3_wgrad(const float* dz, int64_t dz_bs, const float* a, int64_t a_bs, float* dw,
                       int N, int Co, int Ci, int H, int W, float* ws, void* queue, hipStream_t stream) {
    if (!dz || !a || !dw || !ws || N <= 0 || Co <= 0 || Ci <= 0) return AIDE_ERR_ARG;
    if (aide_conv3x3_wgrad_stem_supported(Co, Ci, H, W) && dz_bs % 4 == 0)      // Ci <= 3: taps folded into the GEMM's N
        return aide_conv3x3_wgrad_stem(dz, 0, dz_bs, a, a_bs, dw, N, Co, Ci, H, W, ws,
                                       aide_conv3x3_wgrad_splits(N, Co, Ci, H, W), 0, queue, stream);
    WgradArgs g;
    g.dz = dz; g.a = a; g.slabs = ws; g.dz_bs = dz_bs; g.a_bs = a_bs;
    g.N = N; g.Co = Co; g.Ci = Ci; g.H = H; g.W = W;
    g.tiles_w = (W + 15) / 16; g.tiles_h = (H + 3) / 4;
    g.tiles_total = N * g.tiles_w * g.tiles_h;
    g.splits = aide_conv3x3_wgrad_splits(N, Co, Ci, H, W);
    int rc;
    switch (wgrad_variant(Co, Ci)) {
        case 0: rc = launch_wgrad<2, 2, 1>(g, stream); break;
        case 1: rc = launch_wgrad<1, 2, 2>(g, stream); break;
        case 2: rc = launch_wgrad<2, 1, 2>(g, stream); break;
        default: rc = launch_wgrad<1, 1, 4>(g, stream); break;
    }
    if (rc != 0) return rc;
    return aide_wgrad_reduce_launch(ws, g.splits, Co, Ci, dw, queue, stream);
}

}  // extern "C"

// =====================================================================================================
// Winograd form of the weight gradient:  dW = G^T [ sum_tiles (A Z A^T) (.) (B^T D B) ] G
// (the transposition of F(2x2,3x3): Z = 2x2 tile of dz, D = the 4x4 input patch, 16 multiplies per
// (tile, co, ci) instead of 36).  Per transform position p the sum over tiles is a GEMM
//   M[p][co][ci] += ZT[p][co][tile] * V[p][ci][tile]           (K = tiles, two per v_mfma_f32_32x32x2_f32)
// Workgroup = 64 co x 64 ci, wave = 32 x 32 x 16 positions (256 accumulator registers, 1 wave/SIMD);
// a chunk = 8 tiles of one tile row (2 x 16 output pixels).  Raw dz / input rows go global -> regs ->
// LDS, every lane transforms (channel = lane, tile = wave-slot) patches into the double-buffered
// ZT / V operands ([p][channel][8 tiles], XOR-swizzled so that fragment reads and transform writes are
// conflict-free without padding), all in the shadow of the MFMAs.  The raw LDS area is single
// buffered: a mid-chunk barrier separates its last transform read from the next chunk's store.
namespace {

struct WWArgs {
    const float* dz;
    const float* a;
    float* slabs;
    long dz_bs, a_bs;
    int N, Co, Ci, H, W;
    int rows_t, cols_c, n_co_tiles, n_ci_tiles, splits, chunks_total;
};

constexpr int WW_DSTR = 73, WW_ZSTR = 33;                 // odd raw channel strides (4x18 / 2x16 rows)
constexpr int WW_OP = 16 * 64 * 8;                        // one operand buffer (floats)
constexpr int WW_LDS = 4 * WW_OP + 64 * WW_DSTR + 64 * WW_ZSTR;   // 39552 floats = 154.5 KB

__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_wino_kernel(const WWArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ztb = lds;                       // ZT[2][16][512]
    float* vb = lds + 2 * WW_OP;            // V [2][16][512]
    float* rawD = lds + 4 * WW_OP;
    float* rawZ = rawD + 64 * WW_DSTR;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int wave_co = wid >> 1, wave_ci = wid & 1;

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int ci_tile = b % g.n_ci_tiles; b /= g.n_ci_tiles;
    const int co_tile = b % g.n_co_tiles;
    const int split = b / g.n_co_tiles;
    const int co0 = co_tile * 64, ci0 = ci_tile * 64;
    const int HW = g.H * g.W;
    const int cps = (g.chunks_total + g.splits - 1) / g.splits;
    const int c_begin = split * cps, c_end = min(c_begin + cps, g.chunks_total);

    const __amdgpu_buffer_rsrc_t drs = make_rsrc(g.dz + (long)co0 * HW);
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(g.a + (long)ci0 * HW - (g.W + 1));

    // ---- tile-invariant unit descriptors ----
    // D interior: 64 ci x 4 rows x 4 float4 (4 per thread); D edges: 64 x 4 x 2 dwords (2 per thread);
    // Z: 64 co x 2 rows x 4 float4 (2 per thread)
    unsigned offDi[4], ldsDi[4], rowDi[4], colDi[4], offDe[2], ldsDe[2], rowDe[2], sideDe[2], offZ[2], ldsZ[2], colZ[2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int q = tid + e * 256, c = q >> 4, r = (q >> 2) & 3, s4 = q & 3;
        offDi[e] = (ci0 + c) < g.Ci ? (unsigned)(c * HW + r * g.W + 1 + 4 * s4) * 4u : BUF_OOB;
        ldsDi[e] = (unsigned)(c * WW_DSTR + r * 18 + 1 + 4 * s4);
        rowDi[e] = r; colDi[e] = 4 * s4;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int q = tid + e * 256, c = q >> 3, r = (q >> 1) & 3, side = q & 1;
        offDe[e] = (ci0 + c) < g.Ci ? (unsigned)(c * HW + r * g.W + (side ? 17 : 0)) * 4u : BUF_OOB;
        ldsDe[e] = (unsigned)(c * WW_DSTR + r * 18 + (side ? 17 : 0));
        rowDe[e] = r; sideDe[e] = side;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int q = tid + e * 256, c = q >> 3, r = (q >> 2) & 1, s4 = q & 3;
        offZ[e] = (co0 + c) < g.Co ? (unsigned)(c * HW + r * g.W + 4 * s4) * 4u : BUF_OOB;
        ldsZ[e] = (unsigned)(c * WW_ZSTR + r * 16 + 4 * s4);
        colZ[e] = 4 * s4;
    }

    int th0 = 0, tw0 = 0;
    unsigned dso = 0, aso = 0;
    auto set_chunk = [&](int c) {
        c = min(c, c_end - 1);                 // past the end: refetch the last chunk (never consumed)
        const int cc = c % g.cols_c;
        const int r2 = c / g.cols_c;
        const int tr = r2 % g.rows_t, n = r2 / g.rows_t;
        th0 = 2 * tr; tw0 = 16 * cc;
        dso = (unsigned)((long)n * g.dz_bs + th0 * g.W + tw0) * 4u;
        aso = (unsigned)((long)n * g.a_bs + th0 * g.W + tw0) * 4u;
    };
    // The fp32 MFMA shares the vector-ALU pipe (tools/ubench/mfma_shadow.hip): VALU instructions between
    // MFMAs are never hidden (~5 cycles each + ~13 for the first in a slot) while LDS / VMEM / scalar
    // instructions are.  So per chunk ALL vector-ALU work is issued in two clusters: the halo selects of
    // the eight fetch offsets (prep) and the transform adds (xf_math, packed: the thread's two items sit
    // in the halves of f32x2 registers -> v_pk_add_f32).
    f32x4 rDi[4], rZ[2];
    float rDe[2];
    unsigned vo[8];
    auto prep = [&]() {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int ih = th0 - 1 + (int)rowDi[l];
            vo[l] = (ih >= 0 && ih < g.H && tw0 + (int)colDi[l] < g.W) ? offDi[l] : BUF_OOB;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ih = th0 - 1 + (int)rowDe[e], iw = sideDe[e] ? tw0 + 16 : tw0 - 1;
            vo[4 + e] = (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) ? offDe[e] : BUF_OOB;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) vo[6 + e] = (tw0 + (int)colZ[e] < g.W) ? offZ[e] : BUF_OOB;
    };
    auto fetch = [&](int l) {                  // 8 global loads per thread and chunk
        if (l < 4) rDi[l] = buf_load_f32x4(ars, vo[l], aso);
        else if (l < 6) rDe[l - 4] = buf_load_f32(ars, vo[l], aso);
        else rZ[l - 6] = buf_load_f32x4(drs, vo[l], dso);
    };
    auto put = [&](int w) {                    // 26 LDS stores per thread and chunk
        if (w < 16) rawD[ldsDi[w >> 2] + (w & 3)] = rDi[w >> 2][w & 3];
        else if (w < 18) rawD[ldsDe[w - 16]] = rDe[w - 16];
        else rawZ[ldsZ[(w - 18) >> 2] + ((w - 18) & 3)] = rZ[(w - 18) >> 2][(w - 18) & 3];
    };

    // transforms: item e in {0,1} (register half) -> tile = wid + 4 e of the chunk, channel = lane.
    // Z side: rows of A Z are (z0, z0+z1, z0-z1, -z1) and columns likewise; the two negations are NOT
    // applied here - they flip the sign of whole positions (p >= 12, p % 4 == 3) and are folded into the
    // output transform instead.
    const int swz = (lane >> 2) & 7;
    const int rd0 = lane * WW_DSTR + 2 * wid, rz0 = lane * WW_ZSTR + 2 * wid;
    const int ws0 = lane * 8 + (wid ^ swz), ws1 = lane * 8 + ((wid + 4) ^ swz);
    f32x2 td[16], to[16], tz[4], zo[16];
    auto d_read = [&](int i) {
        td[i].x = rawD[rd0 + (i >> 2) * 18 + (i & 3)];
        td[i].y = rawD[rd0 + (i >> 2) * 18 + (i & 3) + 8];
    };
    auto z_read = [&](int i) {
        tz[i].x = rawZ[rz0 + (i >> 1) * 16 + (i & 1)];
        tz[i].y = rawZ[rz0 + (i >> 1) * 16 + (i & 1) + 8];
    };
    auto xf_math = [&]() {
        f32x2 tt[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            tt[c] = td[c] - td[8 + c]; tt[4 + c] = td[4 + c] + td[8 + c];
            tt[8 + c] = td[8 + c] - td[4 + c]; tt[12 + c] = td[4 + c] - td[12 + c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            to[r * 4] = tt[r * 4] - tt[r * 4 + 2]; to[r * 4 + 1] = tt[r * 4 + 1] + tt[r * 4 + 2];
            to[r * 4 + 2] = tt[r * 4 + 2] - tt[r * 4 + 1]; to[r * 4 + 3] = tt[r * 4 + 1] - tt[r * 4 + 3];
        }
        // (z rows) u[0] = (a0, a1), u[1] = (a0 + b0, a1 + b1), u[2] = (a0 - b0, a1 - b1), u[3] = (b0, b1) [sign folded]
        const f32x2 u0[4] = {tz[0], tz[0] + tz[2], tz[0] - tz[2], tz[2]};
        const f32x2 u1[4] = {tz[1], tz[1] + tz[3], tz[1] - tz[3], tz[3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            zo[r * 4] = u0[r]; zo[r * 4 + 1] = u0[r] + u1[r]; zo[r * 4 + 2] = u0[r] - u1[r]; zo[r * 4 + 3] = u1[r];
        }
    };
    auto d_store = [&](int o, float* vbuf) { vbuf[o * 512 + ws0] = to[o].x; vbuf[o * 512 + ws1] = to[o].y; };
    auto z_store = [&](int o, float* zbuf) { zbuf[o * 512 + ws0] = zo[o].x; zbuf[o * 512 + ws1] = zo[o].y; };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

    // fragment offsets: channel row + swizzled tile slot for the four tile pairs of a chunk
    const int cho_a = (wave_co * 32 + j) * 8, sw_a = ((wave_co * 32 + j) >> 2) & 7;
    const int cho_b = (wave_ci * 32 + j) * 8, sw_b = ((wave_ci * 32 + j) >> 2) & 7;
    int fa[4], fb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { fa[s] = cho_a + ((2 * s + half) ^ sw_a); fb[s] = cho_b + ((2 * s + half) ^ sw_b); }

    // ---- prologue ----
    set_chunk(c_begin);
    prep();
#pragma unroll
    for (int l = 0; l < 8; ++l) fetch(l);
#pragma unroll
    for (int w = 0; w < 26; ++w) put(w);
    set_chunk(c_begin + 1);
    prep();
#pragma unroll
    for (int l = 0; l < 8; ++l) fetch(l);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) d_read(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) z_read(i);
    xf_math();
#pragma unroll
    for (int o = 0; o < 16; ++o) { d_store(o, vb); z_store(o, ztb); }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 26; ++w) put(w);
    __syncthreads();

    int cur = 0;
    for (int c = c_begin; c < c_end; ++c) {
        const float* za = ztb + cur * WW_OP;
        const float* va = vb + cur * WW_OP;
        float* zn = ztb + (cur ^ 1) * WW_OP;
        float* vn = vb + (cur ^ 1) * WW_OP;
        set_chunk(c + 2);
        // four rotating fragment sets, requested three slots before their MFMA (see conv3x3_winograd.hip)
        float af0, bf0, af1, bf1, af2, bf2, af3, bf3;
        auto frag = [&](int st, float& af, float& bf) {
            const int p = st >> 2, s = st & 3;
            af = za[p * 512 + fa[s]];
            bf = va[p * 512 + fb[s]];
        };
        auto slot = [&](int st, float& afc, float& bfc, float& afn, float& bfn) {
            if (st + 3 < 64) frag(st + 3, afn, bfn);
            acc[st >> 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(afc, bfc, acc[st >> 2], 0, 0, 0);
            // schedule (vector-ALU work only in slots 0 and 20):
            //    0      halo selects of the fetch offsets;  0..7 global fetches of raw[c+2]
            //    1..16  input patch reads, 17..18 dz tile reads (both items each)
            //   20      all transform adds;   21 barrier: the raw LDS area is free
            //   22..37  V stores, 38..53 ZT stores (both items each), 28..53 raw[c+2] stores
            if (st == 0) prep();
            if (st < 8) fetch(st);
            if (st >= 1 && st < 17) d_read(st - 1);
            if (st == 17 || st == 18) { z_read(2 * (st - 17)); z_read(2 * (st - 17) + 1); }
            if (st == 20) xf_math();
            if (st == 21) __syncthreads();
            if (st >= 22 && st < 38) d_store(st - 22, vn);
            if (st >= 38 && st < 54) z_store(st - 38, zn);
            if (st >= 28 && st < 54) put(st - 28);
            __builtin_amdgcn_sched_barrier(0);
        };
        frag(0, af0, bf0); frag(1, af1, bf1); frag(2, af2, bf2);
#pragma unroll
        for (int st = 0; st < 64; st += 4) {
            slot(st, af0, bf0, af3, bf3);
            slot(st + 1, af1, bf1, af0, bf0);
            slot(st + 2, af2, bf2, af1, bf1);
            slot(st + 3, af3, bf3, af2, bf2);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- dW = G^T M G per lane (co = accumulator row, ci = lane column), write the split's slab ----
    float* slab = g.slabs + (long)split * 9 * g.Co * g.Ci;
    const int ci = ci0 + wave_ci * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wave_co * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float t[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // positions with p % 4 == 3 and p >= 12 carry a folded sign (see the Z transform above)
            const float sg = c == 3 ? -1.0f : 1.0f;
            const float m0 = sg * acc[c][r], m1 = sg * acc[4 + c][r], m2 = sg * acc[8 + c][r], m3 = -sg * acc[12 + c][r];
            const float hs = 0.5f * (m1 + m2), hd = 0.5f * (m1 - m2);
            t[0][c] = m0 + hs; t[1][c] = hd; t[2][c] = hs + m3;
        }
        if (co < g.Co && ci < g.Ci) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float hs = 0.5f * (t[i][1] + t[i][2]), hd = 0.5f * (t[i][1] - t[i][2]);
                slab[((long)(i * 3 + 0) * g.Co + co) * g.Ci + ci] = t[i][0] + hs;
                slab[((long)(i * 3 + 1) * g.Co + co) * g.Ci + ci] = hd;
                slab[((long)(i * 3 + 2) * g.Co + co) * g.Ci + ci] = hs + t[i][3];
            }
        }
    }
}

}  // namespace

extern "C" {

int aide_conv3x3_wgrad_wino_supported(int Co, int Ci, int H, int W) {
    return (H % 2 == 0 && W % 4 == 0 && Co >= 64 && Ci >= 64) ? 1 : 0;
}

int aide_conv3x3_wgrad_wino_splits(int N, int Co, int Ci, int H, int W) {
    const long blocks = (long)((Co + 63) / 64) * ((Ci + 63) / 64);
    const long chunks = (long)N * (H / 2) * ((W + 15) / 16);
    const long target = 256;
    long s = (target + blocks - 1) / blocks;
    if (s > chunks) s = chunks;
    return (int)(s < 1 ? 1 : s);
}

size_t aide_conv3x3_wgrad_wino_ws_bytes(int N, int Co, int Ci, int H, int W) {
    return (size_t)aide_conv3x3_wgrad_wino_splits(N, Co, Ci, H, W) * 9 * Co * Ci * sizeof(float);
}

// Same contract as aide_conv3x3_wgrad (dw [Co][Ci][3][3]); requires aide_conv3x3_wgrad_wino_supported.
int aide_conv3x3_wgrad_wino(const float* dz, int64_t dz_bs, const float* a, int64_t a_bs, float* dw, int N,
                            int Co, int Ci, int H, int W, float* ws, void* queue, hipStream_t stream) {
    if (!dz || !a || !dw || !ws || !aide_conv3x3_wgrad_wino_supported(Co, Ci, H, W) || dz_bs % 4 || a_bs % 4)
        return AIDE_ERR_ARG;
    static AideLdsOptIn lds_opt;             // per device, status checked (common.h)
    if (int rc = lds_opt.ensure([] {
            return hipFuncSetAttribute((const void*)conv3x3_wgrad_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       WW_LDS * (int)sizeof(float));
        })) return rc;
    WWArgs g;
    g.dz = dz; g.a = a; g.slabs = ws; g.dz_bs = dz_bs; g.a_bs = a_bs;
    g.N = N; g.Co = Co; g.Ci = Ci; g.H = H; g.W = W;
    g.rows_t = H / 2; g.cols_c = (W + 15) / 16;
    g.n_co_tiles = (Co + 63) / 64; g.n_ci_tiles = (Ci + 63) / 64;
    g.chunks_total = N * g.rows_t * g.cols_c;
    g.splits = aide_conv3x3_wgrad_wino_splits(N, Co, Ci, H, W);
    const long nb = (long)g.n_co_tiles * g.n_ci_tiles * g.splits;
    AIDE_LAUNCH_TIMED(AIDE_KT_WGRAD_WINO2, AIDE_CONV_FLOPS(N, H, W, Co, Ci), conv3x3_wgrad_wino_kernel, dim3((unsigned)nb),
                      dim3(256), WW_LDS * sizeof(float), stream, g);
    int rc = aide_launch_status();
    if (rc != 0) return rc;
    return aide_wgrad_reduce_launch(ws, g.splits, Co, Ci, dw, queue, stream);
}

}  // extern "C"
