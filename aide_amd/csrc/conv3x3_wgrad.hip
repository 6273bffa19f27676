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

template <int WAVES_CO, int WAVES_CI, int WAVES_PX>
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_kernel(const WgradArgs g) {
    static_assert(WAVES_CO * WAVES_CI * WAVES_PX == 4, "4 waves");
    constexpr int TCO = WAVES_CO * 32, TCI = WAVES_CI * 32;
    constexpr int PT_H = 4, PT_W = 16, NPX = PT_H * PT_W;
    constexpr int DS = NPX + 1;                       // odd strides: conflict-free fragment reads
    constexpr int RS = PT_W + 2, CSR = (PT_H + 2) * RS, CS = CSR + 1;
    constexpr int DL = TCO * DS, AL = TCI * CS;
    constexpr int ED = (TCO * NPX + 255) / 256;
    constexpr int EA = (TCI * CSR + 255) / 256;
    constexpr int RED = (WAVES_PX > 1) ? (WAVES_CO * WAVES_CI) * 9 * 16 * 64 : 0;
    constexpr int KSTEPS = NPX / 2 / WAVES_PX;         // pixel pairs per wave per tile

    __shared__ float lds[cmax(DL + AL, RED)];
    float* dl = lds;
    float* al = lds + DL;

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

    // Element offsets are recomputed per tile from the thread id (constant divisions on the VALU,
    // which idles beside the MFMA pipe) instead of living in ~90 registers. Loads are SRSRC buffer
    // loads: 32-bit offsets, and out-of-image / out-of-channel elements read 0.0f via BUF_OOB.
    const __amdgpu_buffer_rsrc_t drs = make_rsrc(g.dz + (long)co0 * HW);
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(g.a + (long)ci0 * HW - (g.W + 1));
    float dr[ED], ar[EA];
    auto load_tile = [&](int t) {
        const int tw = t % g.tiles_w;
        const int r2 = t / g.tiles_w;
        const int th = r2 % g.tiles_h, n = r2 / g.tiles_h;
        const int h0 = th * PT_H, w0 = tw * PT_W;
        const unsigned dso = (unsigned)((long)n * g.dz_bs + h0 * g.W + w0) * 4u;
#pragma unroll
        for (int e = 0; e < ED; ++e) {
            const int idx = tid + e * 256;
            const int c = idx / NPX, p = idx - c * NPX;
            const int ph = p / PT_W, pw = p - ph * PT_W;
            const bool ok = idx < TCO * NPX && (co0 + c) < g.Co && (h0 + ph) < g.H && (w0 + pw) < g.W;
            dr[e] = buf_load_f32(drs, ok ? (unsigned)(c * HW + ph * g.W + pw) * 4u : BUF_OOB, dso);
        }
        const unsigned aso = (unsigned)((long)n * g.a_bs + h0 * g.W + w0) * 4u;
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int idx = tid + e * 256;
            const int c = idx / CSR, rem = idx - c * CSR;
            const int r = rem / RS, col = rem - r * RS;
            const int ih = h0 - 1 + r, iw = w0 - 1 + col;
            const bool ok = idx < TCI * CSR && (ci0 + c) < g.Ci && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
            ar[e] = buf_load_f32(ars, ok ? (unsigned)(c * HW + r * g.W + col) * 4u : BUF_OOB, aso);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int e = 0; e < ED; ++e) {
            const int idx = tid + e * 256;
            const int c = idx / NPX, p = idx - c * NPX;
            if (idx < TCO * NPX) dl[c * DS + p] = dr[e];
        }
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int idx = tid + e * 256;
            const int c = idx / CSR, rem = idx - c * CSR;
            if (idx < TCI * CSR) al[c * CS + rem] = ar[e];
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // this wave handles pixel pairs s = wave_px + WAVES_PX*u ; first pixel 2*s + half
    const float* la = dl + (wave_co * 32 + j) * DS + 2 * wave_px + half;
    const float* lb = al + (wave_ci * 32 + j) * CS + 2 * wave_px + half;
    const bool active = (co0 + wave_co * 32) < g.Co && (ci0 + wave_ci * 32) < g.Ci;

    if (t_begin < t_end) load_tile(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (t + 1 < t_end) load_tile(t + 1);
        if (active) {
            // explicit one-step-ahead fragment pipeline; the sched_barrier keeps hipcc from hoisting
            // all KSTEPS*10 LDS reads above the MFMAs (which blew the register budget)
            constexpr int STEP = 2 * WAVES_PX;           // pixels advanced per u (divides PT_W)
            float af[2], bf[2][9];
            auto frag = [&](int u, int buf) {
                const int p = u * STEP;                  // compile-time after unrolling
                const int ph = p / PT_W, pw = p % PT_W;
                af[buf] = la[p];
#pragma unroll
                for (int k = 0; k < 9; ++k) bf[buf][k] = lb[(ph + k / 3) * RS + pw + (k % 3)];
            };
            frag(0, 0);
#pragma unroll
            for (int u = 0; u < KSTEPS; ++u) {
                if (u + 1 < KSTEPS) frag(u + 1, (u + 1) & 1);
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u & 1], bf[u & 1][k], acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
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

    if (wave_px == 0 && active) {
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

// dW[co][ci][t] = sum_s slab[s][t][co][ci]   (thread per (co,ci); fixed order)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, int splits, int Co, int Ci,
                                    float* __restrict__ dw) {
    const long total = (long)Co * Ci;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = slabs[(long)t * total + i];
        for (int s = 1; s < splits; ++s) {
            const float* sl = slabs + (long)s * 9 * total;
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] += sl[(long)t * total + i];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) dw[i * 9 + t] = v[t];
    }
}

template <int WAVES_CO, int WAVES_CI, int WAVES_PX>
int launch_wgrad(WgradArgs g, hipStream_t stream) {
    g.n_co_tiles = (g.Co + WAVES_CO * 32 - 1) / (WAVES_CO * 32);
    g.n_ci_tiles = (g.Ci + WAVES_CI * 32 - 1) / (WAVES_CI * 32);
    const long nb = (long)g.n_co_tiles * g.n_ci_tiles * g.splits;
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<WAVES_CO, WAVES_CI, WAVES_PX>), dim3((unsigned)nb),
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

}  // namespace

extern "C" {

// Number of pixel-range splits (= slabs) used for this problem.
int aide_conv3x3_wgrad_splits(int N, int Co, int Ci, int H, int W) {
    int nco, nci;
    wgrad_tiles(wgrad_variant(Co, Ci), Co, Ci, &nco, &nci);
    const long tiles = (long)N * ((H + 3) / 4) * ((W + 15) / 16);
    long s = (512 + (long)nco * nci - 1) / ((long)nco * nci);   // ~2 workgroups per CU
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
                       int N, int Co, int Ci, int H, int W, float* ws, hipStream_t stream) {
    if (!dz || !a || !dw || !ws || N <= 0 || Co <= 0 || Ci <= 0) return AIDE_ERR_ARG;
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
    const long total = (long)Co * Ci;
    const int blocks = (int)min((total + 255) / 256, (long)2048);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, ws, g.splits, Co, Ci, dw);
    return aide_launch_status();
}

}  // extern "C"
