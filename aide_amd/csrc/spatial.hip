// MaxPool2d(2,2) and bilinear x2 (align_corners=True) up-sampling, forward and backward (gfx950).
//
// Replaces (reference): nn.MaxPool2d(2,2) at models_twomodalinputs/fuseunet.py:13-31 (call sites
// :51-78) and UNet.py:114; nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) at
// netblocks.py:16 / UNet.py:11.  Semantics per SURVEY.md §A.2:
//   * pool backward routes the gradient to the FIRST maximum in row-major window order
//     ((0,0),(0,1),(1,0),(1,1); aten's test is `val > maxval`), ties are common on CHAOS data;
//   * bilinear: src = dst*(in-1)/(out-1) in fp32, i0 = floor(src), i1 = min(i0+1, in-1).
// All kernels are HBM-bound streaming kernels on NCHW planes with explicit batch strides.
#include "common.h"

namespace {

template <typename XT, typename YT>
__global__ __launch_bounds__(256) void maxpool2x2_fwd_kernel(const XT* __restrict__ x, long x_bs,
                                                             YT* __restrict__ y, long y_bs, int C, int H,
                                                             int W, long total) {
    const int Ho = H / 2, Wo = W / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ow2 = (int)(i % (Wo / 2));               // two outputs per thread (float4 in, float2 out)
        long r = i / (Wo / 2);
        const int oh = (int)(r % Ho); r /= Ho;
        const int c = (int)(r % C);
        const long n = r / C;
        const XT* p = x + n * x_bs + (long)c * H * W + (long)(2 * oh) * W + 4 * ow2;
        const f32x4 a = ld4(p);
        const f32x4 b = ld4(p + W);
        f32x2 o;
        o[0] = fmaxf(fmaxf(a[0], a[1]), fmaxf(b[0], b[1]));
        o[1] = fmaxf(fmaxf(a[2], a[3]), fmaxf(b[2], b[3]));
        st2(y + n * y_bs + (long)c * Ho * Wo + (long)oh * Wo + 2 * ow2, o);     // max of bf16 values is exact in bf16
    }
}

__device__ __forceinline__ int first_argmax4(float v0, float v1, float v2, float v3) {
    int k = 0; float m = v0;
    if (v1 > m) { m = v1; k = 1; }
    if (v2 > m) { m = v2; k = 2; }
    if (v3 > m) { m = v3; k = 3; }
    return k;
}

// dx (+)= scatter of dy to the first arg-max of each window (windows do not overlap -> no atomics)
template <typename XT, typename GT, typename DT>
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const XT* __restrict__ x, long x_bs,
                                                             const GT* __restrict__ dy, long dy_bs,
                                                             DT* __restrict__ dx, long dx_bs, int C, int H,
                                                             int W, int accumulate, long total) {
    const int Ho = H / 2, Wo = W / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ow2 = (int)(i % (Wo / 2));
        long r = i / (Wo / 2);
        const int oh = (int)(r % Ho); r /= Ho;
        const int c = (int)(r % C);
        const long n = r / C;
        const long in_off = (long)c * H * W + (long)(2 * oh) * W + 4 * ow2;
        const XT* p = x + n * x_bs + in_off;
        const f32x4 a = ld4(p);
        const f32x4 b = ld4(p + W);
        const f32x2 g2 = ld2(dy + n * dy_bs + (long)c * Ho * Wo + (long)oh * Wo + 2 * ow2);
        const float2 g = make_float2(g2[0], g2[1]);
        const int k0 = first_argmax4(a[0], a[1], b[0], b[1]);
        const int k1 = first_argmax4(a[2], a[3], b[2], b[3]);
        f32x4 ta = {k0 == 0 ? g.x : 0.f, k0 == 1 ? g.x : 0.f, k1 == 0 ? g.y : 0.f, k1 == 1 ? g.y : 0.f};
        f32x4 tb = {k0 == 2 ? g.x : 0.f, k0 == 3 ? g.x : 0.f, k1 == 2 ? g.y : 0.f, k1 == 3 ? g.y : 0.f};
        DT* q = dx + n * dx_bs + in_off;
        if (accumulate) {
            ta += ld4(q);
            tb += ld4(q + W);
        }
        st4(q, ta);
        st4(q + W, tb);
    }
}

__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    // the product is rounded before the subtraction below, as aten does (left to -ffp-contract=fast, `src - i0` becomes an
    // FMA on the exact product in some instantiations and the weight moves by an ulp of src: 1.5e-5 at column 191)
    float src = scale * (float)dst;
    asm volatile("" : "+v"(src));             // (HIP's __fmul_rn is a plain product: it does not stop the contraction)
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l1 = src - (float)i0;
}

__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* __restrict__ x, long x_bs,
                                                             float* __restrict__ y, long y_bs, int C, int H,
                                                             int W, long total) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float sh = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sw = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ow = (int)(i % Wo);
        long r = i / Wo;
        const int oh = (int)(r % Ho); r /= Ho;
        const int c = (int)(r % C);
        const long n = r / C;
        int h0, h1, w0, w1; float lh, lw;
        src_index(oh, sh, H, h0, h1, lh);
        src_index(ow, sw, W, w0, w1, lw);
        const float* p = x + n * x_bs + (long)c * H * W;
        const float v00 = p[h0 * W + w0], v01 = p[h0 * W + w1], v10 = p[h1 * W + w0], v11 = p[h1 * W + w1];
        const float top = (1.f - lw) * v00 + lw * v01, bot = (1.f - lw) * v10 + lw * v11;
        y[n * y_bs + (long)c * Ho * Wo + (long)oh * Wo + ow] = (1.f - lh) * top + lh * bot;
    }
}

// 4 consecutive outputs per thread, one 16-byte store (the scalar form ran at 1.1 TB/s)
// One (n, c) plane per blockIdx.y, 32-bit index arithmetic (the flat 64-bit div/mod form of this kernel was VALU-bound
// at 1.9 TB/s).  A thread produces four consecutive outputs of one row: their sources lie in the four columns
// wl .. wl + 3 of two source rows (3 * scale < 1.5), read once each.
template <typename XT, typename YT>
__global__ __launch_bounds__(256) void upsample2x_fwd_vec_kernel(const XT* __restrict__ x, long x_bs,
                                                                 YT* __restrict__ y, long y_bs, int C, int H,
                                                                 int W, int per_plane4, int gx) {
    const int Ho = 2 * H, Wo = 2 * W, Wo4 = Wo / 4;
    const float sh = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sw = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int plane = blockIdx.x / gx, bx = blockIdx.x - plane * gx, n = plane / C, c = plane - n * C;     // (1-D grid: gridDim.y stops at 65535 planes)
    const XT* xp = x + (long)n * x_bs + (long)c * H * W;
    YT* yp = y + (long)n * y_bs + (long)c * Ho * Wo;
    for (int e = bx * 256 + threadIdx.x; e < per_plane4; e += gx * 256) {
        const int oh = e / Wo4, ow4 = e - oh * Wo4;
        int h0, h1; float lh;
        src_index(oh, sh, H, h0, h1, lh);
        const XT* p0 = xp + h0 * W;
        const XT* p1 = xp + h1 * W;
        int wl, wdummy; float ldummy;
        src_index(4 * ow4, sw, W, wl, wdummy, ldummy);
        float t[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int wc = min(wl + k, W - 1);
            t[k] = ld1(p0 + wc); b[k] = ld1(p1 + wc);
        }
        float v[4];                                   // rows blended first: v[k] = (1 - lh) t[k] + lh b[k]
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (1.f - lh) * t[k] + lh * b[k];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int w0, w1; float lw;
            src_index(4 * ow4 + k, sw, W, w0, w1, lw);
            const int d0 = w0 - wl, d1 = w1 - wl;     // in 0 .. 3
            const float a0 = d0 == 0 ? v[0] : d0 == 1 ? v[1] : d0 == 2 ? v[2] : v[3];
            const float a1 = d1 == 0 ? v[0] : d1 == 1 ? v[1] : d1 == 2 ? v[2] : v[3];
            o[k] = (1.f - lw) * a0 + lw * a1;
        }
        st4(yp + (long)oh * Wo + 4 * ow4, o);
    }
}

// Forward on whole tiles: a workgroup produces 32 x 128 outputs of one (n, c) plane from the <= 18 x 66 source pixels
// they reference, staged once in LDS as fp32 (aligned 16 / 8-byte loads).  A thread owns 8 consecutive output columns
// of two output rows: the column indices / weights are computed once per thread and the row ones once per row, so an
// output costs 4 LDS reads and ~13 VALU instead of 2 global loads and ~25 VALU -- the gather form above is instruction-
// bound (the same elements per second on bf16 storage as on fp32: 1.9 TB/s against 3.4).
constexpr int UF_TR = 32, UF_TC = 128, UF_SR = UF_TR / 2 + 2, UF_SW = 72;     // source window: 18 rows x 72 columns
template <typename XT, typename YT>
__global__ __launch_bounds__(256) void upsample2x_fwd_tiled_kernel(const XT* __restrict__ x, long x_bs,
                                                                   YT* __restrict__ y, long y_bs, int C, int H,
                                                                   int W, int tiles_w, int tiles_h, float sh, float sw) {
    // sh, sw = (in - 1) / (out - 1) come from the host, as aten computes them (a device-side division in this kernel
    // came out one ulp off: 3.5e-5 at column 335)
    __shared__ __attribute__((aligned(16))) float win[UF_SR * UF_SW];
    const int Ho = 2 * H, Wo = 2 * W;
    int b = blockIdx.x;
    const int tw = b % tiles_w; b /= tiles_w;
    const int th = b % tiles_h; const int plane = b / tiles_h;
    const int n = plane / C, c = plane - n * C;
    const XT* xp = x + (long)n * x_bs + (long)c * H * W;
    YT* yp = y + (long)n * y_bs + (long)c * Ho * Wo;
    const int oh0 = th * UF_TR, ow0 = tw * UF_TC;
    int r0, c0, d0; float f0;
    src_index(oh0, sh, H, r0, d0, f0);
    src_index(ow0, sw, W, c0, d0, f0);
    c0 &= ~3;                                             // 16-byte (fp32) / 8-byte (bf16) aligned window start
    // ---- source window -> LDS (rows / columns past the plane are never referenced: the indices are clamped) ----
    for (int u = threadIdx.x; u < UF_SR * (UF_SW / 4); u += 256) {
        const int r = u / (UF_SW / 4), q = u - r * (UF_SW / 4);
        const int ih = r0 + r, iw = c0 + 4 * q;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ih < H && iw < W) v = ld4(xp + ih * W + iw);  // W % 4 == 0: a unit is inside the row or outside
        *reinterpret_cast<f32x4*>(win + r * UF_SW + 4 * q) = v;
    }
    // ---- per-thread columns ----
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    int o0[8], o1[8]; float lw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int w0, w1;
        src_index(ow0 + 8 * cg + k, sw, W, w0, w1, lw[k]);
        o0[k] = w0 - c0; o1[k] = w1 - c0;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int oh = oh0 + rl + 16 * rr;
        int h0, h1; float lh;
        src_index(oh, sh, H, h0, h1, lh);
        const float* t = win + (h0 - r0) * UF_SW;
        const float* bt = win + (h1 - r0) * UF_SW;
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v0 = (1.f - lh) * t[o0[k]] + lh * bt[o0[k]];      // rows blended first, as the gather kernels do
            const float v1 = (1.f - lh) * t[o1[k]] + lh * bt[o1[k]];
            o[k] = (1.f - lw[k]) * v0 + lw[k] * v1;
        }
        YT* q = yp + (long)oh * Wo + ow0 + 8 * cg;
        st4(q, f32x4{o[0], o[1], o[2], o[3]});
        st4(q + 4, f32x4{o[4], o[5], o[6], o[7]});
    }
}

// Tiled, separable form of the transpose for whole planes: a workgroup owns 16 x 64 source pixels of one
// (n, c) plane, stages the 36 x 132 destination pixels that can reference them in LDS (8-byte coalesced
// loads), applies the column weights (<= 6 taps per source column, tables built once per workgroup), then the
// row weights.  12 LDS reads + 12 FMAs per source pixel instead of 36 gathers + 12 index computations.
constexpr int UB_TH = 16, UB_TW = 64, UB_RH = 2 * UB_TH + 4, UB_RW = 2 * UB_TW + 4;

// weights of the (<= 6) destination candidates lo .. lo + 5 (lo = max(0, 2 i - 2)) of source index i
__device__ __forceinline__ int up_bwd_weights(int i, int in_size, int out_size, float sc, float (&wt)[6]) {
    const int lo = max(0, 2 * i - 2);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int a0, a1; float l;
        wt[k] = 0.f;
        if (i < in_size && lo + k < out_size) {
            src_index(lo + k, sc, in_size, a0, a1, l);
            if (a0 == i) wt[k] += 1.f - l;
            if (a1 == i) wt[k] += l;
        }
    }
    return lo;
}

// The window is stored de-interleaved (even / odd destination columns in separate arrays) so that both the 8-byte
// global loads and the six-tap column pass touch consecutive LDS words per lane (the interleaved layout was 2-way
// bank-conflicted on every access); the tap weights live in registers (a thread's source column, and its four source
// rows, are fixed for the whole workgroup), not in LDS tables: 6 + 6 LDS reads per source pixel instead of 13 + 12.
// FAST (destination rows 16-byte aligned: pointer, batch stride and 2 W elements): the window comes in as aligned 16-byte
// pieces of a slightly wider column range (it starts 2 W0 - 8 / 2 W0 - 4 pixels into the row for bf16 / fp32 instead of
// 2 W0 - 2) -- 3 / 5 loads per thread instead of ten 4- / 8-byte ones, and one ds_write_b128 / two ds_write_b64 each: the
// kernel is a stream of the destination tensor and spends most of its life beside the weight gradient on a quarter of the
// CUs, where the bytes a wave has in flight are what it gets done.
// Round 6: a workgroup owns ONE tile position and walks the planes plane0, plane0 + pstep, ... with it: the tap-weight
// tables are built once per workgroup instead of once per tile, and the window of the NEXT plane is loaded into registers
// while the two passes of the current one run (the kernel lives beside the weight gradient on the CUs that one leaves: what
// a wave has in flight there is what it gets done -- one tile per workgroup streamed 1.0 TB/s in the C2 step, 0.5 in C5).
template <typename GT, typename DT, bool FAST>
__global__ __launch_bounds__(256) void upsample2x_bwd_tiled_kernel(const GT* __restrict__ dy, long dy_bs,
                                                                   DT* __restrict__ dx, long dx_bs, int C, int H,
                                                                   int W, int tiles_w, int tiles, int planes, int pstep,
                                                                   int accumulate) {
    constexpr int HW2 = UB_RW / 2;                          // 66 column pairs
    // bf16-stored gradients keep their two bytes in the window: one packed word per column pair instead of two floats --
    // 21 KB of LDS per workgroup instead of 31 KB, i.e. seven resident workgroups per CU instead of five (and more of them
    // beside the weight-gradient workgroups of the side stream)
    constexpr bool PK = sizeof(GT) == 2;
    constexpr int CPX = PK ? 8 : 4;                         // FAST: pixels per 16-byte piece
    constexpr int LEAD = PK ? 6 : 2;                        //       the aligned range starts LEAD pixels left of the window
    constexpr int NCHK = PK ? 18 : 34;                      //       pieces per window row (144 / 136 pixels)
    constexpr int TES = FAST ? 68 : HW2 + 1, TPS = FAST ? 72 : HW2 + 1;      // row strides (dwords)
    constexpr int QL = FAST ? LEAD / 2 : 0;                 // pair index of window column 0 in a stored row
    __shared__ __attribute__((aligned(16))) float te[PK ? 1 : UB_RH][TES];
    __shared__ __attribute__((aligned(16))) float to[PK ? 1 : UB_RH][TES];
    __shared__ __attribute__((aligned(16))) unsigned tp[PK ? UB_RH : 1][TPS];
    __shared__ float hp[UB_RH][UB_TW + 1];
    const int Ho = 2 * H, Wo = 2 * W;
    const float sh = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sw = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int tid = threadIdx.x;
    // 1-D grid, tile index fastest (gridDim.y would stop at 65535 planes: a stacked batch of 4 x 32 images at C = 512)
    const int plane0 = blockIdx.x / tiles, tile = blockIdx.x - plane0 * tiles;
    const int h0 = (tile / tiles_w) * UB_TH, w0 = (tile % tiles_w) * UB_TW;
    const int R0 = 2 * h0 - 2, C0 = 2 * w0 - 2;            // destination window origin (may be -2, always even)
    const int x = tid & 63, rq = tid >> 6;                  // this thread's source column / first source row
    // tap-weight tables, built once per workgroup by 80 threads (every thread building its own cost more VALU time
    // than the whole tile), then held in registers: a thread's column and its four rows never change
    __shared__ float wtab[UB_TH + UB_TW][6];
    __shared__ int otab[UB_TH + UB_TW];
    if (tid < UB_TH + UB_TW) {
        const bool row = tid < UB_TH;
        float wt[6];
        const int lo = row ? up_bwd_weights(h0 + tid, H, Ho, sh, wt) : up_bwd_weights(w0 + tid - UB_TH, W, Wo, sw, wt);
#pragma unroll
        for (int k = 0; k < 6; ++k) wtab[tid][k] = wt[k];
        otab[tid] = lo - (row ? R0 : C0);
    }
    // the window of one plane as this thread's registers: element offsets inside a destination plane (fixed: the tile does not
    // move), -1 = outside the plane (reads a clamped address, zeroed).  All loads of a thread are issued before the first LDS
    // store (a load -> wait -> store loop exposed the full memory latency ten times per workgroup: 0.9 TB/s).
    constexpr int NLD = (UB_RH * HW2 + 255) / 256;
    constexpr int NLF = (UB_RH * NCHK + 255) / 256;
    constexpr int NV = FAST ? NLF : NLD;
    int woff[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int e = tid + k * 256;
        if constexpr (FAST) {
            const int r = e / NCHK, ck = e - r * NCHK;
            const int oh = R0 + r, ow = C0 - LEAD + ck * CPX;          // (a multiple of CPX, as Wo is: a piece is all in or all out)
            woff[k] = (e < UB_RH * NCHK && oh >= 0 && oh < Ho && ow >= 0 && ow + CPX <= Wo) ? oh * Wo + ow : -1;
        } else {
            const int r = e / HW2, c2 = e - r * HW2;
            const int oh = R0 + r, ow = C0 + 2 * c2;
            woff[k] = (e < UB_RH * HW2 && oh >= 0 && oh < Ho && ow >= 0 && ow < Wo) ? oh * Wo + ow : -1;
        }
    }
    u32x4 vf[FAST ? NLF : 1];
    unsigned vp[(!FAST && PK) ? NLD : 1];
    float2 v2[(!FAST && !PK) ? NLD : 1];
    auto fetch = [&](int plane) {
        const int n = plane / C, c = plane - n * C;
        const GT* g = dy + (long)n * dy_bs + (long)c * Ho * Wo;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const bool ok = woff[k] >= 0;
            const GT* q = g + (ok ? woff[k] : 0);
            if constexpr (FAST) {
                const u32x4 t = *reinterpret_cast<const u32x4*>(q);
                vf[k] = ok ? t : u32x4{0u, 0u, 0u, 0u};
            } else if constexpr (PK) {
                const unsigned t = *reinterpret_cast<const unsigned*>(q);
                vp[k] = ok ? t : 0u;
            } else {
                const f32x2 t = ld2(q);
                v2[k] = ok ? make_float2(t[0], t[1]) : make_float2(0.f, 0.f);
            }
        }
    };
    auto put = [&]() {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e = tid + k * 256;
            if constexpr (FAST) {
                if (e < UB_RH * NCHK) {
                    const int r = e / NCHK, ck = e - r * NCHK;
                    if constexpr (PK) {
                        *reinterpret_cast<u32x4*>(&tp[r][ck * 4]) = vf[k];
                    } else {                                    // (e0, o0, e1, o1) -> even / odd column arrays
                        *reinterpret_cast<u32x2*>(&te[r][ck * 2]) = u32x2{vf[k][0], vf[k][2]};
                        *reinterpret_cast<u32x2*>(&to[r][ck * 2]) = u32x2{vf[k][1], vf[k][3]};
                    }
                }
            } else if constexpr (PK) {
                if (e < UB_RH * HW2) tp[e / HW2][e % HW2] = vp[k];
            } else {
                if (e < UB_RH * HW2) {
                    const int r = e / HW2, c2 = e - r * HW2;
                    te[r][c2] = v2[k].x; to[r][c2] = v2[k].y;
                }
            }
        }
    };
    if (plane0 >= planes) return;
    fetch(plane0);
    __syncthreads();                                        // the tables
    float wc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) wc[k] = wtab[UB_TH + x][k];
    const int oc = otab[UB_TH + x];                         // first candidate column inside the window
    const int q0 = (oc >> 1) + QL;                          // oc is even: candidates start at max(0, 2 i - 2), C0 is even
    for (int plane = plane0; plane < planes; plane += pstep) {
        put();
        __syncthreads();
        if (plane + pstep < planes) fetch(plane + pstep);   // in flight while this plane's two passes run
        // column pass: hp[r][x] = sum_l wc[l] * window[r][oc + l]; window column q lives in (q & 1 ? to : te)[r][q >> 1]
        for (int r = rq; r < UB_RH; r += 4) {
            if constexpr (PK) {
                const unsigned a = tp[r][q0], b = tp[r][q0 + 1], d = tp[r][q0 + 2];       // (even | odd << 16) column pairs
                hp[r][x] = wc[0] * __builtin_bit_cast(float, a << 16) + wc[1] * __builtin_bit_cast(float, a & 0xffff0000u) +
                           wc[2] * __builtin_bit_cast(float, b << 16) + wc[3] * __builtin_bit_cast(float, b & 0xffff0000u) +
                           wc[4] * __builtin_bit_cast(float, d << 16) + wc[5] * __builtin_bit_cast(float, d & 0xffff0000u);
            } else {
                hp[r][x] = wc[0] * te[r][q0] + wc[1] * to[r][q0] + wc[2] * te[r][q0 + 1] + wc[3] * to[r][q0 + 1] +
                           wc[4] * te[r][q0 + 2] + wc[5] * to[r][q0 + 2];
            }
        }
        __syncthreads();
        const int n = plane / C, c = plane - n * C;
#pragma unroll
        for (int k4 = 0; k4 < UB_TH / 4; ++k4) {
            const int r = rq + 4 * k4;
            float wr[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) wr[k] = wtab[r][k];     // wave-uniform: LDS broadcast
            const int o = otab[r];
            if (h0 + r < H && w0 + x < W) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 6; ++k) a += wr[k] * hp[o + k][x];
                DT* q = dx + (long)n * dx_bs + (long)c * H * W + (long)(h0 + r) * W + w0 + x;
                st1(q, accumulate ? (ld1(q) + a) : a);
            }
        }
        // (the next put() overwrites te / to / tp, which the column pass has left behind the barrier above; hp is rewritten
        // only after the next put()'s barrier, behind this plane's row pass)
    }
}

// gather form of the transpose: each source pixel sums the destination pixels that referenced it
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dy, long dy_bs,
                                                             float* __restrict__ dx, long dx_bs, int C, int H,
                                                             int W, int accumulate, long total) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float sh = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sw = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int iw = (int)(i % W);
        long r = i / W;
        const int ih = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const long n = r / C;
        // candidate destination rows/cols: dst with floor(scale*dst) in {ih-1, ih}
        float wh[6], ww[6];
        const int oh_lo = max(0, 2 * ih - 2), ow_lo = max(0, 2 * iw - 2);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            int a0, a1; float l;
            const int oh = oh_lo + k;
            wh[k] = 0.f;
            if (oh < Ho) {
                src_index(oh, sh, H, a0, a1, l);
                if (a0 == ih) wh[k] += 1.f - l;
                if (a1 == ih) wh[k] += (a1 != a0) ? l : l;      // i1 == i0 only at the last row (l == 0)
            }
            const int ow = ow_lo + k;
            ww[k] = 0.f;
            if (ow < Wo) {
                src_index(ow, sw, W, a0, a1, l);
                if (a0 == iw) ww[k] += 1.f - l;
                if (a1 == iw) ww[k] += l;
            }
        }
        const float* g = dy + n * dy_bs + (long)c * Ho * Wo;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (wh[k] == 0.f) continue;
            float rowacc = 0.f;
#pragma unroll
            for (int l = 0; l < 6; ++l)
                if (ww[l] != 0.f) rowacc += ww[l] * g[(long)(oh_lo + k) * Wo + ow_lo + l];
            acc += wh[k] * rowacc;
        }
        float* q = dx + n * dx_bs + (long)c * H * W + (long)ih * W + iw;
        *q = accumulate ? (*q + acc) : acc;
    }
}

// scalar forms for widths that are not a multiple of 4 (tiny bottom levels of odd-sized inputs)
__global__ __launch_bounds__(256) void maxpool2x2_fwd_scalar_kernel(const float* __restrict__ x, long x_bs,
                                                                    float* __restrict__ y, long y_bs, int C,
                                                                    int H, int W, long total) {
    const int Ho = H / 2, Wo = W / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ow = (int)(i % Wo);
        long r = i / Wo;
        const int oh = (int)(r % Ho); r /= Ho;
        const int c = (int)(r % C);
        const long n = r / C;
        const float* p = x + n * x_bs + (long)c * H * W + (long)(2 * oh) * W + 2 * ow;
        y[n * y_bs + (long)c * Ho * Wo + (long)oh * Wo + ow] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[W], p[W + 1]));
    }
}

__global__ __launch_bounds__(256) void maxpool2x2_bwd_scalar_kernel(const float* __restrict__ x, long x_bs,
                                                                    const float* __restrict__ dy, long dy_bs,
                                                                    float* __restrict__ dx, long dx_bs, int C,
                                                                    int H, int W, int accumulate, long total) {
    const int Ho = H / 2, Wo = W / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ow = (int)(i % Wo);
        long r = i / Wo;
        const int oh = (int)(r % Ho); r /= Ho;
        const int c = (int)(r % C);
        const long n = r / C;
        const long off = (long)c * H * W + (long)(2 * oh) * W + 2 * ow;
        const float* p = x + n * x_bs + off;
        const float g = dy[n * dy_bs + (long)c * Ho * Wo + (long)oh * Wo + ow];
        const int k = first_argmax4(p[0], p[1], p[W], p[W + 1]);
        float* q = dx + n * dx_bs + off;
        const float v[4] = {k == 0 ? g : 0.f, k == 1 ? g : 0.f, k == 2 ? g : 0.f, k == 3 ? g : 0.f};
        q[0] = accumulate ? q[0] + v[0] : v[0];
        q[1] = accumulate ? q[1] + v[1] : v[1];
        q[W] = accumulate ? q[W] + v[2] : v[2];
        q[W + 1] = accumulate ? q[W + 1] + v[3] : v[3];
    }
}

__global__ __launch_bounds__(256) void fill_zero_scalar_kernel(float* __restrict__ p, long bs, long chw, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / chw, r = i - n * chw;
        p[n * bs + r] = 0.f;
    }
}

// zero a channel slice [N][C][HW] of a strided tensor
__global__ __launch_bounds__(256) void fill_zero_kernel(float* __restrict__ p, long bs, long chw4, long total4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long n = i / chw4, r = i - n * chw4;
        *reinterpret_cast<f32x4*>(p + n * bs + r * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// Inverse augmentation of logit planes (train_files/trainchaos_proposed_30cases1labeled.py:81-95):
// optional horizontal flip, then PIL Image.rotate(angle, BILINEAR) on mode 'F' images.  PIL semantics
// (Image.rotate + Geometry.c affine_transform / bilinear_filter32F): inverse affine map about the
// centre (w/2, h/2) evaluated at pixel centres in double precision, samples outside the source are 0,
// neighbours are clamped in x, the y+1 row falls back to the y row at the bottom edge; multiples of
// 90 degrees take PIL's exact transpose paths.  par[n] = {a, b, c, d, e, f, flip, mode}.
__global__ __launch_bounds__(256) void reverse_aug_kernel(const float* __restrict__ x, long x_bs,
                                                          float* __restrict__ y, long y_bs,
                                                          const double* __restrict__ par, int C, int H, int W,
                                                          long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % W);
        long r = i / W;
        const int oy = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const long n = r / C;
        const double* m = par + n * 8;
        const bool flip = m[6] != 0.0;
        const int mode = (int)m[7];
        const float* src = x + n * x_bs + (long)c * H * W;
        auto at = [&](int yy, int xx) { return src[(long)yy * W + (flip ? W - 1 - xx : xx)]; };
        float v;
        if (mode == 1) v = at(oy, ox);
        else if (mode == 2) v = at(H - 1 - oy, W - 1 - ox);
        else if (mode == 3) v = at(ox, W - 1 - oy);            // ROTATE_90 (square images only)
        else if (mode == 4) v = at(H - 1 - ox, oy);            // ROTATE_270
        else {
            const double xin = m[0] * (ox + 0.5) + m[1] * (oy + 0.5) + m[2];
            const double yin = m[3] * (ox + 0.5) + m[4] * (oy + 0.5) + m[5];
            if (xin < 0.0 || xin >= (double)W || yin < 0.0 || yin >= (double)H) {
                v = 0.f;
            } else {
                const double xi = xin - 0.5, yi = yin - 0.5;
                const double fx = floor(xi), fy = floor(yi);
                const double dx = xi - fx, dy = yi - fy;
                const int x0 = (int)fx, y0 = (int)fy;
                const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
                const int yc0 = min(max(y0, 0), H - 1);
                const double p00 = at(yc0, xc0), p01 = at(yc0, xc1);
                double v1 = p00 + (p01 - p00) * dx, v2 = v1;
                if (y0 + 1 >= 0 && y0 + 1 < H) {
                    const double p10 = at(y0 + 1, xc0), p11 = at(y0 + 1, xc1);
                    v2 = p10 + (p11 - p10) * dx;
                }
                v = (float)(v1 + (v2 - v1) * dy);
            }
        }
        y[n * y_bs + (long)c * H * W + (long)oy * W + ox] = v;
    }
}

int grid_for(long total) { return (int)max(1L, min((total + 255) / 256, 8192L)); }

}  // namespace

extern "C" {

}  // extern "C"

namespace {
template <typename XT, typename YT>
int maxpool_fwd_t(const XT* x, int64_t x_bs, YT* y, int64_t y_bs, int N, int C, int H, int W, hipStream_t stream) {
    const long total = (long)N * C * (H / 2) * (W / 4);
    AIDE_LAUNCH_TIMED(AIDE_KT_POOL, (double)N * C * H * W * (sizeof(XT) + 0.25 * sizeof(YT)), (maxpool2x2_fwd_kernel<XT, YT>), dim3(grid_for(total)), dim3(256), 0, stream, x, (long)x_bs, y,
                       (long)y_bs, C, H, W, total);
    return aide_launch_status();
}
template <typename XT, typename GT, typename DT>
int maxpool_bwd_t(const XT* x, int64_t x_bs, const GT* dy, int64_t dy_bs, DT* dx, int64_t dx_bs, int N, int C, int H,
                  int W, int accumulate, hipStream_t stream) {
    const long total = (long)N * C * (H / 2) * (W / 4);
    // (x to find the arg-max, dy, dx written; an accumulating launch reads dx as well)
    AIDE_LAUNCH_TIMED(AIDE_KT_POOL, (double)N * C * H * W * (sizeof(XT) + 0.25 * sizeof(GT) + (accumulate ? 2.0 : 1.0) * sizeof(DT)),
                      (maxpool2x2_bwd_kernel<XT, GT, DT>), dim3(grid_for(total)), dim3(256), 0, stream, x, (long)x_bs, dy,
                       (long)dy_bs, dx, (long)dx_bs, C, H, W, accumulate, total);
    return aide_launch_status();
}
template <typename XT, typename YT>
int upsample_fwd_t(const XT* x, int64_t x_bs, YT* y, int64_t y_bs, int N, int C, int H, int W, hipStream_t stream) {
    const double kt_bytes = (double)N * C * H * W * (sizeof(XT) + 4.0 * sizeof(YT));      // source read once, 4x the pixels written
    if ((2 * H) % UF_TR == 0 && (2 * W) % UF_TC == 0 && x_bs % 4 == 0 && y_bs % 8 == 0) {   // whole 32 x 128 output tiles
        const int tiles_w = 2 * W / UF_TC, tiles_h = 2 * H / UF_TR;
        AIDE_LAUNCH_TIMED(AIDE_KT_UPSAMPLE, kt_bytes, (upsample2x_fwd_tiled_kernel<XT, YT>), dim3((unsigned)((long)tiles_w * tiles_h * N * C)), dim3(256),
                           0, stream, x, (long)x_bs, y, (long)y_bs, C, H, W, tiles_w, tiles_h,
                           (float)(H - 1) / (float)(2 * H - 1), (float)(W - 1) / (float)(2 * W - 1));
        return aide_launch_status();
    }
    const int per_plane4 = H * W;                       // (2H * 2W) / 4 four-pixel outputs per plane
    const int gx = max(1, min((per_plane4 + 255) / 256, 64));
    AIDE_LAUNCH_TIMED(AIDE_KT_UPSAMPLE, kt_bytes, (upsample2x_fwd_vec_kernel<XT, YT>), dim3((unsigned)((long)gx * N * C)), dim3(256), 0, stream, x,
                      (long)x_bs, y, (long)y_bs, C, H, W, per_plane4, gx);
    return aide_launch_status();
}
}  // namespace

extern "C" {

int aide_maxpool2x2_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int N, int C, int H, int W,
                        hipStream_t stream) {
    if (H % 2 || W % 2) return AIDE_ERR_ARG;
    if (W % 4 || x_bs % 4 || y_bs % 2) {
        const long total = (long)N * C * (H / 2) * (W / 2);
        AIDE_LAUNCH_TIMED(AIDE_KT_POOL, (double)N * C * H * W * 5.0, maxpool2x2_fwd_scalar_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x,
                           (long)x_bs, y, (long)y_bs, C, H, W, total);
        return aide_launch_status();
    }
    return maxpool_fwd_t<float, float>(x, x_bs, y, y_bs, N, C, H, W, stream);
}

// bf16-stored activations (precision='bf16'): x / y element types behind the untyped pointers; W % 4 == 0 required
int aide_maxpool2x2_fwd_mixed(const void* x, int x_bf16, int64_t x_bs, void* y, int y_bf16, int64_t y_bs, int N, int C,
                              int H, int W, hipStream_t stream) {
    if (H % 2 || W % 4 || x_bs % 4 || y_bs % 2) return AIDE_ERR_ARG;
    if (x_bf16) return y_bf16 ? maxpool_fwd_t((const bf16_store_t*)x, x_bs, (bf16_store_t*)y, y_bs, N, C, H, W, stream)
                              : maxpool_fwd_t((const bf16_store_t*)x, x_bs, (float*)y, y_bs, N, C, H, W, stream);
    return y_bf16 ? maxpool_fwd_t((const float*)x, x_bs, (bf16_store_t*)y, y_bs, N, C, H, W, stream)
                  : maxpool_fwd_t((const float*)x, x_bs, (float*)y, y_bs, N, C, H, W, stream);
}

int aide_maxpool2x2_bwd(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, float* dx,
                        int64_t dx_bs, int N, int C, int H, int W, int accumulate, hipStream_t stream) {
    if (H % 2 || W % 2) return AIDE_ERR_ARG;
    if (W % 4 || x_bs % 4 || dx_bs % 4 || dy_bs % 2) {
        const long total = (long)N * C * (H / 2) * (W / 2);
        AIDE_LAUNCH_TIMED(AIDE_KT_POOL, (double)N * C * H * W * (accumulate ? 13.0 : 9.0), maxpool2x2_bwd_scalar_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x,
                           (long)x_bs, dy, (long)dy_bs, dx, (long)dx_bs, C, H, W, accumulate, total);
        return aide_launch_status();
    }
    return maxpool_bwd_t<float, float, float>(x, x_bs, dy, dy_bs, dx, dx_bs, N, C, H, W, accumulate, stream);
}

int aide_maxpool2x2_bwd_mixed(const void* x, int x_bf16, int64_t x_bs, const void* dy, int dy_bf16, int64_t dy_bs, void* dx,
                              int dx_bf16, int64_t dx_bs, int N, int C, int H, int W, int accumulate, hipStream_t stream) {
    if (H % 2 || W % 4 || x_bs % 4 || dx_bs % 4 || dy_bs % 2) return AIDE_ERR_ARG;
#define AIDE_PB(XT, GT, DT) maxpool_bwd_t((const XT*)x, x_bs, (const GT*)dy, dy_bs, (DT*)dx, dx_bs, N, C, H, W, accumulate, stream)
#define AIDE_PB_X(GT, DT) (x_bf16 ? AIDE_PB(bf16_store_t, GT, DT) : AIDE_PB(float, GT, DT))
    if (dy_bf16) return dx_bf16 ? AIDE_PB_X(bf16_store_t, bf16_store_t) : AIDE_PB_X(bf16_store_t, float);
    return dx_bf16 ? AIDE_PB_X(float, bf16_store_t) : AIDE_PB_X(float, float);
#undef AIDE_PB_X
#undef AIDE_PB
}

// workgroups of the tiled up-sampling backward per tile position: each walks planes g, g + groups, ... -- about UB_WGS
// workgroups per launch (five per CU: what its 31 KB of LDS allow)
constexpr int UB_WGS = 1280;
static inline int ub_plane_groups(int tiles, int planes) {
    long g = (UB_WGS + tiles - 1) / tiles;
    if (g > planes) g = planes;
    if (g < 1) g = 1;
    return (int)g;
}

// may the tiled up-sampling backward read its destination rows as aligned 16-byte pieces?  (es = bytes per element)
static inline bool ub_fast(const void* dy, int64_t dy_bs, int W, int es) {
    const int per = 16 / es;
    return (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (dy_bs % per) == 0 && ((2 * W) % per) == 0;
}

int aide_upsample2x_bilinear_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int N, int C, int H,
                                 int W, hipStream_t stream) {
    const long total = (long)N * C * 4 * H * W;
    if ((2 * W) % 4 == 0 && y_bs % 4 == 0) return upsample_fwd_t<float, float>(x, x_bs, y, y_bs, N, C, H, W, stream);
    AIDE_LAUNCH_TIMED(AIDE_KT_UPSAMPLE, (double)N * C * H * W * 20.0, upsample2x_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, (long)x_bs, y,
                       (long)y_bs, C, H, W, total);
    return aide_launch_status();
}

// interpolation in fp32 from fp32 / bf16-stored sources into an fp32 / bf16-stored destination (W % 2 == 0)
int aide_upsample2x_bilinear_fwd_mixed(const void* x, int x_bf16, int64_t x_bs, void* y, int y_bf16, int64_t y_bs, int N,
                                       int C, int H, int W, hipStream_t stream) {
    if ((2 * W) % 4 || y_bs % 4) return AIDE_ERR_ARG;
    if (x_bf16) return y_bf16 ? upsample_fwd_t((const bf16_store_t*)x, x_bs, (bf16_store_t*)y, y_bs, N, C, H, W, stream)
                              : upsample_fwd_t((const bf16_store_t*)x, x_bs, (float*)y, y_bs, N, C, H, W, stream);
    return y_bf16 ? upsample_fwd_t((const float*)x, x_bs, (bf16_store_t*)y, y_bs, N, C, H, W, stream)
                  : upsample_fwd_t((const float*)x, x_bs, (float*)y, y_bs, N, C, H, W, stream);
}

int aide_upsample2x_bilinear_bwd(const float* dy, int64_t dy_bs, float* dx, int64_t dx_bs, int N, int C,
                                 int H, int W, int accumulate, hipStream_t stream) {
    const long total = (long)N * C * H * W;
    const double kt_bytes = (double)total * (16.0 + (accumulate ? 8.0 : 4.0));     // 4x the pixels read, dx written (read: accumulate)
    if (dy_bs % 2 == 0) {                                  // 8-byte loads of the destination rows
        const int tw = (W + UB_TW - 1) / UB_TW, th = (H + UB_TH - 1) / UB_TH;
        const int pstep = ub_plane_groups(tw * th, N * C);
        const dim3 grid((unsigned)((long)tw * th * pstep));
        if (ub_fast(dy, dy_bs, W, 4))
            AIDE_LAUNCH_TIMED(AIDE_KT_UPSAMPLE, kt_bytes, (upsample2x_bwd_tiled_kernel<float, float, true>), grid, dim3(256), 0, stream, dy,
                              (long)dy_bs, dx, (long)dx_bs, C, H, W, tw, tw * th, N * C, pstep, accumulate);
        else
            AIDE_LAUNCH_TIMED(AIDE_KT_UPSAMPLE, kt_bytes, (upsample2x_bwd_tiled_kernel<float, float, false>), grid, dim3(256), 0, stream, dy,
                              (long)dy_bs, dx, (long)dx_bs, C, H, W, tw, tw * th, N * C, pstep, accumulate);
        return aide_launch_status();
    }
    AIDE_LAUNCH_TIMED(AIDE_KT_UPSAMPLE, kt_bytes, upsample2x_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, (long)dy_bs,
                       dx, (long)dx_bs, C, H, W, accumulate, total);
    return aide_launch_status();
}

// the transpose on bf16-stored activation gradients (precision='bf16'); arithmetic in fp32
int aide_upsample2x_bilinear_bwd_mixed(const void* dy, int dy_bf16, int64_t dy_bs, void* dx, int dx_bf16, int64_t dx_bs,
                                       int N, int C, int H, int W, int accumulate, hipStream_t stream) {
    if (dy_bs % 2) return AIDE_ERR_ARG;
    const int tw = (W + UB_TW - 1) / UB_TW, th = (H + UB_TH - 1) / UB_TH;
    const bool fast = ub_fast(dy, dy_bs, W, dy_bf16 ? 2 : 4);
    const int pstep = ub_plane_groups(tw * th, N * C);
    const dim3 grid((unsigned)((long)tw * th * pstep));
    const double kt_bytes = (double)N * C * H * W * (4.0 * (dy_bf16 ? 2 : 4) + (accumulate ? 2.0 : 1.0) * (dx_bf16 ? 2 : 4));
#define AIDE_UB(GT, DT)                                                                                                          \
    do {                                                                                                                         \
        if (fast) AIDE_LAUNCH_TIMED(AIDE_KT_UPSAMPLE, kt_bytes, (upsample2x_bwd_tiled_kernel<GT, DT, true>), grid, dim3(256), 0, stream, \
                                    (const GT*)dy, (long)dy_bs, (DT*)dx, (long)dx_bs, C, H, W, tw, tw * th, N * C, pstep, accumulate); \
        else AIDE_LAUNCH_TIMED(AIDE_KT_UPSAMPLE, kt_bytes, (upsample2x_bwd_tiled_kernel<GT, DT, false>), grid, dim3(256), 0, stream, \
                               (const GT*)dy, (long)dy_bs, (DT*)dx, (long)dx_bs, C, H, W, tw, tw * th, N * C, pstep, accumulate); \
    } while (0)
    if (dy_bf16) { if (dx_bf16) AIDE_UB(bf16_store_t, bf16_store_t); else AIDE_UB(bf16_store_t, float); }
    else { if (dx_bf16) AIDE_UB(float, bf16_store_t); else AIDE_UB(float, float); }
#undef AIDE_UB
    return aide_launch_status();
}

// x, y: [N][C][H][W] (y must not alias x); par: DEVICE array [N][8] doubles {a,b,c,d,e,f,flip,mode},
// mode 0 = affine bilinear, 1 = copy, 2 = rotate 180, 3 = rotate 90, 4 = rotate 270 (3/4: H == W).
int aide_reverse_aug(const float* x, int64_t x_bs, float* y, int64_t y_bs, const double* par, int N, int C,
                     int H, int W, hipStream_t stream) {
    if (!x || !y || !par || x == y) return AIDE_ERR_ARG;
    const long total = (long)N * C * H * W;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, reverse_aug_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, (long)x_bs, y,
                       (long)y_bs, par, C, H, W, total);
    return aide_launch_status();
}

int aide_fill_zero(float* p, int64_t bs, int N, int C, int H, int W, hipStream_t stream) {
    const long chw = (long)C * H * W;
    if (chw % 4 || bs % 4) {
        const long total = (long)N * chw;
        AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, fill_zero_scalar_kernel, dim3(grid_for(total)), dim3(256), 0, stream, p, (long)bs, chw, total);
        return aide_launch_status();
    }
    const long total4 = (long)N * chw / 4;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, fill_zero_kernel, dim3(grid_for(total4)), dim3(256), 0, stream, p, (long)bs, chw / 4,
                       total4);
    return aide_launch_status();
}

}  // extern "C"
