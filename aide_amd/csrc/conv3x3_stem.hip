// Forward of the STEM convolutions (Ci <= 3: the two 3->32 input layers of FuseUNet, 3->64 of UNet) for gfx950.
//
// Replaces (reference): nn.Conv2d(3, C, 3, padding=1) forward at models_twomodalinputs/netblocks.py:24 (modal1_downblock1 /
// modal2_downblock1, fuseunet.py:12,24) and models_singlemodalinput/UNet.py:19 (down_block1).
//
// These layers are bound by WRITING their output (33.5 MB at 256x256 x4, 134 MB bf16 at 512x512 x8) -- 0.45 GFLOP -- but
// on the general kernels they paid a whole channel chunk of mostly zeros per tap (Ci padded to 4 / 16), a filter block
// through LDS and a 256-pixel workgroup each: 47-124 us at 256x256 x4 (7 us of bytes), 64 us alone / 215-270 us inside the
// bf16 step at 512x512 x8, at the head of the step where nothing overlaps them.  Here, as in the stem weight gradient
// (conv3x3_wgrad_stem.hip), the nine taps are folded into the GEMM's K dimension:
//     y[co][p] = b[co] + sum_{k = (ci, tap) < 27} W[co][k] * x[ci][p + tap]          M = co, N = pixels, K = 27 (of 28)
// on v_mfma_f32_32x32x2_f32: 14 MFMAs per 32 co x 32 pixels.  The filters never touch LDS or a pack: lane (co, k parity)
// keeps its 14 values of the MASTER weights w[Co][Ci][3][3] (27 contiguous floats per co) in registers for the whole
// workgroup.  Workgroup = 4 waves, tile = 8 rows x 64 columns of one image (a wave = 2 rows = four 32-pixel blocks), the
// 3 x 10 x 66 halo tile in LDS; the B lane (pixel j, k parity) reads x[ci(k)][row + kh(k)][col + j + kw(k)] through one
// of 14 per-lane address registers + an immediate per pixel block.  8 KB of LDS and ~100 registers: several workgroups
// per CU hide the load -> LDS -> MFMA -> store chain of their neighbours.
// Output fp32 or bf16 storage (precision='bf16': z is stored narrow); ROUND = the bf16 mode's operand contract (image and
// filters rounded to bf16, RNE, when staged: products are then exact in fp32, as on the bf16 matrix pipe); an optional
// per-channel epilogue y = relu?(acc * scale[co] + bias[co]) serves the eval-mode forward (BatchNorm folded into the conv).
#include "common.h"

namespace {

struct StemFwdArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* scale;
    void* y;
    long x_bs, y_bs;
    int N, Ci, H, W, Co, tiles_w, tiles_h, relu;
};

constexpr int SF_R = 8, SF_C = 64;              // tile rows / columns
constexpr int SF_RS = 67;                       // halo row stride: columns -1 .. 64 (+ 1: odd, see the B-operand reads)
constexpr int SF_CS = (SF_R + 2) * SF_RS + 1;   // halo channel stride (odd as well)
constexpr int SF_XN = 3 * (SF_R + 2) * (SF_C + 2);
constexpr int SF_NX = (SF_XN + 255) / 256;

__device__ __forceinline__ float sf_rne_bf16(float v) {     // nearest bf16 (ties to even; finite inputs) as an fp32 value
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return __builtin_bit_cast(float, (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
}

template <int NCOB, bool OUT_BF16, bool ROUND>
__global__ __launch_bounds__(256) void conv3x3_stem_fwd_kernel(const StemFwdArgs a) {
    __shared__ float xs[3 * SF_CS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, j = lane & 31;
    int b = blockIdx.x;
    const int tw = b % a.tiles_w; b /= a.tiles_w;
    const int th = b % a.tiles_h;
    const int n = b / a.tiles_h;
    const int h0 = th * SF_R, w0 = tw * SF_C;
    const int HW = a.H * a.W;

    // ---- halo tile -> LDS (zero outside the image and for channels >= Ci) ----
    const float* xn = a.x + (long)n * a.x_bs;
    float xv[SF_NX];
#pragma unroll
    for (int e = 0; e < SF_NX; ++e) {
        const int q = tid + e * 256;
        const int c = q / ((SF_R + 2) * (SF_C + 2)), rem = q - c * ((SF_R + 2) * (SF_C + 2));
        const int r = rem / (SF_C + 2), col = rem - r * (SF_C + 2);
        const int ih = h0 - 1 + r, iw = w0 - 1 + col;
        const bool ok = q < SF_XN && c < a.Ci && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        xv[e] = ok ? xn[(long)c * HW + (long)ih * a.W + iw] : 0.0f;
    }
    // ---- filters: lane (co = j, k parity = half) keeps W[co][2 s + half], s = 0 .. 13, of every co block ----
    const int K = a.Ci * 9;
    float wa[NCOB][14];
#pragma unroll
    for (int m = 0; m < NCOB; ++m) {
        const float* wr = a.w + (long)(m * 32 + j) * K;
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int k = 2 * s + half;
            float v = k < K ? wr[k] : 0.0f;
            if (ROUND) v = sf_rne_bf16(v);
            wa[m][s] = v;
        }
    }
#pragma unroll
    for (int e = 0; e < SF_NX; ++e) {
        const int q = tid + e * 256;
        if (q < SF_XN) {
            const int c = q / ((SF_R + 2) * (SF_C + 2)), rem = q - c * ((SF_R + 2) * (SF_C + 2));
            const int r = rem / (SF_C + 2), col = rem - r * (SF_C + 2);
            xs[c * SF_CS + r * SF_RS + col] = ROUND ? sf_rne_bf16(xv[e]) : xv[e];
        }
    }
    // B-operand addresses: k = 2 s + half = ci * 9 + kh * 3 + kw -> x[ci][2 wid + kh + (row of the block)][2 j + e + kw]:
    // the two pixel blocks e = 0, 1 of a row INTERLEAVE its 64 columns, so that lane j ends up with the neighbouring pixels
    // 2 j, 2 j + 1 of a channel (one 8-byte fp32 / 4-byte bf16 store per lane, 256 / 128 contiguous bytes per channel and
    // half wave).  The lanes of one half read every second word; the two halves (k, k + 1) are an odd distance apart for
    // every k (kw + 1: 1; next row: RS - 2; next channel: CS - 2 RS - 2), i.e. on the other 32 banks: conflict-free.
    int bo[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int k = min(2 * s + half, 26);               // (k = 27: its filter value is 0; any address will do)
        const int ci = k / 9, t = k - ci * 9, kh = t / 3, kw = t - kh * 3;
        bo[s] = ci * SF_CS + (2 * wid + kh) * SF_RS + 2 * j + kw;
    }
    __syncthreads();

    f32x16 acc[NCOB][4];
#pragma unroll
    for (int m = 0; m < NCOB; ++m)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][p][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 14; ++s) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {                       // pixel block p = (row p >> 1 of the wave's two, column parity p & 1)
            const float bv = xs[bo[s] + (p >> 1) * SF_RS + (p & 1)];
#pragma unroll
            for (int m = 0; m < NCOB; ++m) acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[m][s], bv, acc[m][p], 0, 0, 0);
        }
    }

    // ---- epilogue: D row i = (r & 3) + 8 (r >> 2) + 4 half (output channel), column j (pixel) ----
#pragma unroll
    for (int m = 0; m < NCOB; ++m) {
        float bs[16], sc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            bs[r] = a.bias ? a.bias[co] : 0.0f;
            sc[r] = a.scale ? a.scale[co] : 1.0f;
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int oh = h0 + 2 * wid + rr, ow = w0 + 2 * j;
            if (oh >= a.H) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v0 = __builtin_fmaf(acc[m][2 * rr][r], sc[r], bs[r]), v1 = __builtin_fmaf(acc[m][2 * rr + 1][r], sc[r], bs[r]);
                if (a.relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
                const long o = (long)n * a.y_bs + (long)co * HW + (long)oh * a.W + ow;
                if constexpr (OUT_BF16) *reinterpret_cast<unsigned*>((uint16_t*)a.y + o) = cvt_pk_bf16(v0, v1);
                else *reinterpret_cast<f32x2*>((float*)a.y + o) = f32x2{v0, v1};
            }
        }
    }
}

template <int NCOB>
int launch_stem_fwd(const StemFwdArgs& a, int y_bf16, int round_bf16, hipStream_t stream) {
    const dim3 grid((unsigned)((long)a.tiles_w * a.tiles_h * a.N));
    const double fl = AIDE_CONV_FLOPS(a.N, a.H, a.W, a.Co, a.Ci);
    if (y_bf16) {
        if (round_bf16) AIDE_LAUNCH_TIMED(AIDE_KT_STEM_FWD, fl, (conv3x3_stem_fwd_kernel<NCOB, true, true>), grid, dim3(256), 0, stream, a);
        else AIDE_LAUNCH_TIMED(AIDE_KT_STEM_FWD, fl, (conv3x3_stem_fwd_kernel<NCOB, true, false>), grid, dim3(256), 0, stream, a);
    } else {
        if (round_bf16) AIDE_LAUNCH_TIMED(AIDE_KT_STEM_FWD, fl, (conv3x3_stem_fwd_kernel<NCOB, false, true>), grid, dim3(256), 0, stream, a);
        else AIDE_LAUNCH_TIMED(AIDE_KT_STEM_FWD, fl, (conv3x3_stem_fwd_kernel<NCOB, false, false>), grid, dim3(256), 0, stream, a);
    }
    return aide_launch_status();
}

}  // namespace

extern "C" {

int aide_conv3x3_stem_fwd_supported(int Cin, int H, int W, int Cout) {
    return (Cin >= 1 && Cin <= 3 && (Cout == 32 || Cout == 64) && W % 64 == 0 && H >= 1) ? 1 : 0;
}

//   x : [N][Cin][H][W] fp32 (batch stride x_bs)     w : [Cout][Cin][3][3] fp32 MASTER weights (no pack)
//   y : [N][Cout][H][W] fp32 or bf16 storage (y_bf16; batch stride y_bs in elements, even for bf16)
//   round_bf16 : operands rounded to bf16 when staged (the precision='bf16' contract)
//   epi_scale : NULL, or [Cout]: y = relu?(acc * epi_scale[co] + bias[co]) (eval-mode BatchNorm folded into the conv)
int aide_conv3x3_stem_fwd(const float* x, int64_t x_bs, const float* w, const float* bias, void* y, int y_bf16, int64_t y_bs,
                          int N, int Cin, int H, int W, int Cout, int round_bf16, const float* epi_scale, int epi_relu,
                          hipStream_t stream) {
    if (!x || !w || !y || N <= 0 || !aide_conv3x3_stem_fwd_supported(Cin, H, W, Cout)) return AIDE_ERR_ARG;
    if (y_bf16 && (y_bs % 2)) return AIDE_ERR_ARG;
    StemFwdArgs a;
    a.x = x; a.w = w; a.bias = bias; a.scale = epi_scale; a.y = y; a.x_bs = x_bs; a.y_bs = y_bs;
    a.N = N; a.Ci = Cin; a.H = H; a.W = W; a.Co = Cout;
    a.tiles_w = W / SF_C; a.tiles_h = (H + SF_R - 1) / SF_R;
    a.relu = epi_scale ? epi_relu : 0;
    return Cout == 64 ? launch_stem_fwd<2>(a, y_bf16, round_bf16, stream) : launch_stem_fwd<1>(a, y_bf16, round_bf16, stream);
}

}  // extern "C"
