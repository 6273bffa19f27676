// 3x3 / stride 1 / pad 1 convolution as an fp32 MFMA implicit GEMM for gfx950 (CDNA4).
//
// Replaces (reference): nn.Conv2d(ci, co, 3, padding=1) forward and its autograd dgrad at
//   models_twomodalinputs/netblocks.py:17,24,26 and models_singlemodalinput/UNet.py:12,19,21.
//
// GEMM view (per image):  D[co][pix] = sum_{ci,tap} Wp[ci][tap][co] * X[ci][pix shifted by tap]
//   * output channels sit on the MFMA *row* index i, pixels on the *column* index j, so that
//     one accumulator register of a wave covers 32 consecutive pixels of one NCHW row
//     (128-B contiguous stores, no transposition);
//   * v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 64 cycles/instruction/SIMD; the two K slots
//     of one instruction are an even/odd input-channel pair at the same filter tap, so every
//     LDS fragment address is  lane_base + compile-time immediate  (no VALU in the k-loop);
//   * a workgroup (4 waves) owns TCO output channels x (PT_H x PT_W) pixels of one image and
//     walks the input channels CK at a time: halo tile [CK][PT_H+2][PT_W+2] and filter block
//     [CK][9][TCO] are staged global -> registers -> LDS, the next chunk's global loads are in
//     flight while the MFMAs of the current chunk run;
//   * dgrad is the same kernel on the packed, 180-degree-rotated, channel-transposed filter;
//   * deep layers with few pixels use split-K over input-channel chunks into slabs that a
//     second kernel sums in a fixed order (deterministic).
//
// Weights are consumed in the packed layout Wp[Cin_pad][9][Cout] produced by
// aide_conv3x3_pack_weights (Cin padded with zero rows up to the chunk size).
#include "common.h"

namespace {

struct ConvArgs {
    const float* x;
    const float* wp;
    const float* bias;
    float* y;
    long x_bs, y_bs, split_stride;
    int N, Cin, H, W, Cout, ldw;
    int tiles_w, tiles_h, n_co_tiles, splitk, chunks_total, accumulate;
    const float* scale;   // eval mode (epi_scale of aide_conv3x3_igemm): y = relu?(acc * scale[co] + bias[co]); nullptr: y = acc + bias
    int relu;
};

template <int TCO, int WAVES_M, int WM, int WN, int PT_W, int CK, bool VEC>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(const ConvArgs a) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int RPT = 32 / PT_W;                 // image rows covered by one 32-pixel MFMA tile
    constexpr int PT_H = WAVES_N * WN * RPT;
    // Halo tile rows are laid out [3 pad][col -1][cols 0..PT_W-1][col PT_W][3 pad] so that the
    // interior is 16-byte aligned: it is fetched with 16-byte buffer loads and stored with
    // ds_write_b128; only the two edge columns move as dwords.  (The dword-per-element version of
    // this staging cost 21 % of the kernel: VMEM instruction count, not bytes, was the limiter.)
    constexpr int RS = PT_W + 8;                   // LDS row stride
    constexpr int CS = (PT_H + 2) * RS;            // LDS channel stride
    constexpr int XL = CK * CS;                    // halo tile floats
    constexpr int WL = CK * 9 * TCO;               // filter block floats
    constexpr int UB = CK * (PT_H + 2) * (PT_W / 4);   // interior float4 units per chunk
    constexpr int UC = CK * (PT_H + 2) * 2;            // edge dword units per chunk
    constexpr int NB = (UB + 255) / 256, NC = (UC + 255) / 256;
    constexpr int WV = (WL / 4 + 255) / 256;       // filter float4 per thread
    static_assert(TCO == WAVES_M * WM * 32, "tile mismatch");
    static_assert(CK % 2 == 0 && XL % 4 == 0, "channel pairs / alignment");

    // Two stage buffers: the MFMAs read buffer `cur` while the next channel chunk is fetched to
    // registers (first half of the k-steps) and stored to the other buffer (second half), one staging
    // op per k-step, so no wave ever sits in a load/store burst with the matrix pipe drained.
    constexpr int BUF = XL + WL;
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int wave_m = wid / WAVES_N, wave_n = wid % WAVES_N;

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int co_tile = b % a.n_co_tiles; b /= a.n_co_tiles;
    const int split = b % a.splitk;       b /= a.splitk;
    const int tw = b % a.tiles_w;         b /= a.tiles_w;
    const int th = b % a.tiles_h;
    const int n = b / a.tiles_h;
    const int h0 = th * PT_H, w0 = tw * PT_W, co0 = co_tile * TCO;
    const int HW = a.H * a.W;

    const int cps = (a.chunks_total + a.splitk - 1) / a.splitk;
    const int c_begin = split * cps;
    const int c_end = min(c_begin + cps, a.chunks_total);

    // ---- per-thread staging descriptors (identical for every chunk) ----
    // Descriptor base sits at (row h0-1, col w0-1) so that every unit offset is non-negative; units
    // outside the image carry the BUF_OOB offset and read as 0.0f in hardware (zero padding).
    unsigned offB[NB], ldsB[NB], offC[NC], ldsC[NC], mskB[NB];
#pragma unroll
    for (int e = 0; e < NB; ++e) {
        const int q = tid + e * 256;
        const int c = q / ((PT_H + 2) * (PT_W / 4)), rem = q - c * ((PT_H + 2) * (PT_W / 4));
        const int r = rem / (PT_W / 4), s4 = rem - r * (PT_W / 4);
        const int ih = h0 - 1 + r, iw = w0 + 4 * s4;
        const bool rowok = q < UB && ih >= 0 && ih < a.H;
        offB[e] = (rowok && (VEC ? iw < a.W : true)) ? (unsigned)(c * HW + r * a.W + 1 + 4 * s4) * 4u : BUF_OOB;
        ldsB[e] = q < UB ? (unsigned)(c * CS + r * RS + 4 + 4 * s4) : 0xffffffffu;
        mskB[e] = 0;
        if (!VEC) {           // widths that are not a multiple of 4: per-element column validity
#pragma unroll
            for (int i = 0; i < 4; ++i) mskB[e] |= (rowok && iw + i < a.W) ? (1u << i) : 0u;
        }
    }
#pragma unroll
    for (int e = 0; e < NC; ++e) {
        const int q = tid + e * 256;
        const int c = q / ((PT_H + 2) * 2), rem = q - c * ((PT_H + 2) * 2);
        const int r = rem / 2, side = rem - r * 2;
        const int ih = h0 - 1 + r, iw = side ? w0 + PT_W : w0 - 1;
        const bool ok = q < UC && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        offC[e] = ok ? (unsigned)(c * HW + r * a.W + (side ? PT_W + 1 : 0)) * 4u : BUF_OOB;
        ldsC[e] = q < UC ? (unsigned)(c * CS + r * RS + (side ? 4 + PT_W : 3)) : 0xffffffffu;
    }
    const __amdgpu_buffer_rsrc_t xrs =
        make_rsrc(a.x + (long)n * a.x_bs + (long)h0 * a.W + w0 - (a.W + 1));
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.wp + co0);
    unsigned woff[WV];
#pragma unroll
    for (int v = 0; v < WV; ++v) {
        const int f = tid + v * 256;
        const int row = f / (TCO / 4), c4 = f - row * (TCO / 4);
        // narrow layers (Cout % 32 != 0, the UNet16 ... UNet2 variants): columns past Cout read as zero; a 16-byte piece
        // that straddles Cout picks up the next row's values in accumulator rows the epilogue never stores
        woff[v] = (f < WL / 4 && co0 + c4 * 4 < a.Cout) ? (unsigned)(row * a.ldw + c4 * 4) * 4u : BUF_OOB;
    }
    const bool ragged_c = (a.Cin % CK) != 0;   // only the 3-channel stems

    f32x4 xb[NB];
    float xc[NC];
    f32x4 wr[WV];
    constexpr int NL = NB + NC + WV;           // staging ops per thread per chunk (loads == stores)
    unsigned xs = 0, wsoff = 0;
    bool has_chunk = false;
    int ci0 = 0;
    auto set_chunk = [&](int chunk) {
        has_chunk = chunk < c_end;
        ci0 = chunk * CK;
        xs = (unsigned)ci0 * (unsigned)HW * 4u;
        wsoff = (unsigned)ci0 * 9u * (unsigned)a.ldw * 4u;
    };
    auto fetch = [&](int l) {                  // l is a compile-time op index
        if (l < NB) {
            const int e = l;
            unsigned off = has_chunk ? offB[e] : BUF_OOB;
            if (ragged_c && (ci0 + (tid + e * 256) / ((PT_H + 2) * (PT_W / 4))) >= a.Cin) off = BUF_OOB;
            if (VEC) {
                xb[e] = buf_load_f32x4(xrs, off, xs);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    xb[e][i] = buf_load_f32(xrs, ((mskB[e] >> i) & 1u) && off != BUF_OOB ? off + 4u * i : BUF_OOB, xs);
            }
        } else if (l < NB + NC) {
            const int e = l - NB;
            unsigned off = has_chunk ? offC[e] : BUF_OOB;
            if (ragged_c && (ci0 + (tid + e * 256) / ((PT_H + 2) * 2)) >= a.Cin) off = BUF_OOB;
            xc[e] = buf_load_f32(xrs, off, xs);
        } else {
            const int v = l - NB - NC;
            wr[v] = buf_load_f32x4(wrs, has_chunk ? woff[v] : BUF_OOB, wsoff);
        }
    };
    auto put = [&](int l, float* buf) {
        if (l < NB) {
            if (ldsB[l] != 0xffffffffu) *reinterpret_cast<f32x4*>(buf + ldsB[l]) = xb[l];
        } else if (l < NB + NC) {
            const int e = l - NB;
            if (ldsC[e] != 0xffffffffu) buf[ldsC[e]] = xc[e];
        } else {
            const int v = l - NB - NC, f = tid + v * 256;
            if (f < WL / 4) *reinterpret_cast<f32x4*>(buf + XL + f * 4) = wr[v];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nt][r] = 0.0f;

    // lane offsets of the MFMA operand fragments (pixel column c lives at row index 4 + c)
    const int la_off = XL + half * 9 * TCO + wave_m * WM * 32 + j;
    const int lb_off = half * CS + (wave_n * WN * RPT + j / PT_W) * RS + 3 + (j % PT_W);

    constexpr int STEPS = (CK / 2) * 9;        // (channel pair, tap) k-steps per chunk
    constexpr int HALF = STEPS / 2;
    constexpr int PER = (NL + HALF - 1) / HALF;
    static_assert(PER * HALF >= NL, "staging schedule");

    // prologue: first chunk straight into buffer 0
    set_chunk(c_begin);
#pragma unroll
    for (int l = 0; l < NL; ++l) fetch(l);
#pragma unroll
    for (int l = 0; l < NL; ++l) put(l, lds);
    __syncthreads();

    int cur = 0;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const float* la = lds + cur * BUF + la_off;
        const float* lb = lds + cur * BUF + lb_off;
        float* nxt = lds + (cur ^ 1) * BUF;
        set_chunk(chunk + 1);                  // past the end: every fetch carries BUF_OOB
        // two separately named fragment sets (an array indexed by s&1 can be demoted to memory)
        float afA[WM], bfA[WN], afB[WM], bfB[WN];
        auto frag = [&](int st, float (&af)[WM], float (&bf)[WN]) {
            const int q = st / 9, t = st % 9;            // compile-time after unrolling
            const int kh = t / 3, kw = t % 3;
#pragma unroll
            for (int m = 0; m < WM; ++m) af[m] = la[(2 * q * 9 + t) * TCO + m * 32];
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) bf[nt] = lb[2 * q * CS + (kh + nt * RPT) * RS + kw];
        };
        auto kstep = [&](int st, float (&afc)[WM], float (&bfc)[WN], float (&afn)[WM], float (&bfn)[WN]) {
            if (st + 1 < STEPS) frag(st + 1, afn, bfn);
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int nt = 0; nt < WN; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(afc[m], bfc[nt], acc[m][nt], 0, 0, 0);
            if (st < HALF) {
#pragma unroll
                for (int k = 0; k < PER; ++k)
                    if (st * PER + k < NL) fetch(st * PER + k);
            } else {
#pragma unroll
                for (int k = 0; k < PER; ++k)
                    if ((st - HALF) * PER + k < NL) put((st - HALF) * PER + k, nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        frag(0, afA, bfA);
#pragma unroll
        for (int st = 0; st < STEPS; st += 2) {
            kstep(st, afA, bfA, afB, bfB);
            kstep(st + 1, afB, bfB, afA, bfA);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: D layout row i = (r&3) + 8*(r>>2) + 4*half, col j ----
    float* yn = a.y + (long)split * a.split_stride + (long)n * a.y_bs;
    const bool add_bias = (a.bias != nullptr) && (split == 0);
    // The lane's 16 WM bias values first, as one batch of loads.  (Read next to their use, every one was a dependent
    // global_load -> s_waitcnt vmcnt(0) that also waited for the stores issued before it -- stores count in vmcnt on gfx9 --
    // i.e. one memory round trip per output element.)  The accumulate form batches its 16 old values per tile the same way.
    float bv[WM][16];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + (wave_m * WM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            bv[m][r] = (add_bias && co < a.Cout) ? a.bias[co] : 0.0f;
        }
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
        const int oh = h0 + (wave_n * WN + nt) * RPT + j / PT_W;
        const int ow = w0 + (j % PT_W);
        const bool pok = oh < a.H && ow < a.W;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            float* const pb = yn + (long)(co0 + (wave_m * WM + m) * 32 + 4 * half) * HW + oh * a.W + ow;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[m][nt][r] + bv[m][r];
            if (a.scale) {                     // wave-uniform; never together with accumulate / split-K (the launcher refuses)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (wave_m * WM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float sv = co < a.Cout ? a.scale[co] : 1.0f;
                    const float t = __builtin_fmaf(acc[m][nt][r], sv, bv[m][r]);
                    v[r] = a.relu ? fmaxf(t, 0.0f) : t;
                }
            }
            if (a.accumulate) {
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = (r & 3) + 8 * (r >> 2);
                    old[r] = (pok && co0 + (wave_m * WM + m) * 32 + 4 * half + cl < a.Cout) ? pb[(long)cl * HW] : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += old[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = (r & 3) + 8 * (r >> 2);
                if (pok && co0 + (wave_m * WM + m) * 32 + 4 * half + cl < a.Cout) pb[(long)cl * HW] = v[r];
            }
        }
    }
}

// y[n][c][p] (+)= bias[c] + sum_s slab[s][n][c][p]   (fixed summation order)
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, long split_stride, int splitk,
                                     float* __restrict__ y, long y_bs, int C, int HW,
                                     const float* __restrict__ bias, int accumulate, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long chw = (long)C * HW;
        const long n = i / chw, rem = i - n * chw;
        float v = slabs[i];
        for (int s = 1; s < splitk; ++s) v += slabs[(long)s * split_stride + i];
        if (bias) v += bias[rem / HW];
        float* p = y + n * y_bs + rem;
        if (accumulate) v += *p;
        *p = v;
    }
}

// w[Co][Ci][9] -> wf[Ci_pad][9][Co] (forward) and wd[Co_pad][9][Ci] with the taps reversed
// (dgrad: 180-degree rotation + channel transpose). Padding rows are zero.
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wf,
                                    float* __restrict__ wd, int Co, int Ci, int ci_pad, int co_pad) {
    const long nf = (long)ci_pad * 9 * Co, nd = wd ? (long)co_pad * 9 * Ci : 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd;
         i += (long)gridDim.x * blockDim.x) {
        if (i < nf) {
            const int co = (int)(i % Co);
            const long r = i / Co;
            const int t = (int)(r % 9), ci = (int)(r / 9);
            wf[i] = (ci < Ci) ? w[((long)co * Ci + ci) * 9 + t] : 0.0f;
        } else {
            const long k = i - nf;
            const int ci = (int)(k % Ci);
            const long r = k / Ci;
            const int t = (int)(r % 9), co = (int)(r / 9);
            wd[k] = (co < Co) ? w[((long)co * Ci + ci) * 9 + (8 - t)] : 0.0f;
        }
    }
}

// all filters of a network in ONE launch: a descriptor table (device memory) plus a prefix of
// 256-element blocks per tensor, located by binary search (same scheme as the fused Adam)
struct PackDesc {
    const float* w; float* wf; float* wd;
    int Co, Ci, ci_pad, co_pad;
    long block_start;
};

__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackDesc* __restrict__ descs, int n) {
    const long blk = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block_start <= blk) lo = mid; else hi = mid - 1;
    }
    const PackDesc d = descs[lo];
    const long nf = d.wf ? (long)d.ci_pad * 9 * d.Co : 0, nd = d.wd ? (long)d.co_pad * 9 * d.Ci : 0;
    const long i = (blk - d.block_start) * 256 + threadIdx.x;
    if (i < nf) {
        const int co = (int)(i % d.Co);
        const long r = i / d.Co;
        const int t = (int)(r % 9), ci = (int)(r / 9);
        d.wf[i] = (ci < d.Ci) ? d.w[((long)co * d.Ci + ci) * 9 + t] : 0.0f;
    } else if (i < nf + nd) {
        const long k = i - nf;
        const int ci = (int)(k % d.Ci);
        const long r = k / d.Ci;
        const int t = (int)(r % 9), co = (int)(r / 9);
        d.wd[k] = (co < d.Co) ? d.w[((long)co * d.Ci + ci) * 9 + (8 - t)] : 0.0f;
    }
}

template <int TCO, int WAVES_M, int WM, int WN, int PT_W, int CK>
int launch_cfg(ConvArgs a, hipStream_t stream) {
    constexpr bool CAN_SCALAR = PT_W == 8;      // widths not divisible by 4 are routed to PT_W = 8
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int PT_H = WAVES_N * WN * (32 / PT_W);
    a.tiles_w = (a.W + PT_W - 1) / PT_W;
    a.tiles_h = (a.H + PT_H - 1) / PT_H;
    a.n_co_tiles = (a.Cout + TCO - 1) / TCO;
    a.chunks_total = (a.Cin + CK - 1) / CK;
    const long nb = (long)a.tiles_w * a.tiles_h * a.N * a.n_co_tiles * a.splitk;
    const bool vec = (a.W % 4 == 0) && (a.x_bs % 4 == 0);
    if (vec) {
        AIDE_LAUNCH_TIMED(AIDE_KT_IGEMM, AIDE_CONV_FLOPS(a.N, a.H, a.W, a.Cout, a.Cin),
                          (conv3x3_mfma_kernel<TCO, WAVES_M, WM, WN, PT_W, CK, true>), dim3((unsigned)nb),
                          dim3(256), 0, stream, a);
    } else {
        if constexpr (CAN_SCALAR)
            AIDE_LAUNCH_TIMED(AIDE_KT_IGEMM, AIDE_CONV_FLOPS(a.N, a.H, a.W, a.Cout, a.Cin),
                              (conv3x3_mfma_kernel<TCO, WAVES_M, WM, WN, PT_W, CK, false>), dim3((unsigned)nb),
                              dim3(256), 0, stream, a);
        else
            return AIDE_ERR_ARG;
    }
    return aide_launch_status();
}

template <int PT_W>
int launch_ptw(int variant, int ck, const ConvArgs& a, hipStream_t s) {
    // variant: 0 = 32co x 256px, 1 = 64co x 256px, 2 = 128co x 256px, 3 = 64co x 128px,
    //          4 = 64co x 64px, 5 = 128co x 128px
    if (ck == 4) {
        switch (variant) {
            case 0: return launch_cfg<32, 1, 1, 2, PT_W, 4>(a, s);
            case 1: return launch_cfg<64, 2, 1, 4, PT_W, 4>(a, s);
            default: return AIDE_ERR_ARG;
        }
    }
    switch (variant) {
        case 0: return launch_cfg<32, 1, 1, 2, PT_W, 8>(a, s);
        case 1: return launch_cfg<64, 2, 1, 4, PT_W, 8>(a, s);
        case 2: return launch_cfg<128, 2, 2, 4, PT_W, 8>(a, s);
        case 3: return launch_cfg<64, 2, 1, 2, PT_W, 8>(a, s);
        case 4: return launch_cfg<64, 2, 1, 1, PT_W, 8>(a, s);
        case 5: return launch_cfg<128, 2, 2, 2, PT_W, 8>(a, s);
        default: return AIDE_ERR_ARG;
    }
}

int pick_ptw(int W) {
    if (W % 4 != 0) return 8;                   // only PT_W = 8 carries the per-element (non-16-byte) loader
    int best = 32, waste = ((W + 31) / 32) * 32;
    for (int p : {16, 8}) {
        const int cover = ((W + p - 1) / p) * p;
        if (cover < waste) { waste = cover; best = p; }
    }
    return best;
}

}  // namespace

extern "C" {

// Chunk size (input-channel padding granule) the packed weights must use for a given Cin.
int aide_conv3x3_chunk(int Cin) { return Cin < 8 ? 4 : 8; }

int aide_conv3x3_pack_weights(const float* w, float* wf, float* wd, int Co, int Ci, int ci_pad,
                              int co_pad, hipStream_t stream) {
    const long total = (long)ci_pad * 9 * Co + (wd ? (long)co_pad * 9 * Ci : 0);
    const int blocks = (int)min((total + 255) / 256, (long)4096);
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, pack_weights_kernel, dim3(blocks), dim3(256), 0, stream, w, wf, wd, Co, Ci,
                       ci_pad, co_pad);
    return aide_launch_status();
}

// descs: DEVICE array of n records {w, wf, wd (or 0), Co, Ci, ci_pad, co_pad, block_start} laid out as
// 3 pointers, 4 int32, 1 int64 (40 bytes); total_blocks = sum over tensors of ceil(elems/256).
int aide_conv3x3_pack_weights_multi(const void* descs, int n, int64_t total_blocks, hipStream_t stream) {
    if (!descs || n <= 0 || total_blocks <= 0) return AIDE_ERR_ARG;
    static_assert(sizeof(PackDesc) == 48, "descriptor layout");
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, pack_weights_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream,
                       (const PackDesc*)descs, n);
    return aide_launch_status();
}

// Heuristic plan for one conv problem: returns variant | (splitk << 8).
int aide_conv3x3_plan(int N, int Cin, int H, int W, int Cout) {
    const int ptw = pick_ptw(W);
    auto nblocks = [&](int tco, int px) {
        const int pth = px / ptw;
        return (long)((W + ptw - 1) / ptw) * ((H + pth - 1) / pth) * N * ((Cout + tco - 1) / tco);
    };
    const int ck = aide_conv3x3_chunk(Cin);
    int variant;
    if (Cout % 64 != 0) variant = 0;
    else if (ck == 4) variant = 1;
    else if (Cout % 128 == 0 && nblocks(128, 256) >= 512) variant = 2;
    else if (nblocks(64, 256) >= 512) variant = 1;
    else if (Cout % 128 == 0 && nblocks(128, 128) >= 512) variant = 5;
    else if (nblocks(64, 128) >= 384) variant = 3;
    else variant = 4;
    int splitk = 1;
    if (variant == 4) {
        const long nb = nblocks(64, 64);
        const int chunks = (Cin + ck - 1) / ck;
        while (nb * splitk < 384 && splitk * 2 <= chunks / 8) splitk *= 2;
    }
    return variant | (splitk << 8);
}

size_t aide_conv3x3_ws_bytes(int N, int H, int W, int Cout, int splitk) {
    return splitk > 1 ? (size_t)splitk * N * Cout * H * W * sizeof(float) : 0;
}

// Implicit-GEMM 3x3 convolution on packed weights. Used for forward (wp = forward pack, bias) and
// for dgrad (x = dY, wp = dgrad pack, bias = NULL, Cin/Cout swapped by the caller).
//   x  : [N][Cin][H][W] with batch stride x_bs (floats)    y : [N][Cout][H][W], batch stride y_bs
//   plan: value from aide_conv3x3_plan (or -1 = choose here); ws: split-K slabs (may be NULL if
//   the plan has splitk == 1). accumulate != 0 -> y += result.
int aide_conv3x3_igemm(const float* x, int64_t x_bs, const float* wp, int ldw, const float* bias,
                       float* y, int64_t y_bs, int N, int Cin, int H, int W, int Cout,
                       int accumulate, int plan, float* ws, const float* epi_scale, int epi_relu, hipStream_t stream) {
    const float* aff = epi_scale;                          // eval mode: y = relu?(acc * scale[co] + bias[co]) (nullptr: y = acc + bias)
    const int aff_relu = aff ? epi_relu : 0;
    if (!x || !wp || !y || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return AIDE_ERR_ARG;
    if (plan < 0) plan = aide_conv3x3_plan(N, Cin, H, W, Cout);
    const int variant = plan & 0xff;
    int splitk = plan >> 8;
    if (splitk < 1) splitk = 1;
    if (splitk > 1 && !ws) return AIDE_ERR_ARG;
    const int ck = aide_conv3x3_chunk(Cin);
    const int chunks = (Cin + ck - 1) / ck;
    if (splitk > chunks) splitk = chunks;
    ConvArgs a;
    a.x = x; a.wp = wp; a.x_bs = x_bs; a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
    a.ldw = ldw; a.splitk = splitk;
    a.scale = aff; a.relu = aff_relu;
    if (a.scale && (splitk > 1 || accumulate != 0 || !bias)) return AIDE_ERR_ARG;
    if (splitk > 1) {
        a.y = ws; a.y_bs = (long)Cout * H * W; a.split_stride = (long)N * Cout * H * W;
        a.bias = nullptr; a.accumulate = 0;
    } else {
        a.y = y; a.y_bs = y_bs; a.split_stride = 0; a.bias = bias; a.accumulate = (accumulate == 1);
    }
    int rc;
    switch (pick_ptw(W)) {
        case 32: rc = launch_ptw<32>(variant, ck, a, stream); break;
        case 16: rc = launch_ptw<16>(variant, ck, a, stream); break;
        default: rc = launch_ptw<8>(variant, ck, a, stream); break;
    }
    if (rc != 0) return rc;
    if (splitk > 1 && accumulate != 2) {           // accumulate == 2: the caller consumes the slabs itself
        const long total = (long)N * Cout * H * W;
        const int blocks = (int)min((total + 255) / 256, (long)2048);
        AIDE_LAUNCH_TIMED(AIDE_KT_REDUCE, 0.0, splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, ws,
                           (long)N * Cout * H * W, splitk, y, (long)y_bs, Cout, H * W, bias,
                           accumulate, total);
        rc = aide_launch_status();
    }
    return rc;
}

}  // extern "C"
