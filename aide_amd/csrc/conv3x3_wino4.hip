// 3x3 / pad 1 convolution, Winograd F(4x4, 3x3) on fp32 MFMA (gfx950).  Forward and dgrad of the
// large layers (nn.Conv2d(ci, co, 3, padding=1): models_twomodalinputs/netblocks.py:17,24,26).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A      d: 6x6 input patch, g: 3x3 filter, Y: 4x4 outputs
//
// 36 multiplies per 16 outputs (2.25 per output; F(2x2,3x3) needs 4, the direct form 9).  Per transform
// position p (36 of them) the contraction over input channels is a GEMM
//   M[p][co][tile] += U[p][co][ci] * V[p][tile][ci]        on v_mfma_f32_32x32x2_f32.
// 36 positions x 16 accumulator registers do not fit one wave, so a (32 co x 32 tiles) block is shared by
// TWO waves that own the transform rows 0-2 / 3-5 (18 positions = 288 accumulator registers each):
//   * workgroup = 4 waves = 64 co x 32 tiles (8 x 4 tiles = 32 x 16 pixels) x 36 positions;
//   * stage = 4 input channels: raw halo tile [4][18][41] and V[36][32][4] double buffered in LDS; the
//     filter fragments are loaded global -> registers one stage ahead (each wave is the only consumer of its
//     (positions, co) slice, so LDS staging would only add store traffic: ablation, -12 %); B fragments are
//     ds_read_b64 (channel q pairs with q+2 in the two K slots);
//   * the input transform is split the same way as the positions: a thread produces the 18 values of
//     rows 0-2 (or 3-5) of B^T d B for one (ci, tile) — no exchange, both halves read the 6x6 patch;
//   * the fp32 MFMA shares the vector-ALU pipe (tools/ubench/mfma_shadow.hip), so the loop carries no
//     address arithmetic or selects and the transform's FMAs are issued as one cluster per stage;
//   * output transform: each wave reduces its 18 positions to partial 4x4 outputs, the two partner waves
//     swap halves through LDS (the staging buffers are free by then) and store 16-byte rows.
// Forward error: the transform matrices have entries up to 8 (A) and 5 (B); measured <= 2e-5 relative to
// the output scale on the network's layers (parity tests hold it to 1e-4; the north-star bound is 1e-3).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

template <int V> using ic = std::integral_constant<int, V>;

struct W4Args {
    const float* x;
    const float* u;      // [Cin/4][18][2][Cout][2][2]  (stage, position pair, channel pair, co, position of the pair, channel of the pair)
    const float* bias;
    float* y;
    long x_bs, y_bs, split_stride;
    int N, Cin, H, W, Cout;
    int blocks_w, blocks_h, n_co_tiles, splitk, stages_total, accumulate;
    int gp, gc;          // XCD group: gp pixel tiles x gc co tiles of one split are consecutive logical blocks (see the launcher)
    int pair;            // W == 16: a workgroup tile is 16 rows x (16 columns of image 2n | 16 columns of image 2n + 1)
    float* stats;        // optional [Cout][N * blocks_h * blocks_w][2]: per (channel, workgroup tile) sum / sum of squares of the
                         // pre-bias outputs -- the BatchNorm statistics of the forward pass without a pass over z
    const float* scale;  // AFF variant (eval mode, epi_scale of aide_conv3x3_wino4): y = relu?(acc * scale[co] + bias[co]) -- the
    int relu;            // BatchNorm of running statistics folded into the epilogue, no separate pass over the conv output
    const float* in_tab; // BNIN variant (in_bn_tab of aide_conv3x3_wino4): [N / in_ng][Cin][2] = (scale, shift) per image group and
    int in_ng;           // input channel -- the loader stages relu(x * scale + shift): the BatchNorm + ReLU of the producing
                         // layer applied on the way in, its normalised output never materialised
};

constexpr int F4_BNIN_MAXC = 1024;         // BNIN variant: input channels whose (scale, shift) table + zero copy fit LDS beside the sets
constexpr int F4_NAGPR = 16;               // accumulators (of 18) kept in the AGPR file; the rest are pinned to VGPRs
constexpr int F4_RRS = 41;                 // raw row: [3 pad][-1][0..31][32][4 pad]; odd: conflict-free raw stores
constexpr int F4_RCS = 775;                // raw channel stride: (41, 775) makes patch reads AND raw stores conflict-free
constexpr int F4_RAW = 4 * F4_RCS;         // 3100 floats
constexpr int F4_V = 36 * 32 * 4;          // V[p][2 channel pairs][32 tiles][2]
constexpr int F4_SET = F4_RAW + F4_V;      // 7708 floats = 30832 B; two sets = 60.2 KB (filters never touch LDS)
constexpr int F4_LDS = 2 * F4_SET > 4 * 128 * 64 ? 2 * F4_SET : 4 * 128 * 64;   // epilogue swap needs 32768

// Position pairs: pair q = 3 i + t of transform row i holds columns (1,2), (3,4), (0,5) for t = 0, 1, 2 (the pairs the packed
// input transform produces).  Accumulator slot 2 q + b <-> position F4_P(q, b); slot of column c in row i: F4_SLOT.
__host__ __device__ constexpr int f4_pos(int q, int b) {
    return 6 * (q / 3) + ((q % 3) == 0 ? 1 + b : (q % 3) == 1 ? 3 + b : 5 * b);
}
__host__ __device__ constexpr int f4_slot(int c) { return c == 0 ? 4 : c == 5 ? 5 : c - 1; }   // within a row of 6

// 1-D input transform B^T (F(4,3), points 0, +-1, +-2, inf), all six outputs
__device__ __forceinline__ void bt6(float d0, float d1, float d2, float d3, float d4, float d5, float* o, int st) {
    const float a = __builtin_fmaf(-4.f, d2, d4), b = __builtin_fmaf(4.f, d1, -d3);
    const float c = d4 - d2, e = d3 - d1;
    o[0] = __builtin_fmaf(-5.f, d2, __builtin_fmaf(4.f, d0, d4));
    o[st] = a - b;
    o[2 * st] = a + b;
    o[3 * st] = __builtin_fmaf(2.f, e, c);
    o[4 * st] = __builtin_fmaf(-2.f, e, c);
    o[5 * st] = __builtin_fmaf(-5.f, d3, __builtin_fmaf(4.f, d1, d5));
}

// sums over lanes 0-31 and 32-63 of a wave, valid in lanes 31 and 63.  Vector-ALU only (DPP): no LDS round trips.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float half_total_dpp(float v) {
    v = dpp_add<0x111, 0xf>(v);       // row_shr:1  (lanes shifted in from outside the row of 16 read 0)
    v = dpp_add<0x112, 0xf>(v);       // row_shr:2
    v = dpp_add<0x114, 0xf>(v);       // row_shr:4
    v = dpp_add<0x118, 0xf>(v);       // row_shr:8: lane 15 of every row holds the row total
    v = dpp_add<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    return v;
}

// MODE 1: 16-pixel-wide images, two per workgroup tile (a compile-time variant: the descriptors of the common case keep their
// register allocation -- as a runtime flag the extra live values put a scratch reload into the main loop).
// MODE 2: the workgroup's 32 tile slots as a 5 x 5 canvas of 4x4 tiles = 20 x 20 pixels (25 slots used) instead of 4 x 8 =
// 16 x 32: the 40 x 40 and 20 x 20 planes of the 320 x 320 workload fill 78 % of their tiles instead of 52 % / 39 %.  Only
// the staging descriptors, the patch origin and the output address know the canvas; the main loop is the same code.
// BNIN: the input is the RAW conv output z of the layer before; the staging applies that layer's BatchNorm + ReLU
// (per-channel scale / shift of the image's group, a table the caller derived from the statistics) before the raw tile goes to
// LDS -- the forward-only augmentation passes of the co-teaching step save nothing for a backward pass, so the normalised
// tensor need not exist (SURVEY 8b in_prologue{bn_relu}).  Zero padding must stay zero (BN(0) != 0): a unit that lies
// outside the image reads its (scale, shift) from a ZERO copy of the table, so 0 * 0 + 0 = 0 without a select.  A separate
// instantiation: the training kernels keep their code and register allocation.
template <int MODE, bool AFF = false, bool BNIN = false>
__global__ __launch_bounds__(256, 1) void conv3x3_wino4_kernel(const W4Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform (scalar) roles
    const int half = lane >> 5, j = lane & 31;
    const int ph = wid & 1, cb = wid >> 1;                 // position half (transform rows 3ph..3ph+2), co block

    // xcd_remap gives every XCD a contiguous range of logical blocks.  Their order: [pixel-tile group][split][co group]
    // [co in group][pixel tile in group] -- gp pixel tiles that stream the SAME filter slice (co tile, channel range) and gc
    // co tiles that read the SAME input tile sit in one XCD and share them in its L2.  (co-tile-fastest order made every
    // workgroup of an XCD stream a different filter slice: 512->256 @64x64 fetched its 19 MB of filters 32 times.)
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int p_in = b % a.gp;            b /= a.gp;
    const int co_in = b % a.gc;           b /= a.gc;
    const int ncg = a.n_co_tiles / a.gc;
    const int cog = b % ncg;              b /= ncg;
    const int split = b % a.splitk;       b /= a.splitk;
    const int co_tile = cog * a.gc + co_in;
    int pt = b * a.gp + p_in;                              // pixel tile = (n, th, tw)
    const int tw = pt % a.blocks_w;       pt /= a.blocks_w;
    const int th = pt % a.blocks_h;
    const int n = pt / a.blocks_h;
    constexpr int PAIR = MODE == 1;
    constexpr bool canv = MODE == 2;
    // raw rows per channel / 16-byte units per row / row stride in LDS.  Canvas: 22 rows of [3 pad][-1][0..19][20]; the
    // stride 29 = 5 (mod 8) keeps the patch reads conflict-free (bank = 7 ci + 4 tile for the 8 tiles x 4 channels of a
    // 32-lane group, as (41, 775) does for the 4 x 8 form)
    constexpr int NR = canv ? 22 : 18, NI = canv ? 5 : 8, RRS = canv ? 29 : F4_RRS;
    constexpr int TPH = canv ? 20 : 16, TPW = canv ? 20 : 32;
    static_assert(NR * RRS <= F4_RCS, "raw channel plane");
    const int h0 = th * TPH, w0 = tw * TPW, co0 = co_tile * 64;
    const int HW = a.H * a.W;
    // 16-pixel-wide images (the 16 x 16 bottleneck level): the 32-column tile holds the rows of TWO images side by side, each
    // with its own zero halo columns -- raw row [3 pad][-1][A 0..15][16][-1][B 0..15][16]: image B sits two words further right.
    // Only these stage-invariant descriptors and the output addresses know about it; the main loop is the same.
    constexpr int pair = PAIR;

    const int sps = a.stages_total / a.splitk;             // even, and splitk divides stages_total (host)
    const int s_begin = split * sps;
    const int s_end = min(s_begin + sps, a.stages_total);

    // ---- staging descriptors (stage-invariant) ----
    // raw interior: 4 ci x 18 rows x 8 float4 = 576 units (3 rounds, spare lanes repeat an earlier unit; canvas: 4 x 22 x 5 =
    // 440); raw edges: 4 x 18 x 2 dwords = 144 units (1 round; canvas 176); U: 2304 float4 (9 rounds)
    constexpr int NUI = 4 * NR * NI, NUE = 4 * NR * 2;
    unsigned offB[3], ldsB[3], offC, ldsC;
    int tabB[3], tabC;                     // BNIN: float index of the unit's (scale, shift) entry for stage 0 (valid: table, else zero copy)
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        int q = tid + e * 256;
        while (q >= NUI) q -= 256;
        const int c = q / (NR * NI), rem = q - c * (NR * NI), r = rem / NI, s4 = rem - r * NI;
        const int ih = h0 - 1 + r, iw = w0 + 4 * s4;
        const int imgB = pair && s4 >= 4;
        const bool ok = ih >= 0 && ih < a.H && (pair || iw < a.W);
        offB[e] = ok ? (unsigned)((long)imgB * a.x_bs + c * HW + r * a.W + 1 + 4 * (pair ? (s4 & 3) : s4)) * 4u : BUF_OOB;
        ldsB[e] = (unsigned)(c * F4_RCS + r * RRS + 4 + 4 * s4 + 2 * imgB);
        tabB[e] = F4_LDS + (ok ? 0 : 2 * a.Cin) + 2 * c;
    }
    {
        int q = tid;
        if (q >= NUE) q -= NUE;
        const int c = q / (NR * 2), rem = q - c * (NR * 2), r = rem >> 1, side = rem & 1;
        const int ih = h0 - 1 + r, iw = side ? w0 + TPW : w0 - 1;
        const bool ok = !pair && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        tabC = F4_LDS + (ok ? 0 : 2 * a.Cin) + 2 * c;
        offC = ok ? (unsigned)(c * HW + r * a.W + (side ? TPW + 1 : 0)) * 4u : BUF_OOB;
        // (pair: every halo column is an image border; the edge units keep the two seam columns zero, the outer two are
        // zeroed once in the prologue)
        ldsC = (unsigned)(c * F4_RCS + r * RRS + (pair ? (side ? 21 : 20) : (side ? 4 + TPW : 3)));
    }
    const __amdgpu_buffer_rsrc_t xrs =
        make_rsrc(a.x + (long)(pair ? 2 * n : n) * a.x_bs + (long)h0 * a.W + w0 - (a.W + 1));
    // Filter fragments go global -> registers, never through LDS: wave (cb, ph) is the only consumer of
    // U[p in its half][co in its block], and the packed layout [stage][p][pair][Co][2] makes one position a
    // contiguous 512-byte dwordx2 load in exactly the MFMA A-operand lane order (lane = pair * 32 + co).
    // (Cout % 64 == 32: the upper co block of the last workgroup reads past its rows - into the next run, or past the
    // tensor, where the exact-size descriptor returns zeros; those accumulator rows are never stored)
    // packed layout [stage][position pair q (18)][channel pair][Co][p & 1][2]: one dwordx4 per lane carries the A
    // operands of BOTH K slots of two adjacent positions -- 9 filter loads per stage instead of 18 (vector-memory issue
    // slots are the scarce resource beside the MFMAs: the ablation without any fetch ran 12-14 % faster)
    const long ubase = ((long)9 * ph * 2 * a.Cout + co0 + cb * 32) * 4;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u + ubase), 0, (int)(((long)a.Cin * 36 * a.Cout - ubase) * 4), 0x00020000);
    const unsigned uoff = (unsigned)(half * a.Cout + j) * 16u;
    const unsigned upos = (unsigned)a.Cout * 32u;          // bytes per position pair

    f32x4 rb[3];
    float rc;
    f32x4 ua[2][9];                                        // A fragments of the current / next stage: [q] = {p0k0, p0k1, p1k0, p1k1}
    auto fetch_u = [&](int q, int stage, f32x4* dst) {
        const unsigned us = ((unsigned)min(stage, s_end - 1) * 18u + (unsigned)q) * upos;
        dst[q] = buf_load_f32x4(urs, uoff, us);
    };
    auto fetch = [&](int l, int stage) {                   // 4 raw loads
        const unsigned xs = (unsigned)(min(stage, s_end - 1) * 4) * (unsigned)HW * 4u;
        if (l < 3) rb[l] = buf_load_f32x4(xrs, offB[l], xs);
        else rc = buf_load_f32(xrs, offC, xs);
    };
    auto put_raw = [&](int w, float* raw) {                // 13 dword stores (odd channel stride)
        if (w < 12) raw[ldsB[w >> 2] + (w & 3)] = rb[w >> 2][w & 3];
        else raw[ldsC] = rc;
    };
    // BNIN: (scale, shift) of the four units' channels for one stage -> registers (bn_tab), then relu(x * scale + shift) on
    // the 13 fetched values (bn_in) right before they go to LDS: 6 packed FMAs + 1 FMA + 13 max per stage
    f32x2 tv[4];
    auto bn_tab = [&](int stage) {
        const int so = min(stage, s_end - 1) * 8;
#pragma unroll
        for (int e = 0; e < 3; ++e) tv[e] = *reinterpret_cast<const f32x2*>(lds + tabB[e] + so);
        tv[3] = *reinterpret_cast<const f32x2*>(lds + tabC + so);
    };
    auto bn_in = [&]() {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const f32x2 sc = f32x2{tv[e].x, tv[e].x}, sh = f32x2{tv[e].y, tv[e].y};
            const f32x2 lo = f32x2{rb[e].x, rb[e].y} * sc + sh, hi = f32x2{rb[e].z, rb[e].w} * sc + sh;
            rb[e] = f32x4{fmaxf(lo.x, 0.f), fmaxf(lo.y, 0.f), fmaxf(hi.x, 0.f), fmaxf(hi.y, 0.f)};
        }
        rc = fmaxf(__builtin_fmaf(rc, tv[3].x, tv[3].y), 0.f);
    };
    if constexpr (BNIN) {
        float* tab = lds + F4_LDS;
        const float* src = a.in_tab + (long)(n / a.in_ng) * 2 * a.Cin;
        for (int i = tid; i < 2 * a.Cin; i += 256) { tab[i] = src[i]; tab[2 * a.Cin + i] = 0.f; }
        __syncthreads();
    }

    // ---- input transform: thread = (half hs, ci, tile) ----
    const int item = tid & 127, hs = wid >> 1;             // hs is wave-uniform
    // tile slot -> (tile row, tile column) of the workgroup tile; the 7 spare slots of the canvas repeat its last tile
    const int tslot = canv ? min(item >> 2, 24) : (item >> 2);
    const int trow = canv ? tslot / 5 : tslot >> 3, tcol = canv ? tslot - 5 * trow : tslot & 7;
    const int xr_off = (item & 3) * F4_RCS + trow * 4 * RRS + 3 + 4 * tcol + ((pair && tcol >= 4) ? 2 : 0);
    // patch rows as pairs of columns: the column pass is element-wise over columns -> v_pk_fma_f32 / v_pk_add_f32
    f32x2 tp[6][3];
    f32x2 tq[9];
    auto xf_read = [&](int i, int xo) {                    // xo = patch origin of this thread in a set
        const float v = lds[xo + (i / 6) * RRS + (i % 6)];
        if ((i % 6) & 1) tp[i / 6][(i % 6) >> 1].y = v; else tp[i / 6][(i % 6) >> 1].x = v;
    };
    auto xf_half = [&](auto HS) {
        // column pass: rows 3hs..3hs+2 of B^T d for each pair of columns, then the full B^T along the rows
        constexpr int khs = decltype(HS)::value;
        f32x2 T[3][3];
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {
            const f32x2 d0 = tp[0][cp], d1 = tp[1][cp], d2 = tp[2][cp], d3 = tp[3][cp], d4 = tp[4][cp], d5 = tp[5][cp];
            if (khs == 0) {
                const f32x2 aa = d4 - 4.f * d2, bb = 4.f * d1 - d3;
                T[0][cp] = (4.f * d0 + d4) - 5.f * d2;
                T[1][cp] = bb * f32x2{-1.f, -1.f} + aa;        // (packed subtracts are scalarised by hipcc: v_sub x2 + v_mov x2)
                T[2][cp] = aa + bb;
            } else {
                const f32x2 cc = d2 * f32x2{-1.f, -1.f} + d4, ee = d1 * f32x2{-1.f, -1.f} + d3;
                T[0][cp] = cc + 2.f * ee;
                T[1][cp] = cc - 2.f * ee;
                T[2][cp] = (4.f * d1 + d5) - 5.f * d3;
            }
        }
        // row pass, six packed ops per row: (a, c) (b, e) (o1, o2) (o3, o4) (4 t0 + t4, 4 t1 + t5) (o0, o5)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x2 t01 = T[i][0], t23 = T[i][1], t45 = T[i][2];
            const f32x2 ac = f32x2{t23.x, t23.x} * f32x2{-4.f, -1.f} + f32x2{t45.x, t45.x};
            // (nb, e) = (-(4 t1 - t3), t3 - t1): keeping the first component NEGATED makes every line one packed FMA with
            // splat operands and constant multipliers -- a `{-x, x}` pair costs a v_xor + v_mov on the MFMA's own pipe
            const f32x2 be = f32x2{t01.y, t01.y} * f32x2{-4.f, -1.f} + f32x2{t23.y, t23.y};
            const f32x2 o12 = f32x2{be.x, be.x} * f32x2{1.f, -1.f} + f32x2{ac.x, ac.x};
            const f32x2 o34 = f32x2{be.y, be.y} * f32x2{2.f, -2.f} + f32x2{ac.y, ac.y};
            const f32x2 o05 = t23 * f32x2{-5.f, -5.f} + (t01 * f32x2{4.f, 4.f} + t45);
            // position pairs as they fall out of the packed arithmetic: (1,2) (3,4) (0,5) of transform row i -- the V / U
            // layouts and the accumulator slots use THIS pairing (F4_P0 / F4_P1), so no register shuffling is needed
            tq[3 * i] = o12; tq[3 * i + 1] = o34; tq[3 * i + 2] = o05;
        }
    };
    // The transform half is wave-uniform.  Everything from here on is instantiated once per half and
    // selected by ONE branch: a branch inside the stage makes hipcc drain vmcnt to 0 at the join, i.e.
    // wait for the filter fragments it has just requested.
    auto run = [&](auto HS) {
    auto xf_math = [&]() { xf_half(HS); };
    // V[q = p / 2][channel pair][tile][e = channel of the pair][p & 1]: a thread stores the two positions of a pair as ONE
    // 8-byte store (9 per stage instead of 18) and a lane's 16-byte fragment read carries both K slots of both positions
    // (18 ds_read_b128 per stage instead of 36 ds_read_b64).  Pair 1's run is XOR-swizzled by 16 dwords: the 16-lane
    // store groups then cover all 32 banks; the reads apply the same XOR (a permutation of 16-byte slots: conflict-free).
    const int vitem = ((item >> 1) & 1) * 128 + ((((item >> 2) * 4 + (item & 1) * 2)) ^ (((item >> 1) & 1) * 16));
    auto xf_store = [&](int m, float* vbuf) {
        *reinterpret_cast<f32x2*>(vbuf + (9 * hs + m) * 256 + vitem) = tq[m];
    };

    f32x16 acc[18];
#pragma unroll
    for (int p = 0; p < 18; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
    f32x4 fb[3];                                           // B fragments of a position pair, requested 8 slots ahead

    float* const set0 = lds;
    float* const set1 = lds + F4_SET;
    // per-set patch origins as live registers: with one base + a 30 KB constant hipcc pairs the reads into
    // ds_read2_b32 (8-bit offsets) and re-derives the base with a v_add in every MFMA slot
    int xr0 = xr_off, xr1 = F4_SET + xr_off;
    asm volatile("" : "+v"(xr0), "+v"(xr1));
    // ---- prologue: raw[s0] -> set0, U[s0] -> registers, transform -> set0.V; raw[s0+1] -> set1.raw ----
#pragma unroll
    for (int l = 0; l < 4; ++l) fetch(l, s_begin);
#pragma unroll
    for (int q = 0; q < 9; ++q) fetch_u(q, s_begin, ua[0]);
    if (pair) {                                            // the outer halo columns (-1 of A, 16 of B) of both sets, once
        for (int q = tid; q < 2 * 4 * 18 * 2; q += 256) {
            const int set = q / 144, rem = q - set * 144, c = rem / 36, rr = (rem % 36) >> 1, side = rem & 1;
            lds[set * F4_SET + c * F4_RCS + rr * F4_RRS + (side ? 38 : 3)] = 0.f;
        }
    }
    if constexpr (BNIN) { bn_tab(s_begin); bn_in(); }
#pragma unroll
    for (int w = 0; w < 13; ++w) put_raw(w, set0);
#pragma unroll
    for (int l = 0; l < 4; ++l) fetch(l, s_begin + 1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 36; ++i) xf_read(i, xr0);
    xf_math();
#pragma unroll
    for (int m = 0; m < 9; ++m) xf_store(m, set0 + F4_RAW);
    if constexpr (BNIN) { bn_tab(s_begin + 1); bn_in(); }
#pragma unroll
    for (int w = 0; w < 13; ++w) put_raw(w, set1);
    __syncthreads();

    // ---- main loop.  While the MFMAs consume V of set `sc` and the A fragments ua[CUR] (stage s):
    //   U[s+1] is fetched into ua[1 - CUR]; raw[s+1] (already in sn.raw) is transformed into sn.V;
    //   raw[s+2] is fetched and stored into sc.raw (consumed by the previous stage's transform).
    auto stage = [&](int s, float* sc, float* sn, auto CUR) {
        constexpr int kcur = decltype(CUR)::value;
        // V[p][channel pair][tile][2]: a wave's ds_read_b64 covers 512 contiguous bytes in lane order (a
        // [tile][4] row layout measured 2-way bank conflicts on every fragment read, with or without swizzle)
        const float* lb = sc + F4_RAW + (9 * ph) * 256 + half * 128 + ((j * 4) ^ (half * 16));
        auto frag = [&](int q, int slot3) { fb[slot3] = *reinterpret_cast<const f32x4*>(lb + q * 256); };
        frag(0, 0); frag(1, 1);
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            // slot order inside a position pair g: (2g,k0) (2g+1,k0) (2g,k1) (2g+1,k1)
            const int g = st >> 2, w = st & 3, pi = 2 * g + (w & 1), k = w >> 1;
            const int fs = g % 3;
            if (w == 0 && g + 2 < 9) frag(g + 2, (g + 2) % 3);   // fragments of a pair, two pairs (8 slots) ahead
            // 18 x 16 accumulator registers exceed the 256 AGPRs: positions 16 and 17 are pinned to VGPRs,
            // and the register classes are spelled out (hipcc otherwise shuffles whole accumulators
            // between the two files every stage)
            if (pi < F4_NAGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[pi]) : "v"(ua[kcur][g][(w & 1) * 2 + k]), "v"(fb[fs][k * 2 + (w & 1)]));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[pi]) : "v"(ua[kcur][g][(w & 1) * 2 + k]), "v"(fb[fs][k * 2 + (w & 1)]));
            // staging schedule: only slot 20 carries vector-ALU work
            //   0..3 raw[s+2] global fetches, 4..21 U[s+1] fragment fetches;  0..17 patch reads (2 per slot)
            //   20 transform;  21..29 V stores (2 per slot);  31..35 raw stores (3 per slot)
            if (st < 4) fetch(st, s + 2);
            else if (st < 22 && ((st - 4) & 1) == 0) fetch_u((st - 4) >> 1, s + 1, ua[1 - kcur]);
            constexpr int XM = 20, XS = 21, PR = 31;
            if (st < 18) { xf_read(2 * st, kcur ? xr0 : xr1); xf_read(2 * st + 1, kcur ? xr0 : xr1); }
            if (st == XM) xf_math();
            if (st >= XS && st < XS + 9) xf_store(st - XS, sn + F4_RAW);
            if constexpr (BNIN) {                     // the stage's second (and last) slot with vector-ALU work
                if (st == 24) bn_tab(s + 2);
                if (st == 30) bn_in();
            }
            if (st >= PR && st < PR + 5) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (3 * (st - PR) + q < 13) put_raw(3 * (st - PR) + q, sc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    // two stages per iteration, unconditionally (the host makes the stage count of a split even): with a
    // conditional second stage hipcc reconciles the accumulator registers of the two paths by copying them
    for (int s = s_begin; s < s_end; s += 2) {
        stage(s, set0, set1, ic<0>{});
        stage(s + 1, set1, set0, ic<1>{});
    }

    // ---- output transform.  Partial over this wave's rows i = 3ph..3ph+2:
    //   T[i][b] = sum_c M[i][c] A[c][b];   Yp[a][b] = sum_i A^T[a][i] T[i][b]
    // The wave with ph = 0 finishes accumulator rows r < 8, its partner r >= 8; the other half of the
    // partials travels through LDS ([wave][128][64 lanes]).
    float* xbuf = lds;
    float* yn = a.y + (long)split * a.split_stride + (long)(pair ? 2 * n + ((j & 7) >> 2) : n) * a.y_bs;
    const bool add_bias = (a.bias != nullptr) && (split == 0);
    const int oh = h0 + 4 * (canv ? j / 5 : j >> 3), ow = pair ? 4 * (j & 3) : w0 + 4 * (canv ? j % 5 : j & 7);
    const bool pok = (!canv || j < 25) && oh < a.H && ow < a.W;   // H, W multiples of 4: a tile is in or out
    auto epilogue = [&](auto PH) {
        constexpr int kph = decltype(PH)::value;
        // two accumulator rows (r, r + 1: adjacent registers) per pass as f32x2: the output transform is ~800 scalar VALU
        // instructions per wave otherwise, on the pipe the MFMAs share -- packed, half of that; the partner exchange moves
        // 8-byte pairs
        auto partial = [&](int r, f32x2* yp) {             // r (even) is a compile-time constant after unrolling
            f32x2 T[3][4];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                auto m = [&](int c) { return f32x2{acc[6 * i + f4_slot(c)][r], acc[6 * i + f4_slot(c)][r + 1]}; };
                const f32x2 m0 = m(0), m1 = m(1), m2 = m(2), m3 = m(3), m4 = m(4), m5 = m(5);
                const f32x2 s12 = m1 + m2, d12 = m2 * f32x2{-1.f, -1.f} + m1, s34 = m3 + m4, d34 = m4 * f32x2{-1.f, -1.f} + m3;
                T[i][0] = m0 + s12 + s34;
                T[i][1] = d34 * f32x2{2.f, 2.f} + d12;
                T[i][2] = s34 * f32x2{4.f, 4.f} + s12;
                T[i][3] = d34 * f32x2{8.f, 8.f} + d12 + m5;
            }
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) {
                if (kph == 0) {                             // A^T columns 0,1,2: (1,0,0,0) (1,1,1,1) (1,-1,1,-1)
                    const f32x2 sm = T[1][bq] + T[2][bq], df = T[2][bq] * f32x2{-1.f, -1.f} + T[1][bq];
                    yp[bq] = T[0][bq] + sm; yp[4 + bq] = df; yp[8 + bq] = sm; yp[12 + bq] = df;
                } else {                                    // columns 3,4,5: (1,2,4,8) (1,-2,4,-8) (0,0,0,1)
                    const f32x2 sm = T[0][bq] + T[1][bq], df = T[1][bq] * f32x2{-1.f, -1.f} + T[0][bq];
                    yp[bq] = sm; yp[4 + bq] = df * f32x2{2.f, 2.f}; yp[8 + bq] = sm * f32x2{4.f, 4.f};
                    yp[12 + bq] = df * f32x2{8.f, 8.f} + T[2][bq];
                }
            }
        };
        f32x2* const xbuf2 = reinterpret_cast<f32x2*>(xbuf);       // [wave][row pair 4][output 16][lane] pairs
        // (the main loop ended with a barrier: the staging buffers are free)
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            f32x2 yp[16];
            partial(2 * rp + 8 * (1 - kph), yp);           // the partner's rows
#pragma unroll
            for (int o = 0; o < 16; ++o) xbuf2[((wid * 64) + rp * 16 + o) * 64 + lane] = yp[o];
        }
        __syncthreads();
        // The eight bias values of this lane's output rows are fetched up front.  (Read next to their use they were eight
        // dependent global_load -> s_waitcnt vmcnt(0) round trips per wave, each of which also waited for the output stores
        // issued before it to be acknowledged -- stores count in vmcnt on gfx9: most of the "3.5 us of output stores" of
        // HISTORY §4.8.)
        float bvs[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = q + 8 * kph;
            const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            bvs[q] = (add_bias && co < a.Cout) ? a.bias[co] : 0.f;
        }
        float svs[8];
        if constexpr (AFF) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = q + 8 * kph;
                const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                svs[q] = co < a.Cout ? a.scale[co] : 1.f;
            }
        }
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            f32x2 yp[16];
            partial(2 * rp + 8 * kph, yp);
#pragma unroll
            for (int o = 0; o < 16; ++o) yp[o] += xbuf2[(((wid ^ 1) * 64) + rp * 16 + o) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * rp + e + 8 * kph;
                const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (a.stats) {                                   // wave-uniform
                    float s1 = 0.f, s2 = 0.f;
                    if (pok) {
#pragma unroll
                        for (int o = 0; o < 16; ++o) { s1 += yp[o][e]; s2 = __builtin_fmaf(yp[o][e], yp[o][e], s2); }
                    }
                    // the 32 tiles of this half, total in its last lane (j == 31): inclusive scan inside each row of 16 lanes
                    // (row_shr 1, 2, 4, 8), then the even rows' totals into the odd rows (row_bcast:15) -- five dependent vector
                    // adds per sum.  (As `__shfl_xor` butterflies these were ten ds_bpermute round trips per (row pair, row),
                    // each behind its own s_waitcnt lgkmcnt(0): 80 LDS round trips per workgroup epilogue.)
                    s1 = half_total_dpp(s1);
                    s2 = half_total_dpp(s2);
                    if (j == 31 && co < a.Cout) {
                        const int nparts = a.N * a.blocks_h * a.blocks_w, blk = (n * a.blocks_h + th) * a.blocks_w + tw;
                        *reinterpret_cast<f32x2*>(a.stats + ((long)co * nparts + blk) * 2) = f32x2{s1, s2};
                    }
                }
                if (pok && co < a.Cout) {
                    const float bv = bvs[2 * rp + e];
                    f32x4* const p0 = reinterpret_cast<f32x4*>(yn + (long)co * HW + (long)oh * a.W + ow);
                    f32x4 o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        o[i] = f32x4{yp[4 * i][e] + bv, yp[4 * i + 1][e] + bv, yp[4 * i + 2][e] + bv, yp[4 * i + 3][e] + bv};
                    if constexpr (AFF) {                         // (never with accumulate / split-K: the launcher refuses)
                        const float sv = svs[2 * rp + e];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float v = __builtin_fmaf(yp[4 * i + k][e], sv, bv);
                                o[i][k] = a.relu ? fmaxf(v, 0.f) : v;
                            }
                        }
                    }
                    if (a.accumulate) {                          // (the four old rows as one batch of loads, one wait)
                        f32x4 old[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) old[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p0) + (long)i * a.W);
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] += old[i];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p0) + (long)i * a.W) = o[i];
                }
            }
        }
    };
    if (ph == 0) epilogue(ic<0>{}); else epilogue(ic<1>{});
    };   // run
    if (hs == 0) run(ic<0>{}); else run(ic<1>{});
}

// y[n][c][p] (+)= bias[c] + sum_s slab[s][n][c][p], 16 bytes per thread, fixed summation order.  scale != nullptr (eval mode,
// epi_scale): y = relu?(sum * scale[c] + bias[c])
__global__ __launch_bounds__(256) void w4_splitk_reduce_kernel(const float* __restrict__ slabs, long split_stride,
                                                               int splitk, float* __restrict__ y, long y_bs, int C,
                                                               int HW, const float* __restrict__ bias, int accumulate,
                                                               long total4, const float* __restrict__ scale, int relu) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long e = i * 4, chw = (long)C * HW;
        const long n = e / chw, rem = e - n * chw;
        f32x4 v = *reinterpret_cast<const f32x4*>(slabs + e);
        for (int s = 1; s < splitk; ++s) v += *reinterpret_cast<const f32x4*>(slabs + (long)s * split_stride + e);
        if (scale) {
            const float sv = scale[rem / HW], bv = bias[rem / HW];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = __builtin_fmaf(v[k], sv, bv); v[k] = relu ? fmaxf(t, 0.f) : t; }
        } else if (bias) { const float bv = bias[rem / HW]; v += f32x4{bv, bv, bv, bv}; }
        f32x4* p = reinterpret_cast<f32x4*>(y + n * y_bs + rem);
        if (accumulate) v += *p;
        *p = v;
    }
}

// G g G^T for F(4x4,3x3): 6x6 values per (co, ci) filter
__device__ __forceinline__ void wino4_g(const float g[9], float u[36]) {
    float t[18];                                           // G g : 6x3
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
        const float s = g0 + g2;
        t[c] = 0.25f * g0;
        t[3 + c] = (-1.f / 6.f) * (s + g1);
        t[6 + c] = (-1.f / 6.f) * (s - g1);
        const float q = __builtin_fmaf(4.f, g2, g0);       // g0 + 4 g2
        t[9 + c] = (1.f / 24.f) * (q + 2.f * g1);
        t[12 + c] = (1.f / 24.f) * (q - 2.f * g1);
        t[15 + c] = g2;
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {                          // (G g) G^T : 6x6
        const float a0 = t[r * 3], a1 = t[r * 3 + 1], a2 = t[r * 3 + 2];
        const float s = a0 + a2, q = __builtin_fmaf(4.f, a2, a0);
        u[r * 6] = 0.25f * a0;
        u[r * 6 + 1] = (-1.f / 6.f) * (s + a1);
        u[r * 6 + 2] = (-1.f / 6.f) * (s - a1);
        u[r * 6 + 3] = (1.f / 24.f) * (q + 2.f * a1);
        u[r * 6 + 4] = (1.f / 24.f) * (q - 2.f * a1);
        u[r * 6 + 5] = a2;
    }
}

struct W4PackDesc {
    const float* w; float* uf; float* ud;
    int Co, Ci, pad0, pad1;
    long block_start;
};

// One workgroup transforms a 32 co x 32 ci filter tile (see wino_pack_multi_kernel): one
// (co, ci) pair per thread and group, 36 runs of 128 floats per group of 4 channels:
//   uf [ci/4][36][2][Co][2]   ud [co/4][36][2][Ci][2] (taps reversed): per (channel group, position) the two
//   channel pairs as separate [row][2] runs = the A-operand lane order of the kernel's direct fragment loads
constexpr int P4_T = 32;
__global__ __launch_bounds__(128) void wino4_pack_multi_kernel(const W4PackDesc* __restrict__ descs, int n, long total_blocks) {
    __shared__ __attribute__((aligned(16))) float ot[36 * 128];
    // grid-stride over the filter tiles: the host may launch FEWER workgroups than tiles so that the (HBM-bound) re-layout
    // trickles along on a few CUs beside the first convolutions of the forward pass instead of flooding the chip
    for (long blk = blockIdx.x; blk < total_blocks; blk += gridDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block_start <= blk) lo = mid; else hi = mid - 1;
    }
    const W4PackDesc d = descs[lo];
    const int tid = threadIdx.x;
    const int tiles_ci = (d.Ci + P4_T - 1) / P4_T;
    const int tb = (int)(blk - d.block_start);
    const int co0 = (tb / tiles_ci) * P4_T, ci0 = (tb % tiles_ci) * P4_T;
    const int nci = min(P4_T, d.Ci - ci0);
    float g[9], u[36];
    const int lo2 = tid & 3, hi5 = tid >> 2;
    for (int grp = 0; grp < 16; ++grp) {                   // 8 forward groups of 4 ci, 8 dgrad groups of 4 co
        const bool fwd = grp < 8;
        const int q = grp & 7;
        float* dst = fwd ? d.uf : d.ud;
        if (dst == nullptr) continue;
        const int co = fwd ? hi5 : 4 * q + lo2, ci = fwd ? 4 * q + lo2 : hi5;
        const bool live = fwd ? (ci0 + 4 * q < d.Ci) : (co0 + 4 * q < d.Co);
        if (!live) continue;
        {
            // straight from global: the 32 x 32 tile (36 KB) is read by 16 groups and stays in L1 / L2; the LDS copy of
            // it (37 KB per workgroup) held the kernel at two workgroups per CU (re-layout of a FuseUNet step 0.39 -> 0.31 ms)
            const bool in = co0 + co < d.Co && ci < nci;
            const float* wp = d.w + ((long)(co0 + co) * d.Ci + ci0 + ci) * 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) g[t] = in ? wp[fwd ? t : 8 - t] : 0.f;
        }
        wino4_g(g, u);
        // staging tile ot[p][channel pair][row][2]: row = co (forward) resp. ci (dgrad)
        const int otid = (lo2 >> 1) * 64 + hi5 * 2 + (lo2 & 1);
#pragma unroll
        for (int p = 0; p < 36; ++p) ot[p * 128 + otid] = u[p];
        __syncthreads();
        const int C = fwd ? d.Co : d.Ci, c0 = fwd ? co0 : ci0;
        const long gbase = (long)((fwd ? ci0 : co0) / 4 + q) * 18;
        const int nrow = min(P4_T, C - c0);                // rows that exist
        // output [group][position pair][channel pair][row][p & 1][2]: 16 bytes per row = the kernel's dwordx4 fragment
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int f = tid + k * 128, pq = f >> 6, hh = (f >> 5) & 1, row = f & 31;
            if (row < nrow) {
                const float* o0 = ot + f4_pos(pq, 0) * 128 + hh * 64 + row * 2;
                const float* o1 = ot + f4_pos(pq, 1) * 128 + hh * 64 + row * 2;
                *reinterpret_cast<f32x4*>(dst + (((gbase + pq) * 2 + hh) * C + c0 + row) * 4) =
                    f32x4{o0[0], o0[1], o1[0], o1[1]};
            }
        }
        __syncthreads();
    }
    }
}

}  // namespace

extern "C" {

int aide_conv3x3_wino4_supported(int Cin, int H, int W, int Cout) {
    // W == 16: two images per workgroup tile (the caller's N must be even: checked at launch and in aide_conv3x3_wino4_splitk).
    // 16 < W < 32 (the 20 x 20 bottleneck of the 320 x 320 workload): one tile column, its right part masked -- the same
    // 39 % tile use as F(2x2)'s 16 x 16 tiles there, at 2.25x fewer multiplies.
    return (H % 4 == 0 && W % 4 == 0 && H >= 16 && W >= 16 && Cout % 32 == 0 &&
            Cin % 8 == 0) ? 1 : 0;
}

// workgroup tile of a plane: 0 = 16 x 32 pixels, 1 = two 16-wide images side by side, 2 = the 20 x 20 canvas -- whichever
// covers the plane with fewer tile slots (canvas: 25 of 32 slots used)
static int f4_mode(int H, int W, int* bh, int* bw) {
    const int sh = (H + 15) / 16, sw = (W + 31) / 32, ch = (H + 19) / 20, cw = (W + 19) / 20;
    const int mode = W == 16 ? 1 : ((long)ch * cw < (long)sh * sw) ? 2 : 0;     // (32 slots per workgroup either way)
    *bh = mode == 2 ? ch : sh;
    *bw = mode == 2 ? cw : sw;
    return mode;
}

int aide_conv3x3_wino4_splitk(int N, int Cin, int H, int W, int Cout) {
    int bh, bw;
    f4_mode(H, W, &bh, &bw);
    const long nb = (long)bh * bw * (W == 16 ? N / 2 : N) * ((Cout + 63) / 64);
    const int pairs = Cin / 8;                             // a split gets a whole number of stage pairs
    const long target = 200;                               // workgroups a split launch aims for (sweep: 200 ahead of 112 / 144 / 256)
    int s = 1;
    while (nb * s < target && pairs % (s * 2) == 0 && s * 2 <= pairs / 4) s *= 2;
    return s;
}

// partial-statistics entries per channel written by a forward launch that is given a statistics sink (stats_parts of aide_conv3x3_wino4)
int aide_conv3x3_wino4_stats_parts(int N, int H, int W) {
    int bh, bw;
    f4_mode(H, W, &bh, &bw);
    return W < 32 ? 0 : N * bh * bw;
}

int aide_conv3x3_wino4_pack_blocks(int Co, int Ci) {
    return ((Co + P4_T - 1) / P4_T) * ((Ci + P4_T - 1) / P4_T);
}

// descs: DEVICE array of n 48-byte records {w, uf (or 0), ud (or 0), int32 Co, Ci, 0, 0, int64 block_start}
int aide_conv3x3_wino4_pack_multi(const void* descs, int n, int64_t total_blocks, hipStream_t stream) {
    if (!descs || n <= 0 || total_blocks <= 0) return AIDE_ERR_ARG;
    static_assert(sizeof(W4PackDesc) == 48, "descriptor layout");
    const long grid = total_blocks;                       // one workgroup per tile (a capped grid-stride grid only got slower)
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, wino4_pack_multi_kernel, dim3((unsigned)grid), dim3(128), 0, stream,
                       (const W4PackDesc*)descs, n, (long)total_blocks);
    return aide_launch_status();
}

// y (+)= conv3x3(x) with F(4x4,3x3)-packed filters u [Cin/4][36][Cout][4] (forward pack, or the dgrad pack
// with Cin/Cout swapped by the caller).  splitk from aide_conv3x3_wino4_splitk (or 1); ws: split-K slabs
// of aide_conv3x3_ws_bytes(N, H, W, Cout, splitk) bytes.
int aide_conv3x3_wino4(const float* x, int64_t x_bs, const float* u, const float* bias, float* y,
                       int64_t y_bs, int N, int Cin, int H, int W, int Cout, int accumulate, int splitk,
                       float* ws, float* stats_parts, const float* epi_scale, int epi_relu, const float* in_bn_tab,
                       int in_bn_group_images, hipStream_t stream) {
    const float* aff = epi_scale;
    const int aff_relu = aff ? epi_relu : 0;
    if (!x || !u || !y || !aide_conv3x3_wino4_supported(Cin, H, W, Cout) || x_bs % 4 || y_bs % 4 || (W == 16 && N % 2))
        return AIDE_ERR_ARG;
    static AideLdsOptIn lds_opt;             // per device, status checked (common.h)
    if (int rc = lds_opt.ensure([] {
            constexpr int plain = F4_LDS * (int)sizeof(float), bnin = (F4_LDS + 4 * F4_BNIN_MAXC) * (int)sizeof(float);
            const struct { const void* fn; int bytes; } ks[] = {
                {(const void*)conv3x3_wino4_kernel<0>, plain}, {(const void*)conv3x3_wino4_kernel<1>, plain},
                {(const void*)conv3x3_wino4_kernel<2>, plain}, {(const void*)conv3x3_wino4_kernel<0, true>, plain},
                {(const void*)conv3x3_wino4_kernel<2, true>, plain}, {(const void*)conv3x3_wino4_kernel<0, false, true>, bnin},
                {(const void*)conv3x3_wino4_kernel<2, false, true>, bnin}};
            for (const auto& k : ks) {
                const hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, k.bytes);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        })) return rc;
    W4Args a;
    if (stats_parts && !(splitk <= 1 && accumulate == 0 && W >= 32)) return AIDE_ERR_ARG;   // only a launch that writes final outputs
    a.stats = stats_parts;
    const int mode = f4_mode(H, W, &a.blocks_h, &a.blocks_w);
    if (aff && (accumulate != 0 || !bias || (mode == 1 && splitk <= 1))) return AIDE_ERR_ARG;
    a.scale = splitk > 1 ? nullptr : aff;           // a split launch leaves plain slabs: its reduce applies the epilogue
    a.relu = aff_relu;
    a.pair = mode == 1 ? 1 : 0;
    // input BatchNorm + ReLU in the loader: not with the image-pair tile or the folded epilogue (no caller needs either)
    if (in_bn_tab && (mode == 1 || aff || Cin > F4_BNIN_MAXC || in_bn_group_images < 1 || N % in_bn_group_images)) return AIDE_ERR_ARG;
    a.in_tab = in_bn_tab; a.in_ng = in_bn_tab ? in_bn_group_images : 1;
    a.x = x; a.u = u; a.x_bs = x_bs; a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
    a.n_co_tiles = (Cout + 63) / 64;
    a.stages_total = Cin / 4;
    if (splitk < 1) splitk = 1;
    if ((Cin / 8) % splitk != 0) return AIDE_ERR_ARG;
    if (splitk > 1 && !ws) return AIDE_ERR_ARG;
    a.splitk = splitk;
    if (splitk > 1) {
        a.y = ws; a.y_bs = (long)Cout * H * W; a.split_stride = (long)N * Cout * H * W;
        a.bias = nullptr; a.accumulate = 0;
    } else {
        a.y = y; a.y_bs = y_bs; a.split_stride = 0; a.bias = bias; a.accumulate = (accumulate == 1);
    }
    const long nb = (long)a.blocks_w * a.blocks_h * (a.pair ? N / 2 : N) * a.n_co_tiles * splitk;
    {   // tile group per XCD (nb / 8 consecutive logical blocks).  Memory-side reads of a launch: the filter pack once per
        // pixel-tile GROUP, the input (1.44x with its halo) once per co GROUP:  bytes ~ |U| P / gp + 1.44 |x| C / gc.
        const long P = (long)a.blocks_w * a.blocks_h * (a.pair ? N / 2 : N);
        const int C = a.n_co_tiles;
        const long per_xcd = nb / 8 > 0 ? nb / 8 : 1;
        const double ub = 144.0 * (double)Cin * Cout, xb = 1.44 * 4.0 * (double)N * Cin * H * W;
        long bp = 1; int bc = C;
        double best = 1e300;
        // (layer sweep, tools/bench_conv.py with AIDE_W4_RECT=0/1: the >= 19 GFLOP layers are level or 3-13 % faster with the
        // groups -- 512->512 @32x32 0.077 -> 0.067 ms -- the <= 5 GFLOP split-K layers 5-10 % slower, 32 workgroups of an XCD
        // then stream one small filter slice in step: they keep the co-fastest order)
        const bool grouped = AIDE_CONV_FLOPS(N, H, W, Cout, Cin) >= 8e9;
        for (long gp = 1; gp <= P && gp <= per_xcd && grouped; gp *= 2) {      // (powers of two: a handful of candidates per launch)
            if (P % gp) continue;
            for (int gc = 1; gc <= C; ++gc) {
                if (C % gc || gp * gc > per_xcd) continue;
                const double cost = ub * (double)P / (double)gp + xb * (double)C / (double)gc;
                if (cost < best) { best = cost; bp = gp; bc = gc; }
            }
        }
        a.gp = (int)bp; a.gc = bc;
    }
    if (a.in_tab) {
        const size_t lds_bytes = (F4_LDS + 4 * (size_t)Cin) * sizeof(float);
        if (mode == 2) {
            AIDE_LAUNCH_TIMED(AIDE_KT_WINO4, AIDE_CONV_FLOPS(N, H, W, Cout, Cin), (conv3x3_wino4_kernel<2, false, true>),
                              dim3((unsigned)nb), dim3(256), lds_bytes, stream, a);
        } else {
            AIDE_LAUNCH_TIMED(AIDE_KT_WINO4, AIDE_CONV_FLOPS(N, H, W, Cout, Cin), (conv3x3_wino4_kernel<0, false, true>),
                              dim3((unsigned)nb), dim3(256), lds_bytes, stream, a);
        }
    } else if (a.scale && mode == 2) {
        AIDE_LAUNCH_TIMED(AIDE_KT_WINO4, AIDE_CONV_FLOPS(N, H, W, Cout, Cin), (conv3x3_wino4_kernel<2, true>), dim3((unsigned)nb),
                          dim3(256), F4_LDS * sizeof(float), stream, a);
    } else if (a.scale) {
        AIDE_LAUNCH_TIMED(AIDE_KT_WINO4, AIDE_CONV_FLOPS(N, H, W, Cout, Cin), (conv3x3_wino4_kernel<0, true>), dim3((unsigned)nb),
                          dim3(256), F4_LDS * sizeof(float), stream, a);
    } else if (a.pair) {
        AIDE_LAUNCH_TIMED(AIDE_KT_WINO4, AIDE_CONV_FLOPS(N, H, W, Cout, Cin), conv3x3_wino4_kernel<1>, dim3((unsigned)nb), dim3(256),
                          F4_LDS * sizeof(float), stream, a);
    } else if (mode == 2) {
        AIDE_LAUNCH_TIMED(AIDE_KT_WINO4, AIDE_CONV_FLOPS(N, H, W, Cout, Cin), conv3x3_wino4_kernel<2>, dim3((unsigned)nb), dim3(256),
                          F4_LDS * sizeof(float), stream, a);
    } else {
        AIDE_LAUNCH_TIMED(AIDE_KT_WINO4, AIDE_CONV_FLOPS(N, H, W, Cout, Cin), conv3x3_wino4_kernel<0>, dim3((unsigned)nb), dim3(256),
                          F4_LDS * sizeof(float), stream, a);
    }
    int rc = aide_launch_status();
    if (rc != 0) return rc;
    if (splitk > 1 && accumulate != 2) {           // accumulate == 2: the caller consumes the slabs itself
        const long total4 = (long)N * Cout * H * W / 4;
        AIDE_LAUNCH_TIMED(AIDE_KT_REDUCE, 0.0, w4_splitk_reduce_kernel, dim3((unsigned)min((total4 + 255) / 256, 4096L)), dim3(256), 0,
                           stream, ws, (long)N * Cout * H * W, splitk, y, (long)y_bs, Cout, H * W, bias, accumulate,
                           total4, aff, aff_relu);
        rc = aide_launch_status();
    }
    return rc;
}

}  // extern "C"
