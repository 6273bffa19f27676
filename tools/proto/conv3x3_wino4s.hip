// 3x3 / pad 1 convolution, Winograd F(4x4, 3x3), fp32 arithmetic carried by the bf16 matrix pipe (gfx950).
// Forward and dgrad of the large layers (nn.Conv2d(ci, co, 3, padding=1): models_twomodalinputs/netblocks.py:17,24,26).
//
// Same decomposition, workgroup tile, staging and output transform as conv3x3_wino4.hip; what differs is the contraction
//   M[p][co][tile] += U[p][co][ci] * V[p][tile][ci]
// Every transform-domain operand is written as the exact sum of three bf16 terms, u = uh + um + ul (8 + 8 + 8 significand
// bits, each term the round-to-nearest bf16 of what the terms before it left), and the SIX largest cross products
//   uh vh, uh vm, um vh, um vm, ul vh, uh vl           (dropped: um vl, ul vm, ul vl <= 2^-24 |u v|)
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- 1/16 of the fp32 MFMA's time per multiply, 6/16 with the six
// products.  The 16 K slots of one instruction hold 8 input channels x 2 terms (lanes 0-31 carry K slots 0-7, lanes
// 32-63 slots 8-15), so three instructions per (position, 8 channels) cover the six products:
//   A = [uh | um]  B = [vh | vh]      A = [uh | um]  B = [vm | vm]      A = [ul | uh]  B = [vh | vl]
//   * stage = 8 input channels: raw halo tile [8][18][41] fp32 and V[36][3 terms][32 tiles][8 ch] bf16, double buffered
//     (2 x 80 KB of the CU's 160 KB);
//   * a transform thread owns (tile, channel pair (c, c + 4), transform rows 0-2 | 3-5): the two channels of a pair are
//     the two halves of one bf16x2 word, so the split results are packed without any cross-lane traffic and every V store
//     is a lane-linear ds_write_b32;
//   * the filter terms are pre-split by the pack kernel into [stage][position][uh, um, ul, uh][Co][8] bf16 and stream
//     global -> registers as two 16-byte loads per position (a ring of six positions);
//   * accumulator layout = the fp32 kernel's (the C/D map of the 32x32 MFMA does not depend on the operand type), so the
//     output transform, the partner exchange and the BatchNorm statistics of the epilogue are the same code.
// Error against float64 of the same fp32 inputs: profiles/r05_split_error.txt (below the fp32 F(4x4) kernel's own).
#include "../../aide_amd/csrc/common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

template <int V> using ic = std::integral_constant<int, V>;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// compile-time loop: f(ic<0>{}), f(ic<1>{}), ... -- the schedule of a stage is a function of the slot number, which must be a
// constant in every slot (a `#pragma unroll` loop of this size is silently left rolled by the optimiser)
template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>) { (f(ic<I>{}), ...); }

struct S4Args {
    const float* x;
    const unsigned short* u;   // [Cin/8][36][4 planes: uh, um, ul, uh][Cout][8] bf16 (K slot 2 d + e <-> channel d + 4 e)
    const float* bias;
    float* y;
    long x_bs, y_bs, split_stride;
    int N, Cin, H, W, Cout;
    int blocks_w, blocks_h, n_co_tiles, splitk, stages_total, accumulate;
    int gp, gc;
    float* stats;
};

constexpr int S4_NAGPR = 16;
constexpr int S4_RRS = 41;                 // raw row: [3 pad][-1][0..31][32][4 pad]
constexpr int S4_RCS = 775;                // raw channel stride (conflict-free patch reads, see conv3x3_wino4.hip)
constexpr int S4_RAW = 8 * S4_RCS;         // 6200 dwords
constexpr int S4_V = 36 * 3 * 128;         // dwords: V[position 36][term 3][tile 32][4 dwords = 8 bf16]
constexpr int S4_LDS = 2 * (S4_RAW + S4_V); // 160192 B: [raw 0][raw 1][V 0][V 1] (the epilogue swap needs 131072)
static_assert(S4_LDS * 4 <= 160 * 1024, "LDS");
static_assert((S4_RAW % 4) == 0 && (S4_V % 4) == 0, "16-byte aligned V");

__host__ __device__ constexpr int s4_slot(int c) { return c == 0 ? 4 : c == 5 ? 5 : c - 1; }   // within a row of 6

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float s4_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float s4_half_total(float v) {
    v = s4_dpp_add<0x111, 0xf>(v);
    v = s4_dpp_add<0x112, 0xf>(v);
    v = s4_dpp_add<0x114, 0xf>(v);
    v = s4_dpp_add<0x118, 0xf>(v);
    v = s4_dpp_add<0x142, 0xa>(v);
    return v;
}

__global__ __launch_bounds__(256, 1) void conv3x3_wino4s_kernel(const S4Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int ph = wid & 1, cb = wid >> 1;                 // position half (transform rows 3ph..3ph+2), co block

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int p_in = b % a.gp;            b /= a.gp;
    const int co_in = b % a.gc;           b /= a.gc;
    const int ncg = a.n_co_tiles / a.gc;
    const int cog = b % ncg;              b /= ncg;
    const int split = b % a.splitk;       b /= a.splitk;
    const int co_tile = cog * a.gc + co_in;
    int pt = b * a.gp + p_in;
    const int tw = pt % a.blocks_w;       pt /= a.blocks_w;
    const int th = pt % a.blocks_h;
    const int n = pt / a.blocks_h;
    constexpr int NR = 18, NI = 8, RRS = S4_RRS;
    const int h0 = th * 16, w0 = tw * 32, co0 = co_tile * 64;
    const int HW = a.H * a.W;

    const int sps = a.stages_total / a.splitk;             // even (host)
    const int s_begin = split * sps;
    const int s_end = s_begin + sps;

    // ---- staging descriptors: interior 8 ci x 18 rows x 8 float4 = 1152 units (5 rounds; spare lanes repeat a unit),
    // edges 8 x 18 x 2 dwords = 288 units (2 rounds)
    constexpr int NUI = 8 * NR * NI, NUE = 8 * NR * 2;
    unsigned offB[5], ldsB[5], offC[2], ldsC[2];
#pragma unroll
    for (int e = 0; e < 5; ++e) {
        int q = tid + e * 256;
        while (q >= NUI) q -= 256;
        const int c = q / (NR * NI), rem = q - c * (NR * NI), r = rem / NI, s4 = rem - r * NI;
        const int ih = h0 - 1 + r, iw = w0 + 4 * s4;
        const bool ok = ih >= 0 && ih < a.H && iw < a.W;
        offB[e] = ok ? (unsigned)(c * HW + r * a.W + 1 + 4 * s4) * 4u : BUF_OOB;
        ldsB[e] = (unsigned)(c * S4_RCS + r * RRS + 4 + 4 * s4);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        int q = tid + e * 256;
        if (q >= NUE) q -= NUE;
        const int c = q / (NR * 2), rem = q - c * (NR * 2), r = rem >> 1, side = rem & 1;
        const int ih = h0 - 1 + r, iw = side ? w0 + 32 : w0 - 1;
        const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        offC[e] = ok ? (unsigned)(c * HW + r * a.W + (side ? 33 : 0)) * 4u : BUF_OOB;
        ldsC[e] = (unsigned)(c * S4_RCS + r * RRS + (side ? 36 : 3));
    }
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.x + (long)n * a.x_bs + (long)h0 * a.W + w0 - (a.W + 1));
    // filter terms: per (stage, position) four planes [uh, um, ul, uh] of [Cout][8] bf16; lane (half, co) reads 16 bytes of
    // plane `half` (X = [uh | um]) and of plane 2 + half (Y = [ul | uh])
    const long ubase = (long)(co0 + cb * 32) * 8;          // in bf16 elements
    const long utotal = (long)a.Cin * 36 * 4 * a.Cout;     // bf16 elements of the pack
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(a.u + ubase), 0, (int)((utotal - ubase) * 2), 0x00020000);
    const unsigned uoff = (unsigned)(half * a.Cout + j) * 16u;
    const unsigned upos = (unsigned)a.Cout * 64u;          // bytes per (stage, position): 4 planes
    const unsigned uy = (unsigned)a.Cout * 32u;            // planes 2, 3

    f32x4 rb[5];
    float rc[2];
    u32x4 uX[6], uY[6];                                    // ring of six positions
    auto fetch_u = [&](int pi, int stage) {                // pi: position of this wave's half, compile-time ring slot
        const unsigned us = ((unsigned)min(stage, s_end - 1) * 36u + (unsigned)(18 * ph + pi)) * upos;
        uX[pi % 6] = __builtin_bit_cast(u32x4, buf_load_f32x4(urs, uoff, us));
        uY[pi % 6] = __builtin_bit_cast(u32x4, buf_load_f32x4(urs, uoff, us + uy));
    };
    auto fetch = [&](int l, int stage) {                   // 7 raw loads
        const unsigned xs = (unsigned)(min(stage, s_end - 1) * 8) * (unsigned)HW * 4u;
        if (l < 5) rb[l] = buf_load_f32x4(xrs, offB[l], xs);
        else rc[l - 5] = buf_load_f32(xrs, offC[l - 5], xs);
    };
    // LDS: [raw set 0][raw set 1][V set 0][V set 1].  Both raw sets lie within the 64 KB a DS instruction's immediate offset
    // reaches from one address register; the V sets get their own (opaque) base registers
    auto put_raw = [&](int w, int set) {                   // 22 dword stores
        if (w < 20) lds[ldsB[w >> 2] + set * S4_RAW + (w & 3)] = rb[w >> 2][w & 3];
        else lds[ldsC[w - 20] + set * S4_RAW] = rc[w - 20];
    };

    // ---- input transform: thread = (half hs, tile, channel pair cp: channels cp and cp + 4) ----
    const int item = tid & 127, hs = wid >> 1;             // hs is wave-uniform
    const int tslot = item >> 2;
    const int trow = tslot >> 3, tcol = tslot & 7;
    int xr_off = (item & 3) * S4_RCS + trow * 4 * RRS + 3 + 4 * tcol;
    asm volatile("" : "+v"(xr_off));
    // The patch is consumed one COLUMN PAIR at a time: 12 values (6 rows x 2 columns) -> rows 3hs..3hs+2 of B^T d for that
    // pair (3 packed values); two 12-value buffers so that the reads run a column pair ahead of the arithmetic
    f32x2 tp[1][6];
    f32x2 T[2][3][3];                                      // [channel of the pair][row of the half][column pair]
    auto xf_read = [&](int u, int k, int xo) {             // u = 3 channel + column pair (0..5), k = 2 row + (column of the pair)
        const int ch = u / 3, cp = u % 3;
        const float v = lds[xo + ch * 4 * S4_RCS + (k >> 1) * RRS + 2 * cp + (k & 1)];
        if (k & 1) tp[0][k >> 1].y = v; else tp[0][k >> 1].x = v;
    };
    auto xf_col = [&](auto HS, int u) {
        constexpr int khs = decltype(HS)::value;
        const int ch = u / 3, cp = u % 3;
        const f32x2 d0 = tp[0][0], d1 = tp[0][1], d2 = tp[0][2], d3 = tp[0][3], d4 = tp[0][4], d5 = tp[0][5];
        if (khs == 0) {
            const f32x2 aa = d4 - 4.f * d2, bb = 4.f * d1 - d3;
            T[ch][0][cp] = (4.f * d0 + d4) - 5.f * d2;
            T[ch][1][cp] = bb * f32x2{-1.f, -1.f} + aa;
            T[ch][2][cp] = aa + bb;
        } else {
            const f32x2 cc = d2 * f32x2{-1.f, -1.f} + d4, ee = d1 * f32x2{-1.f, -1.f} + d3;
            T[ch][0][cp] = cc + 2.f * ee;
            T[ch][1][cp] = cc - 2.f * ee;
            T[ch][2][cp] = (4.f * d1 + d5) - 5.f * d3;
        }
    };
    f32x2 tq[2][3];                                        // one transform row of both channels: accumulator slots 6 i + 0..5
    auto xf_row = [&](int i) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const f32x2 t01 = T[ch][i][0], t23 = T[ch][i][1], t45 = T[ch][i][2];
            const f32x2 ac = f32x2{t23.x, t23.x} * f32x2{-4.f, -1.f} + f32x2{t45.x, t45.x};
            const f32x2 be = f32x2{t01.y, t01.y} * f32x2{-4.f, -1.f} + f32x2{t23.y, t23.y};
            tq[ch][0] = f32x2{be.x, be.x} * f32x2{1.f, -1.f} + f32x2{ac.x, ac.x};
            tq[ch][1] = f32x2{be.y, be.y} * f32x2{2.f, -2.f} + f32x2{ac.y, ac.y};
            tq[ch][2] = t23 * f32x2{-5.f, -5.f} + (t01 * f32x2{4.f, 4.f} + t45);
        }
    };
    // three-term split of the two channels' values at slot q of the row in tq, packed (channel cp | channel cp + 4) and stored
    // lane-linear: V[position][term][tile][4 dwords], dword = channel pair.  The remainders come from v_dot2c_f32_bf16 with
    // a (-1, 0) / (0, -1) selector: v - float(bf16 half) in ONE instruction, exact (the difference is representable)
    // (the selectors are opaque SGPR values: given the constant 0x0000bf80 hipcc encodes the inline constant -1.0, which the
    // hardware replicates into BOTH bf16 halves -- tools/ubench/dot2_bf16_split.hip; the SGPR and literal forms are exact,
    // subnormal remainders included)
    unsigned ksel_lo = 0x0000bf80u, ksel_hi = 0xbf800000u;
    asm volatile("" : "+s"(ksel_lo), "+s"(ksel_hi));
    auto split_store = [&](int i, int q, unsigned* vst) {  // vst: this thread's dword of (position 18 hs, term 0)
        float va = (q & 1) ? tq[0][q >> 1].y : tq[0][q >> 1].x, vb = (q & 1) ? tq[1][q >> 1].y : tq[1][q >> 1].x;
        const int k = 6 * i + q;
        const bf16x2 sel_lo = __builtin_bit_cast(bf16x2, ksel_lo), sel_hi = __builtin_bit_cast(bf16x2, ksel_hi);
        const bf16x2 H = __builtin_convertvector(f32x2{va, vb}, bf16x2);
        va = __builtin_amdgcn_fdot2_f32_bf16(sel_lo, H, va, false);
        vb = __builtin_amdgcn_fdot2_f32_bf16(sel_hi, H, vb, false);
        const bf16x2 M = __builtin_convertvector(f32x2{va, vb}, bf16x2);
        va = __builtin_amdgcn_fdot2_f32_bf16(sel_lo, M, va, false);
        vb = __builtin_amdgcn_fdot2_f32_bf16(sel_hi, M, vb, false);
        const bf16x2 L = __builtin_convertvector(f32x2{va, vb}, bf16x2);
        vst[(k * 3 + 0) * 128] = __builtin_bit_cast(unsigned, H);
        vst[(k * 3 + 1) * 128] = __builtin_bit_cast(unsigned, M);
        vst[(k * 3 + 2) * 128] = __builtin_bit_cast(unsigned, L);
    };
    // the transform of one stage as 54 schedule steps (one per MFMA slot):
    //   reads of unit u (channel, column pair: 12 values) in steps 3u..3u+2 (4 per step), its column pass in step 3u+4;
    //   then per transform row i: row pass of both channels in step 20 + 11 i, its six splits spread over the ten steps after
    auto xf_sched = [&](auto HS, int st, int xo, unsigned* vst) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (st == 3 * u + 3) xf_col(HS, u);
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (st >= 3 * u && st < 3 * u + 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) xf_read(u, 4 * (st - 3 * u) + e, xo);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int base = 20 + 11 * i;
            if (st == base) xf_row(i);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (st == base + 1 + (q * 5) / 3) split_store(i, q, vst);
        }
    };

    auto run = [&](auto HS) {
    f32x16 acc[18];
#pragma unroll
    for (int p = 0; p < 18; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
    u32x4 fb[4][3];                                        // B fragments of a position: [ring of 4][vh|vh, vm|vm, vh|vl]

    // per-set BYTE offsets into the dynamic LDS as opaque live registers (a base + 55 KB constant would be re-derived with a
    // v_add per access; opaque POINTERS would lose their address space and turn every access into a flat one)
    constexpr int V0 = 2 * S4_RAW, V1 = 2 * S4_RAW + S4_V;                 // dword offsets of the V sets
    unsigned vst0b = (unsigned)(V0 + (18 * hs * 3) * 128 + item) * 4u;    // this thread's V dword (position 18 hs, term 0)
    unsigned vst1b = (unsigned)(V1 + (18 * hs * 3) * 128 + item) * 4u;
    unsigned lb0b = (unsigned)(V0 + (18 * ph * 3) * 128 + j * 4) * 4u;    // this lane's fragment of position 18 ph
    unsigned lb1b = (unsigned)(V1 + (18 * ph * 3) * 128 + j * 4) * 4u;
    unsigned lc0b = lb0b + (unsigned)half * 1024u;         // lanes 32-63 read term vl where lanes 0-31 read vh
    unsigned lc1b = lb1b + (unsigned)half * 1024u;
    asm volatile("" : "+v"(vst0b), "+v"(vst1b), "+v"(lb0b), "+v"(lb1b), "+v"(lc0b), "+v"(lc1b));
    char* const ldsc = reinterpret_cast<char*>(lds);
    unsigned* const vst0 = reinterpret_cast<unsigned*>(ldsc + vst0b);
    unsigned* const vst1 = reinterpret_cast<unsigned*>(ldsc + vst1b);
    const u32x4* const lb0 = reinterpret_cast<const u32x4*>(ldsc + lb0b);
    const u32x4* const lb1 = reinterpret_cast<const u32x4*>(ldsc + lb1b);
    const u32x4* const lc0 = reinterpret_cast<const u32x4*>(ldsc + lc0b);
    const u32x4* const lc1 = reinterpret_cast<const u32x4*>(ldsc + lc1b);

    // ---- prologue: raw[s0] -> set0, raw[s0+1] -> set1, raw[s0+2] -> registers; V[s0] -> set0; U ring positions 0..4
#pragma unroll
    for (int l = 0; l < 7; ++l) fetch(l, s_begin);
#pragma unroll
    for (int pi = 0; pi < 5; ++pi) fetch_u(pi, s_begin);
#pragma unroll
    for (int w = 0; w < 22; ++w) put_raw(w, 0);
#pragma unroll
    for (int l = 0; l < 7; ++l) fetch(l, s_begin + 1);
#pragma unroll
    for (int w = 0; w < 22; ++w) put_raw(w, 1);
#pragma unroll
    for (int l = 0; l < 7; ++l) fetch(l, s_begin + 2);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 6; ++u) {
#pragma unroll
        for (int k = 0; k < 12; ++k) xf_read(u, k, xr_off);
        xf_col(HS, u);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        xf_row(i);
#pragma unroll
        for (int q = 0; q < 6; ++q) split_store(i, q, vst0);
    }
    __syncthreads();

    // ---- main loop.  Stage s consumes V of set `sc` and the filter ring; meanwhile
    //   raw[s+2] (in registers since the stage before) is stored into sc.raw (its old content was transformed a stage ago),
    //   raw[s+1] (in sn.raw) is transformed and split into sn.V, raw[s+3] is fetched into registers.
    auto stage = [&](int s, auto SC) {
        constexpr int ksc = decltype(SC)::value;          // the set being consumed
        const u32x4* lb = ksc ? lb1 : lb0;
        const u32x4* lc = ksc ? lc1 : lc0;
        unsigned* vst = ksc ? vst0 : vst1;
        const int xrn = xr_off + (1 - ksc) * S4_RAW;
        // positions in pairs, their three instructions interleaved: (2g, m0) (2g+1, m0) (2g, m1) (2g+1, m1) (2g, m2) (2g+1, m2) -- an
        // MFMA never waits for the one before it; B fragments in a ring of four positions (pair parity runs across both stages
        // of a loop iteration), the fragments of pair g + 1 requested in the first two slots of pair g
        auto frag = [&](int pi) {
            constexpr int dummy = 0; (void)dummy;
            const int r = (18 * ksc + pi) % 4;
            fb[r][0] = lb[(pi * 3 + 0) * 32];
            fb[r][1] = lb[(pi * 3 + 1) * 32];
            fb[r][2] = lc[(pi * 3 + 0) * 32];
        };
        frag(0); frag(1);
        static_for([&](auto ST) {
            constexpr int st = decltype(ST)::value, g = st / 6, w = st % 6, pi = 2 * g + (w & 1), m = w >> 1;
            if (w < 2 && 2 * g + 2 + w < 18) frag(2 * g + 2 + w);
            if (w == 0) { if (2 * g + 5 < 18) fetch_u(2 * g + 5, s); else fetch_u(2 * g + 5 - 18, s + 1); }
            if (w == 5) { if (2 * g + 6 < 18) fetch_u(2 * g + 6, s); else fetch_u(2 * g + 6 - 18, s + 1); }
            const u32x4 av = m == 2 ? uY[pi % 6] : uX[pi % 6];
            const u32x4 bv = fb[(18 * ksc + pi) % 4][m];
            if (pi < S4_NAGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[pi]) : "v"(av), "v"(bv));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[pi]) : "v"(av), "v"(bv));
            // staging schedule: raw[s+2] stores in slots 0..10 (2 per slot), the transform (xf_sched), raw[s+3] fetches in 24..30
            if (st < 11) { put_raw(2 * st, ksc); put_raw(2 * st + 1, ksc); }
            xf_sched(HS, st, xrn, vst);
            if (st >= 24 && st < 31) fetch(st - 24, s + 3);
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 54>{});
        __syncthreads();
    };
    for (int s = s_begin; s < s_end; s += 2) {
        stage(s, ic<0>{});
        stage(s + 1, ic<1>{});
    }

    // ---- output transform (conv3x3_wino4.hip: same accumulator layout) ----
    float* xbuf = lds;
    float* yn = a.y + (long)split * a.split_stride + (long)n * a.y_bs;
    const bool add_bias = (a.bias != nullptr) && (split == 0);
    const int oh = h0 + 4 * (j >> 3), ow = w0 + 4 * (j & 7);
    const bool pok = oh < a.H && ow < a.W;
    auto epilogue = [&](auto PH) {
        constexpr int kph = decltype(PH)::value;
        auto partial = [&](int r, f32x2* yp) {
            f32x2 T[3][4];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                auto mm = [&](int c) { return f32x2{acc[6 * i + s4_slot(c)][r], acc[6 * i + s4_slot(c)][r + 1]}; };
                const f32x2 m0 = mm(0), m1 = mm(1), m2 = mm(2), m3 = mm(3), m4 = mm(4), m5 = mm(5);
                const f32x2 s12 = m1 + m2, d12 = m2 * f32x2{-1.f, -1.f} + m1, s34 = m3 + m4, d34 = m4 * f32x2{-1.f, -1.f} + m3;
                T[i][0] = m0 + s12 + s34;
                T[i][1] = d34 * f32x2{2.f, 2.f} + d12;
                T[i][2] = s34 * f32x2{4.f, 4.f} + s12;
                T[i][3] = d34 * f32x2{8.f, 8.f} + d12 + m5;
            }
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) {
                if (kph == 0) {
                    const f32x2 sm = T[1][bq] + T[2][bq], df = T[2][bq] * f32x2{-1.f, -1.f} + T[1][bq];
                    yp[bq] = T[0][bq] + sm; yp[4 + bq] = df; yp[8 + bq] = sm; yp[12 + bq] = df;
                } else {
                    const f32x2 sm = T[0][bq] + T[1][bq], df = T[1][bq] * f32x2{-1.f, -1.f} + T[0][bq];
                    yp[bq] = sm; yp[4 + bq] = df * f32x2{2.f, 2.f}; yp[8 + bq] = sm * f32x2{4.f, 4.f};
                    yp[12 + bq] = df * f32x2{8.f, 8.f} + T[2][bq];
                }
            }
        };
        f32x2* const xbuf2 = reinterpret_cast<f32x2*>(xbuf);
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            f32x2 yp[16];
            partial(2 * rp + 8 * (1 - kph), yp);
#pragma unroll
            for (int o = 0; o < 16; ++o) xbuf2[((wid * 64) + rp * 16 + o) * 64 + lane] = yp[o];
        }
        __syncthreads();
        float bvs[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = q + 8 * kph;
            const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            bvs[q] = (add_bias && co < a.Cout) ? a.bias[co] : 0.f;
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
                if (a.stats) {
                    float s1 = 0.f, s2 = 0.f;
                    if (pok) {
#pragma unroll
                        for (int o = 0; o < 16; ++o) { s1 += yp[o][e]; s2 = __builtin_fmaf(yp[o][e], yp[o][e], s2); }
                    }
                    s1 = s4_half_total(s1);
                    s2 = s4_half_total(s2);
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
                    if (a.accumulate) {
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

}  // namespace

extern "C" {

int aide_conv3x3_wino4s_supported(int Cin, int H, int W, int Cout) {
    return (H % 4 == 0 && W % 4 == 0 && H >= 16 && W >= 32 && Cout % 32 == 0 && Cin % 16 == 0) ? 1 : 0;
}

// y (+)= conv3x3(x) with the split-bf16 F(4x4,3x3) filter pack u [Cin/8][36][4][Cout][8] bf16
int aide_conv3x3_wino4s(const float* x, int64_t x_bs, const void* u, const float* bias, float* y, int64_t y_bs, int N, int Cin,
                        int H, int W, int Cout, int accumulate, int splitk, float* ws, float* stats_parts, hipStream_t stream) {
    if (!x || !u || !y || !aide_conv3x3_wino4s_supported(Cin, H, W, Cout) || x_bs % 4 || y_bs % 4) return AIDE_ERR_ARG;
    static const int attr_rc = (int)hipFuncSetAttribute((const void*)conv3x3_wino4s_kernel,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, S4_LDS * (int)sizeof(float));
    if (attr_rc != 0) return attr_rc;
    S4Args a;
    if (stats_parts && !(splitk <= 1 && accumulate == 0)) return AIDE_ERR_ARG;
    a.stats = stats_parts;
    a.blocks_h = (H + 15) / 16; a.blocks_w = (W + 31) / 32;
    a.x = x; a.u = (const unsigned short*)u; a.x_bs = x_bs; a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
    a.n_co_tiles = (Cout + 63) / 64;
    a.stages_total = Cin / 8;
    if (splitk < 1) splitk = 1;
    if ((Cin / 16) % splitk != 0) return AIDE_ERR_ARG;
    if (splitk > 1 && !ws) return AIDE_ERR_ARG;
    a.splitk = splitk;
    if (splitk > 1) {
        a.y = ws; a.y_bs = (long)Cout * H * W; a.split_stride = (long)N * Cout * H * W;
        a.bias = nullptr; a.accumulate = 0;
    } else {
        a.y = y; a.y_bs = y_bs; a.split_stride = 0; a.bias = bias; a.accumulate = (accumulate == 1);
    }
    const long nb = (long)a.blocks_w * a.blocks_h * N * a.n_co_tiles * splitk;
    {
        const long P = (long)a.blocks_w * a.blocks_h * N;
        const int C = a.n_co_tiles;
        const long per_xcd = nb / 8 > 0 ? nb / 8 : 1;
        const double ub = 288.0 * (double)Cin * Cout, xb = 1.44 * 4.0 * (double)N * Cin * H * W;
        long bp = 1; int bc = C;
        double best = 1e300;
        const bool grouped = AIDE_CONV_FLOPS(N, H, W, Cout, Cin) >= 8e9;
        for (long gp = 1; gp <= P && gp <= per_xcd && grouped; gp *= 2) {
            if (P % gp) continue;
            for (int gc = 1; gc <= C; ++gc) {
                if (C % gc || gp * gc > per_xcd) continue;
                const double cost = ub * (double)P / (double)gp + xb * (double)C / (double)gc;
                if (cost < best) { best = cost; bp = gp; bc = gc; }
            }
        }
        a.gp = (int)bp; a.gc = bc;
    }
    AIDE_LAUNCH_TIMED(AIDE_KT_WINO4, AIDE_CONV_FLOPS(N, H, W, Cout, Cin), conv3x3_wino4s_kernel, dim3((unsigned)nb), dim3(256),
                      S4_LDS * sizeof(float), stream, a);
    return aide_launch_status();
}

}  // extern "C"
