// 3x3 convolution weight gradient, transposed Winograd F(4x4,3x3) on fp32 MFMA (gfx950), for the large layers
// (autograd of nn.Conv2d(ci, co, 3, padding=1): models_twomodalinputs/netblocks.py:17,24,26).
//
//   dW = G^T [ sum_tiles (A Z A^T) (.) (B^T D B) ] G      Z: 4x4 tile of dz, D: the 6x6 input patch around it
//
// 36 multiplies per (tile, co, ci) for 16 output pixels (direct: 144, F(2x2): 64).  Per transform position the sum
// over tiles is a GEMM  M[p][co][ci] += ZT[p][co][tile] * V[p][ci][tile]  (K = tiles, two per MFMA).
//   * workgroup = 64 co x 32 ci x 36 positions; as in conv3x3_wino4.hip a (32 x 32) block belongs to two partner
//     waves that own transform rows 0-2 / 3-5 (18 accumulators: 16 in AGPRs, 2 pinned to VGPRs);
//   * chunk = 4 horizontally adjacent tiles (4 rows x 16 pixels).  dz tiles go global -> registers (a thread
//     owns one (co, tile) and produces all 36 values of A Z A^T); input patches go through a double-buffered raw
//     LDS tile and are transformed by (ci, tile, row-half) threads; ZT[36][64][4] and V[36][32][4] are double
//     buffered as [p][tile pair][channel][2], so every ds_read_b64 fragment is 512 contiguous bytes;
//   * vector-ALU work (it shares the pipe with the fp32 MFMA) sits in three clusters per chunk, the
//     wave-uniform role branch is hoisted around the whole loop;
//   * output: each wave reduces its 18 positions to a partial 3x3 (G^T M G is linear in the rows of M), partner
//     waves swap halves through LDS, the split's slab [9][Co][Ci] is written coalesced along ci and reduced by
//     the fixed-order slab reduce (launch_wgrad_reduce, conv3x3_wgrad.hip).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

int aide_wgrad_reduce_launch(const float* ws, int splits, int Co, int Ci, float* dw, void* queue, hipStream_t stream);

namespace {

template <int V> using ic = std::integral_constant<int, V>;

struct G4Args {
    const float* dz;
    const float* a;
    float* slabs;
    long dz_bs, a_bs;
    int N, Co, Ci, H, W;
    int rows_t, cols_c, n_co_tiles, n_ci_tiles, splits, chunks_total;
    int gco, gci;        // XCD group: gco x gci neighbouring (co, ci) tiles are consecutive logical blocks (see the launcher)
    int b_off;           // first logical block of this launch (a layer of >= 2 x target blocks runs as several launches)
};

constexpr int G4_NAGPR = 16;
// Position pairs (as conv3x3_wino4.hip): pair q = 3 i + t of transform row i holds columns (1,2), (3,4), (0,5) for t = 0, 1, 2
// -- the pairs both packed transforms produce.  Accumulator slot of column c within a row of 6: g4_slot.
__host__ __device__ constexpr int g4_slot(int c) { return c == 0 ? 4 : c == 5 ? 5 : c - 1; }
constexpr int G4_RS = 20;                  // raw input row: [-1][0..15][16][2 pad]: patch of tile t starts at 4 t (16-byte aligned)
constexpr int G4_DS = 144;                 // raw channel stride (= 16 mod 64: conflict-free b128 patch reads)
constexpr int G4_RAW = 32 * G4_DS;         // 4608 floats
constexpr int G4_ZT = 36 * 64 * 4;         // ZT[18 position pairs][2 tile pairs][64 co][tile & 1][p & 1]
constexpr int G4_V = 36 * 32 * 4;          // V[18 position pairs][2 tile pairs][32 ci][tile & 1][p & 1]
constexpr int G4_SET = G4_ZT + G4_V;       // 13824 floats
constexpr int G4_LDS = 2 * G4_SET + 2 * G4_RAW;   // 36864 floats = 144 KB (epilogue swap needs 4 x 72 x 64 = 18432)

__global__ __launch_bounds__(256, 1) void conv3x3_wgrad4_kernel(const G4Args g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int ph = wid & 1, cb = wid >> 1;                 // position half, co block of the MFMA role
    const int hs = wid >> 1, vt_hi = wid & 1;              // input-transform role: row half, tile pair

    // xcd_remap gives every XCD a contiguous range of logical blocks; inside a pixel split they are ordered by RECTANGLES of
    // gco x gci tiles, so the workgroups of one XCD (which run in step: same chunk count) re-use gco slices of dz and gci
    // slices of the input from their private L2 instead of one dz slice and every input slice
    const int b = g.b_off + xcd_remap(blockIdx.x, gridDim.x);
    const int per_split = g.n_co_tiles * g.n_ci_tiles;
    const int split = b / per_split, rb = b - split * per_split;
    const int gsz = g.gco * g.gci, grp = rb / gsz, within = rb - grp * gsz;
    const int ngci = g.n_ci_tiles / g.gci;
    const int co_tile = (grp / ngci) * g.gco + within / g.gci;
    const int ci_tile = (grp % ngci) * g.gci + within % g.gci;
    const int co0 = co_tile * 64, ci0 = ci_tile * 32;
    const int HW = g.H * g.W;
    const int cps = (g.chunks_total + g.splits - 1) / g.splits;
    const int c_begin = split * cps, c_end = min(c_begin + cps, g.chunks_total);
    const int c_stop = c_begin + ((max(c_end - c_begin, 0) + 1) & ~1);     // chunks are processed in pairs

    const __amdgpu_buffer_rsrc_t drs = make_rsrc(g.dz + (long)co0 * HW);
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(g.a + (long)ci0 * HW - (g.W + 1));
    const __amdgpu_buffer_rsrc_t nul = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dz), 0, 0, 0x00020000);

    // ---- dz tiles: thread = (co_l, tile), four 16-byte row loads straight into registers ----
    const int zco = 16 * wid + (lane >> 2), zt = lane & 3;
    // (Co % 64 == 32: the trailing half tile reads zeros for its upper 32 channels and drops their rows in the epilogue)
    const unsigned offZ = co0 + zco < g.Co ? (unsigned)(zco * HW + 4 * zt) * 4u : BUF_OOB;
    // ---- raw input units: interior 32 ci x 6 rows x 4 float4 (3 rounds), edges 32 x 6 x 2 dwords (2 rounds) ----
    // Which lanes must read zeros depends on the chunk (top / bottom image row, first / last column block) but
    // the lane sets are fixed: they are kept as 64-bit lane masks in scalar registers, the per-chunk halo
    // logic is scalar, and each of the 6 load offsets costs ONE v_cndmask per chunk.
    typedef unsigned long long u64;
    const int w_last = g.W - 16 * (g.cols_c - 1);          // valid pixels in the last column block (4..16)
    unsigned offDi[3], ldsDi[3], offDe[2], ldsDe[2];
    u64 mTopI[3], mBotI[3], mColI[3], mTopE[2], mBotE[2], mLeft[2], mRight[2];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int q = tid + e * 256, c = q / 24, rem = q - c * 24, r = rem >> 2, s4 = rem & 3;
        offDi[e] = (unsigned)(c * HW + r * g.W + 1 + 4 * s4) * 4u;
        ldsDi[e] = (unsigned)(c * G4_DS + r * G4_RS + 1 + 4 * s4);
        mTopI[e] = __builtin_amdgcn_ballot_w64(r == 0);
        mBotI[e] = __builtin_amdgcn_ballot_w64(r == 5);
        mColI[e] = __builtin_amdgcn_ballot_w64(4 * s4 >= w_last);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        int q = tid + e * 256;
        if (q >= 384) q -= 256;                            // spare lanes repeat a unit
        const int c = q / 12, rem = q - c * 12, r = rem >> 1, side = rem & 1;
        offDe[e] = (unsigned)(c * HW + r * g.W + (side ? 17 : 0)) * 4u;
        ldsDe[e] = (unsigned)(c * G4_DS + r * G4_RS + (side ? 17 : 0));
        mTopE[e] = __builtin_amdgcn_ballot_w64(r == 0);
        mBotE[e] = __builtin_amdgcn_ballot_w64(r == 5);
        mLeft[e] = __builtin_amdgcn_ballot_w64(side == 0);
        mRight[e] = __builtin_amdgcn_ballot_w64(side == 1);
    }
    const u64 mColZ = __builtin_amdgcn_ballot_w64(4 * zt >= w_last);

    // chunk cursors (scalar, advanced incrementally: no divisions in the loop): Z runs one chunk ahead of the
    // MFMAs, D two chunks ahead
    struct Cur { int c, cc, tr, n; };
    auto cur_init = [&](int c) {
        Cur k; k.c = c;
        const int cl = min(c, g.chunks_total - 1);
        k.cc = cl % g.cols_c; const int r2 = cl / g.cols_c; k.tr = r2 % g.rows_t; k.n = r2 / g.rows_t;
        return k;
    };
    auto cur_next = [&](Cur& k) {
        ++k.c;
        if (++k.cc == g.cols_c) { k.cc = 0; if (++k.tr == g.rows_t) { k.tr = 0; ++k.n; } }
    };
    auto sel = [&](unsigned off, u64 bad) {                // off where the lane's bit in `bad` is clear, else OOB
        unsigned r;
        asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(off), "v"(BUF_OOB), "s"(bad));
        return r;
    };
    f32x4 rz[4], rDi[3];
    float rDe[2];
    unsigned voZ, voD[5], dso = 0, aso = 0;
    int liveZ = 0, liveD = 0;
    // chunks past the end of the split (odd tail of the pair loop) read through an empty descriptor: zeros
    auto prep_z = [&](const Cur& k) {
        liveZ = k.c < c_end;
        dso = (unsigned)((long)k.n * g.dz_bs + 4 * k.tr * g.W + 16 * k.cc) * 4u;
        voZ = sel(offZ, k.cc == g.cols_c - 1 ? mColZ : 0ull);
    };
    auto prep_d = [&](const Cur& k) {
        liveD = k.c < c_end;
        aso = (unsigned)((long)k.n * g.a_bs + 4 * k.tr * g.W + 16 * k.cc) * 4u;
        const bool top = k.tr == 0, bot = k.tr == g.rows_t - 1, first = k.cc == 0, last = k.cc == g.cols_c - 1;
        const bool rbad = 16 * k.cc + 16 >= g.W;
#pragma unroll
        for (int l = 0; l < 3; ++l)
            voD[l] = sel(offDi[l], (top ? mTopI[l] : 0ull) | (bot ? mBotI[l] : 0ull) | (last ? mColI[l] : 0ull));
#pragma unroll
        for (int e = 0; e < 2; ++e)
            voD[3 + e] = sel(offDe[e], (top ? mTopE[e] : 0ull) | (bot ? mBotE[e] : 0ull) | (first ? mLeft[e] : 0ull) |
                                           (rbad ? mRight[e] : 0ull));
    };
    auto fetch_z = [&](int r) { rz[r] = buf_load_f32x4(liveZ ? drs : nul, voZ, dso + (unsigned)(r * g.W) * 4u); };
    auto fetch_d = [&](int l) {
        if (l < 3) rDi[l] = buf_load_f32x4(liveD ? ars : nul, voD[l], aso);
        else rDe[l - 3] = buf_load_f32(liveD ? ars : nul, voD[l], aso);
    };
    auto put_d = [&](int w, float* raw) {                  // 14 dword stores
        if (w < 12) raw[ldsDi[w >> 2] + (w & 3)] = rDi[w >> 2][w & 3];
        else raw[ldsDe[w - 12]] = rDe[w - 12];
    };

    // ---- input transform (B^T D B), thread = (row half hs, ci, tile) ----
    const int vtile = lane & 3, vci = (lane >> 2) + 16 * vt_hi;
    int xr0 = 2 * G4_SET + vci * G4_DS + 4 * vtile, xr1 = xr0 + G4_RAW;
    asm volatile("" : "+v"(xr0), "+v"(xr1));
    f32x2 tp[6][3];
    f32x2 tq[9];
    auto v_read = [&](int r, int xo) {                     // one 16-byte + one 8-byte read per patch row
        // (xo is a multiple of four floats by construction, but it travels through an opaque register: without the alignment
        // spelled out hipcc emitted three ds_read2_b32 per row -- 18 per chunk, each 4-way bank-conflicted: the lanes' row
        // starts are 16 bytes apart, i.e. 8 of the 32 dword banks -- where the layout was built for a conflict-free
        // ds_read_b128 (G4_DS = 16 mod 64) plus one ds_read_b64)
        const float* row = static_cast<const float*>(__builtin_assume_aligned(lds + xo + r * G4_RS, 16));
        const f32x4 q = *reinterpret_cast<const f32x4*>(row);
        const f32x2 e = *reinterpret_cast<const f32x2*>(row + 4);
        tp[r][0] = f32x2{q[0], q[1]}; tp[r][1] = f32x2{q[2], q[3]}; tp[r][2] = e;
    };
    auto v_math = [&](auto HS) {
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
            tq[3 * i] = o12; tq[3 * i + 1] = o34; tq[3 * i + 2] = o05;     // position pairs (1,2) (3,4) (0,5)
        }
    };
    // Operand layouts [position pair q][tile pair][channel][tile & 1][p & 1]: a lane's 16-byte fragment read carries both K
    // slots (tiles) of both positions of a pair -- 18 + 18 ds_read_b128 per chunk instead of 36 + 36 ds_read_b64 (which hipcc
    // merged into ds_read2st64_b64 at half the LDS rate) -- and a thread stores the two positions of a pair as one 8-byte
    // store.  Tile pair 1's run is XOR-swizzled by 16 dwords so that the 16-lane store groups cover all 32 banks; the reads
    // apply the same XOR (a permutation of 16-byte slots: conflict-free).
    const int vpos = (vtile >> 1) * 128 + ((vci * 4 + (vtile & 1) * 2) ^ ((vtile >> 1) * 16));
    // per-set store bases as live registers (as xr0 / xr1): with one base + a > 64 KB constant the offsets do not fit the
    // 16-bit DS immediate and hipcc re-derives the address with a v_add per store (on the MFMA's pipe)
    // (bases in units of 8 bytes: the stores are then provably 8-byte aligned -> ds_write_b64 with a 16-bit immediate)
    f32x2* const lds2 = reinterpret_cast<f32x2*>(lds);
    int vb0 = (G4_ZT + 9 * hs * 256 + vpos) / 2, vb1 = (G4_SET + G4_ZT + 9 * hs * 256 + vpos) / 2;
    asm volatile("" : "+v"(vb0), "+v"(vb1));
    auto v_store = [&](int m, int set) { lds2[(set ? vb1 : vb0) + m * 128] = tq[m]; };

    // ---- dz transform (A Z A^T), thread = (co_l, tile), all 36 values ----
    f32x2 zq[12];               // position pairs (1,2) and (3,4) of the six transform rows
    float z0[6], z5[6];          // columns 0 and 5 (the third pair of a row), stored as two dwords
    auto z_math = [&]() {
        // column pass over column pairs: y0 = z0, y1/y2 = (z0 + z2) +- (z1 + z3), y3/y4 = (z0 + 4 z2) +- 2 (z1 + 4 z3), y5 = z3
        f32x2 Y[6][2];
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const f32x2 z0 = f32x2{rz[0][2 * cp], rz[0][2 * cp + 1]}, z1 = f32x2{rz[1][2 * cp], rz[1][2 * cp + 1]},
                        z2 = f32x2{rz[2][2 * cp], rz[2][2 * cp + 1]}, z3 = f32x2{rz[3][2 * cp], rz[3][2 * cp + 1]};
            const f32x2 s02 = z0 + z2, s13 = z1 + z3, aa = z0 + 4.f * z2, bb = z1 + 4.f * z3;
            Y[0][cp] = z0; Y[1][cp] = s02 + s13; Y[2][cp] = s13 * f32x2{-1.f, -1.f} + s02;    // (one packed FMA; a packed subtract is scalarised)
            Y[3][cp] = aa + 2.f * bb; Y[4][cp] = aa - 2.f * bb; Y[5][cp] = z3;
        }
        // row pass, four packed ops per row: (s02, a) (s13, b) (o1, o2) (o3, o4); o0 = t0, o5 = t3
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x2 t01 = Y[i][0], t23 = Y[i][1];
            const f32x2 sa = f32x2{t23.x, t23.x} * f32x2{1.f, 4.f} + f32x2{t01.x, t01.x};
            const f32x2 sb = f32x2{t23.y, t23.y} * f32x2{1.f, 4.f} + f32x2{t01.y, t01.y};
            const f32x2 o12 = f32x2{sb.x, sb.x} * f32x2{1.f, -1.f} + f32x2{sa.x, sa.x};
            const f32x2 o34 = f32x2{sb.y, sb.y} * f32x2{2.f, -2.f} + f32x2{sa.y, sa.y};
            zq[2 * i] = o12; zq[2 * i + 1] = o34; z0[i] = t01.x; z5[i] = t23.y;
        }
    };
    const int zpos = (zt >> 1) * 256 + ((zco * 4 + (zt & 1) * 2) ^ ((zt >> 1) * 16));   // ZT[q][tile pair][co 64][tile & 1][p & 1]
    int zb0 = zpos / 2, zb1 = (G4_SET + zpos) / 2;
    asm volatile("" : "+v"(zb0), "+v"(zb1));
    // 24 stores per chunk: k = 4 i + {0, 1}: the 8-byte pairs of row i; 4 i + {2, 3}: columns 0 and 5 (pair q = 3 i + 2)
    auto z_store = [&](int k, int set) {
        const int i = k >> 2, t = k & 3, zb = set ? zb1 : zb0;
        if (t < 2) lds2[zb + (3 * i + t) * 256] = zq[2 * i + t];
        else lds[2 * zb + (3 * i + 2) * 512 + (t - 2)] = (t == 2) ? z0[i] : z5[i];
    };

    float* const set0 = lds;
    float* const set1 = lds + G4_SET;
    float* const raw0 = lds + 2 * G4_SET;
    float* const raw1 = raw0 + G4_RAW;

    auto run = [&](auto HS) {
    f32x16 acc[18];
#pragma unroll
    for (int p = 0; p < 18; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

    // ---- prologue: chunk c0 -> set0 (ZT, V); raw input of c0 + 1 -> raw1; dz of c0 + 1 is fetched in the loop ----
    Cur kz = cur_init(c_begin), kd = cur_init(c_begin);
    prep_z(kz); prep_d(kd);
#pragma unroll
    for (int r = 0; r < 4; ++r) fetch_z(r);
#pragma unroll
    for (int l = 0; l < 5; ++l) fetch_d(l);
#pragma unroll
    for (int w = 0; w < 14; ++w) put_d(w, raw0);
    cur_next(kd);
    prep_d(kd);
#pragma unroll
    for (int l = 0; l < 5; ++l) fetch_d(l);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 6; ++r) v_read(r, xr0);
    v_math(HS);
#pragma unroll
    for (int m = 0; m < 9; ++m) v_store(m, 0);
    z_math();
#pragma unroll
    for (int k = 0; k < 24; ++k) z_store(k, 0);
#pragma unroll
    for (int w = 0; w < 14; ++w) put_d(w, raw1);
    __syncthreads();

    // ---- main loop.  While the MFMAs consume set `sc` (chunk c):
    //   dz tiles of c + 1 are fetched into registers, transformed and stored to sn.ZT;
    //   the raw input of c + 1 (in rawn) is transformed into sn.V;
    //   the raw input of c + 2 is fetched and stored into rawc (consumed by the previous chunk's transform).
    auto chunk = [&](int c, float* sc, float* sn, float* rawc, auto CUR) {
        constexpr int kcur = decltype(CUR)::value;
        const float* la = sc + (9 * ph) * 512 + half * 256 + (((cb * 32 + j) * 4) ^ (half * 16));     // ZT[q][tile pair][co 64][4]
        const float* lb = sc + G4_ZT + (9 * ph) * 256 + half * 128 + ((j * 4) ^ (half * 16));        // V[q][tile pair][ci 32][4]
        f32x4 fa[3], fb[3];                                 // fragments of a position pair: {t0p0, t0p1, t1p0, t1p1}
        auto frag = [&](int q, int s3) {
            fa[s3] = *reinterpret_cast<const f32x4*>(la + q * 512);
            fb[s3] = *reinterpret_cast<const f32x4*>(lb + q * 256);
        };
        frag(0, 0); frag(1, 1);
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            const int gq = st >> 2, w = st & 3, pi = 2 * gq + (w & 1), k = w >> 1;
            const int fs = gq % 3, e = k * 2 + (w & 1);
            if (w == 0 && gq + 2 < 9) frag(gq + 2, (gq + 2) % 3);
            if (pi < G4_NAGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[pi]) : "v"(fa[fs][e]), "v"(fb[fs][e]));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[pi]) : "v"(fa[fs][e]), "v"(fb[fs][e]));
            // schedule (vector-ALU work only in slots 0, 14 and 20):
            //   0 offsets;  0..3 dz[c+1] loads, 4..8 raw[c+2] loads;  1..6 patch reads
            //   14 input transform, 15..23 V stores;  20 dz transform, 21..32 ZT stores (3 per slot)
            //   31..35 raw[c+2] stores (3 per slot)
            if (st == 0) { cur_next(kz); prep_z(kz); }
            if (st < 4) fetch_z(st);
            if (st == 3) { cur_next(kd); prep_d(kd); }
            if (st >= 4 && st < 9) fetch_d(st - 4);
            if (st >= 1 && st < 7) v_read(st - 1, kcur ? xr0 : xr1);
            if (st == 14) v_math(HS);
            if (st >= 15 && st < 24) v_store(st - 15, 1 - kcur);
            if (st == 20) z_math();
            if (st >= 21 && st < 33) {
#pragma unroll
                for (int q = 0; q < 2; ++q) z_store(2 * (st - 21) + q, 1 - kcur);
            }
            if (st >= 31) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (3 * (st - 31) + q < 14) put_d(3 * (st - 31) + q, rawc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    for (int c = c_begin; c < c_stop; c += 2) {
        chunk(c, set0, set1, raw0, ic<0>{});
        chunk(c + 1, set1, set0, raw1, ic<1>{});
    }

    // ---- dW = G^T M G.  Partial over this wave's rows i = 3ph..3ph+2:
    //   T[i][b] = sum_c M[i][c] G[c][b];   dWp[a][b] = sum_i G[i][a] T[i][b]
    float* xbuf = lds;
    float* slab = g.slabs + (long)split * 9 * g.Co * g.Ci;
    const int ci = ci0 + j;
    auto epilogue = [&](auto PH) {
        constexpr int kph = decltype(PH)::value;
        auto partial = [&](int r, float* wp) {
            float T[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float m0 = acc[6 * i + g4_slot(0)][r], m1 = acc[6 * i + g4_slot(1)][r], m2 = acc[6 * i + g4_slot(2)][r],
                            m3 = acc[6 * i + g4_slot(3)][r], m4 = acc[6 * i + g4_slot(4)][r], m5 = acc[6 * i + g4_slot(5)][r];
                const float s12 = m1 + m2, d21 = m2 - m1, s34 = m3 + m4, d34 = m3 - m4;
                T[i][0] = 0.25f * m0 - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
                T[i][1] = (1.f / 6.f) * d21 + (1.f / 12.f) * d34;
                T[i][2] = (1.f / 6.f) * (s34 - s12) + m5;
            }
#pragma unroll
            for (int bq = 0; bq < 3; ++bq) {
                if (kph == 0) {                             // G rows 0,1,2: (1/4,0,0) (-1/6,-1/6,-1/6) (-1/6,1/6,-1/6)
                    const float sm = T[1][bq] + T[2][bq];
                    wp[bq] = 0.25f * T[0][bq] - (1.f / 6.f) * sm;
                    wp[3 + bq] = (1.f / 6.f) * (T[2][bq] - T[1][bq]);
                    wp[6 + bq] = -(1.f / 6.f) * sm;
                } else {                                    // G rows 3,4,5: (1/24,1/12,1/6) (1/24,-1/12,1/6) (0,0,1)
                    const float sm = T[0][bq] + T[1][bq];
                    wp[bq] = (1.f / 24.f) * sm;
                    wp[3 + bq] = (1.f / 12.f) * (T[0][bq] - T[1][bq]);
                    wp[6 + bq] = (1.f / 6.f) * sm + T[2][bq];
                }
            }
        };
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            float wp[9];
            partial(rr + 8 * (1 - kph), wp);               // the partner's rows
#pragma unroll
            for (int o = 0; o < 9; ++o) xbuf[((wid * 72) + rr * 9 + o) * 64 + lane] = wp[o];
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = rr + 8 * kph;
            float wp[9];
            partial(r, wp);
            const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
            for (int o = 0; o < 9; ++o)
                if (co < g.Co) slab[((long)o * g.Co + co) * g.Ci + ci] = wp[o] + xbuf[(((wid ^ 1) * 72) + rr * 9 + o) * 64 + lane];
        }
    };
    if (ph == 0) epilogue(ic<0>{}); else epilogue(ic<1>{});
    };   // run
    if (hs == 0) run(ic<0>{}); else run(ic<1>{});
}

}  // namespace

extern "C" {

int aide_conv3x3_wgrad_wino4_supported(int Co, int Ci, int H, int W) {
    return (H % 4 == 0 && W % 4 == 0 && H >= 8 && W >= 16 && Co % 32 == 0 && Ci % 32 == 0) ? 1 : 0;
}

// Workgroups per launch.  One per CU (256) is NOT the optimum for a kernel that lives on the side stream: its 144 KB
// workgroups own a CU and keep the main stream's kernels off it.  With 128 the weight gradient takes half of the chip
// for twice as long and the dependent chain (BatchNorm backward, dgrad, pooling / up-sampling backward) runs on the
// other half undisturbed: same-box C2 step 537 -> 566 images/s (256 -> 128; 192: 552, 144: 550, 112: 552, 96: 525,
// 64: 448, 384: 526, 512: 520), co-teaching step 123.7 -> 126.9.  (Probe switch AIDE_WG4_TARGET.)
// target <= 0: that default.  A caller that knows nothing is left to share the chip with (the last such launch of a
// backward pass whose dependent chain has already ended) asks for 256.
int aide_conv3x3_wgrad_wino4_splits_t(int N, int Co, int Ci, int H, int W, int target_wgs) {
    const long blocks = (long)((Co + 63) / 64) * (Ci / 32);
    const long chunks = (long)N * (H / 4) * ((W + 15) / 16);
    const long target = target_wgs > 0 ? target_wgs : 128;      // default: half of the chip (HISTORY §4.6)
    long s = (target + blocks - 1) / blocks;
    if (s > chunks / 2) s = chunks / 2;
    if (s < 1) s = 1;
    return (int)s;
}

int aide_conv3x3_wgrad_wino4_splits(int N, int Co, int Ci, int H, int W) {
    return aide_conv3x3_wgrad_wino4_splits_t(N, Co, Ci, H, W, 0);
}

size_t aide_conv3x3_wgrad_wino4_ws_bytes_t(int N, int Co, int Ci, int H, int W, int target_wgs) {
    return (size_t)aide_conv3x3_wgrad_wino4_splits_t(N, Co, Ci, H, W, target_wgs) * 9 * Co * Ci * sizeof(float);
}

size_t aide_conv3x3_wgrad_wino4_ws_bytes(int N, int Co, int Ci, int H, int W) {
    return aide_conv3x3_wgrad_wino4_ws_bytes_t(N, Co, Ci, H, W, 0);
}

// dw [Co][Ci][3][3] = sum over images and pixels of dz (x) shifted input; ws: aide_conv3x3_wgrad_wino4_ws_bytes()
int aide_conv3x3_wgrad_wino4_t(const float* dz, int64_t dz_bs, const float* a, int64_t a_bs, float* dw, int N,
                               int Co, int Ci, int H, int W, float* ws, int target_wgs, void* queue, hipStream_t stream) {
    if (!dz || !a || !dw || !ws || N <= 0 || !aide_conv3x3_wgrad_wino4_supported(Co, Ci, H, W) || dz_bs % 4 || a_bs % 4)
        return AIDE_ERR_ARG;
    static AideLdsOptIn lds_opt;             // per device, status checked (common.h)
    if (int rc = lds_opt.ensure([] {
            return hipFuncSetAttribute((const void*)conv3x3_wgrad4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G4_LDS * (int)sizeof(float));
        })) return rc;
    G4Args g;
    g.dz = dz; g.a = a; g.slabs = ws; g.dz_bs = dz_bs; g.a_bs = a_bs;
    g.N = N; g.Co = Co; g.Ci = Ci; g.H = H; g.W = W;
    g.rows_t = H / 4; g.cols_c = (W + 15) / 16;
    g.n_co_tiles = (Co + 63) / 64; g.n_ci_tiles = Ci / 32;
    g.chunks_total = N * g.rows_t * g.cols_c;
    g.splits = aide_conv3x3_wgrad_wino4_splits_t(N, Co, Ci, H, W, target_wgs);
    const long nb = (long)g.n_co_tiles * g.n_ci_tiles * g.splits;
    // Tile rectangle per XCD (nb / 8 consecutive logical blocks).  Memory-side reads of a launch: every dz slice once per
    // rectangle COLUMN that needs it, every input slice (1.5x with its halo rows) once per rectangle ROW:
    //   bytes ~ |dz| * n_ci / gci + 1.5 |x| * n_co / gco,  gco * gci <= blocks per XCD  ->  2 / gci + 1.5 / gco minimal.
    // (ci-fastest order = a 1 x n_ci rectangle read 1024->512 @32x32's input 8 times: 208 MB of fetches for 34 MB of operands.)
    // A layer with >= 2 x target (co, ci) tiles (1024 -> 512: 256 tiles; the 1024-channel layers of the U-Net: 512) puts a
    // 144 KB workgroup on EVERY CU and the main stream's kernels wait for CUs meanwhile (~100 us stalls of the dependent chain
    // at 4.80 / 4.92 ms of the C2 step, profiles/r06_timeline_c2.txt).  Running such a layer as consecutive launches of
    // `target` workgroups (b_off) was built and measured in round 6 -- same-box step A/B, split / one launch: C2 614.7 / 627.5,
    // C4 382.1 / 397.3, C3 155.0 / 155.7 images/s (profiles/r06_wgrad4_split_ab.txt): the stall costs less than the second
    // launch's ramp and the halved rate of these, the most efficient, layers.  One launch.
    const long per_launch = nb;
    {
        const long per_xcd = per_launch / 8 > 0 ? per_launch / 8 : 1;
        int bco = 1, bci = g.n_ci_tiles;
        double best = 1e30;
        for (int gco = 1; gco <= g.n_co_tiles; ++gco) {
            if (g.n_co_tiles % gco) continue;
            for (int gci = 1; gci <= g.n_ci_tiles; ++gci) {
                if (g.n_ci_tiles % gci || (long)gco * gci > per_xcd) continue;
                const double cost = 2.0 / gci + 1.5 / gco;
                if (cost < best - 1e-12) { best = cost; bco = gco; bci = gci; }
            }
        }
        g.gco = bco; g.gci = bci;
    }
    for (long off = 0; off < nb; off += per_launch) {
        const long cnt = nb - off < per_launch ? nb - off : per_launch;
        g.b_off = (int)off;
        AIDE_LAUNCH_TIMED(AIDE_KT_WGRAD4, AIDE_CONV_FLOPS(N, H, W, Co, Ci) * (double)cnt / (double)nb, conv3x3_wgrad4_kernel,
                          dim3((unsigned)cnt), dim3(256), G4_LDS * sizeof(float), stream, g);
    }
    const int rc = aide_launch_status();
    if (rc != 0) return rc;
    return aide_wgrad_reduce_launch(ws, g.splits, Co, Ci, dw, queue, stream);
}

}  // extern "C"
