// bf16-MFMA mode of the 3x3 / stride 1 / pad 1 convolution (BASELINE config 5):
// bf16 operands in LDS, v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 tensors in HBM, fp32 master
// weights / bias / BatchNorm statistics / loss.  Replaces (reference) nn.Conv2d(ci, co, 3, padding=1) forward and
// autograd's convolution_backward at models_twomodalinputs/netblocks.py:17,24,26 and
// models_singlemodalinput/UNet.py:12,19,21 when the engine runs with precision='bf16'.
//
// Numerics contract: every operand (activation, gradient, filter) is rounded to bf16 (RNE, v_cvt_pk_bf16_f32) when it
// is staged; products are exact in fp32 and the sums are fp32 fma chains, so the result equals an fp32 convolution
// of the bf16-rounded operands up to summation order (tests/test_gpu_bf16.py checks exactly that, <= 2e-5).
//
// Forward / dgrad (conv3x3_bf16_kernel): implicit GEMM  D[co][pix] = sum_{ci,tap} W[co][ci,tap] X[ci][pix+tap].
//   * one MFMA consumes 16 input channels at ONE tap: lanes 0-31 hold channels 0-7, lanes 32-63 channels 8-15 of
//     their row (A: output channel) / column (B: pixel) -- 8 bf16 = one 16-byte LDS slot per lane per operand;
//   * LDS image of the input halo tile is pixel-major: slot[g][row][col] = the 8 channels of group g at one pixel,
//     so the B fragment of any tap is 32 consecutive slots (512 contiguous bytes per half wave: conflict free) and
//     a tap shift is an immediate offset.  Staging converts NCHW fp32 -> that image in registers: a thread owns a
//     pixel PAIR x 8 channels (eight coalesced 8-byte loads, eight v_cvt_pk, two ds_write_b128);
//   * filters are pre-packed once per optimizer step to bf16 [Ci/16][tap][g][Co][8] = the A-fragment slot order, so
//     a filter block moves global -> LDS as plain 16-byte pieces;
//   * a workgroup (4 waves, one per SIMD) owns TCO x (TH rows x 32 pixels); the pipeline is two deep in registers:
//     while chunk c is multiplied out of LDS buffer `cur`, the registers holding chunk c+1 (fetched one whole stage
//     earlier) are converted and stored into the other buffer, and each freed register set immediately re-issues
//     its loads for chunk c+2 -- one staging op per tap, so HBM latency has a full stage (~1 us) to land;
//   * dgrad = the same kernel on the rotated, channel-transposed pack with an accumulate epilogue; split-K over
//     channel chunks into slabs with a fixed-order reduce for the deep layers.
//
// Weight gradient (conv3x3_wgrad_bf16_kernel): dW[co][ci][tap] = sum_pix dz[co][pix] x[ci][pix+tap], K = pixels.
//   * an MFMA consumes 16 consecutive pixels of one image row: NCHW rows are already K-contiguous, so both LDS
//     images stay channel-major (dz [co][4 rows][-1..38 px], x [ci][6 rows][32 px], bf16);
//   * the +-1 column shift of the taps is carried by the dz operand (dW[kh][kw] = sum_q dz[q - (kw-1)] x[q + (kh-1) W] over
//     the aligned pixels q): one shifted dz fragment serves all three kh.  A 2-byte shift would need misaligned 16-byte LDS
//     reads; instead the dz rows are stored starting at column -1, one aligned 32-byte window per (row, k-step) is read and
//     the three fragments are formed in registers (kw=2: the window itself, kw=1: four v_alignbit, kw=0: a register slice);
//     the x rows are plain aligned copies with a row halo;
//   * workgroup = 64 co x 64 ci x 9 taps (a wave owns 32 x 32 x 9 = 144 accumulators), stage = 4 image rows x 32
//     columns, pixel range split over workgroups into slabs [split][tap][co][ci] reduced in fixed order
//     (aide_wgrad_reduce_launch, shared with the fp32 kernels) -> bit-reproducible.
#include "common.h"

extern "C" int aide_conv3x3_wgrad_stem_splits(int N, int H, int W);
extern "C" int aide_conv3x3_wgrad_stem_supported(int Co, int Ci, int H, int W);                  // conv3x3_wgrad_stem.hip
extern "C" int aide_conv3x3_wgrad_stem(const void* dz, int dz_bf16, int64_t dz_bs, const float* x, int64_t x_bs, float* dw,
                                       int N, int Co, int Ci, int H, int W, float* ws, int splits, int round_bf16,
                                       void* queue, hipStream_t stream);

int aide_wgrad_reduce_launch(const float* ws, int splits, int Co, int Ci, float* dw, void* queue, hipStream_t stream);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ u32x4 buf_load_u32x4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}

// ------------------------------------------------------------------------------------------ forward / dgrad
struct BfArgs {
    const void* x;            // fp32, or bf16 when IN_BF16 (dgrad reading a bf16-stored dz)
    const uint16_t* wp;
    const float* bias;
    void* y;                  // fp32, or bf16 when OUT_BF16 (forward writing a bf16-stored z)
    long x_bs, y_bs, split_stride;
    int N, Cin, H, W, Cout;
    int tiles_w, tiles_h, n_co_tiles, splitk, chunks_total, accumulate;
};


// NW waves per workgroup: 4 (tile 32 / 64 co x 512 px, two workgroups per CU) or 8 (128 co x 512 px, one workgroup per
// CU: the two co halves share ONE halo tile, which halves the input bytes through L1 and LDS per MFMA -- the fp32 input
// stream, 59 B/clk/CU at full MFMA rate for the 64-co tile, is what bounds the deep layers)
template <int WM, int WAVES_M, int WN, int OCC, bool RAGGED, bool IN_BF16, bool OUT_BF16, int TW, int NW>
__global__ __launch_bounds__(64 * NW, OCC) void conv3x3_bf16_kernel(const BfArgs a) {
    constexpr int NT = 64 * NW;                    // threads per workgroup
    constexpr unsigned ES = IN_BF16 ? 2u : 4u;     // bytes per input element
    constexpr int WAVES_N = NW / WAVES_M;
    constexpr int TCO = 32 * WM * WAVES_M;
    // pixel tile = TH rows x TW columns (TW = 32 or 64): the WAVES_N * WN 32-pixel MFMA column blocks are laid out
    // CB = TW / 32 per image row.  The wide form reads 272-byte and writes 128 / 256-byte row pieces (fp32 / bf16
    // in, bf16 / fp32 out) instead of 144 and 64 / 128: fewer, fuller DRAM bursts on the HBM-bound big planes.
    constexpr int CB = TW / 32;
    constexpr int TH = WAVES_N * WN / CB;          // image rows per tile
    constexpr int BF_PITCH = TW + 4;               // slots per halo-tile row: columns -2 .. TW + 1
    constexpr int NPR = BF_PITCH / 2;              // pixel pairs per halo-tile row
    constexpr int PLANE = (TH + 2) * BF_PITCH;     // slots per channel group
    constexpr int XS = 2 * PLANE;                  // halo-tile slots (2 groups of 8 channels)
    constexpr int AS = 18 * TCO;                   // filter-block slots: [tap][g][TCO]
    constexpr int BUF = XS + AS + 2;               // + 2 dump slots: idle staging lanes store there (no branches)
    constexpr int DUMP = XS + AS;
    constexpr int UX = 2 * (TH + 2) * NPR;         // pixel-pair units per stage
    constexpr int NUX = (UX + NT - 1) / NT;
    constexpr int NUA = (AS + NT - 1) / NT;
    constexpr int NOPA0 = (9 - NUX) < NUA ? (9 - NUX) : NUA;
    constexpr int APG = (NUA + NOPA0 - 1) / NOPA0;      // filter pieces per staging op
    constexpr int NOPA = (NUA + APG - 1) / APG;
    constexpr int NOPS = NUX + NOPA;                    // one staging op per tap
    static_assert(NOPS <= 9, "staging schedule");
    static_assert(WN % CB == 0, "a wave's column blocks must cover whole tile rows");
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];   // 2 * BUF slots + TCO floats (this tile's bias)
    // The epilogue adds the bias from LDS.  (Read from global memory inside the epilogue it was one dependent
    // global_load -> s_waitcnt vmcnt(0) per accumulator pair, 64 times per thread: 30-48 % of the short-K layers.)
    float* sbias = reinterpret_cast<float*>(lds + 2 * BUF);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int wave_m = wid / WAVES_N, wave_n = wid % WAVES_N;

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int co_tile = b % a.n_co_tiles; b /= a.n_co_tiles;
    const int split = b % a.splitk;       b /= a.splitk;
    const int tw = b % a.tiles_w;         b /= a.tiles_w;
    const int th = b % a.tiles_h;
    const int n = b / a.tiles_h;
    const int h0 = th * TH, w0 = tw * TW, co0 = co_tile * TCO;
    const int HW = a.H * a.W;

    const int cps = (a.chunks_total + a.splitk - 1) / a.splitk;
    const int c_begin = split * cps;
    const int c_end = min(c_begin + cps, a.chunks_total);
    if (tid < TCO) sbias[tid] = (a.bias != nullptr && split == 0) ? a.bias[co0 + tid] : 0.0f;   // (visible after the prologue barrier)

    // ---- per-thread staging descriptors (the same for every chunk) ----
    unsigned offX[NUX], ldsX[NUX];
#pragma unroll
    for (int e = 0; e < NUX; ++e) {
        const int u = tid + e * NT;
        const int g = u / ((TH + 2) * NPR), rem = u - g * ((TH + 2) * NPR);
        const int row = rem / NPR, pr = rem - row * NPR;
        const int ih = h0 - 1 + row, iw = w0 - 2 + 2 * pr;
        const bool ok = u < UX && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        offX[e] = ok ? (unsigned)(g * 8 * HW + row * a.W + 2 * pr) * ES : BUF_OOB;
        ldsX[e] = u < UX ? (unsigned)(g * PLANE + row * BF_PITCH + 2 * pr) : (unsigned)DUMP;
    }
    // filter pieces: piece v of a thread is NT slots after piece v-1 = (NT / TCO) [tap][g] rows further on, a
    // wave-uniform distance that rides in the scalar offset; only the last (partial) piece needs its own mask
    static_assert(NT % TCO == 0, "filter piece stride");
    const unsigned offA0 = (unsigned)((tid / TCO) * a.Cout + (tid % TCO)) * 16u;
    const unsigned strideA = (unsigned)(NT / TCO) * (unsigned)a.Cout * 16u;
    constexpr bool A_TAIL = (AS % NT) != 0;
    const unsigned offAt = (tid + (NUA - 1) * NT < AS) ? offA0 : BUF_OOB;
    const unsigned ldsAt = (tid + (NUA - 1) * NT < AS) ? (unsigned)(XS + tid + (NUA - 1) * NT) : (unsigned)DUMP;
    // chunks past the end of this split read through an empty descriptor (every load returns 0, no per-load select)
    const char* xbase = (const char*)a.x + ((long)n * a.x_bs + (long)(h0 - 1) * a.W + (w0 - 2)) * (long)ES;
    const uint16_t* wbase = a.wp + (long)co0 * 8;

    f32x2 xr[NUX][8];
    u32x4 wr[NUA];
    auto fetch = [&](int op, int chunk) {          // op is a compile-time index
        const unsigned nrec = chunk < c_end ? BUF_OOB : 0u;
        const int ci0 = chunk * 16;
        if (op < NUX) {
            const int e = op;
            const __amdgpu_buffer_rsrc_t xrs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase), 0, nrec, 0x00020000);
            const unsigned xs = (unsigned)ci0 * (unsigned)HW * ES;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                unsigned off = offX[e];
                if (RAGGED && ci0 + (ldsX[e] >= (unsigned)PLANE && ldsX[e] < (unsigned)XS ? 8 : 0) + c >= a.Cin)
                    off = BUF_OOB;
                if constexpr (IN_BF16) xr[e][c].x = buf_load_f32(xrs, off, xs + (unsigned)c * (unsigned)HW * ES);   // 2 pixels
                else xr[e][c] = buf_load_f32x2(xrs, off, xs + (unsigned)c * (unsigned)HW * ES);
            }
        } else {
            const __amdgpu_buffer_rsrc_t wrs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wbase), 0, nrec, 0x00020000);
            const unsigned ws = (unsigned)chunk * 18u * (unsigned)a.Cout * 16u;
#pragma unroll
            for (int k = 0; k < APG; ++k) {
                const int v = (op - NUX) * APG + k;
                if (v < NUA)
                    wr[v] = buf_load_u32x4(wrs, (A_TAIL && v == NUA - 1) ? offAt : offA0, ws + (unsigned)v * strideA);
            }
        }
    };
    auto put = [&](int op, u32x4* buf) {
        if (op < NUX) {
            const int e = op;
            u32x4 s0, s1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (IN_BF16) {       // dword = (pixel 0, pixel 1) of one channel -> (channel 2q, 2q+1) of one pixel
                    const unsigned lo = __builtin_bit_cast(unsigned, xr[e][2 * q].x);
                    const unsigned hi = __builtin_bit_cast(unsigned, xr[e][2 * q + 1].x);
                    s0[q] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
                    s1[q] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
                } else {
                    s0[q] = pk_bf16(xr[e][2 * q].x, xr[e][2 * q + 1].x);
                    s1[q] = pk_bf16(xr[e][2 * q].y, xr[e][2 * q + 1].y);
                }
            }
            buf[ldsX[e]] = s0;
            buf[ldsX[e] + 1] = s1;
        } else {
#pragma unroll
            for (int k = 0; k < APG; ++k) {
                const int v = (op - NUX) * APG + k;
                if (v < NUA) buf[(A_TAIL && v == NUA - 1) ? ldsAt : (unsigned)(XS + tid + v * NT)] = wr[v];
            }
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nt][r] = 0.0f;

    // lane slots of the operand fragments: pixel column c sits at slot c + 2 of its tile row
    const int la = XS + half * TCO + wave_m * WM * 32 + j;
    const int lb = half * PLANE + j + 1;

    // prologue: chunk c_begin -> buffer 0, chunk c_begin + 1 stays in registers
#pragma unroll
    for (int op = 0; op < NOPS; ++op) fetch(op, c_begin);
#pragma unroll
    for (int op = 0; op < NOPS; ++op) put(op, lds);
#pragma unroll
    for (int op = 0; op < NOPS; ++op) fetch(op, c_begin + 1);
    __syncthreads();

    int cur = 0;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const u32x4* pa = lds + cur * BUF + la;
        const u32x4* pb = lds + cur * BUF + lb + (wave_n * WN / CB) * BF_PITCH;
        u32x4* nxt = lds + (cur ^ 1) * BUF;
        bf16x8 afA[WM], bfA[WN], afB[WM], bfB[WN];
        auto frag = [&](int t, bf16x8 (&af)[WM], bf16x8 (&bf)[WN]) {
            const int kh = t / 3, kw = t % 3;
#pragma unroll
            for (int m = 0; m < WM; ++m) af[m] = __builtin_bit_cast(bf16x8, pa[t * 2 * TCO + m * 32]);
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
                bf[nt] = __builtin_bit_cast(bf16x8, pb[((nt / CB) + kh) * BF_PITCH + (nt % CB) * 32 + kw]);
        };
        auto kstep = [&](int t, bf16x8 (&afc)[WM], bf16x8 (&bfc)[WN], bf16x8 (&afn)[WM], bf16x8 (&bfn)[WN]) {
            if (t + 1 < 9) frag(t + 1, afn, bfn);
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int nt = 0; nt < WN; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[m], bfc[nt], acc[m][nt], 0, 0, 0);
            if (t < NOPS) {
                put(t, nxt);               // chunk + 1 (fetched one stage ago) -> the other buffer
                fetch(t, chunk + 2);       // its registers re-issue their loads at once
            }
            // a wave issues in order: left alone, the staging instructions queue up behind the last MFMA and the
            // matrix pipe drains.  Interleave: after each MFMA one fragment read, two conversions, one LDS store,
            // one global load (groups without a matching instruction are skipped).
#pragma unroll
            for (int i = 0; i < WM * WN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        frag(0, afA, bfA);
        kstep(0, afA, bfA, afB, bfB);
        kstep(1, afB, bfB, afA, bfA);
        kstep(2, afA, bfA, afB, bfB);
        kstep(3, afB, bfB, afA, bfA);
        kstep(4, afA, bfA, afB, bfB);
        kstep(5, afB, bfB, afA, bfA);
        kstep(6, afA, bfA, afB, bfB);
        kstep(7, afB, bfB, afA, bfA);
        kstep(8, afA, bfA, afB, bfB);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: D row i = (r&3) + 8*(r>>2) + 4*half (output channel), column j (pixel) ----
    if constexpr (OUT_BF16) {
        // bf16 output through LDS.  Registers r, r+1 are channels i, i+1 at pixel j: neighbouring lanes swap one value
        // (DPP) so that an even lane owns channel i at pixels (j, j+1) and an odd lane channel i+1 at (j-1, j), one
        // packed dword each -- stored straight to HBM those dwords form 64-byte pieces (half cache lines), and the
        // ablation (tools/ab_bf16_probes.sh) showed the stores, not the loads or the MFMAs, bounding the <= 128-channel
        // layers (64->64 @512x512: 0.286 ms, 0.135 ms without the stores).  So the tile is first assembled in the (now
        // idle) stage buffers as [co][row][pixel] and leaves as 16-byte pieces per lane: 128 / 64 contiguous bytes per
        // (channel, row), eight such runs per instruction, a quarter of the store instructions.
        constexpr int EP = TH * TW / 2 + 16;          // dwords per channel (+16: odd / even lanes hit disjoint banks)
        static_assert(WM * 32 * EP * 4 <= 2 * BUF * 16, "epilogue tile must fit the stage buffers");
        unsigned* ep = reinterpret_cast<unsigned*>(lds);
        uint16_t* yn = (uint16_t*)a.y + (long)n * a.y_bs;
        const int odd = j & 1;
        if (a.accumulate) {
            // a second / third writer of a bf16-stored gradient (skip paths): fp32 add to the stored value, ONE rounding
            // -- straight from the accumulators (the LDS tile holds already-rounded values)
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
                const int oh = h0 + (wave_n * WN + nt) / CB;
                const int ow = w0 + ((wave_n * WN + nt) % CB) * 32 + (j & ~1);
#pragma unroll
                for (int m = 0; m < WM; ++m) {
                    unsigned olds[8];              // the eight stored pairs first, as one batch of loads (one wait, no store in between)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int co = co0 + (wave_m * WM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half + odd;
                        olds[r >> 1] = oh < a.H ? *reinterpret_cast<const unsigned*>(yn + (long)co * HW + oh * a.W + ow) : 0u;
                    }
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float own0 = acc[m][nt][r], own1 = acc[m][nt][r + 1];
                        const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own0), 0xB1, 0xf, 0xf, true));
                        const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own1), 0xB1, 0xf, 0xf, true));
                        const int co = co0 + (wave_m * WM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half + odd;
                        float lo = odd ? t1 : own0, hi = odd ? own1 : t0;
                        { const float b = sbias[co - co0]; lo += b; hi += b; }
                        if (oh < a.H) {
                            unsigned* q = reinterpret_cast<unsigned*>(yn + (long)co * HW + oh * a.W + ow);
                            const unsigned old = olds[r >> 1];
                            *q = pk_bf16(lo + bf16_lo(old), hi + bf16_hi(old));
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int pass = 0; pass < WAVES_M; ++pass) {   // 64 channels (one wave_m row of waves) per pass
            if (wave_m == pass) {
#pragma unroll
                for (int nt = 0; nt < WN; ++nt) {
                    const int row = (wave_n * WN + nt) / CB;
                    const int colp = ((wave_n * WN + nt) % CB) * 16 + (j >> 1);
#pragma unroll
                    for (int m = 0; m < WM; ++m) {
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const float own0 = acc[m][nt][r], own1 = acc[m][nt][r + 1];
                            const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own0), 0xB1, 0xf, 0xf, true));
                            const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own1), 0xB1, 0xf, 0xf, true));
                            const int cl = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half + odd;      // channel within the pass
                            float lo = odd ? t1 : own0, hi = odd ? own1 : t0;
                            { const float b = sbias[pass * WM * 32 + cl]; lo += b; hi += b; }
                            ep[cl * EP + row * (TW / 2) + colp] = pk_bf16(lo, hi);
                        }
                    }
                }
            }
            __syncthreads();
            constexpr int SEG = TW / 8;                // 16-byte pieces per tile row
            constexpr int NCH = WM * 32 * TH * SEG;    // pieces of this pass
#pragma unroll
            for (int k = 0; k < (NCH + NT - 1) / NT; ++k) {
                const int q = tid + k * NT;
                const int seg = q % SEG, row = (q / SEG) % TH, cl = q / (SEG * TH);
                const int oh = h0 + row;
                if (q < NCH && oh < a.H) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(ep + cl * EP + row * (TW / 2) + seg * 4);
                    *reinterpret_cast<u32x4*>(yn + (long)(co0 + pass * WM * 32 + cl) * HW + (long)oh * a.W + w0 + seg * 8) = v;
                }
            }
            if (pass + 1 < WAVES_M) __syncthreads();
        }
    } else {
        float* yn = (float*)a.y + (long)split * a.split_stride + (long)n * a.y_bs;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            const int oh = h0 + (wave_n * WN + nt) / CB;
            const int ow = w0 + ((wave_n * WN + nt) % CB) * 32 + j;
            const bool pok = oh < a.H;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (wave_m * WM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (pok) {
                        float v = acc[m][nt][r] + sbias[co - co0];
                        float* p = yn + (long)co * HW + oh * a.W + ow;
                        if (a.accumulate) v += *p;
                        *p = v;
                    }
                }
            }
        }
    }
}

// y[n][c][p] (+)= bias[c] + sum_s slab[s][n][c][p]   (fixed summation order, 4 pixels per thread; y fp32 or bf16)
template <bool OUT_BF16>
__global__ void bf16_splitk_reduce_kernel(const float* __restrict__ slabs, long split_stride, int splitk,
                                          void* __restrict__ yv, long y_bs, int C, int HW,
                                          const float* __restrict__ bias, int accumulate, long total4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4, chw = (long)C * HW;
        const long n = e / chw, rem = e - n * chw;
        f32x4 v = *reinterpret_cast<const f32x4*>(slabs + e);
        for (int s = 1; s < splitk; ++s) v += *reinterpret_cast<const f32x4*>(slabs + (long)s * split_stride + e);
        if (bias) v += bias[rem / HW];
        if constexpr (OUT_BF16) {
            u32x2* q = reinterpret_cast<u32x2*>((uint16_t*)yv + n * y_bs + rem);
            if (accumulate) {
                const u32x2 old = *q;
                v[0] += bf16_lo(old[0]); v[1] += bf16_hi(old[0]); v[2] += bf16_lo(old[1]); v[3] += bf16_hi(old[1]);
            }
            *q = u32x2{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])};
        } else {
            f32x4* p = reinterpret_cast<f32x4*>((float*)yv + n * y_bs + rem);
            if (accumulate) v += *p;
            *p = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ filter pack
// w[Co][Ci][9] fp32 -> uf[ceil(Ci/16)][9][2][Co][8] bf16 (forward) and ud[ceil(Co/16)][9 reversed][2][Ci][8]
// (dgrad: 180-degree rotation + channel transpose); padding channels are zero.
// A workgroup owns a 64 co x 16 ci block of one filter: its 64 x 144 floats are 64 contiguous 576-byte runs of w
// (coalesced dword loads), they pass through LDS, and both packs leave as 16-byte slots in runs of 64 (forward: 1 KB per
// (tap, channel group)) or 16 (dgrad: 256 B per (co chunk, tap, group)).  (One thread per slot gathering its eight
// floats 36 bytes apart from global memory ran the re-layout of a FuseUNet at 1.2 TB/s: 181 us at the head of every
// bf16 step, beside the two stem convolutions.)
struct BfPackDesc {
    const float* w; uint16_t* uf; uint16_t* ud;
    int Co, Ci, r0, r1;
    long block_start;
};
constexpr int BP_CO = 64, BP_CI = 16, BP_ROW = BP_CI * 9 + 1;      // LDS row: 144 floats + 1 (odd stride: conflict-free columns)

__global__ __launch_bounds__(256) void bf16_pack_multi_kernel(const BfPackDesc* __restrict__ descs, int n) {
    __shared__ float tile[BP_CO * BP_ROW];
    const long blk = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block_start <= blk) lo = mid; else hi = mid - 1;
    }
    const BfPackDesc d = descs[lo];
    const int nci = (d.Ci + BP_CI - 1) / BP_CI;
    const int lb = (int)(blk - d.block_start);
    const int co0 = (lb / nci) * BP_CO, chunk = lb % nci, ci0 = chunk * BP_CI;
    const int tid = threadIdx.x;
    const int valid = min(BP_CI, d.Ci - ci0) * 9;                  // floats of a row that exist
#pragma unroll 4
    for (int e = tid; e < BP_CO * BP_CI * 9; e += 256) {
        const int r = e / (BP_CI * 9), c = e - r * (BP_CI * 9);
        tile[r * BP_ROW + c] = (co0 + r < d.Co && c < valid) ? d.w[((long)(co0 + r) * d.Ci + ci0) * 9 + c] : 0.0f;
    }
    __syncthreads();
    if (d.uf) {                                                    // slot = (tap, group, co): 8 input channels of one tap
        for (int s = tid; s < 9 * 2 * BP_CO; s += 256) {
            const int co = s % BP_CO, g = (s / BP_CO) & 1, t = s / (2 * BP_CO);
            if (co0 + co >= d.Co) continue;
            const float* p = tile + co * BP_ROW + g * 72 + t;
            u32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = pk_bf16(p[18 * q], p[18 * q + 9]);
            reinterpret_cast<u32x4*>(d.uf)[(((long)chunk * 9 + t) * 2 + g) * d.Co + co0 + co] = v;
        }
    }
    if (d.ud) {                                                    // slot = (co chunk, tap', group, ci): 8 output channels, tap 8 - tap'
        const int nchd = (d.Co + 15) / 16;
        for (int s = tid; s < 4 * 9 * 2 * BP_CI; s += 256) {
            const int ci = s % BP_CI, g = (s / BP_CI) & 1, t = (s / (2 * BP_CI)) % 9, cq = s / (2 * BP_CI * 9);
            const int chd = co0 / 16 + cq;
            if (ci0 + ci >= d.Ci || chd >= nchd) continue;
            const float* p = tile + (cq * 16 + g * 8) * BP_ROW + ci * 9 + (8 - t);
            u32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = pk_bf16(p[(2 * q) * BP_ROW], p[(2 * q + 1) * BP_ROW]);
            reinterpret_cast<u32x4*>(d.ud)[(((long)chd * 9 + t) * 2 + g) * d.Ci + ci0 + ci] = v;
        }
    }
}

// 64-column tiles wherever the width allows (ahead by 3-10 % on every layer of the sweep from W = 64 up)
bool bf16_wide_tile(int W, int H) { return W >= 64 && W % 64 == 0; }

template <int WM, int WAVES_M, int WN, int OCC, bool RAGGED, bool IN_BF16, bool OUT_BF16, int TW, int NW>
int launch_bf16_t(BfArgs a, hipStream_t stream) {
    constexpr int WAVES_N = NW / WAVES_M, TCO = 32 * WM * WAVES_M, TH = WAVES_N * WN / (TW / 32);
    constexpr int BUF = 2 * (TH + 2) * (TW + 4) + 18 * TCO + 2;
    constexpr int LDS_BYTES = 2 * BUF * 16 + TCO * 4;
    static AideLdsOptIn lds_opt;             // per device, status checked (common.h)
    if (int rc = lds_opt.ensure([] {
            return hipFuncSetAttribute((const void*)conv3x3_bf16_kernel<WM, WAVES_M, WN, OCC, RAGGED, IN_BF16, OUT_BF16, TW, NW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        })) return rc;
    a.tiles_w = a.W / TW;
    a.tiles_h = (a.H + TH - 1) / TH;
    a.n_co_tiles = a.Cout / TCO;
    const long nb = (long)a.tiles_w * a.tiles_h * a.N * a.n_co_tiles * a.splitk;
    AIDE_LAUNCH_TIMED(AIDE_KT_BF16, AIDE_CONV_FLOPS(a.N, a.H, a.W, a.Cout, a.Cin),
                      (conv3x3_bf16_kernel<WM, WAVES_M, WN, OCC, RAGGED, IN_BF16, OUT_BF16, TW, NW>), dim3((unsigned)nb), dim3(64 * NW),
                      LDS_BYTES, stream, a);
    return aide_launch_status();
}

template <int WM, int WAVES_M, int WN, int OCC, bool RAGGED, bool IN_BF16, bool OUT_BF16, int NW>
int launch_bf16_r(const BfArgs& a, hipStream_t stream) {
    if (bf16_wide_tile(a.W, a.H)) return launch_bf16_t<WM, WAVES_M, WN, OCC, RAGGED, IN_BF16, OUT_BF16, 64, NW>(a, stream);
    return launch_bf16_t<WM, WAVES_M, WN, OCC, RAGGED, IN_BF16, OUT_BF16, 32, NW>(a, stream);
}

// storage combinations: fp32 -> fp32 (stand-alone operator), fp32 -> bf16 (forward from an fp32 tensor into a bf16 z),
// bf16 -> bf16 (forward from a bf16-stored activation), bf16 -> fp32 (dgrad from a bf16 dz); ragged channel counts
// only occur on the fp32 network inputs
template <int WM, int WAVES_M, int WN, int OCC, int NW>
int launch_bf16(const BfArgs& a, int in_bf16, int out_bf16, hipStream_t stream) {
    if (in_bf16) {                      // bf16-stored activations (forward) or conv-output gradients (dgrad)
        if (a.Cin & 15) return AIDE_ERR_ARG;
        return out_bf16 ? launch_bf16_r<WM, WAVES_M, WN, OCC, false, true, true, NW>(a, stream)
                        : launch_bf16_r<WM, WAVES_M, WN, OCC, false, true, false, NW>(a, stream);
    }
    if (out_bf16)
        return (a.Cin & 15) ? launch_bf16_r<WM, WAVES_M, WN, OCC, true, false, true, NW>(a, stream)
                            : launch_bf16_r<WM, WAVES_M, WN, OCC, false, false, true, NW>(a, stream);
    return (a.Cin & 15) ? launch_bf16_r<WM, WAVES_M, WN, OCC, true, false, false, NW>(a, stream)
                        : launch_bf16_r<WM, WAVES_M, WN, OCC, false, false, false, NW>(a, stream);
}

// variant: 0 = 32 co, 1 = 64 co (4 waves, two workgroups per CU); pixel tile 512 = 16 rows x 32 or 8 rows x 64 columns.
// (Dropped after measurement: a 4-wave 128 co x 256 px tile at one workgroup per CU, 1.0-1.3x slower on every layer, and a
// 128 co tile on 8 waves whose two co halves share the halo tile, level or behind once the epilogues stopped stalling.)
long bf16_blocks(int variant, int N, int H, int W, int Cout) {
    const int tco = variant == 1 ? 64 : 32;
    const int tw = bf16_wide_tile(W, H) ? 64 : 32, th = 512 / tw;
    return (long)(W / tw) * ((H + th - 1) / th) * N * (Cout / tco);
}
// 32 or 64 output channels per workgroup, two workgroups per CU.  (The 128-co tile on 8 waves, one workgroup per CU, was
// 4-10 % ahead on the >= 256-workgroup layers while every epilogue stalled on its bias loads; with those batched the
// two-workgroup form is level on the long-K layers and 5-10 % ahead on the short-K ones -- dgrad 64->128 @512x512 0.409 ->
// 0.371 ms, 128->128 @256x256 0.147 -> 0.135 -- whose prologue and epilogue nothing overlaps at one workgroup per CU:
// round 3, tools/bench_bf16.py with AIDE_BF16_V=1.)
int bf16_variant(int N, int H, int W, int Cout) { return Cout % 64 == 0 ? 1 : 0; }

// ------------------------------------------------------------------------------------------ weight gradient
struct BgArgs {
    const void* dz;           // fp32, or bf16 when DZ_BF16
    const void* x;            // fp32, or bf16 when X_BF16 (bf16-stored activations)
    float* slabs;
    long dz_bs, x_bs;
    int N, Co, Ci, H, W;
    int n_co_tiles, n_ci_tiles, splits, chunks_total, segs_w, bands_h;
};

// R = image rows of dz per stage (4, or 2: half the staging registers -> two workgroups per CU); NWCO = co blocks of 32
// per workgroup: 2 (64 co x 64 ci, 4 waves) or 4 (128 co x 64 ci, 8 waves = two per SIMD: the four co blocks share one
// x tile, halving the x bytes through L2 / L1 / LDS per MFMA, and a second wave per SIMD covers the other's stalls), or
// 1 (32 co x 64 ci, 2 waves, two workgroups per CU) for the 32-channel first-level layers: in the 64 x 64 tile three of four
// waves multiplied zeros there (32->32 @512x512 x8: 0.195 ms for 268 MB of operands); a wave whose 32 input channels lie
// beyond Ci skips its MFMAs, and the layer is bound by streaming dz and x once.
//
// Which operand carries the +-1 column shift of the taps (round 3): dW[kh][kw] = sum_q dz[q - (kw - 1)] x[q + (kh - 1) W], q
// running over the ALIGNED pixels of the stage -- the shift sits on dz, whose shifted fragment serves all three kh, instead
// of on x, where each of the three row windows needed its own two shifted copies: 2 instead of 6 shifted fragments (6
// instead of 18 vector instructions) and 5 instead of 7 LDS reads per k-step of 9 MFMAs.  So the dz rows are stored from
// column -1 (one aligned 32-byte window per (row, k-step): kw = 2 is the window, kw = 1 four v_alignbit, kw = 0 a register
// slice), zero where the neighbouring column lies outside the image, and the x rows are plain aligned copies with their row halo.
template <int R, int NWCO> struct GCfg {
    static constexpr int NT = 64 * (NWCO == 1 ? 2 : 2 * NWCO);   // threads: NWCO co blocks x 2 ci blocks of 32
    static constexpr int TCO = 32 * NWCO;            // output channels per workgroup
    static constexpr int DZP = R * 5 + 1;            // slots per dz channel: R rows x 5 slots (columns -1 .. 38); odd: conflict-free
    static constexpr int XP = (R + 2) * 4 + 1;       // slots per x channel: R + 2 rows x 4 slots (columns 0 .. 31)
    static constexpr int DZS = TCO * DZP;
    static constexpr int BUF = DZS + 64 * XP + 1;    // slots per stage buffer (+ 1 dump slot for idle staging lanes)
    static constexpr int NDM = TCO * R * 4 / NT;                 // dz main units per thread (TCO co x R rows x 4 slots)
    static constexpr int NDE = (TCO * R + NT - 1) / NT;          // dz edge units per thread (columns 31, 32)
    static constexpr int NX = 64 * (R + 2) * 4 / NT;             // x units per thread       (64 ci x (R + 2) rows x 4 blocks)
    static constexpr int NOPS = NDM + NDE + NX;
    static constexpr int KS = 2 * R;                 // k-steps per stage
    static_assert(TCO * R * 4 % NT == 0 && 64 * (R + 2) * 4 % NT == 0, "staging units must divide evenly");
};

template <int R, bool DZ_BF16, bool X_BF16, int NWCO>
__global__ __launch_bounds__(128 * NWCO, (R == 2 && NWCO == 2) ? 2 : 1) void conv3x3_wgrad_bf16_kernel(const BgArgs g) {
    constexpr unsigned XE = X_BF16 ? 2u : 4u;        // bytes per x element
    constexpr unsigned DE = DZ_BF16 ? 2u : 4u;       // bytes per dz element
    using G = GCfg<R, NWCO>;
    constexpr int NDM = G::NDM, NDE = G::NDE, NX = G::NX, NOPS = G::NOPS, KS = G::KS, NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];   // 2 * G::BUF slots
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int wco = wid >> 1, wci = wid & 1;

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int ci_tile = b % g.n_ci_tiles;    b /= g.n_ci_tiles;   // tiles of one pixel range are neighbours: they
    const int co_tile = b % g.n_co_tiles;                         // share its dz / x rows in the XCD's L2
    const int split = b / g.n_co_tiles;
    const int co0 = co_tile * G::TCO, ci0 = ci_tile * 64;
    const int HW = g.H * g.W;
    const int cps = (g.chunks_total + g.splits - 1) / g.splits;
    const int c_begin = split * cps;
    const int c_end = min(c_begin + cps, g.chunks_total);

    // ---- staging descriptors ----
    // dz main: unit = (co, row, slot s): columns 8s-1 .. 8s+6 of the segment -> one slot.  fp32: descriptor base = column -1,
    // the unit's 16-byte loads start at column 8s; bf16: base = column -2 (the (8s-2, 8s-1) pair is one aligned dword), the
    // unit's 16-byte load starts at column 8s.  ownL: the unit owns column -1 (zero when the segment starts the image row)
    unsigned offDM[NDM], ldsDM[NDM], ownL[NDM];
#pragma unroll
    for (int e = 0; e < NDM; ++e) {
        const int u = tid + e * NT;
        const int sl = u & 3, rr = u >> 2;
        const int row = rr % R, co = rr / R;
        offDM[e] = (co0 + co < g.Co) ? (unsigned)(co * HW + row * g.W + sl * 8 + (DZ_BF16 ? 2 : 1)) * DE : BUF_OOB;
        ldsDM[e] = (unsigned)(co * G::DZP + row * 5 + sl);
        ownL[e] = sl == 0 ? 1u : 0u;
    }
    // dz edge: unit = (co, row): columns 31, 32 -> first dword of slot 4
    unsigned offDE[NDE], ldsDE[NDE];
#pragma unroll
    for (int e = 0; e < NDE; ++e) {
        const int u = tid + e * NT;
        const int row = u % R, co = u / R;
        const bool ok = u < G::TCO * R && co0 + co < g.Co;
        offDE[e] = ok ? (unsigned)(co * HW + row * g.W + 32) * DE : BUF_OOB;   // fp32: column 31 (base = column -1); bf16: pair (30, 31) (base = column -2)
        ldsDE[e] = u < G::TCO * R ? (unsigned)(co * G::DZP + row * 5 + 4) : (unsigned)(G::BUF - 1);   // dump slot
    }
    // x: unit = (ci, tile row, 8-pixel block): 16 (bf16) or 2 x 16 (fp32) bytes -> one slot.  flagX: bit 0 = top halo row,
    // bit 1 = bottom halo row
    unsigned offX[NX], ldsX[NX], flagX[NX];
#pragma unroll
    for (int e = 0; e < NX; ++e) {
        const int u = tid + e * NT;
        const int blk = u & 3, rr = u >> 2;
        const int row = rr % (R + 2), ci = rr / (R + 2);
        offX[e] = (ci0 + ci < g.Ci) ? (unsigned)(ci * HW + row * g.W + blk * 8) * XE : BUF_OOB;
        ldsX[e] = (unsigned)(G::DZS + ci * G::XP + row * 4 + blk);
        flagX[e] = (row == 0 ? 1u : 0u) | (row == R + 1 ? 2u : 0u);
    }

    float dme[NDM];
    f32x4 dmr[NDM][2];
    float der[NDE][2];
    f32x4 xr[NX][2];
    // Per-chunk scalars (descriptor bases, halo masks) are derived ONCE per stage.  A chunk past the end of the split reads
    // through empty descriptors; image-border rows / columns are one v_cndmask per unit.
    __amdgpu_buffer_rsrc_t rsD, rsX;
    unsigned edge_mask = 0;                      // bit 0: top row outside, 1: bottom row outside
    bool left_out = false, right_ok = false;
    // chunk cursor (segment, band, image), advanced incrementally: set_chunk is called for consecutive chunks, and the
    // div / mod form of it was ~40 scalar instructions per stage of a loop that is bound by instruction issue
    int cu_seg = c_begin % g.segs_w, cu_band = (c_begin / g.segs_w) % g.bands_h, cu_n = c_begin / g.segs_w / g.bands_h;
    auto set_chunk = [&](int chunk) {
        const unsigned nrec = chunk < c_end ? BUF_OOB : 0u;
        const int seg = cu_seg, band = cu_band, n = cu_n;
        if (++cu_seg == g.segs_w) { cu_seg = 0; if (++cu_band == g.bands_h) { cu_band = 0; ++cu_n; } }
        const int h0 = band * R, w0 = seg * 32;
        const long doff = (long)n * g.dz_bs + (long)co0 * HW + (long)h0 * g.W + w0 - (DZ_BF16 ? 2 : 1);
        rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>((const char*)g.dz + doff * (long)DE), 0, nrec, 0x00020000);
        const long xoff = (long)n * g.x_bs + (long)ci0 * HW + (long)(h0 - 1) * g.W + w0;
        rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>((const char*)g.x + xoff * (long)XE), 0, nrec, 0x00020000);
        edge_mask = (h0 == 0 ? 1u : 0u) | (h0 + R >= g.H ? 2u : 0u);
        left_out = w0 == 0;
        right_ok = w0 + 32 < g.W;
    };
    // ops 0 .. NDM-1 = dz main units, then NDE dz edge units, then NX x units
    auto fetch = [&](int op) {
        if (op < NDM) {
            const unsigned off = offDM[op];
            dme[op] = buf_load_f32(rsD, ((ownL[op] && left_out) || off == BUF_OOB) ? BUF_OOB : off - 4u, 0);
            dmr[op][0] = buf_load_f32x4(rsD, off, 0);          // bf16: columns 8s .. 8s+7 as stored, dme = pair (8s-2, 8s-1)
            if constexpr (!DZ_BF16) dmr[op][1] = buf_load_f32x4(rsD, off, 16);
        } else if (op < NDM + NDE) {
            const int e = op - NDM;
            // fp32: columns 31 and 32; bf16: the pairs (30, 31) and (32, 33)
            der[e][0] = buf_load_f32(rsD, offDE[e], 0);
            der[e][1] = buf_load_f32(rsD, (right_ok && offDE[e] != BUF_OOB) ? offDE[e] + 4u : BUF_OOB, 0);
        } else {
            const int e = op - NDM - NDE;
            const unsigned off = (flagX[e] & edge_mask) ? BUF_OOB : offX[e];
            xr[e][0] = buf_load_f32x4(rsX, off, 0);                // bf16: the 8 pixels of the slot as stored
            if constexpr (!X_BF16) xr[e][1] = buf_load_f32x4(rsX, off, 16);
        }
    };
    auto put = [&](int op, u32x4* buf) {
        if (op < NDM) {
            u32x4 s;
            if constexpr (DZ_BF16) {       // shift the stored pairs by one pixel: slot = columns 8s-1 .. 8s+6
                const unsigned d = __builtin_bit_cast(unsigned, dme[op]);
                const u32x4 q = __builtin_bit_cast(u32x4, dmr[op][0]);
                s[0] = __builtin_amdgcn_alignbit(q[0], d, 16);    s[1] = __builtin_amdgcn_alignbit(q[1], q[0], 16);
                s[2] = __builtin_amdgcn_alignbit(q[2], q[1], 16); s[3] = __builtin_amdgcn_alignbit(q[3], q[2], 16);
            } else {
                s[0] = pk_bf16(dme[op], dmr[op][0].x);          s[1] = pk_bf16(dmr[op][0].y, dmr[op][0].z);
                s[2] = pk_bf16(dmr[op][0].w, dmr[op][1].x);     s[3] = pk_bf16(dmr[op][1].y, dmr[op][1].z);
            }
            buf[ldsDM[op]] = s;
        } else if (op < NDM + NDE) {
            const int e = op - NDM;
            if constexpr (DZ_BF16)         // (column 31 = high half of the first pair, column 32 = low half of the second)
                reinterpret_cast<unsigned*>(buf + ldsDE[e])[0] = __builtin_amdgcn_alignbit(
                    __builtin_bit_cast(unsigned, der[e][1]), __builtin_bit_cast(unsigned, der[e][0]), 16);
            else
                reinterpret_cast<unsigned*>(buf + ldsDE[e])[0] = pk_bf16(der[e][0], der[e][1]);
        } else {
            const int e = op - NDM - NDE;
            u32x4 s;
            if constexpr (X_BF16) {
                s = __builtin_bit_cast(u32x4, xr[e][0]);
            } else {
                s[0] = pk_bf16(xr[e][0].x, xr[e][0].y); s[1] = pk_bf16(xr[e][0].z, xr[e][0].w);
                s[2] = pk_bf16(xr[e][1].x, xr[e][1].y); s[3] = pk_bf16(xr[e][1].z, xr[e][1].w);
            }
            buf[ldsX[e]] = s;
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int la = (wco * 32 + j) * G::DZP + half;
    const int lb = G::DZS + (wci * 32 + j) * G::XP + half;
    // (NWCO == 1) a wave whose whole ci block lies beyond Ci has nothing to multiply: it only stages
    const bool mm = NWCO != 1 || ci0 + wci * 32 < g.Ci;

    set_chunk(c_begin);
#pragma unroll
    for (int op = 0; op < NOPS; ++op) fetch(op);
#pragma unroll
    for (int op = 0; op < NOPS; ++op) put(op, lds);
    set_chunk(c_begin + 1);
#pragma unroll
    for (int op = 0; op < NOPS; ++op) fetch(op);
    __syncthreads();

    constexpr int OPK = (NOPS + KS - 1) / KS;        // staging ops per k-step
    int cur = 0;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const u32x4* pa = lds + cur * G::BUF + la;
        const u32x4* pb = lds + cur * G::BUF + lb;
        u32x4* nxt = lds + (cur ^ 1) * G::BUF;
        set_chunk(chunk + 2);
        // Operand registers are double buffered by hand: the LDS reads of k-step ks + 1 are ISSUED FIRST in k-step ks
        // (one group, ahead of its MFMAs) and consumed a whole k-step later (one wave per SIMD: nothing else would hide
        // the LDS round trip, ~150 cycles, 40 times per stage).
        u32x4 dA[2], dB[2];
        bf16x8 xA[3], xB[3];
        auto load_ops = [&](int ks, u32x4 (&dw)[2], bf16x8 (&xf)[3]) {
            const int r = ks >> 1, hs = ks & 1;
            dw[0] = pa[r * 5 + hs * 2];
            dw[1] = pa[r * 5 + hs * 2 + 1];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) xf[kh] = __builtin_bit_cast(bf16x8, pb[(r + kh) * 4 + hs * 2]);
        };
        load_ops(0, dA, xA);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {             // k-step = (dz row r, 16-pixel half segment hs)
            u32x4 (&dw)[2] = (ks & 1) ? dB : dA;
            bf16x8 (&xf)[3] = (ks & 1) ? xB : xA;
            if (ks + 1 < KS) { if (ks & 1) load_ops(ks + 1, dA, xA); else load_ops(ks + 1, dB, xB); }
            const u32x4 w0v = dw[0], w1v = dw[1];    // dz columns q-1 .. q+14 of this lane's 8 aligned pixels q .. q+7
            u32x4 s1, s0;
            s1[0] = __builtin_amdgcn_alignbit(w0v[1], w0v[0], 16);       // kw = 1: dz[q .. q+7]
            s1[1] = __builtin_amdgcn_alignbit(w0v[2], w0v[1], 16);
            s1[2] = __builtin_amdgcn_alignbit(w0v[3], w0v[2], 16);
            s1[3] = __builtin_amdgcn_alignbit(w1v[0], w0v[3], 16);
            s0[0] = w0v[1]; s0[1] = w0v[2]; s0[2] = w0v[3]; s0[3] = w1v[0];   // kw = 0: dz[q+1 .. q+8]
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, w0v), a1 = __builtin_bit_cast(bf16x8, s1), a0 = __builtin_bit_cast(bf16x8, s0);
            if (mm) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, xf[kh], acc[kh * 3 + 0], 0, 0, 0);
                    acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xf[kh], acc[kh * 3 + 1], 0, 0, 0);
                    acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xf[kh], acc[kh * 3 + 2], 0, 0, 0);
                }
            }
            // staging: chunk + 1 registers -> the other buffer, then re-issue their loads for chunk + 2
#pragma unroll
            for (int k = 0; k < OPK; ++k) {
                const int op = ks * OPK + k;
                if (op < NOPS) put(op, nxt);
                if (op < NOPS) fetch(op);
            }
            // issue order: ALL operand reads of the next k-step first, then per MFMA two VALU (shifts / conversions),
            // one LDS store, one global load
            __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);       // DS reads (next k-step's operands)
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        cur ^= 1;
    }

    float* slab = g.slabs + (long)split * 9 * g.Co * g.Ci;
    const int ci = ci0 + wci * 32 + j;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < g.Co && ci < g.Ci) slab[((long)t * g.Co + co) * g.Ci + ci] = acc[t][r];
        }
    }
}

template <int R, bool DZ_BF16, bool X_BF16, int NWCO>
int launch_wgrad_bf16(BgArgs g, hipStream_t stream) {
    constexpr int LDS_BYTES = 2 * GCfg<R, NWCO>::BUF * 16;
    static AideLdsOptIn lds_opt;             // per device, status checked (common.h)
    if (int rc = lds_opt.ensure([] {
            return hipFuncSetAttribute((const void*)conv3x3_wgrad_bf16_kernel<R, DZ_BF16, X_BF16, NWCO>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        })) return rc;
    g.bands_h = g.H / R;
    g.chunks_total = g.N * g.segs_w * g.bands_h;
    g.n_co_tiles = (g.Co + 32 * NWCO - 1) / (32 * NWCO);
    const long nb = (long)g.n_co_tiles * g.n_ci_tiles * g.splits;
    AIDE_LAUNCH_TIMED(AIDE_KT_WGRAD_BF16, AIDE_CONV_FLOPS(g.N, g.H, g.W, g.Co, g.Ci),
                      (conv3x3_wgrad_bf16_kernel<R, DZ_BF16, X_BF16, NWCO>), dim3((unsigned)nb), dim3(128 * NWCO), LDS_BYTES,
                      stream, g);
    return aide_launch_status();
}

// co blocks of 32 per workgroup: the 128 x 64 tile (8 waves) pays on the large layers (>= 150 GFLOP: 1.06 vs 0.84 PFLOP/s on
// 1024->512 @64x64 x8); below that it doubles the split count for nothing.  co_blocks: the caller's choice (2 or 4; 0 = this
// rule).  (R = 2 rows per stage -- two workgroups per CU -- measured 3.5 % behind on the C5 step and is not instantiated.)
int wgrad_bf16_nwco(int N, int Co, int Ci, int H, int W, int co_blocks) {
    if (co_blocks == 1 || (co_blocks == 0 && Co <= 32)) return 1;
    if (co_blocks == 2 || Co % 128 != 0) return 2;
    if (co_blocks == 4) return 4;
    return 18.0 * N * H * W * (double)Co * Ci >= 1.5e11 ? 4 : 2;
}

}  // namespace

extern "C" {

// bf16 mode covers every layer whose width is a multiple of 32 and whose output channels are a multiple of 32
// (all FuseUNet / UNet layers at 512x512 and the >= 32-wide levels of smaller inputs); the engine keeps the fp32
// kernels for the rest.
int aide_conv3x3_bf16_supported(int Cin, int H, int W, int Cout) {
    return Cin >= 1 && H >= 1 && W >= 32 && W % 32 == 0 && Cout % 32 == 0;
}

// split factor over 16-channel chunks for layers that cannot fill 256 CUs with pixel x channel tiles
int aide_conv3x3_bf16_splitk(int N, int Cin, int H, int W, int Cout) {
    if (!aide_conv3x3_bf16_supported(Cin, H, W, Cout)) return 1;
    const long nb = bf16_blocks(bf16_variant(N, H, W, Cout), N, H, W, Cout);
    const int chunks = (Cin + 15) / 16;
    int s = 1;
    while (nb * s < 256 && s * 2 <= chunks / 4) s *= 2;
    return s;
}

size_t aide_conv3x3_bf16_pack_elems(int Cout, int Cin) {    // bf16 elements of one direction's pack
    return (size_t)((Cin + 15) / 16) * 18 * Cout * 8;
}

// workgroups of one filter in aide_conv3x3_bf16_pack_multi: 64 co x 16 ci blocks
int aide_conv3x3_bf16_pack_blocks(int Cout, int Cin) { return ((Cout + BP_CO - 1) / BP_CO) * ((Cin + BP_CI - 1) / BP_CI); }

// descs: DEVICE array of n 48-byte records {const float* w; uint16* uf (or 0); uint16* ud (or 0); int32 Co, Ci, 0, 0;
// int64 block_start}; an entry occupies aide_conv3x3_bf16_pack_blocks(Co, Ci) workgroups.
int aide_conv3x3_bf16_pack_multi(const void* descs, int n, int64_t total_blocks, hipStream_t stream) {
    if (!descs || n <= 0 || total_blocks <= 0) return AIDE_ERR_ARG;
    static_assert(sizeof(BfPackDesc) == 48, "descriptor layout");
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, bf16_pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream,
                       (const BfPackDesc*)descs, n);
    return aide_launch_status();
}

// y (+)= conv3x3(x) with bf16-packed filters u (forward pack, or the dgrad pack with Cin/Cout swapped by the
// caller).  x, y: NCHW with batch strides (elements), fp32 or -- x_bf16 / y_bf16 -- bf16 storage (the forward writing
// a bf16 z from an fp32 or bf16-stored activation, the dgrad reading a bf16 dz and writing / accumulating a bf16-stored
// activation gradient).
// ws: split-K slabs (splitk * N*Cout*H*W floats).
int aide_conv3x3_bf16_mixed(const void* x, int x_bf16, int64_t x_bs, const uint16_t* u, const float* bias, void* y,
                            int y_bf16, int64_t y_bs, int N, int Cin, int H, int W, int Cout, int accumulate, int splitk,
                            float* ws, hipStream_t stream) {
    if (!x || !u || !y || N <= 0 || !aide_conv3x3_bf16_supported(Cin, H, W, Cout)) return AIDE_ERR_ARG;
    if ((x_bf16 && (x_bs % 2)) || (y_bf16 && (y_bs % 2))) return AIDE_ERR_ARG;
    BfArgs a;
    a.x = x; a.wp = u; a.x_bs = x_bs; a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
    a.chunks_total = (Cin + 15) / 16;
    if (splitk < 1) splitk = 1;
    if (splitk > a.chunks_total) splitk = a.chunks_total;
    if (splitk > 1 && (!ws || (y_bs % 4) != 0)) return AIDE_ERR_ARG;
    a.splitk = splitk;
    int kernel_out_bf16 = y_bf16;
    if (splitk > 1) {
        a.y = ws; a.y_bs = (long)Cout * H * W; a.split_stride = (long)N * Cout * H * W;
        a.bias = nullptr; a.accumulate = 0;
        kernel_out_bf16 = 0;                     // slabs are fp32; the reduce narrows
    } else {
        a.y = y; a.y_bs = y_bs; a.split_stride = 0; a.bias = bias; a.accumulate = (accumulate == 1);
    }
    int rc;
    switch (bf16_variant(N, H, W, Cout)) {
        case 1: rc = launch_bf16<2, 1, 4, 2, 4>(a, x_bf16, kernel_out_bf16, stream); break;
        default: rc = launch_bf16<1, 1, 4, 2, 4>(a, x_bf16, kernel_out_bf16, stream); break;
    }
    if (rc != 0) return rc;
    if (splitk > 1 && accumulate != 2) {           // accumulate == 2: the caller consumes the slabs itself
        const long total4 = (long)N * Cout * H * W / 4;
        const int blocks = (int)min((total4 + 255) / 256, (long)2048);
        if (y_bf16)
            AIDE_LAUNCH_TIMED(AIDE_KT_REDUCE, 0.0, bf16_splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, stream, ws,
                               (long)N * Cout * H * W, splitk, y, (long)y_bs, Cout, H * W, bias, accumulate, total4);
        else
            AIDE_LAUNCH_TIMED(AIDE_KT_REDUCE, 0.0, bf16_splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, stream, ws,
                               (long)N * Cout * H * W, splitk, y, (long)y_bs, Cout, H * W, bias, accumulate, total4);
        rc = aide_launch_status();
    }
    return rc;
}

int aide_conv3x3_wgrad_bf16_supported(int Co, int Ci, int H, int W) {
    // any Ci: input channels beyond Ci are masked in the loaders and in the slab store (the 3-channel network inputs
    // cost a mostly idle 64-wide ci tile, but that layer is bound by streaming dz, not by the MFMAs)
    return Co % 32 == 0 && Ci >= 1 && W % 32 == 0 && H % 4 == 0;
}

int aide_conv3x3_wgrad_bf16_splits(int N, int Co, int Ci, int H, int W, int co_blocks) {
    if (aide_conv3x3_wgrad_stem_supported(Co, Ci, H, W))        // folded-tap kernel (conv3x3_wgrad_stem.hip)
        return aide_conv3x3_wgrad_stem_splits(N, H, W);
    const int tco = 32 * wgrad_bf16_nwco(N, Co, Ci, H, W, co_blocks);
    const long tiles = (long)((Co + tco - 1) / tco) * ((Ci + 63) / 64);
    const long chunks = (long)N * (H / 4) * (W / 32);
    const long target = tco == 32 ? 512 : 192;   // (the 32-co tile: two workgroups per CU, and these first-level layers are the tail of
                                                 // the backward pass -- nothing is left on the main stream to leave CUs to)
    long s = (target + tiles - 1) / tiles;       // fewer workgroups than CUs: the kernel runs on the side stream and leaves
                                                 // CUs to the dependent chain (same-box C5 step: 256 -> 439.9, 224 -> 442.9,
                                                 // 192 -> 444.4, 128 -> 428; 512 / 1024 measured 6 % / 16 % slower: twice the
                                                 // slab bytes for the fixed-order reduce, twice the prologues)
    if (s > chunks / 2) s = chunks / 2;          // at least two stages per workgroup
    if (s < 1) s = 1;
    return (int)s;
}

size_t aide_conv3x3_wgrad_bf16_ws_bytes(int N, int Co, int Ci, int H, int W, int co_blocks) {
    return (size_t)aide_conv3x3_wgrad_bf16_splits(N, Co, Ci, H, W, co_blocks) * 9 * Co * Ci * sizeof(float);
}

//   dz : [N][Co][H][W] (batch stride dz_bs; fp32, or bf16 storage when dz_bf16)   a : [N][Ci][H][W] (batch stride a_bs;
//   fp32, or bf16 storage when a_bf16)   dw : [Co][Ci][3][3] fp32
int aide_conv3x3_wgrad_bf16_mixed(const void* dz, int dz_bf16, int64_t dz_bs, const void* a, int a_bf16, int64_t a_bs,
                                  float* dw, int N, int Co, int Ci, int H, int W, float* ws, int co_blocks, void* queue,
                                  hipStream_t stream) {
    if (!dz || !a || !dw || !ws || N <= 0 || !aide_conv3x3_wgrad_bf16_supported(Co, Ci, H, W)) return AIDE_ERR_ARG;
    if (co_blocks != 0 && co_blocks != 1 && co_blocks != 2 && co_blocks != 4) return AIDE_ERR_ARG;
    if ((dz_bs % (dz_bf16 ? 8 : 4)) || (a_bs % (a_bf16 ? 8 : 4))) return AIDE_ERR_ARG;
    if (a_bf16 && aide_conv3x3_wgrad_stem_supported(Co, Ci, H, W)) return AIDE_ERR_ARG;   // stems read the fp32 images
    if (aide_conv3x3_wgrad_stem_supported(Co, Ci, H, W))                        // Ci <= 3
        return aide_conv3x3_wgrad_stem(dz, dz_bf16, dz_bs, (const float*)a, a_bs, dw, N, Co, Ci, H, W, ws,
                                       aide_conv3x3_wgrad_bf16_splits(N, Co, Ci, H, W, co_blocks), 1, queue, stream);
    BgArgs g;
    g.dz = dz; g.x = a; g.slabs = ws; g.dz_bs = dz_bs; g.x_bs = a_bs;
    g.N = N; g.Co = Co; g.Ci = Ci; g.H = H; g.W = W;
    g.n_co_tiles = (Co + 63) / 64; g.n_ci_tiles = (Ci + 63) / 64;
    g.segs_w = W / 32;
    g.splits = aide_conv3x3_wgrad_bf16_splits(N, Co, Ci, H, W, co_blocks);
    int rc;
#define AIDE_WG(RR, NW) (dz_bf16 ? (a_bf16 ? launch_wgrad_bf16<RR, true, true, NW>(g, stream) : launch_wgrad_bf16<RR, true, false, NW>(g, stream)) \
                                 : (a_bf16 ? launch_wgrad_bf16<RR, false, true, NW>(g, stream) : launch_wgrad_bf16<RR, false, false, NW>(g, stream)))
    const int nwco = wgrad_bf16_nwco(N, Co, Ci, H, W, co_blocks);
    if (nwco == 4) rc = AIDE_WG(4, 4);
    else if (nwco == 1) rc = AIDE_WG(4, 1);
    else rc = AIDE_WG(4, 2);
#undef AIDE_WG
    if (rc != 0) return rc;
    return aide_wgrad_reduce_launch(ws, g.splits, Co, Ci, dw, queue, stream);
}

}  // extern "C"
