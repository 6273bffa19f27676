// Training-mode BatchNorm2d + ReLU, forward and backward, as HBM-bound streaming kernels (gfx950).
//
// Replaces (reference): nn.BatchNorm2d (train) + nn.ReLU at models_twomodalinputs/netblocks.py:
// 25,27,28,18 / UNet.py:20,22,23,13 and their autograd backward.
//
//   forward : mean_c, var_c (biased) over (N,H,W);  a = relu(gamma*(z-mean)*rstd + beta)
//             running_mean <- (1-m) rm + m mean ; running_var <- (1-m) rv + m var n/(n-1) ; nbt += 1
//   backward: dy = dA * (a > 0);  dbeta = sum dy;  dgamma = sum dy*xhat
//             dz = gamma*rstd*(dy - dbeta/n - xhat*dgamma/n);  dbias_conv = sum dz (== 0 up to rounding)
//
// All tensors are NCHW planes with an explicit batch stride, so a tensor may be a channel slice of a
// larger (concatenation) buffer; this is how torch.cat is eliminated (fuseunet.py:49, netblocks.py:145).
// Reductions: fp64 accumulation per thread -> fixed-order tree -> per-(channel,split) partials ->
// finalize in channel order.  No atomics: results are bit-reproducible run to run.
#include "common.h"
#include <type_traits>

namespace {

// planes whose size is not a multiple of 4 (e.g. the 3x2 bottom level of a 48x32 input) take the
// scalar instantiation V = 1; everything on the BASELINE shapes runs the 16-byte V = 4 form
// V = 8: the bf16-stored tensors of the precision='bf16' mode move 16 bytes per lane as well (8 values; an fp32 partner
// tensor moves two 16-byte pieces) -- with V = 4 their loads were 8 bytes per lane and the two-stream reduce ran at 3.1 TB/s
template <int V> __device__ __forceinline__ void ldv(const float* p, float (&o)[V]) {
    if constexpr (V == 8) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p), w = *reinterpret_cast<const f32x4*>(p + 4);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; o[4] = w[0]; o[5] = w[1]; o[6] = w[2]; o[7] = w[3];
    } else if constexpr (V == 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(p); o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
    else { o[0] = p[0]; }
}
template <int V> __device__ __forceinline__ void stv(float* p, const float (&o)[V]) {
    if constexpr (V == 8) {
        *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{o[4], o[5], o[6], o[7]};
    } else if constexpr (V == 4) { *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]}; }
    else { p[0] = o[0]; }
}
// bf16 storage (precision='bf16': conv outputs z and their gradients dz live in HBM as bf16, the arithmetic
// here stays fp32/fp64): widening is exact, narrowing is round-to-nearest-even (v_cvt_pk_bf16_f32)
typedef uint16_t bf16_t;
template <int V> __device__ __forceinline__ void ldv(const bf16_t* p, float (&o)[V]) {
    if constexpr (V == 8) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[2 * k] = __builtin_bit_cast(float, v[k] << 16); o[2 * k + 1] = __builtin_bit_cast(float, v[k] & 0xffff0000u);
        }
    } else if constexpr (V == 4) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(p);
        o[0] = __builtin_bit_cast(float, v[0] << 16); o[1] = __builtin_bit_cast(float, v[0] & 0xffff0000u);
        o[2] = __builtin_bit_cast(float, v[1] << 16); o[3] = __builtin_bit_cast(float, v[1] & 0xffff0000u);
    } else { o[0] = __builtin_bit_cast(float, (unsigned)p[0] << 16); }
}
__device__ __forceinline__ unsigned bn_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
template <int V> __device__ __forceinline__ void stv(bf16_t* p, const float (&o)[V]) {
    if constexpr (V == 8) {
        *reinterpret_cast<u32x4*>(p) = u32x4{bn_pk_bf16(o[0], o[1]), bn_pk_bf16(o[2], o[3]), bn_pk_bf16(o[4], o[5]), bn_pk_bf16(o[6], o[7])};
    } else if constexpr (V == 4) { *reinterpret_cast<u32x2*>(p) = u32x2{bn_pk_bf16(o[0], o[1]), bn_pk_bf16(o[2], o[3])}; }
    else { p[0] = (bf16_t)(bn_pk_bf16(o[0], 0.f) & 0xffffu); }
}

// Workspace (aide_bn_ws_bytes): one block of BN_WS_STRIDE doubles per channel, at the same address whatever the C of the call
// (an engine shares one workspace between its layers): [0] epoch (int64: generations completed on this channel), [1] broadcast
// values (two floats), [2] broadcast generation (int64), then BN_SLOTS slots of {v0, v1, v2, generation (int64)}.
constexpr int BN_SLOTS = 256, BN_WS_HDR = 8, BN_WS_STRIDE = BN_WS_HDR + 4 * BN_SLOTS;
__device__ __forceinline__ double* ws_chan(double* ws, int c) { return ws + (long)c * BN_WS_STRIDE; }
__device__ __forceinline__ const double* ws_chan(const double* ws, int c) { return ws + (long)c * BN_WS_STRIDE; }
__device__ __forceinline__ double* ws_slot(double* ws, int c, int slot) { return ws_chan(ws, c) + BN_WS_HDR + 4 * slot; }
__device__ __forceinline__ const double* ws_slot(const double* ws, int c, int slot) { return ws_chan(ws, c) + BN_WS_HDR + 4 * slot; }

// Cursor over the V-element units [beg, end) of one channel of an [N][C][HW] tensor, stride 256 units per step:
// (image, unit inside the plane) advance incrementally -- a 64-bit `i / hw4` per step cost more than the arithmetic
// of the step (the reductions ran at 2.8 TB/s, the element-wise pass with the same loop at 5.2).
struct PlaneCursor {
    int n, p, left;                   // image, unit inside the plane, units left for this thread (stride 256)
    __device__ __forceinline__ PlaneCursor(long beg, long end, int hw4) {
        const long i = beg + threadIdx.x;
        n = (int)(i / hw4);
        p = (int)(i - (long)n * hw4);
        left = i < end ? (int)((end - i + 255) >> 8) : 0;
    }
    __device__ __forceinline__ void next(int hw4) {
        p += 256;
        while (p >= hw4) { p -= hw4; ++n; }
        --left;
    }
};

// ---------------------------------------------------------------- statistics (sum, sum of squares)
// Split-K convolutions of a training forward leave their partial results as slabs [split][N][C][HW] (fp32): the
// BatchNorm that follows sums them itself (fixed order s = 0, 1, ...; + conv bias), stores z for the backward pass and
// takes its statistics from the stored value -- the separate split-reduce launch and one pass over z disappear.
struct SlabSrc {
    const float* slabs;               // nullptr: z is the input
    const float* bias;
    long split_stride, slab_bs;       // elements between splits / between images inside a slab
    int splitk;
};
// The gradient of a 2x2 max-pooling of this layer's activation, folded into its BatchNorm backward (round 6): the layer's dA is
// dA (the other readers' gradient, e.g. the decoder's skip path) + the pooled gradient routed to the arg-max of every window -- the
// activation is recomputed from z (the same fmaf + max as the forward pass: bit-identical), first maximum in row-major window order as
// nn.MaxPool2d's backward picks it.  pdy: [N][C][H/2][W/2] fp32 at this layer's channel 0, batch stride pdy_bs.
struct PoolSrc {
    const float* pdy;
    long pdy_bs;
    int W;
    // ... or (HEAD) the layer's activation fed the 1x1 head: dA = sum_k w[k][c] * dlogits[n][k][p], computed on the way in (the head's
    // data-gradient pass -- 67 MB written and read back per C2 step -- never runs); same products and order as head_dgrad_kernel
    const float* hw;          // [K][C] head weights
    int K;
};
template <int V> __device__ __forceinline__ void slab_sum(const SlabSrc& sl, long off, int c, float (&o)[V]) {
    ldv<V>(sl.slabs + off, o);
    for (int s = 1; s < sl.splitk; ++s) {
        float t[V];
        ldv<V>(sl.slabs + (long)s * sl.split_stride + off, t);
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] += t[k];
    }
    if (sl.bias) {
        const float b = sl.bias[c];
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] += b;
    }
}
// value as the backward pass will read it back from a ZT-typed z
__device__ __forceinline__ float as_stored(float v, const float*) { return v; }
__device__ __forceinline__ float as_stored(float v, const bf16_t*) { return __builtin_bit_cast(float, bn_pk_bf16(v, 0.f) << 16); }

template <int V, typename ZT, bool SLABS>
__global__ __launch_bounds__(256) void bn_stats_kernel(const ZT* __restrict__ z, long z_bs, int N, int C,
                                                       int HW, int splits, double* __restrict__ partials,
                                                       const SlabSrc sl) {
    __shared__ double sm[2 * 4];
    const int c = blockIdx.x % C, s = blockIdx.x / C;
    // blockIdx.y: group of a stacked batch (aide_bn_train_fwd_groups; N = images per group, a plain launch has one group)
    const int n0 = blockIdx.y * N;
    const int slot0 = blockIdx.y * splits;         // partials: slot group * splits + s of the channel's workspace block
    const long total4 = (long)N * HW / V;
    const long per = (total4 + splits - 1) / splits;
    const long beg = s * per, end = min(beg + per, total4);
    const int hw4 = HW / V;
    double acc[2] = {0.0, 0.0};
    const ZT* zc = z + (long)n0 * z_bs + (long)c * HW;
#pragma unroll 2
    for (PlaneCursor cur(beg, end, hw4); cur.left > 0; cur.next(hw4)) {
        float v[V];
        if constexpr (SLABS) {
            slab_sum<V>(sl, (long)(n0 + cur.n) * sl.slab_bs + (long)c * HW + cur.p * V, c, v);
            stv<V>(const_cast<ZT*>(zc) + (long)cur.n * z_bs + cur.p * V, v);
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] = as_stored(v[k], (const ZT*)nullptr);
        } else {
            ldv<V>(zc + (long)cur.n * z_bs + cur.p * V, v);
        }
        float s1 = 0.0f;                           // the V-element group sum in fp32, promoted once (squares stay fp64:
#pragma unroll                                     // E[z^2] - mean^2 cancels)
        for (int k = 0; k < V; ++k) {
            const double d = (double)v[k];
            s1 += v[k];
            acc[1] = fma(d, d, acc[1]);
        }
        acc[0] += (double)s1;
    }
    block_sum_d<2>(acc, sm);
    if (threadIdx.x == 0) {
        ws_slot(partials, c, slot0 + s)[0] = acc[0];
        ws_slot(partials, c, slot0 + s)[1] = acc[1];
    }
}

// ---- statistics emitted by the convolution itself.  A forward conv launch that finds a sink armed writes, per output
// channel and workgroup tile, the fp32 sum and sum of squares of its PRE-BIAS outputs: parts[c][nparts][2].  The apply
// kernel below then derives mean / variance from them (fp64 across the parts, fixed order) -- the statistics pass over z
// (bn_stats_kernel: one more read of the tensor and a launch) disappears.  Pre-bias sums keep E[y^2] - E[y]^2 benign: the
// conv part of a BatchNorm input is close to zero-mean, the bias only shifts the mean.
// Training forward, second pass: every block re-derives its channel's (scale, shift) from the fp64
// partials (one wave, <= 64 splits) and applies a = relu(z*scale + shift); the (x == 0, n == 0) block
// of each channel also publishes mean / rstd / scale / shift and updates the running statistics.
// (Folding the finalize step in here removes one ~5 us launch per BatchNorm.)
template <int V, typename ZT, typename AT>
__global__ __launch_bounds__(256) void bn_train_apply_kernel(
    const ZT* __restrict__ z, long z_bs, AT* __restrict__ a, long a_bs, int C, int HW,
    const double* __restrict__ partials, int splits, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
    float* __restrict__ running_var, long long* __restrict__ nbt, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, float* __restrict__ scale_out, float* __restrict__ shift_out, int relu,
    const float* __restrict__ fparts, int nparts, const float* __restrict__ cbias, int pstride, int ng, int groups) {
    __shared__ float coef[2];
    // grid = (planes, chunks of a plane): the plane index rides on gridDim.x (limit 2^31 - 1) -- a stacked batch of 4 groups x 32
    // images at C = 512 is 65 536 planes, one more than gridDim.y may hold
    const int plane = blockIdx.x;                 // n*C + c
    const int n = plane / C, c = plane - n * C;
    // Stacked batch (aide_bn_train_fwd_groups): `groups` runs of ng images, each normalised with ITS statistics (`count`
    // = elements per channel of one group; partials [group][C][splits][2], the conv-epilogue entries of group g at
    // [c][g * nparts ..]).  The publishing block of a channel walks the groups in order -- the running statistics after
    // `groups` sequential forwards, bit for bit -- and leaves the last group's mean / rstd / scale / shift.
    const int mine = n / ng;
    const bool publisher = blockIdx.y == 0 && n == 0;
    if (threadIdx.x < 64) {
        for (int gi = publisher ? 0 : mine; gi <= (publisher ? groups - 1 : mine); ++gi) {
            double s = 0.0, ss = 0.0;
            if (fparts) {                             // statistics from the convolution's epilogue: [c][pstride][2], pre-bias
                for (int i = threadIdx.x; i < nparts; i += 64) {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(fparts + ((long)c * pstride + (long)gi * nparts + i) * 2);
                    s += (double)v[0];
                    ss += (double)v[1];
                }
            } else if ((int)threadIdx.x < splits) {
                s = ws_slot(partials, c, gi * splits + threadIdx.x)[0];
                ss = ws_slot(partials, c, gi * splits + threadIdx.x)[1];
            }
            s = wave_sum_d(s);
            ss = wave_sum_d(ss);
            if (threadIdx.x == 0) {
                double mean = s / count;
                double var = ss / count - mean * mean;
                if (fparts && cbias) mean += (double)cbias[c];           // the sums are of z - bias
                if (var < 0.0) var = 0.0;
                const float rstd = (float)(1.0 / sqrt(var + (double)eps));
                const float g = gamma ? gamma[c] : 1.0f, bb = beta ? beta[c] : 0.0f;
                const float sc = g * rstd, sh = bb - (float)mean * sc;
                if (gi == mine) { coef[0] = sc; coef[1] = sh; }
                if (publisher) {
                    if (gi == groups - 1) { mean_out[c] = (float)mean; rstd_out[c] = rstd; scale_out[c] = sc; shift_out[c] = sh; }
                    if (running_mean) {
                        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
                        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
                    }
                    if (c == 0 && nbt) *nbt += 1;
                }
            }
        }
    }
    __syncthreads();
    const float sc = coef[0], sh = coef[1];
    const ZT* zp = z + (long)n * z_bs + (long)c * HW;
    AT* ap = a + (long)n * a_bs + (long)c * HW;
    const int hw4 = HW / V;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < hw4; i += gridDim.y * 256) {
        float v[V];
        ldv<V>(zp + i * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float y = fmaf(v[k], sc, sh);
            v[k] = relu ? fmaxf(y, 0.0f) : y;
        }
        stv<V>(ap + i * V, v);
    }
}

// Statistics -> coefficients only (no pass over z): the conv epilogue's partial sums of a STACKED batch become, per group
// and channel, the (scale, shift) pair of tab[group][tab_c][2] at channel offset tab_c0 -- the table the consumer
// convolution's loader applies (conv3x3_wino4_kernel<.., BNIN>) -- and the running statistics move once per group, in order
// (the arithmetic of bn_train_apply_kernel's publishing block, bit for bit).  One wave per channel.
__global__ __launch_bounds__(256) void bn_finalize_groups_kernel(
    const float* __restrict__ fparts, int nparts, int pstride, const float* __restrict__ cbias, double count, int groups,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, float* __restrict__ scale_out, float* __restrict__ shift_out,
    float* __restrict__ tab, int tab_c, int tab_c0) {
    // four waves per channel: wave w sums the entries of groups w, w + 4, ... (each group exactly as the one-wave form did: lane i
    // takes entries i, i + 64, ...), then thread 0 walks the groups in order.  (One wave per channel walking the groups took 45 us
    // per launch in the co-teaching step: 64 .. 512 waves on the chip, each a chain of dependent loads.)
    __shared__ double sums[32][2];
    const int c = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int g0 = 0; g0 < groups; g0 += 32) {
        const int ng = min(32, groups - g0);
        for (int gi = w; gi < ng; gi += 4) {
            double s = 0.0, ss = 0.0;
            for (int i = lane; i < nparts; i += 64) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(fparts + ((long)c * pstride + (long)(g0 + gi) * nparts + i) * 2);
                s += (double)v[0];
                ss += (double)v[1];
            }
            s = wave_sum_d(s);
            ss = wave_sum_d(ss);
            if (lane == 0) { sums[gi][0] = s; sums[gi][1] = ss; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int gi = 0; gi < ng; ++gi) {
                double mean = sums[gi][0] / count;
                double var = sums[gi][1] / count - mean * mean;
                if (cbias) mean += (double)cbias[c];                     // the sums are of z - bias
                if (var < 0.0) var = 0.0;
                const float rstd = (float)(1.0 / sqrt(var + (double)eps));
                const float g = gamma ? gamma[c] : 1.0f, bb = beta ? beta[c] : 0.0f;
                const float sc = g * rstd, sh = bb - (float)mean * sc;
                *reinterpret_cast<f32x2*>(tab + ((long)(g0 + gi) * tab_c + tab_c0 + c) * 2) = f32x2{sc, sh};
                if (g0 + gi == groups - 1) { mean_out[c] = (float)mean; rstd_out[c] = rstd; scale_out[c] = sc; shift_out[c] = sh; }
                if (running_mean) {
                    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
                    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
                }
                if (c == 0 && nbt) *nbt += 1;
            }
        }
        __syncthreads();
    }
}

// eval mode with the BatchNorm folded into the convolution's epilogue: a = relu(acc * scale + fbias), fbias = conv_bias * scale + shift
__global__ void bn_eval_fold_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                    const float* __restrict__ conv_bias, float* __restrict__ scale_out,
                                    float* __restrict__ shift_out, float* __restrict__ fbias_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = 1.0f / sqrtf(rv[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.0f) * rstd;
    const float sh = (beta ? beta[c] : 0.0f) - rm[c] * sc;
    scale_out[c] = sc;
    shift_out[c] = sh;
    fbias_out[c] = __builtin_fmaf(conv_bias ? conv_bias[c] : 0.0f, sc, sh);
}

// ---------------------------------------------------------------- a = relu(z*scale + shift)
template <int V, typename ZT, typename AT>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const ZT* __restrict__ z, long z_bs,
                                                            AT* __restrict__ a, long a_bs, int C, int HW,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu) {
    const int plane = blockIdx.x;                 // n*C + c (grid = (planes, chunks): see bn_train_apply_kernel)
    const int n = plane / C, c = plane - n * C;
    const float sc = scale[c], sh = shift[c];
    const ZT* zp = z + (long)n * z_bs + (long)c * HW;
    AT* ap = a + (long)n * a_bs + (long)c * HW;
    const int hw4 = HW / V;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < hw4; i += gridDim.y * 256) {
        float v[V];
        ldv<V>(zp + i * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float y = fmaf(v[k], sc, sh);
            v[k] = relu ? fmaxf(y, 0.0f) : y;
        }
        stv<V>(ap + i * V, v);
    }
}

// ---------------------------------------------------------------- backward reductions
// partials[c][s] = { sum dy, sum dy*xhat, sum xhat } with dy = dA * (z*scale+shift > 0)
template <int V, typename ZT, typename GT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const GT* __restrict__ dA, long d_bs,
                                                            const ZT* __restrict__ z, long z_bs, int N,
                                                            int C, int HW, int splits,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu,
                                                            double* __restrict__ partials) {
    __shared__ double sm[3 * 4];
    const int c = blockIdx.x % C, s = blockIdx.x / C;
    const long total4 = (long)N * HW / V;
    const long per = (total4 + splits - 1) / splits;
    const long beg = s * per, end = min(beg + per, total4);
    const int hw4 = HW / V;
    const float mu = mean[c], rs = rstd[c], sc = scale[c], sh = shift[c];
    double acc[3] = {0.0, 0.0, 0.0};
    const ZT* zc = z + (long)c * HW;
    const GT* dc = dA + (long)c * HW;
#pragma unroll 2
    for (PlaneCursor cur(beg, end, hw4); cur.left > 0; cur.next(hw4)) {
        float zv[V], dv[V];
        ldv<V>(zc + (long)cur.n * z_bs + cur.p * V, zv);
        ldv<V>(dc + (long)cur.n * d_bs + cur.p * V, dv);
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;     // V-element group sums in fp32, promoted once per group
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const bool on = !relu || fmaf(zv[k], sc, sh) > 0.0f;
            const float dy = on ? dv[k] : 0.0f;
            const float xh = (zv[k] - mu) * rs;
            t0 += dy;
            t1 = fmaf(dy, xh, t1);
            t2 += xh;
        }
        acc[0] += (double)t0;
        acc[1] += (double)t1;
        acc[2] += (double)t2;
    }
    block_sum_d<3>(acc, sm);
    if (threadIdx.x == 0) {
        ws_slot(partials, c, s)[0] = acc[0];
        ws_slot(partials, c, s)[1] = acc[1];
        ws_slot(partials, c, s)[2] = acc[2];
    }
}

// dz = scale*(dy - c0 - xhat*c1) with c0 = sum(dy)/n, c1 = sum(dy*xhat)/n re-derived per block from the
// fp64 partials; the split-0 block of each channel also writes dgamma, dbeta and the conv-bias gradient.
// The latter is sum(dz), which is zero in exact arithmetic; it is evaluated from the same sums,
//   sum dz = scale * ((sum dy - n c0) - c1 sum xhat),
// i.e. as the rounding residue it is (the reference's autograd value is the same kind of ~1e-8 noise).
template <int V, typename ZT, typename DT, typename GT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const GT* __restrict__ dA, long d_bs,
                                                           const ZT* __restrict__ z, long z_bs,
                                                           DT* __restrict__ dz, long dz_bs, int N, int C,
                                                           int HW, int splits, double count,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu,
                                                           const double* __restrict__ partials,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ dbias) {
    __shared__ float coef[2];
    const int c = blockIdx.x % C, s = blockIdx.x / C;
    const float mu = mean[c], rs = rstd[c], sc = scale[c], sh = shift[c];
    if (threadIdx.x < 64) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        if ((int)threadIdx.x < splits) {
            a0 = ws_slot(partials, c, threadIdx.x)[0];
            a1 = ws_slot(partials, c, threadIdx.x)[1];
            a2 = ws_slot(partials, c, threadIdx.x)[2];
        }
        a0 = wave_sum_d(a0); a1 = wave_sum_d(a1); a2 = wave_sum_d(a2);
        if (threadIdx.x == 0) {
            const float c0f = (float)(a0 / count), c1f = (float)(a1 / count);
            coef[0] = c0f; coef[1] = c1f;
            if (s == 0) {
                if (dbeta) dbeta[c] = (float)a0;
                if (dgamma) dgamma[c] = (float)a1;
                if (dbias) dbias[c] = (float)((double)sc * ((a0 - count * (double)c0f) - (double)c1f * a2));
            }
        }
    }
    __syncthreads();
    const float c0 = coef[0], c1 = coef[1];
    const long total4 = (long)N * HW / V;
    const long per = (total4 + splits - 1) / splits;
    const long beg = s * per, end = min(beg + per, total4);
    const int hw4 = HW / V;
    const ZT* zc = z + (long)c * HW;
    const GT* dc = dA + (long)c * HW;
    DT* oc = dz + (long)c * HW;
#pragma unroll 2
    for (PlaneCursor cur(beg, end, hw4); cur.left > 0; cur.next(hw4)) {
        float zv[V], dv[V], o[V];
        ldv<V>(zc + (long)cur.n * z_bs + cur.p * V, zv);
        ldv<V>(dc + (long)cur.n * d_bs + cur.p * V, dv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const bool on = !relu || fmaf(zv[k], sc, sh) > 0.0f;
            const float dy = on ? dv[k] : 0.0f;
            const float xh = (zv[k] - mu) * rs;
            o[k] = sc * (dy - c0 - xh * c1);
        }
        stv<V>(oc + (long)cur.n * dz_bs + cur.p * V, o);
    }
}

// ---------------------------------------------------------------- one pass, several workgroups per channel
// Every channel is owned by S workgroups (blockIdx = c * S + s): each reads its share of the channel ONCE into registers
// (Q units of V values per thread, all loads in flight at once) and reduces it to fp64 partial sums.  Every workgroup but one hands
// its partials to the channel's LEADER (the last one, coop_leader) and waits for the coefficients; the leader sums the S partials in slot order
// (bit-reproducible: no dependence on arrival order), finishes the statistics and broadcasts the two coefficients every
// workgroup needs; all of them normalise from their registers.  Forward 8 B / element instead of 12 (statistics pass + apply
// pass), backward 12 instead of 20, one launch per direction -- and S x the workgroups of the round-5
// one-workgroup-per-channel kernels, which left half of the chip idle at C = 128 and ran at 0.2 of the HBM roofline.
//
// The exchange uses no read-modify-write atomics and no cache maintenance (both measured: a counter per channel cost 0.5-1 us
// per arrival -- same-address device-scope atomics serialise at the memory side -- and an agent-scope release / acquire pair
// per workgroup writes back / invalidates the XCD's whole L2: the C = 32 @256x256 forward took 106 us and 262 us).  Partials,
// generation flags and the broadcast move with agent-scope RELAXED atomic stores and loads (sc1: performed at the
// device-coherent level, past the per-XCD L2s); "payload, then flag" is the order of ONE thread's own stores with
// s_waitcnt vmcnt(0) between them (a store is counted until the level it was sent to acknowledges it), and a reader loads the
// payload after the flag's value came back.  Generations: every channel has an epoch in the workspace that only its leader
// advances (by `groups` per launch); a flag or broadcast is valid when it carries exactly the generation the launch expects,
// so nothing is ever reset and stale entries of earlier launches (any S, any C) can never match.  The workspace must be
// zero-filled once and used by one stream at a time.
//
// The waits cannot deadlock: the partners of a workgroup are its index neighbours (< S apart), the hardware dispatches
// workgroups in index order, so the oldest waiting group's missing members are next in line on their XCDs and only workgroups
// of older (never waiting) groups are ahead of them.  Several such kernels at once (the two lanes of a forward pass, the two
// networks of the co-teaching step): the waiting workgroups of a kernel all belong to its ONE partly dispatched group (< S of them), so
// k concurrent kernels hold fewer than k * S workgroup slots while they wait -- with S <= 130 on every shape the size rule below admits
// (<= 36 on the BASELINE shapes) and >= 1024 slots on the chip (4 workgroups per CU at <= 128 VGPRs) there is always room for the
// missing members once the other kernels on the CUs (which never wait for these) retire.  The spins are bounded anyway: a wait that
// runs out poisons the result with NaN instead of hanging the queue.
constexpr int BN_MAX_S = BN_SLOTS;
// which workgroup of a channel leads: the LAST one (s = S - 1) is dispatched last and tends to finish its loads last -- as the leader it
// finds the other partials already posted and the broadcast leaves one poll round trip earlier than with s = 0
constexpr bool BN_LEADER_LAST = true;
__device__ __forceinline__ int coop_leader(int S) { return BN_LEADER_LAST ? S - 1 : 0; }
constexpr int BN_SPIN_LIMIT = 1 << 21;

__device__ __forceinline__ void coop_store(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coop_store_i(double* p, long long v) {
    __hip_atomic_store(reinterpret_cast<long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double coop_load(const double* p) {
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ long long coop_load_i(const double* p) {
    return __hip_atomic_load(reinterpret_cast<const long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coop_fence() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ bool coop_wait(const double* flag, long long gen) {
    for (int spins = 0; spins < BN_SPIN_LIMIT; ++spins) {
        if (coop_load_i(flag) == gen) {
            asm volatile("" ::: "memory");
            return true;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    return false;
}
// One generation of the exchange.  In: thread 0 of every workgroup holds its NV partial sums in acc.  Out (leader): thread 0
// holds the channel totals in acc and returns true from is_leader(); the leader then calls coop_publish(v0, v1), the others
// coop_receive(v0, v1) (thread 0 each).  `ok` turns false when a bounded wait ran out.
template <int NV>
__device__ __forceinline__ void coop_gather(double* ws, int c, int s, int S, long long gen, double (&acc)[NV], double* sm, int* okf) {
    if (s != coop_leader(S)) {
        if (threadIdx.x == 0) {
            double* slot = ws_slot(ws, c, s);
#pragma unroll
            for (int i = 0; i < NV; ++i) coop_store(slot + i, acc[i]);
            coop_fence();
            coop_store_i(slot + 3, gen);
        }
        return;
    }
    // leader: thread t collects slot t (its own partials stand in for slot 0); totals in slot order
    __syncthreads();                            // (thread 0 is done with sm from the block sum that produced acc)
    double t[NV];
    bool ok = true;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) t[i] = acc[i];
    } else if ((int)threadIdx.x < S) {
        // (thread 0 holds the leader's own partials; thread t > 0 collects the t-th of the other workgroups, in index order)
        const int other = BN_LEADER_LAST ? (int)threadIdx.x - 1 : (int)threadIdx.x;
        const double* slot = ws_slot(ws, c, other);
        ok = coop_wait(slot + 3, gen);
#pragma unroll
        for (int i = 0; i < NV; ++i) t[i] = coop_load(slot + i);
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) t[i] = 0.0;
    }
    if (!ok) *okf = 0;                          // (benign race: every writer stores 0)
    block_sum_d<NV>(t, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = t[i];
    }
}
__device__ __forceinline__ void coop_publish(double* ws, int c, long long gen, float v0, float v1) {
    double* h = ws_chan(ws, c);
    coop_store(h + 1, __builtin_bit_cast(double, f32x2{v0, v1}));
    coop_fence();
    coop_store_i(h + 2, gen);
}
__device__ __forceinline__ bool coop_receive(const double* ws, int c, long long gen, float& v0, float& v1) {
    const double* h = ws_chan(ws, c);
    const bool ok = coop_wait(h + 2, gen);
    const f32x2 v = __builtin_bit_cast(f32x2, coop_load(h + 1));
    v0 = v[0]; v1 = v[1];
    return ok;
}

// Q units of V values of slab-resident data: all Q loads of a slab in flight together, slabs in order s = 0, 1, ...
template <int V, int Q>
__device__ __forceinline__ void slab_sum_q(const SlabSrc& sl, const long (&off)[Q], const bool (&on)[Q], int c, float (&o)[Q][V]) {
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        if (on[k]) ldv<V>(sl.slabs + off[k], o[k]);
        else {
#pragma unroll
            for (int e = 0; e < V; ++e) o[k][e] = 0.f;
        }
    }
#pragma unroll(Q * V <= 8 ? 4 : Q <= 2 ? 2 : 1)
    for (int s = 1; s < sl.splitk; ++s) {
        float t[Q][V];
#pragma unroll
        for (int k = 0; k < Q; ++k) if (on[k]) ldv<V>(sl.slabs + (long)s * sl.split_stride + off[k], t[k]);
#pragma unroll
        for (int k = 0; k < Q; ++k) if (on[k]) {
#pragma unroll
            for (int e = 0; e < V; ++e) o[k][e] += t[k][e];
        }
    }
    if (sl.bias) {
        const float b = sl.bias[c];
#pragma unroll
        for (int k = 0; k < Q; ++k) if (on[k]) {
#pragma unroll
            for (int e = 0; e < V; ++e) o[k][e] += b;
        }
    }
}

template <int V, int Q, typename ZT, typename AT, bool SLABS, bool POOL = false>
__global__ __launch_bounds__(256) void bn_fwd_coop_kernel(
    const ZT* __restrict__ z, long z_bs, AT* __restrict__ a, long a_bs, int N, int C, int HW, int S, int per, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, float* __restrict__ scale_out,
    float* __restrict__ shift_out, int relu, const SlabSrc sl, int groups, double* __restrict__ ws, const PoolSrc pl) {
    // POOL (round 6): the layer's activation also feeds nn.MaxPool2d(2, 2) -- the threads of the even image rows fetch the row below from
    // z (cache: its own thread is 32 .. 40 units away), normalise it as well and store the four window maxima of their unit to the pooled
    // tensor pl.pdy ([N][C][H/2][W/2] at this layer's channel 0): aide_maxpool2x2_fwd's pass disappears, same values (a maximum is exact)
    __shared__ double sm[2 * 4];
    __shared__ float coef[2];
    __shared__ int okf;
    const int c = blockIdx.x / S, s = blockIdx.x - c * S;
    const int hwv = HW / V, total = N * hwv;
    const int beg = s * per, end = min(beg + per, total);
    __shared__ long long e0s;
    if (threadIdx.x == 0) {
        okf = 1;
        e0s = S > 1 ? coop_load_i(ws_chan(ws, c)) : 0;        // generations completed on this channel before this launch
    }
    int nn[Q], pp[Q];
    bool on[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int u = beg + (int)threadIdx.x + k * 256;
        on[k] = u < end;
        nn[k] = on[k] ? u / hwv : 0;
        pp[k] = on[k] ? (u - nn[k] * hwv) * V : 0;
    }
    for (int gi = 0; gi < groups; ++gi) {
        const long n0 = (long)gi * N;
        float v[Q][V];
        if constexpr (SLABS) {
            long off[Q];
#pragma unroll
            for (int k = 0; k < Q; ++k) off[k] = (n0 + nn[k]) * sl.slab_bs + (long)c * HW + pp[k];
            slab_sum_q<V, Q>(sl, off, on, c, v);
#pragma unroll
            for (int k = 0; k < Q; ++k) if (on[k]) {
                stv<V>(const_cast<ZT*>(z) + (n0 + nn[k]) * z_bs + (long)c * HW + pp[k], v[k]);
#pragma unroll
                for (int e = 0; e < V; ++e) v[k][e] = as_stored(v[k][e], (const ZT*)nullptr);
            }
        } else {
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                if (on[k]) ldv<V>(z + (n0 + nn[k]) * z_bs + (long)c * HW + pp[k], v[k]);
                else {
#pragma unroll
                    for (int e = 0; e < V; ++e) v[k][e] = 0.f;
                }
            }
        }
        double acc[2] = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < Q; ++k) {
#pragma unroll
            for (int e = 0; e < V; ++e) { const double d = (double)v[k][e]; acc[0] += d; acc[1] = fma(d, d, acc[1]); }
        }
        block_sum_d<2>(acc, sm);
        const long long e0 = e0s, gen = e0 + 1 + gi;          // (every thread: block_sum_d synchronised behind thread 0's store)
        if (S > 1) coop_gather<2>(ws, c, s, S, gen, acc, sm, &okf);
        if (threadIdx.x == 0) {
            if (s == coop_leader(S)) {
                const double mean = acc[0] / count;
                double var = acc[1] / count - mean * mean;
                if (var < 0.0) var = 0.0;
                const float rstd = (float)(1.0 / sqrt(var + (double)eps));
                const float g = gamma ? gamma[c] : 1.0f, bb = beta ? beta[c] : 0.0f;
                const float sc = okf ? g * rstd : __builtin_nanf(""), sh = bb - (float)mean * sc;
                coef[0] = sc; coef[1] = sh;
                if (S > 1) coop_publish(ws, c, gen, sc, sh);
                if (gi == groups - 1) { mean_out[c] = (float)mean; rstd_out[c] = rstd; scale_out[c] = sc; shift_out[c] = sh; }
                if (running_mean) {
                    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
                    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
                }
                if (c == 0 && nbt) *nbt += 1;
            } else {
                float sc, sh;
                if (!coop_receive(ws, c, gen, sc, sh)) sc = sh = __builtin_nanf("");
                coef[0] = sc; coef[1] = sh;
            }
        }
        __syncthreads();
        const float sc = coef[0], sh = coef[1];
#pragma unroll
        for (int k = 0; k < Q; ++k) if (on[k]) {
            float o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) { const float y = fmaf(v[k][e], sc, sh); o[e] = relu ? fmaxf(y, 0.0f) : y; }
            stv<V>(a + (n0 + nn[k]) * a_bs + (long)c * HW + pp[k], o);
            if constexpr (POOL) {
                static_assert(V == 8 && !SLABS, "pool-fused forward: units of 8, z as it is");
                const int row = pp[k] / pl.W, col = pp[k] - row * pl.W;
                if (!(row & 1)) {
                    float zp[V];
                    ldv<V>(z + (n0 + nn[k]) * z_bs + (long)c * HW + pp[k] + pl.W, zp);
                    float m[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float q0 = fmaf(zp[2 * j], sc, sh), q1 = fmaf(zp[2 * j + 1], sc, sh);
                        if (relu) { q0 = fmaxf(q0, 0.0f); q1 = fmaxf(q1, 0.0f); }
                        m[j] = fmaxf(fmaxf(o[2 * j], o[2 * j + 1]), fmaxf(q0, q1));
                    }
                    stv<4>(const_cast<float*>(pl.pdy) + (n0 + nn[k]) * pl.pdy_bs + (long)c * (HW >> 2) + (row >> 1) * (pl.W >> 1) + (col >> 1), m);
                }
            }
        }
    }
    if (S > 1 && s == coop_leader(S) && threadIdx.x == 0) coop_store_i(ws_chan(ws, c), e0s + groups);
}

// SLABS: dA is still in the split-K slabs of the data-gradient convolution that produced it ([split][N][C][HW], fp32):
// the kernel sums them itself in the order of the split reduce (s = 0, 1, ...) -- that launch and its pass disappear.
template <int V, int Q, typename ZT, typename DT, typename GT, bool SLABS, bool POOL = false, bool HEAD = false>
__global__ __launch_bounds__(256) void bn_bwd_coop_kernel(
    const GT* __restrict__ dA, long d_bs, const ZT* __restrict__ z, long z_bs, DT* __restrict__ dz, long dz_bs, int N, int C,
    int HW, int S, int per, double count, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ scale, const float* __restrict__ shift, int relu, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ dbias, const SlabSrc sl, double* __restrict__ ws, const PoolSrc pl) {
    __shared__ double sm[3 * 4];
    __shared__ float coef[2];
    __shared__ int okf;
    const int c = blockIdx.x / S, s = blockIdx.x - c * S;
    const int hwv = HW / V, total = N * hwv;
    const int beg = s * per, end = min(beg + per, total);
    __shared__ long long e0s;
    if (threadIdx.x == 0) {
        okf = 1;
        e0s = S > 1 ? coop_load_i(ws_chan(ws, c)) : 0;
    }
    const float mu = mean[c], rs = rstd[c], sc = scale[c], sh = shift[c];
    int nn[Q], pp[Q];
    bool on[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int u = beg + (int)threadIdx.x + k * 256;
        on[k] = u < end;
        nn[k] = on[k] ? u / hwv : 0;
        pp[k] = on[k] ? (u - nn[k] * hwv) * V : 0;
    }
    float dy[Q][V], xh[Q][V];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        if (on[k]) ldv<V>(z + (long)nn[k] * z_bs + (long)c * HW + pp[k], xh[k]);
        else {
#pragma unroll
            for (int e = 0; e < V; ++e) xh[k][e] = 0.f;
        }
    }
    if constexpr (SLABS) {
        long off[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) off[k] = (long)nn[k] * sl.slab_bs + (long)c * HW + pp[k];
        slab_sum_q<V, Q>(sl, off, on, c, dy);
    } else if constexpr (HEAD) {
        // pl.pdy = dlogits [N][K][HW] (batch stride pl.pdy_bs), pl.hw = head weights [K][C]
#pragma unroll
        for (int k = 0; k < Q; ++k) {
#pragma unroll
            for (int e = 0; e < V; ++e) dy[k][e] = 0.f;
        }
        for (int kc = 0; kc < pl.K; ++kc) {
            const float wk = pl.hw[kc * C + c];
#pragma unroll
            for (int k = 0; k < Q; ++k) if (on[k]) {
                float g[V];
                ldv<V>(pl.pdy + (long)nn[k] * pl.pdy_bs + (long)kc * HW + pp[k], g);
#pragma unroll
                for (int e = 0; e < V; ++e) dy[k][e] = kc ? __builtin_fmaf(wk, g[e], dy[k][e]) : wk * g[e];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            if (on[k]) ldv<V>(dA + (long)nn[k] * d_bs + (long)c * HW + pp[k], dy[k]);
            else {
#pragma unroll
                for (int e = 0; e < V; ++e) dy[k][e] = 0.f;
            }
        }
    }
    if constexpr (POOL) {
        // (V = 8, W % 8 == 0: a unit lies in one image row and covers four windows' columns; the partner row comes from L1 / L2 --
        // its own thread is 32 .. 40 units away)
        static_assert(V == 8, "pool-fused backward: units of 8");
#pragma unroll
        for (int k = 0; k < Q; ++k) if (on[k]) {
            const int row = pp[k] / pl.W, col = pp[k] - row * pl.W;
            const bool lower = row & 1;
            float zp[V], g4[4];
            ldv<V>(z + (long)nn[k] * z_bs + (long)c * HW + pp[k] + (lower ? -pl.W : pl.W), zp);
            ldv<4>(pl.pdy + (long)nn[k] * pl.pdy_bs + (long)c * (HW >> 2) + (row >> 1) * (pl.W >> 1) + (col >> 1), g4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float o0 = fmaf(xh[k][2 * j], sc, sh), o1 = fmaf(xh[k][2 * j + 1], sc, sh);
                float q0 = fmaf(zp[2 * j], sc, sh), q1 = fmaf(zp[2 * j + 1], sc, sh);
                if (relu) { o0 = fmaxf(o0, 0.f); o1 = fmaxf(o1, 0.f); q0 = fmaxf(q0, 0.f); q1 = fmaxf(q1, 0.f); }
                const float t0 = lower ? q0 : o0, t1 = lower ? q1 : o1, b0 = lower ? o0 : q0, b1 = lower ? o1 : q1;
                int idx = 0;
                float m = t0;
                if (t1 > m) { m = t1; idx = 1; }
                if (b0 > m) { m = b0; idx = 2; }
                if (b1 > m) { m = b1; idx = 3; }
                const int mine = lower ? 2 : 0;
                if (idx == mine) dy[k][2 * j] += g4[j];
                else if (idx == mine + 1) dy[k][2 * j + 1] += g4[j];
            }
        }
    }
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < Q; ++k) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float zv = xh[k][e];
            const bool live = on[k] && (!relu || fmaf(zv, sc, sh) > 0.0f);
            dy[k][e] = live ? dy[k][e] : 0.0f;
            xh[k][e] = on[k] ? (zv - mu) * rs : 0.0f;
            acc[0] += (double)dy[k][e];
            acc[1] = fma((double)dy[k][e], (double)xh[k][e], acc[1]);
            acc[2] += (double)xh[k][e];
        }
    }
    block_sum_d<3>(acc, sm);
    const long long e0 = e0s;
    if (S > 1) coop_gather<3>(ws, c, s, S, e0 + 1, acc, sm, &okf);
    if (threadIdx.x == 0) {
        if (s == coop_leader(S)) {
            const float c0f = okf ? (float)(acc[0] / count) : __builtin_nanf(""), c1f = (float)(acc[1] / count);
            coef[0] = c0f; coef[1] = c1f;
            if (S > 1) coop_publish(ws, c, e0 + 1, c0f, c1f);
            if (dbeta) dbeta[c] = (float)acc[0];
            if (dgamma) dgamma[c] = (float)acc[1];
            if (dbias) dbias[c] = (float)((double)sc * ((acc[0] - count * (double)c0f) - (double)c1f * acc[2]));
        } else {
            float c0f, c1f;
            if (!coop_receive(ws, c, e0 + 1, c0f, c1f)) c0f = c1f = __builtin_nanf("");
            coef[0] = c0f; coef[1] = c1f;
        }
    }
    __syncthreads();
    const float c0 = coef[0], c1 = coef[1];
#pragma unroll
    for (int k = 0; k < Q; ++k) if (on[k]) {
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = sc * (dy[k][e] - c0 - xh[k][e] * c1);
        stv<V>(dz + (long)nn[k] * dz_bs + (long)c * HW + pp[k], o);
    }
    if (S > 1 && s == coop_leader(S) && threadIdx.x == 0) coop_store_i(ws_chan(ws, c), e0 + 1);
}

// which one-pass instantiation covers a channel of N * HW values: units of V values, Q per thread, S workgroups per channel.
// V = 8 whenever the plane and every batch stride are multiples of 8 -- whatever the storage types, so that the bf16-stored
// and the fp32-stored call partition the channel identically (bit-identical statistics) -- else V = 4.
struct CoopPlan { int V, Q, S, per; };
// Granularity from same-box step A/B (profiles/r06_bn_granularity_ab.txt): values per thread 8 / 16 / 32 / 64 -> C2 619 / 632 / 639 /
// 630 images/s, C4 390 / 397 / 399 / 397; smallest launch 2048 / 1024 / 512 workgroups -> C2 630 / 632 / 633 (with 32 values: 635).
constexpr int BN_COOP_MIN_WGS = 512;           // workgroups a launch should have before a thread takes more than 8 values
// Tensors beyond this many values take the two-pass kernels: the one-pass form keeps the channel in registers, so at most the
// register file's worth of a tensor (~10 M values) is in flight and every workgroup sits through the exchange (~10 us); on
// tensors several times that size the two streaming passes win.  Same-box step A/B, limit none / 34 M / 17 M / 9 M / 0
// (profiles/r06_bn_onepass_ab.txt): C2 630 / 631 / 632 / 629 / 614 images/s, C4 399 / 399 / 397 / 396 / 394 (fp32 storage),
// C5 465 / 491 / 501 / 504 / 504 (bf16 storage: half the bytes in flight per value).
constexpr long BN_ONEPASS_MAX_VALUES = 34L << 20, BN_ONEPASS_MAX_VALUES_NARROW = 9L << 20;
__host__ inline bool coop_plan(int N, int C, int HW, bool mod8, CoopPlan& p, bool narrow = false) {
    // (a channel of fewer than 2048 values: units of 4, so that all 256 threads of its workgroup hold one)
    p.V = (mod8 && HW % 8 == 0 && (long)N * HW >= 2048) ? 8 : 4;
    if ((long)N * C * HW > (narrow ? BN_ONEPASS_MAX_VALUES_NARROW : BN_ONEPASS_MAX_VALUES)) return false;
    if (HW % p.V || (long)N * HW / p.V > (long)BN_MAX_S * 256 * 8) return false;
    const int units = N * HW / p.V;
    int q = 32 / p.V, s = (units + 256 * q - 1) / (256 * q);           // 32 values per thread ...
    while (s > BN_MAX_S && q < 8) { q <<= 1; s = (units + 256 * q - 1) / (256 * q); }
    if (s > BN_MAX_S) return false;
    while (q > 1 && C * s < BN_COOP_MIN_WGS && s * 2 <= BN_MAX_S) { q >>= 1; s = (units + 256 * q - 1) / (256 * q); }   // ... fewer while the launch is small
    p.S = s;
    p.per = (units + s - 1) / s;
    const int need = (p.per + 255) / 256;
    p.Q = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
    return true;
}
// (the stacked passes and the conv-epilogue statistics keep their round-5 meaning of "small plane": see aide_bn_two_pass)
__host__ inline bool bn_fused_ok(int N, int C, int HW) { return (long)N * HW <= 256L * 16 * 4 && C >= 64; }

int pick_splits(int N, int C, int HW) {
    const long total4 = (long)N * HW / ((HW % 4 == 0) ? 4 : 1);
    int s = (int)((2048 + C - 1) / C);
    const long maxs = (total4 + 255) / 256;       // at least one float4 per thread
    if (s > maxs) s = (int)maxs;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

}  // namespace

extern "C" {

// workspace doubles needed by the BN kernels for a C-channel tensor
// (one block of BN_WS_STRIDE doubles per channel: see the top of this file).  ZERO-FILLED once by the caller.
size_t aide_bn_ws_bytes(int C) { return (size_t)C * BN_WS_STRIDE * sizeof(double); }

}  // extern "C"

namespace {

// N: images per group; groups > 1: a stacked batch of `groups` runs of N images, each with its own batch statistics, the
// running statistics updated once per group in order (one launch sequence instead of `groups`)
template <typename ZT, typename AT, bool SLABS = false>
int bn_train_fwd_t(const ZT* z, int64_t z_bs, AT* a, int64_t a_bs, int N, int C, int H, int W,
                   const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                   float* running_var, long long* num_batches_tracked, float* mean, float* rstd,
                   float* scale, float* shift, int relu, void* ws, hipStream_t stream, SlabSrc sl = SlabSrc{}, int groups = 1) {
    const int HW = H * W;
    if (!z || !a || !ws || groups < 1) return AIDE_ERR_ARG;
    const bool v4 = HW % 4 == 0 && z_bs % 4 == 0 && a_bs % 4 == 0;
    if (SLABS && (!sl.slabs || sl.splitk < 1 || !v4)) return AIDE_ERR_ARG;
    int splits = pick_splits(N, C, HW);
    if (groups > 1 && splits > 96 / groups) splits = 96 / groups > 0 ? 96 / groups : 1;   // partials: slots group * splits + s of a channel's block
    if ((long)groups * splits > 96) return AIDE_ERR_ARG;
    double* partials = (double*)ws;
    const double count = (double)N * HW;
    const int gx = max(1, min((HW / (v4 ? 4 : 1) + 255) / 256, 16));
    const int NT = N * groups;
    // algorithmic bytes of the layer (kernel timer): the input read once (z, or the slabs + the z it writes), a written once
    const double kt_bytes = (double)NT * C * HW * ((SLABS ? 4.0 * sl.splitk : 0.0) + sizeof(ZT) + sizeof(AT));
    // one pass, S workgroups per channel (all BASELINE shapes); the two-pass kernels below remain for planes that are not a
    // multiple of 4 values and for channels beyond BN_MAX_S * 2048 units
    {
        CoopPlan cp;
        constexpr bool narrow = !(std::is_same<ZT, float>::value && std::is_same<AT, float>::value);
        if (v4 && coop_plan(N, C, HW, z_bs % 8 == 0 && a_bs % 8 == 0 && (!SLABS || sl.split_stride % 8 == 0), cp, narrow)) {
#define AIDE_BN_FC(VV, QQ)                                                                                                    \
            AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, kt_bytes, (bn_fwd_coop_kernel<VV, QQ, ZT, AT, SLABS>), dim3(C * cp.S), dim3(256), 0, stream, z, \
                               (long)z_bs, a, (long)a_bs, N, C, HW, cp.S, cp.per, count, gamma, beta, eps, momentum,          \
                               running_mean, running_var, num_batches_tracked, mean, rstd, scale, shift, relu, sl, groups,    \
                               (double*)ws, PoolSrc{})
#define AIDE_BN_FQ(VV) do { if (cp.Q == 1) AIDE_BN_FC(VV, 1); else if (cp.Q == 2) AIDE_BN_FC(VV, 2); else if (cp.Q == 4) AIDE_BN_FC(VV, 4); else AIDE_BN_FC(VV, 8); } while (0)
            if (cp.V == 8) AIDE_BN_FQ(8); else AIDE_BN_FQ(4);
#undef AIDE_BN_FQ
#undef AIDE_BN_FC
            return aide_launch_status();
        }
    }
    // 8 values per lane: 16-byte accesses for the bf16-stored tensors of the precision='bf16' mode
    // (for every storage type: the bf16-storage kernels stay bit-identical to the fp32-storage ones on the widened tensor)
    const bool v8 = v4 && !SLABS && HW % 8 == 0 && z_bs % 8 == 0 && a_bs % 8 == 0;
    if (v8) {
        const int gx8 = max(1, min((HW / 8 + 255) / 256, 16));
        AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, 0.0, (bn_stats_kernel<8, ZT, false>), dim3(C * splits, groups), dim3(256), 0, stream, z, (long)z_bs, N, C, HW, splits, partials, sl);
        AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, kt_bytes, (bn_train_apply_kernel<8, ZT, AT>), dim3(NT * C, gx8), dim3(256), 0, stream, z, (long)z_bs, a, (long)a_bs, C, HW,
                           partials, splits, count, gamma, beta, eps, momentum, running_mean, running_var,
                           num_batches_tracked, mean, rstd, scale, shift, relu, (const float*)nullptr, 0, (const float*)nullptr, 0, N, groups);
    } else if (v4) {
        AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, 0.0, (bn_stats_kernel<4, ZT, SLABS>), dim3(C * splits, groups), dim3(256), 0, stream, z, (long)z_bs, N, C, HW, splits, partials, sl);
        AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, kt_bytes, (bn_train_apply_kernel<4, ZT, AT>), dim3(NT * C, gx), dim3(256), 0, stream, z, (long)z_bs, a, (long)a_bs, C, HW,
                           partials, splits, count, gamma, beta, eps, momentum, running_mean, running_var,
                           num_batches_tracked, mean, rstd, scale, shift, relu, (const float*)nullptr, 0, (const float*)nullptr, 0, N, groups);
    } else {
        AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, 0.0, (bn_stats_kernel<1, ZT, false>), dim3(C * splits, groups), dim3(256), 0, stream, z, (long)z_bs, N, C, HW, splits, partials, sl);
        AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, kt_bytes, (bn_train_apply_kernel<1, ZT, AT>), dim3(NT * C, gx), dim3(256), 0, stream, z, (long)z_bs, a, (long)a_bs, C, HW,
                           partials, splits, count, gamma, beta, eps, momentum, running_mean, running_var,
                           num_batches_tracked, mean, rstd, scale, shift, relu, (const float*)nullptr, 0, (const float*)nullptr, 0, N, groups);
    }
    return aide_launch_status();
}

template <typename ZT, typename AT>
int bn_relu_apply_t(const ZT* z, int64_t z_bs, AT* a, int64_t a_bs, int N, int C, int H, int W,
                    const float* scale, const float* shift, int relu, hipStream_t stream) {
    const int HW = H * W;
    const bool v4 = HW % 4 == 0 && z_bs % 4 == 0 && a_bs % 4 == 0;
    const int gx = max(1, min((HW / (v4 ? 4 : 1) + 255) / 256, 16));
    if (v4) AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, (double)N * C * HW * (sizeof(ZT) + sizeof(AT)), (bn_relu_apply_kernel<4, ZT, AT>), dim3(N * C, gx), dim3(256), 0, stream, z, (long)z_bs, a, (long)a_bs, C, HW, scale, shift, relu);
    else AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, (double)N * C * HW * (sizeof(ZT) + sizeof(AT)), (bn_relu_apply_kernel<1, ZT, AT>), dim3(N * C, gx), dim3(256), 0, stream, z, (long)z_bs, a, (long)a_bs, C, HW, scale, shift, relu);
    return aide_launch_status();
}

template <typename ZT, typename DT, typename GT>
int bn_relu_bwd_t(const GT* dA, int64_t d_bs, const ZT* z, int64_t z_bs, DT* dz, int64_t dz_bs,
                  int N, int C, int H, int W, const float* mean, const float* rstd, const float* scale,
                  const float* shift, int relu, float* dgamma, float* dbeta, float* dbias, void* ws, void* done,
                  hipStream_t stream) {
    const int HW = H * W;
    if (!ws) return AIDE_ERR_ARG;
    const bool v4 = HW % 4 == 0 && z_bs % 4 == 0 && d_bs % 4 == 0 && dz_bs % 4 == 0;
    const int splits = pick_splits(N, C, HW);
    double* partials = (double*)ws;
    const double count = (double)N * HW;
    const double kt_bytes = (double)N * C * HW * (sizeof(GT) + sizeof(ZT) + sizeof(DT));     // dA, z read once, dz written once
    {
        CoopPlan cp;
        constexpr bool narrow = !(std::is_same<ZT, float>::value && std::is_same<DT, float>::value && std::is_same<GT, float>::value);
        if (v4 && coop_plan(N, C, HW, z_bs % 8 == 0 && d_bs % 8 == 0 && dz_bs % 8 == 0, cp, narrow)) {
#define AIDE_BN_BC(VV, QQ)                                                                                                    \
            AIDE_LAUNCH_DONE_TIMED(AIDE_KT_BN_BWD, kt_bytes, done, (bn_bwd_coop_kernel<VV, QQ, ZT, DT, GT, false>), dim3(C * cp.S), dim3(256), 0, \
                             stream, dA, (long)d_bs, z, (long)z_bs, dz, (long)dz_bs, N, C, HW, cp.S, cp.per, count, mean,     \
                             rstd, scale, shift, relu, dgamma, dbeta, dbias, SlabSrc{}, (double*)ws, PoolSrc{})
#define AIDE_BN_BQ(VV) do { if (cp.Q == 1) AIDE_BN_BC(VV, 1); else if (cp.Q == 2) AIDE_BN_BC(VV, 2); else if (cp.Q == 4) AIDE_BN_BC(VV, 4); else AIDE_BN_BC(VV, 8); } while (0)
            if (cp.V == 8) AIDE_BN_BQ(8); else AIDE_BN_BQ(4);
#undef AIDE_BN_BQ
#undef AIDE_BN_BC
            return aide_launch_status();
        }
    }
    const bool v8 = v4 && HW % 8 == 0 && z_bs % 8 == 0 && d_bs % 8 == 0 && dz_bs % 8 == 0;
    if (v8) {
        AIDE_LAUNCH_TIMED(AIDE_KT_BN_BWD, 0.0, (bn_bwd_reduce_kernel<8, ZT, GT>), dim3(C * splits), dim3(256), 0, stream, dA, (long)d_bs, z, (long)z_bs, N, C, HW, splits, mean, rstd, scale, shift, relu, partials);
        AIDE_LAUNCH_DONE_TIMED(AIDE_KT_BN_BWD, kt_bytes, done, (bn_bwd_apply_kernel<8, ZT, DT, GT>), dim3(C * splits), dim3(256), 0, stream, dA, (long)d_bs, z, (long)z_bs, dz, (long)dz_bs, N, C, HW, splits, count, mean, rstd, scale, shift, relu, partials, dgamma, dbeta, dbias);
    } else if (v4) {
        AIDE_LAUNCH_TIMED(AIDE_KT_BN_BWD, 0.0, (bn_bwd_reduce_kernel<4, ZT, GT>), dim3(C * splits), dim3(256), 0, stream, dA, (long)d_bs, z, (long)z_bs, N, C, HW, splits, mean, rstd, scale, shift, relu, partials);
        AIDE_LAUNCH_DONE_TIMED(AIDE_KT_BN_BWD, kt_bytes, done, (bn_bwd_apply_kernel<4, ZT, DT, GT>), dim3(C * splits), dim3(256), 0, stream, dA, (long)d_bs, z, (long)z_bs, dz, (long)dz_bs, N, C, HW, splits, count, mean, rstd, scale, shift, relu, partials, dgamma, dbeta, dbias);
    } else {
        AIDE_LAUNCH_TIMED(AIDE_KT_BN_BWD, 0.0, (bn_bwd_reduce_kernel<1, ZT, GT>), dim3(C * splits), dim3(256), 0, stream, dA, (long)d_bs, z, (long)z_bs, N, C, HW, splits, mean, rstd, scale, shift, relu, partials);
        AIDE_LAUNCH_DONE_TIMED(AIDE_KT_BN_BWD, kt_bytes, done, (bn_bwd_apply_kernel<1, ZT, DT, GT>), dim3(C * splits), dim3(256), 0, stream, dA, (long)d_bs, z, (long)z_bs, dz, (long)dz_bs, N, C, HW, splits, count, mean, rstd, scale, shift, relu, partials, dgamma, dbeta, dbias);
    }
    return aide_launch_status();
}

}  // namespace

extern "C" {

// Training-mode forward of relu(bn(z)): batch statistics (pass 1), then finalize + apply (pass 2).
// Outputs mean/rstd/scale/shift [C] are kept by the caller for the backward.
// the same operators on bf16-stored conv outputs / conv-output gradients (precision='bf16'); z_bf16 / dz_bf16 select
// the storage type of the untyped pointers, everything else is unchanged
int aide_bn_train_fwd_mixed(const void* z, int z_bf16, int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int C,
                            int H, int W, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, long long* num_batches_tracked, float* mean,
                            float* rstd, float* scale, float* shift, int relu, void* ws, hipStream_t stream) {
#define AIDE_BN_FWD(ZT, AT) bn_train_fwd_t<ZT, AT>((const ZT*)z, z_bs, (AT*)a, a_bs, N, C, H, W, gamma, beta, eps, momentum, \
                                                   running_mean, running_var, num_batches_tracked, mean, rstd, scale,      \
                                                   shift, relu, ws, stream)
    if (z_bf16) return a_bf16 ? AIDE_BN_FWD(bf16_t, bf16_t) : AIDE_BN_FWD(bf16_t, float);
    return a_bf16 ? AIDE_BN_FWD(float, bf16_t) : AIDE_BN_FWD(float, float);
#undef AIDE_BN_FWD
}

// The same operator fed by the split-K slabs of the convolution before it (launched with accumulate = 2): sums
// slabs [splitk][N][C][H][W] (fp32, dense) + bias[c] in split order, WRITES z (kept for the backward pass), normalises.
// With group batching the caller offsets `slabs` / z / a by whole images; `split_stride` stays the full slab size.
int aide_bn_train_fwd_slabs(const float* slabs, int splitk, int64_t split_stride, const float* bias, void* z, int z_bf16,
                            int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int C, int H, int W,
                            const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                            float* running_var, long long* num_batches_tracked, float* mean, float* rstd, float* scale,
                            float* shift, int relu, void* ws, hipStream_t stream) {
    SlabSrc sl;
    sl.slabs = slabs; sl.bias = bias; sl.split_stride = split_stride; sl.slab_bs = (long)C * H * W; sl.splitk = splitk;
#define AIDE_BN_FWD_S(ZT, AT) bn_train_fwd_t<ZT, AT, true>((const ZT*)z, z_bs, (AT*)a, a_bs, N, C, H, W, gamma, beta, eps,   \
                                                           momentum, running_mean, running_var, num_batches_tracked, mean, \
                                                           rstd, scale, shift, relu, ws, stream, sl)
    if (z_bf16) return a_bf16 ? AIDE_BN_FWD_S(bf16_t, bf16_t) : AIDE_BN_FWD_S(bf16_t, float);
    return a_bf16 ? AIDE_BN_FWD_S(float, bf16_t) : AIDE_BN_FWD_S(float, float);
#undef AIDE_BN_FWD_S
}

// N images per group, `groups` groups stacked along the batch; group g's entries are parts[c][g * nparts .. (g + 1) * nparts)
static int bn_parts_launch(const void* z, int z_bf16, int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int groups, int C,
                           int H, int W, const float* parts, int nparts, int parts_stride, const float* conv_bias,
                           const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                           float* running_var, long long* num_batches_tracked, float* mean, float* rstd,
                           float* scale, float* shift, int relu, hipStream_t stream) {
    const int HW = H * W;
    if (!z || !a || !parts || nparts <= 0 || groups < 1 || parts_stride < nparts * groups || HW % 4 || z_bs % 4 || a_bs % 4)
        return AIDE_ERR_ARG;
    const double count = (double)N * HW;
    const int gx = max(1, min((HW / 4 + 255) / 256, 16));
#define AIDE_BN_PARTS(ZT, AT)                                                                                                  \
    AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, (double)N * groups * C * HW * (sizeof(ZT) + sizeof(AT)), (bn_train_apply_kernel<4, ZT, AT>), dim3(N * groups * C, gx), dim3(256), 0, stream, (const ZT*)z, (long)z_bs, \
                       (AT*)a, (long)a_bs, C, HW, (const double*)nullptr, 0, count, gamma, beta, eps, momentum, running_mean, \
                       running_var, num_batches_tracked, mean, rstd, scale, shift, relu, parts, nparts, conv_bias, parts_stride, N, groups)
    if (z_bf16) { if (a_bf16) AIDE_BN_PARTS(bf16_t, bf16_t); else AIDE_BN_PARTS(bf16_t, float); }
    else { if (a_bf16) AIDE_BN_PARTS(float, bf16_t); else AIDE_BN_PARTS(float, float); }
#undef AIDE_BN_PARTS
    return aide_launch_status();
}

// BatchNorm(train)+ReLU whose statistics were emitted by the convolution's own epilogue (stats_parts of aide_conv3x3_wino4): parts
// [C][nparts][2] fp32 = per channel and conv workgroup tile the sum and sum of squares of z - conv_bias.  One launch, one
// read of z.  (H*W % 4 == 0 and 16-byte aligned batch strides.)
// parts_stride: entries per channel in `parts` (>= nparts).  A group of a stacked batch passes the pointer to ITS first
// entry of channel 0 and the count of its own entries (the conv writes the entries of an image contiguously).
int aide_bn_train_fwd_parts_strided(const void* z, int z_bf16, int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int C,
                                    int H, int W, const float* parts, int nparts, int parts_stride, const float* conv_bias,
                                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                    float* running_var, long long* num_batches_tracked, float* mean, float* rstd,
                                    float* scale, float* shift, int relu, hipStream_t stream);

int aide_bn_train_fwd_parts_strided(const void* z, int z_bf16, int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int C,
                                    int H, int W, const float* parts, int nparts, int parts_stride, const float* conv_bias,
                                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                    float* running_var, long long* num_batches_tracked, float* mean, float* rstd,
                                    float* scale, float* shift, int relu, hipStream_t stream) {
    return bn_parts_launch(z, z_bf16, z_bs, a, a_bf16, a_bs, N, 1, C, H, W, parts, nparts, parts_stride, conv_bias, gamma, beta,
                           eps, momentum, running_mean, running_var, num_batches_tracked, mean, rstd, scale, shift, relu,
                           stream);
}

// two-pass (statistics kernel + apply kernel) or single small-plane kernel?  1 = two passes: only then do epilogue
// statistics save a launch
int aide_bn_two_pass(int N, int C, int H, int W) { return bn_fused_ok(N, C, H * W) ? 0 : 1; }

// does the one-pass form (S workgroups per channel, values held in registers) cover a channel of N * H * W values?
// (batch strides that are multiples of 8 elements assumed when H * W is)
int aide_bn_one_pass(int N, int C, int H, int W) {
    CoopPlan cp;
    return coop_plan(N, C, H * W, true, cp) ? 1 : 0;
}

// BatchNorm(train) WITHOUT its pass over z: from the conv epilogue's statistics of a stacked batch (parts as in
// aide_bn_train_fwd_groups) to the per-group (scale, shift) table tab[groups][tab_C][2], entries [tab_c0, tab_c0 + C), that
// the consumer convolution's loader applies (in_bn_tab of aide_conv3x3_wino4), plus the running-statistics updates of
// `groups` sequential forwards.  N = images per group.
int aide_bn_finalize_groups(int N, int groups, int C, int H, int W, const float* parts, int nparts, int parts_stride,
                            const float* conv_bias, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, long long* num_batches_tracked, float* mean,
                            float* rstd, float* scale, float* shift, float* tab, int tab_C, int tab_c0, hipStream_t stream) {
    if (!parts || !tab || N < 1 || groups < 1 || C < 1 || nparts < 1 || parts_stride < nparts * groups || tab_c0 < 0 ||
        tab_c0 + C > tab_C || !mean || !rstd || !scale || !shift)
        return AIDE_ERR_ARG;
    AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, 0.0, bn_finalize_groups_kernel, dim3(C), dim3(256), 0, stream, parts, nparts, parts_stride, conv_bias,
                       (double)N * H * W, groups, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
                       mean, rstd, scale, shift, tab, tab_C, tab_c0);
    return aide_launch_status();
}

// BatchNorm(train)+ReLU of a STACKED batch: `groups` independent batches of N images each, stacked along the batch
// dimension of z / a (the four detached augmentation forwards of the co-teaching loop run as one pass,
// trainchaos_proposed_30cases1labeled.py:265-269).  Every group is normalised with its own batch statistics and the
// running statistics / num_batches_tracked are updated once per group, in order -- exactly `groups` sequential
// train-mode forwards -- in one launch sequence instead of `groups`.  mean / rstd / scale / shift receive the LAST
// group's values.  Input, one of: z as it is (slabs == parts == NULL); split-K slabs [splitk][N * groups][C][H][W] of the
// conv before it (slabs, splitk, split_stride, slab_bias: as aide_bn_train_fwd_slabs; z is written); the conv epilogue's
// statistics (parts [C][parts_stride][2] with group g's nparts entries at g * nparts, conv_bias: as
// aide_bn_train_fwd_parts).  groups <= 96 / 2 (workspace: aide_bn_ws_bytes).
int aide_bn_train_fwd_groups(const void* z, int z_bf16, int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int groups,
                             int C, int H, int W, const float* slabs, int splitk, int64_t split_stride,
                             const float* slab_bias, const float* parts, int nparts, int parts_stride,
                             const float* conv_bias, const float* gamma, const float* beta, float eps, float momentum,
                             float* running_mean, float* running_var, long long* num_batches_tracked, float* mean,
                             float* rstd, float* scale, float* shift, int relu, void* ws, hipStream_t stream) {
    if (groups < 1 || N < 1 || (slabs && parts)) return AIDE_ERR_ARG;
    if (parts)
        return bn_parts_launch(z, z_bf16, z_bs, a, a_bf16, a_bs, N, groups, C, H, W, parts, nparts, parts_stride, conv_bias,
                               gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, mean, rstd, scale,
                               shift, relu, stream);
    SlabSrc sl = SlabSrc{};
    if (slabs) { sl.slabs = slabs; sl.bias = slab_bias; sl.split_stride = split_stride; sl.slab_bs = (long)C * H * W; sl.splitk = splitk; }
#define AIDE_BN_FWD_G(ZT, AT)                                                                                                  \
    (slabs ? bn_train_fwd_t<ZT, AT, true>((const ZT*)z, z_bs, (AT*)a, a_bs, N, C, H, W, gamma, beta, eps, momentum, running_mean, \
                                          running_var, num_batches_tracked, mean, rstd, scale, shift, relu, ws, stream, sl, groups) \
           : bn_train_fwd_t<ZT, AT, false>((const ZT*)z, z_bs, (AT*)a, a_bs, N, C, H, W, gamma, beta, eps, momentum, running_mean, \
                                           running_var, num_batches_tracked, mean, rstd, scale, shift, relu, ws, stream, sl, groups))
    if (z_bf16) return a_bf16 ? AIDE_BN_FWD_G(bf16_t, bf16_t) : AIDE_BN_FWD_G(bf16_t, float);
    return a_bf16 ? AIDE_BN_FWD_G(float, bf16_t) : AIDE_BN_FWD_G(float, float);
#undef AIDE_BN_FWD_G
}

int aide_bn_relu_apply_mixed(const void* z, int z_bf16, int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int C,
                             int H, int W, const float* scale, const float* shift, int relu, hipStream_t stream) {
#define AIDE_BN_APPLY(ZT, AT) bn_relu_apply_t<ZT, AT>((const ZT*)z, z_bs, (AT*)a, a_bs, N, C, H, W, scale, shift, relu, stream)
    if (z_bf16) return a_bf16 ? AIDE_BN_APPLY(bf16_t, bf16_t) : AIDE_BN_APPLY(bf16_t, float);
    return a_bf16 ? AIDE_BN_APPLY(float, bf16_t) : AIDE_BN_APPLY(float, float);
#undef AIDE_BN_APPLY
}

int aide_bn_relu_bwd_mixed(const void* dA, int dA_bf16, int64_t d_bs, const void* z, int z_bf16, int64_t z_bs, void* dz,
                           int dz_bf16, int64_t dz_bs, int N, int C, int H, int W, const float* mean, const float* rstd,
                           const float* scale, const float* shift, int relu, float* dgamma, float* dbeta, float* dbias,
                           void* ws, void* done, hipStream_t stream) {
#define AIDE_BN_BWD(ZT, DT, GT) bn_relu_bwd_t<ZT, DT, GT>((const GT*)dA, d_bs, (const ZT*)z, z_bs, (DT*)dz, dz_bs, N, C, H, W, \
                                                          mean, rstd, scale, shift, relu, dgamma, dbeta, dbias, ws, done, stream)
#define AIDE_BN_BWD_G(ZT, DT) (dA_bf16 ? AIDE_BN_BWD(ZT, DT, bf16_t) : AIDE_BN_BWD(ZT, DT, float))
    if (z_bf16) return dz_bf16 ? AIDE_BN_BWD_G(bf16_t, bf16_t) : AIDE_BN_BWD_G(bf16_t, float);
    return dz_bf16 ? AIDE_BN_BWD_G(float, bf16_t) : AIDE_BN_BWD_G(float, float);
#undef AIDE_BN_BWD_G
#undef AIDE_BN_BWD
}

// Backward of relu(bn(z)) with dA taken from the split-K slabs [splitk][N][C][H][W] of the data-gradient convolution that
// produced it (launched with accumulate = 2).  Shapes of the one-pass form: aide_bn_one_pass(N, C, H, W) == 1.
int aide_bn_relu_bwd_slabs(const float* slabs, int splitk, int64_t split_stride, const float* z, int64_t z_bs, float* dz,
                           int64_t dz_bs, int N, int C, int H, int W, const float* mean, const float* rstd,
                           const float* scale, const float* shift, int relu, float* dgamma, float* dbeta, float* dbias,
                           void* ws, void* done, hipStream_t stream) {
    const int HW = H * W;
    CoopPlan cp;
    if (!slabs || splitk < 1 || !z || !dz || !ws || HW % 4 || z_bs % 4 || dz_bs % 4 || split_stride % 4 ||
        !coop_plan(N, C, HW, z_bs % 8 == 0 && dz_bs % 8 == 0 && split_stride % 8 == 0, cp))
        return AIDE_ERR_ARG;
    SlabSrc sl;
    sl.slabs = slabs; sl.bias = nullptr; sl.split_stride = split_stride; sl.slab_bs = (long)C * HW; sl.splitk = splitk;
#define AIDE_BN_BS(VV, QQ)                                                                                                    \
    AIDE_LAUNCH_DONE_TIMED(AIDE_KT_BN_BWD, (double)N * C * HW * (4.0 * splitk + 8.0), done, (bn_bwd_coop_kernel<VV, QQ, float, float, float, true>), dim3(C * cp.S), dim3(256), 0, stream, \
                     (const float*)nullptr, 0L, z, (long)z_bs, dz, (long)dz_bs, N, C, HW, cp.S, cp.per, (double)N * HW, mean, \
                     rstd, scale, shift, relu, dgamma, dbeta, dbias, sl, (double*)ws, PoolSrc{})
#define AIDE_BN_BQ(VV) do { if (cp.Q == 1) AIDE_BN_BS(VV, 1); else if (cp.Q == 2) AIDE_BN_BS(VV, 2); else if (cp.Q == 4) AIDE_BN_BS(VV, 4); else AIDE_BN_BS(VV, 8); } while (0)
    if (cp.V == 8) AIDE_BN_BQ(8); else AIDE_BN_BQ(4);
#undef AIDE_BN_BQ
#undef AIDE_BN_BS
    return aide_launch_status();
}

// BatchNorm(train)+ReLU forward of a layer whose activation also feeds nn.MaxPool2d(2, 2) (fuseunet.py:51-78, UNet.py:114): writes the
// activation AND the pooled tensor [N * groups][C][H/2][W/2] (at this layer's channel 0, batch stride pooled_bs) -- the pooling pass does
// not run.  fp32 storage, z as it is (no slabs), shapes of aide_bn_relu_bwd_pool_supported(N, C, H, W) (N = images per group).
int aide_bn_train_fwd_pool(const float* z, int64_t z_bs, float* a, int64_t a_bs, float* pooled, int64_t pooled_bs, int N, int groups,
                           int C, int H, int W, const float* gamma, const float* beta, float eps, float momentum,
                           float* running_mean, float* running_var, long long* num_batches_tracked, float* mean, float* rstd,
                           float* scale, float* shift, int relu, void* ws, hipStream_t stream) {
    const int HW = H * W;
    CoopPlan cp;
    if (!z || !a || !pooled || !ws || groups < 1 || H % 2 || W % 8 || z_bs % 8 || a_bs % 8 || pooled_bs % 4 ||
        !coop_plan(N, C, HW, true, cp) || cp.V != 8)
        return AIDE_ERR_ARG;
    PoolSrc pl;
    pl.pdy = pooled; pl.pdy_bs = pooled_bs; pl.W = W; pl.hw = nullptr; pl.K = 0;
    const double kt_bytes = (double)N * groups * C * HW * 9.0;      // z in, a out, a quarter of it once more
#define AIDE_BN_FP(QQ)                                                                                                        \
    AIDE_LAUNCH_TIMED(AIDE_KT_BN_FWD, kt_bytes, (bn_fwd_coop_kernel<8, QQ, float, float, false, true>), dim3(C * cp.S), dim3(256), 0,   \
                      stream, z, (long)z_bs, a, (long)a_bs, N, C, HW, cp.S, cp.per, (double)N * HW, gamma, beta, eps, momentum,          \
                      running_mean, running_var, num_batches_tracked, mean, rstd, scale, shift, relu, SlabSrc{}, groups, (double*)ws, pl)
    if (cp.Q == 1) AIDE_BN_FP(1); else if (cp.Q == 2) AIDE_BN_FP(2); else if (cp.Q == 4) AIDE_BN_FP(4); else AIDE_BN_FP(8);
#undef AIDE_BN_FP
    return aide_launch_status();
}

// Backward of relu(bn(z)) whose activation also fed a MaxPool2d(2, 2): dA (the gradient from the activation's other readers, fp32) +
// the pooled gradient pdy [N][C][H/2][W/2] routed to every window's arg-max -- the max-pooling backward never runs as a pass of its
// own.  fp32 storage, one-pass shapes with units of 8 (aide_bn_relu_bwd_pool_supported).
int aide_bn_relu_bwd_pool_supported(int N, int C, int H, int W) {
    CoopPlan cp;
    return (H % 2 == 0 && W % 8 == 0 && coop_plan(N, C, H * W, true, cp) && cp.V == 8) ? 1 : 0;
}

int aide_bn_relu_bwd_pool(const float* dA, int64_t d_bs, const float* pdy, int64_t pdy_bs, const float* z, int64_t z_bs, float* dz,
                          int64_t dz_bs, int N, int C, int H, int W, const float* mean, const float* rstd, const float* scale,
                          const float* shift, int relu, float* dgamma, float* dbeta, float* dbias, void* ws, void* done,
                          hipStream_t stream) {
    const int HW = H * W;
    CoopPlan cp;
    if (!dA || !pdy || !z || !dz || !ws || !aide_bn_relu_bwd_pool_supported(N, C, H, W) || z_bs % 8 || d_bs % 8 || dz_bs % 8 ||
        pdy_bs % 4 || !coop_plan(N, C, HW, true, cp))
        return AIDE_ERR_ARG;
    PoolSrc pl;
    pl.pdy = pdy; pl.pdy_bs = pdy_bs; pl.W = W; pl.hw = nullptr; pl.K = 0;
    // dA, z read once, dz written once, the pooled gradient read once (the partner rows of z come from cache)
    const double kt_bytes = (double)N * C * HW * 12.0 + (double)N * C * (HW / 4) * 4.0;
#define AIDE_BN_BP(QQ)                                                                                                        \
    AIDE_LAUNCH_DONE_TIMED(AIDE_KT_BN_BWD, kt_bytes, done, (bn_bwd_coop_kernel<8, QQ, float, float, float, false, true>),     \
                           dim3(C * cp.S), dim3(256), 0, stream, dA, (long)d_bs, z, (long)z_bs, dz, (long)dz_bs, N, C, HW, cp.S, \
                           cp.per, (double)N * HW, mean, rstd, scale, shift, relu, dgamma, dbeta, dbias, SlabSrc{}, (double*)ws, pl)
    if (cp.Q == 1) AIDE_BN_BP(1); else if (cp.Q == 2) AIDE_BN_BP(2); else if (cp.Q == 4) AIDE_BN_BP(4); else AIDE_BN_BP(8);
#undef AIDE_BN_BP
    return aide_launch_status();
}

// Backward of relu(bn(z)) of the layer whose activation feeds the 1x1 head (fuseunet.py:41, :88; UNet.py:120): its dA is the head's data
// gradient sum_k w[k][c] dlogits[n][k][p], formed while dlogits is read -- aide_head1x1_bwd's dx pass (a write and a read of the widest
// feature map) does not run.  fp32 storage, one-pass shapes, 1 <= K <= 8.
int aide_bn_relu_bwd_head(const float* dlogits, int64_t dl_bs, const float* head_w, int K, const float* z, int64_t z_bs, float* dz,
                          int64_t dz_bs, int N, int C, int H, int W, const float* mean, const float* rstd, const float* scale,
                          const float* shift, int relu, float* dgamma, float* dbeta, float* dbias, void* ws, void* done,
                          hipStream_t stream) {
    const int HW = H * W;
    CoopPlan cp;
    if (!dlogits || !head_w || K < 1 || K > 8 || !z || !dz || !ws || HW % 4 || z_bs % 4 || dz_bs % 4 || dl_bs % 4 ||
        !coop_plan(N, C, HW, z_bs % 8 == 0 && dz_bs % 8 == 0 && dl_bs % 8 == 0, cp))
        return AIDE_ERR_ARG;
    PoolSrc pl;
    pl.pdy = dlogits; pl.pdy_bs = dl_bs; pl.W = W; pl.hw = head_w; pl.K = K;
    const double kt_bytes = (double)N * C * HW * 8.0 + (double)N * K * HW * 4.0;       // z, dz; dlogits once (re-read from cache per channel)
#define AIDE_BN_BH(VV, QQ)                                                                                                    \
    AIDE_LAUNCH_DONE_TIMED(AIDE_KT_BN_BWD, kt_bytes, done, (bn_bwd_coop_kernel<VV, QQ, float, float, float, false, false, true>), \
                           dim3(C * cp.S), dim3(256), 0, stream, (const float*)nullptr, 0L, z, (long)z_bs, dz, (long)dz_bs, N, C, HW, \
                           cp.S, cp.per, (double)N * HW, mean, rstd, scale, shift, relu, dgamma, dbeta, dbias, SlabSrc{}, (double*)ws, pl)
#define AIDE_BN_BQ(VV) do { if (cp.Q == 1) AIDE_BN_BH(VV, 1); else if (cp.Q == 2) AIDE_BN_BH(VV, 2); else if (cp.Q == 4) AIDE_BN_BH(VV, 4); else AIDE_BN_BH(VV, 8); } while (0)
    if (cp.V == 8) AIDE_BN_BQ(8); else AIDE_BN_BQ(4);
#undef AIDE_BN_BQ
#undef AIDE_BN_BH
    return aide_launch_status();
}

int aide_bn_eval_fold(int C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, const float* conv_bias, float* scale, float* shift,
                      float* fbias, hipStream_t stream) {
    if (C <= 0 || !running_mean || !running_var || !scale || !shift || !fbias) return AIDE_ERR_ARG;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, bn_eval_fold_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, C, gamma, beta,
                       running_mean, running_var, eps, conv_bias, scale, shift, fbias);
    return aide_launch_status();
}

// Backward of relu(bn(z)): dA -> dz, dgamma, dbeta, and the (mathematically zero) conv-bias grad.
}  // extern "C"
