// Spatial_Attention branch of the attention variants (fuseunetsa / UNetsa), forward and backward, gfx950.
//
// Replaces (reference): Spatial_Attention.forward  models_twomodalinputs/netblocks.py:68-89 (dup
// models_singlemodalinput/UNet.py:85-107) and its use `y = sa(y) * y` (fuseunet.py:139-141, UNet.py:191-200):
//     t1 = conv1x1(y; C -> R = C/16)      t2 = conv3x3(t1; dilation 4, pad 4)     t3 = conv3x3(t2; same)
//     t4 = conv1x1(t3; R -> 1)            g  = sigmoid(BatchNorm2d(1)(t4))        out[c] = g * y[c]
// and autograd's backward of that chain.
//
// Roofline: the branch works on R = 2..64 channels (C/16), i.e. 94 MFLOP per FuseUNet forward -- nothing for the
// matrix cores.  The cost is the three passes over the C-channel tensor (conv1 forward, the gate multiply, and in
// the backward the channel dot + conv1 dgrad/wgrad), all HBM streams: every kernel here is a plain coalesced
// VALU kernel, one output channel (or one weight) per workgroup row, fp64 block reductions in fixed order
// (bit-reproducible, like the BatchNorm kernels).
#include "common.h"

namespace {

template <int V> struct Vec;
template <> struct Vec<4> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec<1> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return f32x4{*p, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *p = v[0]; }
};

// ---- 1x1 convolution, any C -> one output channel per blockIdx.y --------------------------------------------
// y[n][r][p] = b[r] + sum_c w[r][c] x[n][c][p]
template <int V>
__global__ __launch_bounds__(256) void pw_fwd_kernel(const float* __restrict__ x, long x_bs,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     float* __restrict__ y, long y_bs, int C, int HW, long total) {
    extern __shared__ float ws[];                       // w[r][0..C)
    const int r = blockIdx.y;
    for (int i = threadIdx.x; i < C; i += 256) ws[i] = w[(long)r * C + i];
    __syncthreads();
    const int hwv = HW / V;
    const float bias = b ? b[r] : 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / hwv, p = (i - n * hwv) * V;
        const float* xp = x + n * x_bs + p;
        f32x4 acc = {bias, bias, bias, bias};
        for (int c = 0; c < C; ++c) acc += ws[c] * Vec<V>::ld(xp + (long)c * HW);
        Vec<V>::st(y + n * y_bs + (long)r * HW + p, acc);
    }
}

// dx[n][c][p] (+)= gate[n][p] * dout[n][c][p] + sum_r w[r][c] dt[n][r][p]     (gate / dout optional)
template <int V>
__global__ __launch_bounds__(256) void pw_dgrad_kernel(const float* __restrict__ dt, long dt_bs,
                                                       const float* __restrict__ w, const float* __restrict__ gate,
                                                       const float* __restrict__ dout, long dout_bs,
                                                       float* __restrict__ dx, long dx_bs, int C, int R, int HW,
                                                       long total, int accumulate) {
    extern __shared__ float ws[];                       // w[0..R)[c]
    const int c = blockIdx.y;
    for (int i = threadIdx.x; i < R; i += 256) ws[i] = w[(long)i * C + c];
    __syncthreads();
    const int hwv = HW / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / hwv, p = (i - n * hwv) * V;
        const float* tp = dt + n * dt_bs + p;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) acc += ws[r] * Vec<V>::ld(tp + (long)r * HW);
        if (gate) acc += Vec<V>::ld(gate + n * HW + p) * Vec<V>::ld(dout + n * dout_bs + (long)c * HW + p);
        float* xp = dx + n * dx_bs + (long)c * HW + p;
        if (accumulate) acc += Vec<V>::ld(xp);
        Vec<V>::st(xp, acc);
    }
}

// dw[r][c] = sum_{n,p} dt[n][r][p] x[n][c][p];  column c == C is the bias: db[r] = sum dt[n][r][p]
__global__ __launch_bounds__(256) void pw_wgrad_kernel(const float* __restrict__ dt, long dt_bs,
                                                       const float* __restrict__ x, long x_bs, int N, int C, int HW,
                                                       float* __restrict__ dw, float* __restrict__ db) {
    __shared__ double sm[4];
    const int c = blockIdx.x, r = blockIdx.y;
    double acc[1] = {0.0};
    const long total = (long)N * HW;
    for (long i = threadIdx.x; i < total; i += 256) {
        const long n = i / HW, p = i - n * HW;
        const float g = dt[n * dt_bs + (long)r * HW + p];
        const float v = c < C ? x[n * x_bs + (long)c * HW + p] : 1.f;
        acc[0] += (double)(g * v);
    }
    block_sum_d<1>(acc, sm);
    if (threadIdx.x == 0) {
        if (c < C) dw[(long)r * C + c] = (float)acc[0];
        else if (db) db[r] = (float)acc[0];
    }
}

// ---- dilated 3x3 convolution on R <= 64 channels (dense [N][R][H][W] tensors) ------------------------------------
// forward   : y[n][o][h][w] = b[o] + sum_{i,kh,kw} w[o][i][kh][kw] x[n][i][h+(kh-1)d][w+(kw-1)d]
// transposed: y[n][i][h][w] =        sum_{o,kh,kw} w[o][i][kh][kw] x[n][o][h-(kh-1)d][w-(kw-1)d]      (dgrad)
__global__ __launch_bounds__(256) void dconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ b, float* __restrict__ y, int N,
                                                    int Cin, int Cout, int H, int W, int dil, int transposed) {
    extern __shared__ float ws[];                       // [Cin][9] for this output channel
    const int o = blockIdx.y;
    for (int i = threadIdx.x; i < Cin * 9; i += 256) {
        const int ci = i / 9, t = i - ci * 9;
        ws[i] = transposed ? w[((long)ci * Cout + o) * 9 + (8 - t)] : w[((long)o * Cin + ci) * 9 + t];
    }
    __syncthreads();
    const int HW = H * W;
    const long total = (long)N * HW;
    const float bias = b ? b[o] : 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW;
        const int p = (int)(i - n * HW), h = p / W, wc = p - h * W;
        const float* xn = x + n * (long)Cin * HW;
        float acc = bias;
        for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ih = h + (t / 3 - 1) * dil, iw = wc + (t % 3 - 1) * dil;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) acc += ws[ci * 9 + t] * xn[(long)ci * HW + ih * W + iw];
            }
        }
        y[n * (long)Cout * HW + (long)o * HW + p] = acc;
    }
}

// dw[o][i][t] = sum_{n,h,w} dy[n][o][h][w] x[n][i][h+(kh-1)d][w+(kw-1)d];  blockIdx.x == Cin*9: db[o] = sum dy
__global__ __launch_bounds__(256) void dconv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ dw, float* __restrict__ db, int N,
                                                          int Cout, int Cin, int H, int W, int dil) {
    __shared__ double sm[4];
    const int o = blockIdx.y, it = blockIdx.x;
    const bool is_bias = it == Cin * 9;
    const int ci = it / 9, t = it - ci * 9;
    const int dh = (t / 3 - 1) * dil, dwc = (t % 3 - 1) * dil;
    const int HW = H * W;
    double acc[1] = {0.0};
    const long total = (long)N * HW;
    for (long i = threadIdx.x; i < total; i += 256) {
        const long n = i / HW;
        const int p = (int)(i - n * HW), h = p / W, wc = p - h * W;
        const float g = dy[(n * Cout + o) * (long)HW + p];
        float v = 1.f;
        if (!is_bias) {
            const int ih = h + dh, iw = wc + dwc;
            v = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? x[(n * Cin + ci) * (long)HW + ih * W + iw] : 0.f;
        }
        acc[0] += (double)(g * v);
    }
    block_sum_d<1>(acc, sm);
    if (threadIdx.x == 0) {
        if (!is_bias) dw[((long)o * Cin + ci) * 9 + t] = (float)acc[0];
        else if (db) db[o] = (float)acc[0];
    }
}

// ---- BatchNorm2d(1) + sigmoid gate ---------------------------------------------------------------------------
// stat[0] = mean, stat[1] = rstd (training: batch statistics + running-stat update; eval: running statistics)
__global__ __launch_bounds__(1024) void bn1_stats_kernel(const float* __restrict__ t, long M, float eps,
                                                         float momentum, int training, float* __restrict__ rm,
                                                         float* __restrict__ rv, long long* __restrict__ nbt,
                                                         float* __restrict__ stat) {
    __shared__ double sm[2 * 16];
    if (!training) {
        if (threadIdx.x == 0) { stat[0] = rm[0]; stat[1] = 1.0f / sqrtf(rv[0] + eps); }
        return;
    }
    double v[2] = {0.0, 0.0};
    for (long i = threadIdx.x; i < M; i += 1024) { const double a = t[i]; v[0] += a; v[1] += a * a; }
    block_sum_d<2>(v, sm);
    if (threadIdx.x == 0) {
        const double mean = v[0] / (double)M;
        double var = v[1] / (double)M - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[0] = (float)mean;
        stat[1] = (float)(1.0 / sqrt(var + (double)eps));
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        rm[0] = (1.f - momentum) * rm[0] + momentum * (float)mean;
        rv[0] = (1.f - momentum) * rv[0] + momentum * (float)unbiased;
        if (nbt) nbt[0] += 1;
    }
}

__global__ void sa_gate_kernel(const float* __restrict__ t, const float* __restrict__ stat,
                               const float* __restrict__ gamma, const float* __restrict__ beta,
                               float* __restrict__ gate, long M) {
    const float mean = stat[0], rstd = stat[1], g = gamma[0], b = beta[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long)gridDim.x * blockDim.x) {
        const float z = g * ((t[i] - mean) * rstd) + b;
        gate[i] = 1.0f / (1.0f + expf(-z));
    }
}

// out[n][c][p] = gate[n][p] * y[n][c][p]
template <int V>
__global__ __launch_bounds__(256) void sa_mul_kernel(const float* __restrict__ gate, const float* __restrict__ y,
                                                     long y_bs, float* __restrict__ out, long out_bs, int C, int HW,
                                                     long total) {
    const int hwv = HW / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / hwv, p = (i - n * hwv) * V;
        const f32x4 g = Vec<V>::ld(gate + n * HW + p);
        const float* yp = y + n * y_bs + p;
        float* op = out + n * out_bs + p;
        for (int c = 0; c < C; ++c) Vec<V>::st(op + (long)c * HW, g * Vec<V>::ld(yp + (long)c * HW));
    }
}

// ds[n][p] = (sum_c dout[n][c][p] y[n][c][p]) * g (1 - g)        (gradient w.r.t. the BatchNorm output)
template <int V>
__global__ __launch_bounds__(256) void sa_mul_bwd_kernel(const float* __restrict__ dout, long dout_bs,
                                                         const float* __restrict__ y, long y_bs,
                                                         const float* __restrict__ gate, float* __restrict__ ds,
                                                         int C, int HW, long total) {
    const int hwv = HW / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / hwv, p = (i - n * hwv) * V;
        const float* dp = dout + n * dout_bs + p;
        const float* yp = y + n * y_bs + p;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) acc += Vec<V>::ld(dp + (long)c * HW) * Vec<V>::ld(yp + (long)c * HW);
        const f32x4 g = Vec<V>::ld(gate + n * HW + p);
        Vec<V>::st(ds + n * HW + p, acc * g * (1.0f - g));
    }
}

// sums[0] = sum ds, sums[1] = sum ds * xhat;  dgamma = sums[1], dbeta = sums[0]
__global__ __launch_bounds__(1024) void bn1_bwd_reduce_kernel(const float* __restrict__ ds, const float* __restrict__ t,
                                                              const float* __restrict__ stat, long M,
                                                              double* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
    __shared__ double sm[2 * 16];
    const float mean = stat[0], rstd = stat[1];
    double v[2] = {0.0, 0.0};
    for (long i = threadIdx.x; i < M; i += 1024) {
        const double d = ds[i];
        v[0] += d;
        v[1] += d * (double)((t[i] - mean) * rstd);
    }
    block_sum_d<2>(v, sm);
    if (threadIdx.x == 0) {
        sums[0] = v[0]; sums[1] = v[1];
        dbeta[0] = (float)v[0];
        dgamma[0] = (float)v[1];
    }
}

// dt[i] = gamma * rstd * (ds[i] - mean(ds) - xhat[i] * mean(ds * xhat))
__global__ void bn1_bwd_apply_kernel(const float* __restrict__ ds, const float* __restrict__ t,
                                     const float* __restrict__ stat, const float* __restrict__ gamma,
                                     const double* __restrict__ sums, float* __restrict__ dt, long M) {
    const float mean = stat[0], rstd = stat[1];
    const float m1 = (float)(sums[0] / (double)M), m2 = (float)(sums[1] / (double)M);
    const float k = gamma[0] * rstd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long)gridDim.x * blockDim.x) {
        const float xh = (t[i] - mean) * rstd;
        dt[i] = k * (ds[i] - m1 - xh * m2);
    }
}

inline unsigned grid_for(long total) { return (unsigned)min((total + 255) / 256, (long)4096); }

}  // namespace

extern "C" {

// 1x1 convolution C -> R (bias optional).  x: [N][C][HW] batch stride x_bs; y: [N][R][HW] batch stride y_bs.
int aide_pwconv_fwd(const float* x, int64_t x_bs, const float* w, const float* b, float* y, int64_t y_bs, int N,
                    int C, int R, int HW, hipStream_t stream) {
    if (!x || !w || !y || N <= 0 || C <= 0 || R <= 0 || HW <= 0 || C > 8192) return AIDE_ERR_ARG;
    const bool vec = HW % 4 == 0 && x_bs % 4 == 0 && y_bs % 4 == 0;
    const long total = (long)N * (vec ? HW / 4 : HW);
    const dim3 grid(grid_for(total), R);
    if (vec) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, pw_fwd_kernel<4>, grid, dim3(256), C * sizeof(float), stream, x, (long)x_bs, w, b, y, (long)y_bs, C, HW, total);
    else AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, pw_fwd_kernel<1>, grid, dim3(256), C * sizeof(float), stream, x, (long)x_bs, w, b, y, (long)y_bs, C, HW, total);
    return aide_launch_status();
}

// dx (+)= gate * dout + W^T dt   (gate, dout may both be NULL: plain 1x1 dgrad).  gate: dense [N][HW].
int aide_pwconv_dgrad(const float* dt, int64_t dt_bs, const float* w, const float* gate, const float* dout,
                      int64_t dout_bs, float* dx, int64_t dx_bs, int N, int C, int R, int HW, int accumulate,
                      hipStream_t stream) {
    if (!dt || !w || !dx || N <= 0 || C <= 0 || R <= 0 || HW <= 0 || R > 8192 || (!gate != !dout)) return AIDE_ERR_ARG;
    const bool vec = HW % 4 == 0 && dt_bs % 4 == 0 && dx_bs % 4 == 0 && (!dout || dout_bs % 4 == 0);
    const long total = (long)N * (vec ? HW / 4 : HW);
    const dim3 grid(grid_for(total), C);
    if (vec) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, pw_dgrad_kernel<4>, grid, dim3(256), R * sizeof(float), stream, dt, (long)dt_bs, w, gate, dout, (long)dout_bs, dx, (long)dx_bs, C, R, HW, total, accumulate);
    else AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, pw_dgrad_kernel<1>, grid, dim3(256), R * sizeof(float), stream, dt, (long)dt_bs, w, gate, dout, (long)dout_bs, dx, (long)dx_bs, C, R, HW, total, accumulate);
    return aide_launch_status();
}

int aide_pwconv_wgrad(const float* dt, int64_t dt_bs, const float* x, int64_t x_bs, float* dw, float* db, int N,
                      int C, int R, int HW, hipStream_t stream) {
    if (!dt || !x || !dw || N <= 0 || C <= 0 || R <= 0 || HW <= 0) return AIDE_ERR_ARG;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, pw_wgrad_kernel, dim3(C + 1, R), dim3(256), 0, stream, dt, (long)dt_bs, x, (long)x_bs, N, C, HW, dw, db);
    return aide_launch_status();
}

// dilated 3x3 convolution (padding = dilation) on dense small-channel tensors; transposed != 0: the dgrad form
// (x = dy [N][Cout][H][W] -> y = dx [N][Cin][H][W], no bias), with w always [Cout][Cin][3][3].
int aide_dconv3x3_small(const float* x, const float* w, const float* b, float* y, int N, int Cin, int Cout, int H,
                        int W, int dilation, int transposed, hipStream_t stream) {
    if (!x || !w || !y || N <= 0 || Cin <= 0 || Cout <= 0 || Cin > 1024 || Cout > 1024 || dilation < 1) return AIDE_ERR_ARG;
    const long total = (long)N * H * W;
    if (transposed)     // reads Cout planes, writes Cin planes
        AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, dconv_kernel, dim3(grid_for(total), Cin), dim3(256), Cout * 9 * sizeof(float), stream, x, w,
                           (const float*)nullptr, y, N, Cout, Cin, H, W, dilation, 1);
    else
        AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, dconv_kernel, dim3(grid_for(total), Cout), dim3(256), Cin * 9 * sizeof(float), stream, x, w, b,
                           y, N, Cin, Cout, H, W, dilation, 0);
    return aide_launch_status();
}

int aide_dconv3x3_small_wgrad(const float* dy, const float* x, float* dw, float* db, int N, int Cout, int Cin, int H,
                              int W, int dilation, hipStream_t stream) {
    if (!dy || !x || !dw || N <= 0 || Cin <= 0 || Cout <= 0) return AIDE_ERR_ARG;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, dconv_wgrad_kernel, dim3(Cin * 9 + 1, Cout), dim3(256), 0, stream, dy, x, dw, db, N, Cout, Cin, H,
                       W, dilation);
    return aide_launch_status();
}

// gate = sigmoid(BatchNorm2d(1)(t4)); stat (2 floats) receives mean / rstd for the backward.
int aide_sa_gate_fwd(const float* t4, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float eps, float momentum, int training, float* stat, float* gate,
                     int64_t M, hipStream_t stream) {
    if (!t4 || !gamma || !beta || !running_mean || !running_var || !stat || !gate || M <= 0) return AIDE_ERR_ARG;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, bn1_stats_kernel, dim3(1), dim3(1024), 0, stream, t4, (long)M, eps, momentum, training,
                       running_mean, running_var, (long long*)num_batches_tracked, stat);
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, sa_gate_kernel, dim3(grid_for(M)), dim3(256), 0, stream, t4, stat, gamma, beta, gate, (long)M);
    return aide_launch_status();
}

// out[n][c][p] = gate[n][p] * y[n][c][p]
int aide_sa_mul(const float* gate, const float* y, int64_t y_bs, float* out, int64_t out_bs, int N, int C, int HW,
                hipStream_t stream) {
    if (!gate || !y || !out || N <= 0 || C <= 0 || HW <= 0) return AIDE_ERR_ARG;
    const bool vec = HW % 4 == 0 && y_bs % 4 == 0 && out_bs % 4 == 0;
    const long total = (long)N * (vec ? HW / 4 : HW);
    if (vec) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, sa_mul_kernel<4>, dim3(grid_for(total)), dim3(256), 0, stream, gate, y, (long)y_bs, out, (long)out_bs, C, HW, total);
    else AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, sa_mul_kernel<1>, dim3(grid_for(total)), dim3(256), 0, stream, gate, y, (long)y_bs, out, (long)out_bs, C, HW, total);
    return aide_launch_status();
}

// backward of  out = sigmoid(bn(t4)) * y  w.r.t. t4 (-> dt4, dgamma, dbeta).  ws: M floats + 16 bytes.
int aide_sa_gate_bwd(const float* dout, int64_t dout_bs, const float* y, int64_t y_bs, const float* gate,
                     const float* t4, const float* stat, const float* gamma, float* dgamma, float* dbeta, float* dt4,
                     int N, int C, int HW, float* ws, hipStream_t stream) {
    if (!dout || !y || !gate || !t4 || !stat || !gamma || !dgamma || !dbeta || !dt4 || !ws) return AIDE_ERR_ARG;
    const long M = (long)N * HW;
    double* sums = reinterpret_cast<double*>(ws);
    float* ds = ws + 4;
    const bool vec = HW % 4 == 0 && dout_bs % 4 == 0 && y_bs % 4 == 0;
    const long total = (long)N * (vec ? HW / 4 : HW);
    if (vec) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, sa_mul_bwd_kernel<4>, dim3(grid_for(total)), dim3(256), 0, stream, dout, (long)dout_bs, y, (long)y_bs, gate, ds, C, HW, total);
    else AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, sa_mul_bwd_kernel<1>, dim3(grid_for(total)), dim3(256), 0, stream, dout, (long)dout_bs, y, (long)y_bs, gate, ds, C, HW, total);
    AIDE_LAUNCH_TIMED(AIDE_KT_REDUCE, 0.0, bn1_bwd_reduce_kernel, dim3(1), dim3(1024), 0, stream, ds, t4, stat, M, sums, dgamma, dbeta);
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, bn1_bwd_apply_kernel, dim3(grid_for(M)), dim3(256), 0, stream, ds, t4, stat, gamma, sums, dt4, M);
    return aide_launch_status();
}

}  // extern "C"
