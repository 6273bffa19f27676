// 1x1 head convolution (64 -> num_classes) forward/backward and fused multi-tensor Adam(amsgrad).
//
// Replaces (reference):
//   last_conv1 = nn.Conv2d(64, num_classes, 1)  models_twomodalinputs/fuseunet.py:41,89,
//                                               models_singlemodalinput/UNet.py:150,164
//   torch.optim.Adam(net.parameters(), lr, amsgrad=True)  train_files/trainchaos_comparison_1case.py:170
// Both are HBM-bound: the head reads 64 planes once (MFMA is pointless at N = 2 output channels);
// Adam streams 20 B/param in and 16 B/param out.
#include "common.h"

namespace {

constexpr int MAXK = 8;      // num_classes supported by the head kernels (the reference's scripts use 2) = aide_seg_max_classes()
constexpr int HEAD_WG_BLOCKS = 1024;   // workgroups (= fp64 partial rows) of the head weight gradient: four per CU

template <int K, typename XT>
__global__ __launch_bounds__(256) void head_fwd_kernel(const XT* __restrict__ x, long x_bs,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       float* __restrict__ y, long y_bs, int C, int HW,
                                                       long total4, const float* __restrict__ in_scale,
                                                       const float* __restrict__ in_shift) {
    // in_scale / in_shift (round 6): x is the RAW output z of the layer under the head and its training-mode BatchNorm + ReLU is
    // applied here, a = max(fma(z, scale[c], shift[c]), 0) -- the same expression as the BatchNorm apply kernels, bit for bit -- so
    // that layer's normalising pass (134 MB per C2 step) never runs and its activation is never stored
    extern __shared__ float ws[];                 // [K][C]
    for (int i = threadIdx.x; i < K * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int hw4 = HW / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long n = i / hw4, p = i - n * hw4;
        const XT* xp = x + n * x_bs + p * 4;
        f32x4 acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { const float bk = b ? b[k] : 0.f; acc[k] = f32x4{bk, bk, bk, bk}; }
        for (int c = 0; c < C; ++c) {
            f32x4 v = ld4(xp + (long)c * HW);
            if (in_scale) {
                const float sc = in_scale[c], sh = in_shift[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(__builtin_fmaf(v[e], sc, sh), 0.0f);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += ws[k * C + c] * v;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) *reinterpret_cast<f32x4*>(y + n * y_bs + (long)k * HW + p * 4) = acc[k];
    }
}

// dx[n][c][p] = sum_k dy[n][k][p] w[k][c]
template <int K, typename DT>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dy, long dy_bs,
                                                         const float* __restrict__ w, DT* __restrict__ dx,
                                                         long dx_bs, int C, int HW, long total4) {
    extern __shared__ float ws[];
    for (int i = threadIdx.x; i < K * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int hw4 = HW / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long n = i / hw4, p = i - n * hw4;
        f32x4 g[K];
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = *reinterpret_cast<const f32x4*>(dy + n * dy_bs + (long)k * HW + p * 4);
        DT* xp = dx + n * dx_bs + p * 4;
        for (int c = 0; c < C; ++c) {
            f32x4 v = ws[c] * g[0];                    // (explicit fused multiply-adds, k ascending: aide_bn_relu_bwd_head forms
#pragma unroll                                         //  the same sums inside the BatchNorm backward, bit for bit)
            for (int k = 1; k < K; ++k) {
                const float wk = ws[k * C + c];
                v = f32x4{__builtin_fmaf(wk, g[k][0], v[0]), __builtin_fmaf(wk, g[k][1], v[1]), __builtin_fmaf(wk, g[k][2], v[2]),
                          __builtin_fmaf(wk, g[k][3], v[3])};
            }
            st4(xp + (long)c * HW, v);
        }
    }
}

// partial[b][k][c] = sum over this block's pixels of dy[k] * x[c];  partial[b][K*C + k] = sum dy[k]
// Channels are walked in groups of 8 ("channel" -1 = the bias: x == 1): eight 16-byte x loads in flight
// per pixel quad and ONE block reduction (two barriers) per group instead of per channel.
template <int K, typename XT>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ dy, long dy_bs,
                                                         const XT* __restrict__ x, long x_bs, int C, int HW,
                                                         long total4, double* __restrict__ partials,
                                                         const float* __restrict__ in_scale, const float* __restrict__ in_shift) {
    constexpr int G = K > 4 ? 4 : 8;          // (G x K fp64 accumulators per thread)
    __shared__ double sm[4][G * K];
    const int hw4 = HW / 4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int stride = gridDim.x * 256;
    for (int c0 = -1; c0 < C; c0 += G) {
        double acc[G][K];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[g][k] = 0.0;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
            const long n = i / hw4, p = i - n * hw4;
            f32x4 gk[K];
#pragma unroll
            for (int k = 0; k < K; ++k) gk[k] = *reinterpret_cast<const f32x4*>(dy + n * dy_bs + (long)k * HW + p * 4);
            f32x4 v[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int c = c0 + g;
                v[g] = f32x4{1.f, 1.f, 1.f, 1.f};
                if (c >= 0 && c < C) {
                    v[g] = ld4(x + n * x_bs + (long)c * HW + p * 4);
                    if (in_scale) {                    // (x is z: the activation is recomputed as head_fwd_kernel computed it)
                        const float sc = in_scale[c], sh = in_shift[c];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] = fmaxf(__builtin_fmaf(v[g][e], sc, sh), 0.0f);
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int k = 0; k < K; ++k)
                    // the pixel quad's four products are summed in fp32 and promoted once (per-product fp64 adds made
                    // the kernel VALU-bound: 565 us at 512x512 x8 on bf16 storage against 70 us of HBM time)
                    acc[g][k] += (double)(gk[k][0] * v[g][0] + gk[k][1] * v[g][1] + (gk[k][2] * v[g][2] + gk[k][3] * v[g][3]));
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[g][k] = wave_sum_d(acc[g][k]);
        __syncthreads();
        if (lane == 0)
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int k = 0; k < K; ++k) sm[wid][g * K + k] = acc[g][k];
        __syncthreads();
        if (threadIdx.x < G * K) {
            const int g = threadIdx.x / K, k = threadIdx.x - g * K, c = c0 + g;
            if (c < C) {
                const double t = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
                const int slot = (c >= 0) ? k * C + c : K * C + k;
                partials[(long)blockIdx.x * (K * C + K) + slot] = t;
            }
        }
    }
}

// one wave per output: lanes stride over the per-block partials, fixed-order shuffle tree
__global__ __launch_bounds__(64) void head_wgrad_finalize_kernel(const double* __restrict__ partials, int nblocks,
                                                                 int KC, int K, float* __restrict__ dw,
                                                                 float* __restrict__ db) {
    const int i = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partials[(long)b * (KC + K) + i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) {
        if (i < KC) dw[i] = (float)s;
        else if (db) db[i - KC] = (float)s;
    }
}

// ---------------------------------------------------------------- Adam (amsgrad), multi-tensor
struct AdamArgs {
    float* const* p; const float* const* g; float* const* m; float* const* v; float* const* vmax;
    const long* sizes; const long* block_start;      // prefix of 1024-element blocks per tensor
    int ntensors;
    float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt;
    int amsgrad;
};

__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
    // binary search the tensor owning this 1024-element block
    const long blk = blockIdx.x;
    int lo = 0, hi = a.ntensors - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.block_start[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    const int t = lo;
    const long base = (blk - a.block_start[t]) * 1024;
    const long n = a.sizes[t];
    float* p = a.p[t]; const float* g = a.g[t];
    float* m = a.m[t]; float* v = a.v[t]; float* vm = a.vmax[t];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long i = base + k * 256 + threadIdx.x;
        if (i < n) {
            float gi = g[i];
            const float pi = p[i];
            if (a.weight_decay != 0.f) gi += a.weight_decay * pi;
            const float mi = a.beta1 * m[i] + (1.f - a.beta1) * gi;
            const float vi = a.beta2 * v[i] + (1.f - a.beta2) * gi * gi;
            m[i] = mi; v[i] = vi;
            float vhat = vi;
            if (a.amsgrad) { vhat = fmaxf(vm[i], vi); vm[i] = vhat; }
            const float denom = sqrtf(vhat) / a.bc2_sqrt + a.eps;
            p[i] = pi - (a.lr / a.bc1) * (mi / denom);
        }
    }
}

int grid_for(long total) { return (int)max(1L, min((total + 255) / 256, 4096L)); }

template <int K, typename XT>
int head_wgrad_launch(const float* dy, long dy_bs, const XT* x, long x_bs, int C, int HW, long total4,
                      double* partials, int nblocks, hipStream_t s, const float* in_scale = nullptr, const float* in_shift = nullptr) {
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, (head_wgrad_kernel<K, XT>), dim3(nblocks), dim3(256), 0, s, dy, dy_bs, x, x_bs, C, HW,
                       total4, partials, in_scale, in_shift);
    return aide_launch_status();
}

template <typename XT>
int head_fwd_t(const XT* x, int64_t x_bs, const float* w, const float* b, float* y, int64_t y_bs, int N, int C, int K,
               int H, int W, hipStream_t stream, const float* in_scale = nullptr, const float* in_shift = nullptr) {
    const int HW = H * W;
    if (K < 1 || K > MAXK || HW % 4 || x_bs % 4 || y_bs % 4) return AIDE_ERR_ARG;
    const long total4 = (long)N * HW / 4;
    const int grid = grid_for(total4);
    const size_t sh = (size_t)K * C * sizeof(float);
#define AIDE_HEAD_FWD(KK) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, (head_fwd_kernel<KK, XT>), dim3(grid), dim3(256), sh, stream, x, (long)x_bs, w, b, y, (long)y_bs, C, HW, total4, in_scale, in_shift)
    switch (K) { case 1: AIDE_HEAD_FWD(1); break; case 2: AIDE_HEAD_FWD(2); break; case 3: AIDE_HEAD_FWD(3); break; case 4: AIDE_HEAD_FWD(4); break;
                 case 5: AIDE_HEAD_FWD(5); break; case 6: AIDE_HEAD_FWD(6); break; case 7: AIDE_HEAD_FWD(7); break; default: AIDE_HEAD_FWD(8); }
#undef AIDE_HEAD_FWD
    return aide_launch_status();
}

template <typename XT, typename DT>
int head_bwd_t(const float* dy, int64_t dy_bs, const XT* x, int64_t x_bs, const float* w, DT* dx, int64_t dx_bs,
               float* dw, float* db, int N, int C, int K, int H, int W, void* ws, hipStream_t stream,
               const float* in_scale = nullptr, const float* in_shift = nullptr) {
    const int HW = H * W;
    if (K < 1 || K > MAXK || HW % 4 || x_bs % 4 || dy_bs % 4 || (dx && dx_bs % 4) || !ws) return AIDE_ERR_ARG;
    const long total4 = (long)N * HW / 4;
    const size_t sh = (size_t)K * C * sizeof(float);
    if (dx) {
        const int grid = grid_for(total4);
#define AIDE_HEAD_DG(KK) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, (head_dgrad_kernel<KK, DT>), dim3(grid), dim3(256), sh, stream, dy, (long)dy_bs, w, dx, (long)dx_bs, C, HW, total4)
        switch (K) { case 1: AIDE_HEAD_DG(1); break; case 2: AIDE_HEAD_DG(2); break; case 3: AIDE_HEAD_DG(3); break; case 4: AIDE_HEAD_DG(4); break;
                     case 5: AIDE_HEAD_DG(5); break; case 6: AIDE_HEAD_DG(6); break; case 7: AIDE_HEAD_DG(7); break; default: AIDE_HEAD_DG(8); }
#undef AIDE_HEAD_DG
    }
    if (!dw) return aide_launch_status();      // data gradient only (the weight gradient is issued on another stream)
    const int nblocks = (int)max(1L, min((total4 + 255) / 256, (long)HEAD_WG_BLOCKS));
    int rc;
    switch (K) {
        case 1: rc = head_wgrad_launch<1>(dy, dy_bs, x, x_bs, C, HW, total4, (double*)ws, nblocks, stream, in_scale, in_shift); break;
        case 2: rc = head_wgrad_launch<2>(dy, dy_bs, x, x_bs, C, HW, total4, (double*)ws, nblocks, stream, in_scale, in_shift); break;
        case 3: rc = head_wgrad_launch<3>(dy, dy_bs, x, x_bs, C, HW, total4, (double*)ws, nblocks, stream, in_scale, in_shift); break;
        case 4: rc = head_wgrad_launch<4>(dy, dy_bs, x, x_bs, C, HW, total4, (double*)ws, nblocks, stream, in_scale, in_shift); break;
        case 5: rc = head_wgrad_launch<5>(dy, dy_bs, x, x_bs, C, HW, total4, (double*)ws, nblocks, stream, in_scale, in_shift); break;
        case 6: rc = head_wgrad_launch<6>(dy, dy_bs, x, x_bs, C, HW, total4, (double*)ws, nblocks, stream, in_scale, in_shift); break;
        case 7: rc = head_wgrad_launch<7>(dy, dy_bs, x, x_bs, C, HW, total4, (double*)ws, nblocks, stream, in_scale, in_shift); break;
        default: rc = head_wgrad_launch<8>(dy, dy_bs, x, x_bs, C, HW, total4, (double*)ws, nblocks, stream, in_scale, in_shift);
    }
    if (rc) return rc;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, head_wgrad_finalize_kernel, dim3(K * C + K), dim3(64), 0, stream,
                       (const double*)ws, nblocks, K * C, K, dw, db);
    return aide_launch_status();
}

}  // namespace

extern "C" {

size_t aide_head1x1_ws_bytes(int C, int K) { return (size_t)HEAD_WG_BLOCKS * (K * C + K) * sizeof(double); }

// dy: [N][K][HW] -> dx [N][C][HW] (may be NULL), dw [K][C] + db [K] (dw may be NULL: data gradient only)
// the head on a bf16-stored feature map (precision='bf16'); logits and every gradient stay fp32
int aide_head1x1_fwd_mixed(const void* x, int x_bf16, int64_t x_bs, const float* w, const float* b, float* y,
                           int64_t y_bs, int N, int C, int K, int H, int W, hipStream_t stream) {
    return x_bf16 ? head_fwd_t((const bf16_store_t*)x, x_bs, w, b, y, y_bs, N, C, K, H, W, stream)
                  : head_fwd_t((const float*)x, x_bs, w, b, y, y_bs, N, C, K, H, W, stream);
}
int aide_head1x1_bwd_mixed(const float* dy, int64_t dy_bs, const void* x, int x_bf16, int64_t x_bs, const float* w,
                           void* dx, int dx_bf16, int64_t dx_bs, float* dw, float* db, int N, int C, int K, int H, int W,
                           void* ws, hipStream_t stream) {
#define AIDE_HB(XT, DT) head_bwd_t(dy, dy_bs, (const XT*)x, x_bs, w, (DT*)dx, dx_bs, dw, db, N, C, K, H, W, ws, stream)
    if (x_bf16) return dx_bf16 ? AIDE_HB(bf16_store_t, bf16_store_t) : AIDE_HB(bf16_store_t, float);
    return dx_bf16 ? AIDE_HB(float, bf16_store_t) : AIDE_HB(float, float);
#undef AIDE_HB
}

// The head on the RAW conv output z of the layer under it, that layer's training-mode BatchNorm + ReLU applied on the way in
// (in_scale / in_shift [C]: the layer's scale / shift, e.g. from aide_bn_finalize_groups): forward, and the weight / bias gradient
// (the data gradient of such a layer is formed by aide_bn_relu_bwd_head).  fp32.
int aide_head1x1_fwd_bn(const float* z, int64_t z_bs, const float* in_scale, const float* in_shift, const float* w, const float* b,
                        float* y, int64_t y_bs, int N, int C, int K, int H, int W, hipStream_t stream) {
    if (!z || !in_scale || !in_shift) return AIDE_ERR_ARG;
    return head_fwd_t(z, z_bs, w, b, y, y_bs, N, C, K, H, W, stream, in_scale, in_shift);
}
int aide_head1x1_wgrad_bn(const float* dy, int64_t dy_bs, const float* z, int64_t z_bs, const float* in_scale, const float* in_shift,
                          float* dw, float* db, int N, int C, int K, int H, int W, void* ws, hipStream_t stream) {
    if (!z || !in_scale || !in_shift || !dw) return AIDE_ERR_ARG;
    return head_bwd_t<float, float>(dy, dy_bs, z, z_bs, nullptr, nullptr, 0, dw, db, N, C, K, H, W, ws, stream, in_scale, in_shift);
}

// Pointer tables (device memory): p,g,m,v,vmax [ntensors]; sizes, block_start [ntensors] (int64).
// `step` is the 1-based step count; bias corrections are folded on the host like torch.optim.Adam.
int aide_adam_amsgrad_multi(float* const* p, const float* const* g, float* const* m, float* const* v,
                            float* const* vmax, const int64_t* sizes, const int64_t* block_start,
                            int ntensors, int64_t total_blocks, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int amsgrad, int64_t step, hipStream_t stream) {
    if (ntensors <= 0 || total_blocks <= 0 || step < 1) return AIDE_ERR_ARG;
    AdamArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.vmax = vmax;
    a.sizes = (const long*)sizes; a.block_start = (const long*)block_start; a.ntensors = ntensors;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.amsgrad = amsgrad;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, adam_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, a);
    return aide_launch_status();
}

}  // extern "C"
