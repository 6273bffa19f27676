// ConvTranspose2d(kernel 2, stride 2) forward / dgrad / wgrad (gfx950) — the learned_bilinear=True
// up path.  Replaces (reference): nn.ConvTranspose2d(ci, co, kernel_size=2, stride=2) at
// models_twomodalinputs/netblocks.py:12 and models_singlemodalinput/UNet.py:7.
//
//   y[n][co][2h+kh][2w+kw] = b[co] + sum_ci x[n][ci][h][w] * W[ci][co][kh][kw]     (windows never overlap)
//
// The four taps are four independent 1x1 GEMMs.  No shipped train script enables this variant; it is served
// by one LDS-tiled GEMM skeleton on the fp32 MATRIX cores (v_mfma_f32_32x32x2_f32: 64 x 64 tile, four waves of 32 x 32,
// K staged 16 at a time) with problem-specific load / store functors:
//   forward  D[co][pix]      = sum_ci  W[ci][co][t] X[ci][pix]            (per tap t; pixel-shuffle store)
//   dgrad    D[ci][pix]      = sum_(co,t) W[ci][co][t] dY[co][2h+kh][2w+kw]
//   wgrad    D[ci][(co,t)]   = sum_pix X[ci][pix] dY[co][2h+kh][2w+kw]     (split over pixels, fixed-order slab sum)
// The MFMA is a k-ordered fp32 fmaf chain (exact fp32, no TF32 on gfx950): results are bit-identical to the scalar-FMA
// form of this kernel.
#include "common.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

// C[m][n] = sum_k A(k,m) * B(k,n) over k in [k0, k1)
template <class LoadA, class LoadB, class Store>
__device__ __forceinline__ void tile_gemm(int m0, int n0, int M, int Nn, int k0, int k1, LoadA la, LoadB lb,
                                          Store st) {
    __shared__ float As[TK][TM + 4];
    __shared__ float Bs[TK][TN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32;          // this wave's 32 x 32 block of the 64 x 64 tile
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kb = k0; kb < k1; kb += TK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const int kk = idx / TM, mm = idx % TM;          // consecutive threads -> consecutive m / n
            const int k = kb + kk;
            As[kk][mm] = (k < k1 && m0 + mm < M) ? la(k, m0 + mm) : 0.f;
            Bs[kk][mm] = (k < k1 && n0 + mm < Nn) ? lb(k, n0 + mm) : 0.f;
        }
        __syncthreads();
        // one MFMA per K pair: A[i = j][k = half], B[k = half][n = j] (rows of 68 floats: the two K slots of a wave read
        // two different rows, 32 consecutive floats each)
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + half][wm + j], Bs[kk + half][wn + j], acc, 0, 0, 0);
        __syncthreads();
    }
    // D layout: row = (r & 3) + 8 (r >> 2) + 4 half, column = j
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * half, n = n0 + wn + j;
        if (m < M && n < Nn) st(m, n, acc[r]);
    }
}

__global__ __launch_bounds__(256) void convt_fwd_kernel(const float* __restrict__ x, long x_bs,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ y, long y_bs, int Ci, int Co, int H,
                                                        int W) {
    const int HW = H * W, n = blockIdx.z >> 2, t = blockIdx.z & 3, kh = t >> 1, kw = t & 1;
    const float* xn = x + (long)n * x_bs;
    float* yn = y + (long)n * y_bs;
    tile_gemm(blockIdx.y * TM, blockIdx.x * TN, Co, HW, 0, Ci,
              [&](int ci, int co) { return w[((long)ci * Co + co) * 4 + t]; },
              [&](int ci, int p) { return xn[(long)ci * HW + p]; },
              [&](int co, int p, float v) {
                  const int h = p / W, ww = p - h * W;
                  yn[(long)co * 4 * HW + (long)(2 * h + kh) * 2 * W + 2 * ww + kw] = v + (b ? b[co] : 0.f);
              });
}

__global__ __launch_bounds__(256) void convt_dgrad_kernel(const float* __restrict__ dy, long dy_bs,
                                                          const float* __restrict__ w, float* __restrict__ dx,
                                                          long dx_bs, int Ci, int Co, int H, int W) {
    const int HW = H * W, n = blockIdx.z;
    const float* gn = dy + (long)n * dy_bs;
    float* dn = dx + (long)n * dx_bs;
    tile_gemm(blockIdx.y * TM, blockIdx.x * TN, Ci, HW, 0, Co * 4,
              [&](int k, int ci) { return w[(long)ci * Co * 4 + k]; },
              [&](int k, int p) {
                  const int co = k >> 2, t = k & 3, h = p / W, ww = p - h * W;
                  return gn[(long)co * 4 * HW + (long)(2 * h + (t >> 1)) * 2 * W + 2 * ww + (t & 1)];
              },
              [&](int ci, int p, float v) { dn[(long)ci * HW + p] = v; });
}

// slab[split][ci][co*4+t] = sum over this split's (n, pixel) range
__global__ __launch_bounds__(256) void convt_wgrad_kernel(const float* __restrict__ x, long x_bs,
                                                          const float* __restrict__ dy, long dy_bs,
                                                          float* __restrict__ slabs, int N, int Ci, int Co, int H,
                                                          int W, int splits) {
    const int HW = H * W, s = blockIdx.z;
    const long K = (long)N * HW;
    const long per = ((K + splits - 1) / splits + TK - 1) / TK * TK;
    const int k0 = (int)min((long)s * per, K), k1 = (int)min((long)(s + 1) * per, K);
    float* slab = slabs + (long)s * Ci * Co * 4;
    tile_gemm(blockIdx.y * TM, blockIdx.x * TN, Ci, Co * 4, k0, k1,
              [&](int k, int ci) { const int n = k / HW, p = k - n * HW; return x[(long)n * x_bs + (long)ci * HW + p]; },
              [&](int k, int j) {
                  const int n = k / HW, p = k - n * HW, co = j >> 2, t = j & 3, h = p / W, ww = p - h * W;
                  return dy[(long)n * dy_bs + (long)co * 4 * HW + (long)(2 * h + (t >> 1)) * 2 * W + 2 * ww + (t & 1)];
              },
              [&](int ci, int j, float v) { slab[(long)ci * Co * 4 + j] = v; });
}

__global__ void slab_sum_kernel(const float* __restrict__ slabs, int splits, long n, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = slabs[i];
        for (int s = 1; s < splits; ++s) v += slabs[(long)s * n + i];
        out[i] = v;
    }
}

int wgrad_splits(int N, int Ci, int Co, int H, int W) {
    const long tiles = (long)((Ci + TM - 1) / TM) * ((Co * 4 + TN - 1) / TN);
    long s = (1024 + tiles - 1) / tiles;
    const long kmax = ((long)N * H * W + TK - 1) / TK;
    if (s > kmax) s = kmax;
    return (int)max(1L, min(s, 256L));
}

}  // namespace

extern "C" {

// x: [N][Ci][H][W] -> y: [N][Co][2H][2W];  w: [Ci][Co][2][2]
int aide_convT2x2_fwd(const float* x, int64_t x_bs, const float* w, const float* b, float* y, int64_t y_bs,
                      int N, int Ci, int Co, int H, int W, hipStream_t stream) {
    const int HW = H * W;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, convt_fwd_kernel, dim3((HW + TN - 1) / TN, (Co + TM - 1) / TM, N * 4), dim3(256), 0,
                       stream, x, (long)x_bs, w, b, y, (long)y_bs, Ci, Co, H, W);
    return aide_launch_status();
}

// dy: [N][Co][2H][2W] -> dx: [N][Ci][H][W]
int aide_convT2x2_dgrad(const float* dy, int64_t dy_bs, const float* w, float* dx, int64_t dx_bs, int N,
                        int Ci, int Co, int H, int W, hipStream_t stream) {
    const int HW = H * W;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, convt_dgrad_kernel, dim3((HW + TN - 1) / TN, (Ci + TM - 1) / TM, N), dim3(256), 0,
                       stream, dy, (long)dy_bs, w, dx, (long)dx_bs, Ci, Co, H, W);
    return aide_launch_status();
}

size_t aide_convT2x2_wgrad_ws_bytes(int N, int Ci, int Co, int H, int W) {
    return (size_t)wgrad_splits(N, Ci, Co, H, W) * Ci * Co * 4 * sizeof(float);
}

int aide_convT2x2_wgrad(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, float* dw, int N,
                        int Ci, int Co, int H, int W, float* ws, hipStream_t stream) {
    if (!ws) return AIDE_ERR_ARG;
    const int splits = wgrad_splits(N, Ci, Co, H, W);
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, convt_wgrad_kernel, dim3((Co * 4 + TN - 1) / TN, (Ci + TM - 1) / TM, splits), dim3(256),
                       0, stream, x, (long)x_bs, dy, (long)dy_bs, ws, N, Ci, Co, H, W, splits);
    const long n = (long)Ci * Co * 4;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, slab_sum_kernel, dim3((unsigned)min((n + 255) / 256, 2048L)), dim3(256), 0, stream, ws,
                       splits, n, dw);
    return aide_launch_status();
}

}  // extern "C"
