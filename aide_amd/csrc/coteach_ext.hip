// The remaining co-teaching operators of utils/coteach_loss.py, two classes, gfx950:
//   KLbidirection (:85-92), Coteachingloss_dropregionce (:163-196), the pixel-level term of
//   Coteachingloss_dropimagedroppixel (:221-252).
// Building blocks: per-pixel maps with their backward, a deterministic "k smallest" selection (radix select on
// the float bits, ties by lower index like a stable argsort) that also sums a second array over the selected
// set in fp64, and the max-pooled region cross entropy with its scatter backward.  HBM-bound streaming kernels;
// no atomics on floating-point data, so results are bit-reproducible.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float d) { return 1.0f / (1.0f + expf(-d)); }
// cross entropy of a two-class logit pair for target t in {0,1}: softplus(-(+-d)), d = z1 - z0
__device__ __forceinline__ float ce2(float d, int t) {
    const float x = t ? -d : d;
    return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));
}
// KL(p1||p2) + KL(p2||p1) of two-class softmaxes = (s1 - s2)(d1 - d2)
__device__ __forceinline__ float kl2(float d1, float d2) { return (sigmoidf_(d1) - sigmoidf_(d2)) * (d1 - d2); }
__device__ __forceinline__ void kl2_grad(float d1, float d2, float& g1, float& g2) {
    const float s1 = sigmoidf_(d1), s2 = sigmoidf_(d2), dd = d1 - d2, ds = s1 - s2;
    g1 = s1 * (1.0f - s1) * dd + ds;
    g2 = -(s2 * (1.0f - s2) * dd + ds);
}

// ---------------------------------------------------------------- KLbidirection map (+ backward)
__global__ __launch_bounds__(256) void kl_map_kernel(const float* __restrict__ z1, long b1, const float* __restrict__ z2,
                                                     long b2, int HW, long total, float* __restrict__ out,
                                                     const float* __restrict__ gout, float* __restrict__ g1,
                                                     long gb1, float* __restrict__ g2, long gb2) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, p = i - n * HW;
        const float d1 = z1[n * b1 + HW + p] - z1[n * b1 + p], d2 = z2[n * b2 + HW + p] - z2[n * b2 + p];
        if (out) out[i] = kl2(d1, d2);
        if (gout) {
            float a, b;
            kl2_grad(d1, d2, a, b);
            const float g = gout[i];
            g1[n * gb1 + p] = -g * a; g1[n * gb1 + HW + p] = g * a;
            g2[n * gb2 + p] = -g * b; g2[n * gb2 + HW + p] = g * b;
        }
    }
}

// ---------------------------------------------------------------- region CE: 2x2 max-pooled logits / targets
// aux byte: bits 0-1 window index of the class-0 maximum (first maximum in scan order, like max_pool2d),
// bits 2-3 the class-1 maximum, bit 4 the pooled target, bit 5 ignored
__global__ __launch_bounds__(256) void region_ce_kernel(const float* __restrict__ z, long zb,
                                                        const long long* __restrict__ t, long tb, int H, int W,
                                                        int ignore_index, long total, float* __restrict__ loss,
                                                        unsigned char* __restrict__ aux) {
    const int Wp = W / 2, P = (H / 2) * Wp, HW = H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / P;
        const int p = (int)(i - n * P), ph = p / Wp, pw = p - ph * Wp;
        const long o = (long)(2 * ph) * W + 2 * pw;
        float m[2]; int am[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* q = z + n * zb + (long)c * HW + o;
            const float v0 = q[0], v1 = q[1], v2 = q[W], v3 = q[W + 1];
            m[c] = v0; am[c] = 0;
            if (v1 > m[c]) { m[c] = v1; am[c] = 1; }
            if (v2 > m[c]) { m[c] = v2; am[c] = 2; }
            if (v3 > m[c]) { m[c] = v3; am[c] = 3; }
        }
        const long long* tq = t + n * tb + o;
        long long tp = tq[0];
        tp = tq[1] > tp ? tq[1] : tp; tp = tq[W] > tp ? tq[W] : tp; tp = tq[W + 1] > tp ? tq[W + 1] : tp;
        const bool ign = tp == ignore_index;
        loss[i] = ign ? 0.0f : ce2(m[1] - m[0], tp != 0);
        aux[i] = (unsigned char)(am[0] | (am[1] << 2) | ((tp != 0) << 4) | (ign << 5));
    }
}

// dz = coeff * mask * dCE/d(pooled logits) scattered to the arg-max positions; every other position 0
__global__ __launch_bounds__(256) void region_ce_bwd_kernel(const float* __restrict__ z, long zb,
                                                            const unsigned char* __restrict__ aux,
                                                            const unsigned char* __restrict__ mask,
                                                            const float* __restrict__ coeff, int H, int W,
                                                            long total, float* __restrict__ dz, long db) {
    const int Wp = W / 2, P = (H / 2) * Wp, HW = H * W;
    const float cf = coeff[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / P;
        const int p = (int)(i - n * P), ph = p / Wp, pw = p - ph * Wp;
        const long o = (long)(2 * ph) * W + 2 * pw;
        const int a = aux[i], a0 = a & 3, a1 = (a >> 2) & 3, tp = (a >> 4) & 1;
        float g = 0.0f;
        if (mask[i] && !(a & 32)) {
            const float* q = z + n * zb;
            const float m0 = q[o + (a0 >> 1) * W + (a0 & 1)], m1 = q[HW + o + (a1 >> 1) * W + (a1 & 1)];
            g = cf * (sigmoidf_(m1 - m0) - (float)tp);          // d/dm1; d/dm0 = -g
        }
        float* d0 = dz + n * db + o;
        float* d1 = d0 + HW;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d0[(k >> 1) * W + (k & 1)] = (k == a0) ? -g : 0.0f;
            d1[(k >> 1) * W + (k & 1)] = (k == a1) ? g : 0.0f;
        }
    }
}

// ---------------------------------------------------------------- region CE, any pooling window and class count
// Coteachingloss_dropregionce(scale) pools logits (per class) and targets with MaxPool2d(kernel = stride = (KH, KW),
// ceil_mode=True) (utils/coteach_loss.py:171-177): ceil(H / KH) x ceil(W / KW) regions, the last row / column of windows
// clipped at the image border.  aux: C + 1 words per region -- the plane offset of every class' maximum (first maximum in
// scan order, like max_pool2d) and (pooled target | ignored << 16).
template <int MAXC>
__global__ __launch_bounds__(256) void region_ce_win_kernel(const float* __restrict__ z, long zb,
                                                            const long long* __restrict__ t, long tb, int C, int H, int W,
                                                            int KH, int KW, int ignore_index, long total,
                                                            float* __restrict__ loss, int* __restrict__ aux) {
    const int Wp = (W + KW - 1) / KW, Hp = (H + KH - 1) / KH, P = Hp * Wp, HW = H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / P;
        const int p = (int)(i - n * P), ph = p / Wp, pw = p - ph * Wp;
        const int h0 = ph * KH, w0 = pw * KW, h1 = min(h0 + KH, H), w1 = min(w0 + KW, W);
        float m[MAXC];
        for (int c = 0; c < C; ++c) {
            const float* q = z + n * zb + (long)c * HW;
            float best = q[h0 * W + w0]; int at = h0 * W + w0;
            for (int hh = h0; hh < h1; ++hh)
                for (int ww = w0; ww < w1; ++ww) {
                    const float v = q[hh * W + ww];
                    if (v > best) { best = v; at = hh * W + ww; }
                }
            m[c] = best;
            aux[i * (C + 1) + c] = at;
        }
        long long tp = t[n * tb + h0 * W + w0];
        for (int hh = h0; hh < h1; ++hh)
            for (int ww = w0; ww < w1; ++ww) { const long long v = t[n * tb + hh * W + ww]; tp = v > tp ? v : tp; }
        const bool ign = tp == ignore_index;
        float l = 0.f;
        if (!ign) {                                   // lse(m) - m[tp], the arithmetic of log_softmax + nll_loss
            float mx = m[0];
            for (int c = 1; c < C; ++c) mx = fmaxf(mx, m[c]);
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += expf(m[c] - mx);
            l = (mx + logf(s)) - m[(int)tp];
        }
        loss[i] = l;
        aux[i * (C + 1) + C] = (int)(ign ? 0 : tp) | (ign ? 1 << 16 : 0);
    }
}

// dz = coeff * mask * dCE / d(pooled logits) at the arg-max positions, 0 everywhere else (the windows tile the plane)
template <int MAXC>
__global__ __launch_bounds__(256) void region_ce_win_bwd_kernel(const float* __restrict__ z, long zb,
                                                                const int* __restrict__ aux,
                                                                const unsigned char* __restrict__ mask,
                                                                const float* __restrict__ coeff, int C, int H, int W, int KH,
                                                                int KW, long total, float* __restrict__ dz, long db) {
    const int Wp = (W + KW - 1) / KW, Hp = (H + KH - 1) / KH, P = Hp * Wp, HW = H * W;
    const float cf = coeff[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / P;
        const int p = (int)(i - n * P), ph = p / Wp, pw = p - ph * Wp;
        const int h0 = ph * KH, w0 = pw * KW, h1 = min(h0 + KH, H), w1 = min(w0 + KW, W);
        const int* a = aux + i * (C + 1);
        const int tw = a[C], tp = tw & 0xffff;
        const bool live = mask[i] && !(tw >> 16);
        float m[MAXC], g[MAXC];
        if (live) {
            float mx = -3.4e38f;
            for (int c = 0; c < C; ++c) { m[c] = z[n * zb + (long)c * HW + a[c]]; mx = fmaxf(mx, m[c]); }
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += expf(m[c] - mx);
            for (int c = 0; c < C; ++c) g[c] = cf * (expf(m[c] - mx) / s - (c == tp ? 1.f : 0.f));
        }
        for (int c = 0; c < C; ++c) {
            float* d = dz + n * db + (long)c * HW;
            for (int hh = h0; hh < h1; ++hh)
                for (int ww = w0; ww < w1; ++ww) d[hh * W + ww] = (live && hh * W + ww == a[c]) ? g[c] : 0.f;
        }
    }
}

// ---------------------------------------------------------------- k smallest of a segment (+ sum of a second array)
__device__ __forceinline__ unsigned fkey(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);     // total order of floats as unsigned keys
}

// One workgroup (1024 threads) per segment of M values.  Candidates: all values, or only those > 0.
// k = *k_in (device) if given, else (long)(rr * candidates) if rr >= 0, else k_host; clamped to the candidates.
// Selected = the k candidates with the smallest values (ties: lower index first).  Outputs: mask[M] (0/1),
// sums[seg] = fp64 sum of sum_vals over the selection, ks[seg] = k.
__global__ __launch_bounds__(1024) void select_smallest_kernel(const float* __restrict__ sel_vals,
                                                               const float* __restrict__ sum_vals, long seg_stride,
                                                               int M, long k_host, double rr,
                                                               const long long* __restrict__ k_in, int only_positive,
                                                               unsigned char* __restrict__ mask,
                                                               double* __restrict__ sums, long long* __restrict__ ks) {
    __shared__ unsigned hist[256];
    __shared__ unsigned pre[1024];
    __shared__ double red[1024];
    __shared__ unsigned sh_prefix, sh_rem;
    __shared__ long long sh_k;
    const int tid = threadIdx.x, seg = blockIdx.x;
    const float* sv = sel_vals + (long)seg * seg_stride;
    const float* uv = sum_vals + (long)seg * seg_stride;
    unsigned char* mk = mask + (long)seg * seg_stride;
    const int cpt = (M + 1023) / 1024, beg = min(tid * cpt, M), end = min(beg + cpt, M);

    // candidates and k
    unsigned cnt = 0;
    for (int i = beg; i < end; ++i) cnt += (!only_positive || sv[i] > 0.0f) ? 1u : 0u;
    pre[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        long long c = 0;
        for (int i = 0; i < 1024; ++i) c += pre[i];
        long long k = k_in ? *k_in : (rr >= 0.0 ? (long long)(rr * (double)c) : (long long)k_host);
        if (k > c) k = c;
        if (k < 0) k = 0;
        sh_k = k; sh_prefix = 0; sh_rem = (unsigned)k;
    }
    __syncthreads();
    const long long k = sh_k;
    // radix select of the k-th smallest key, 8 bits per pass
    unsigned keymask = 0;
    for (int shift = 24; shift >= 0 && k > 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = sh_prefix;
        for (int i = beg; i < end; ++i) {
            const float v = sv[i];
            if (only_positive && !(v > 0.0f)) continue;
            const unsigned key = fkey(v);
            if ((key & keymask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned rem = sh_rem, cum = 0;
            int d = 0;
            for (; d < 256; ++d) {
                if (cum + hist[d] >= rem) break;
                cum += hist[d];
            }
            sh_prefix = prefix | ((unsigned)d << shift);
            sh_rem = rem - cum;
        }
        keymask |= 255u << shift;
        __syncthreads();
    }
    const unsigned tau = sh_prefix, r_eq = sh_rem;         // take r_eq of the values equal to tau (lowest indices)
    // rank of equal keys in index order: per-thread counts -> exclusive scan
    unsigned eq = 0;
    if (k > 0)
        for (int i = beg; i < end; ++i) {
            const float v = sv[i];
            if (only_positive && !(v > 0.0f)) continue;
            eq += fkey(v) == tau ? 1u : 0u;
        }
    __syncthreads();
    pre[tid] = eq;
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;
        for (int i = 0; i < 1024; ++i) { const unsigned c = pre[i]; pre[i] = run; run += c; }
    }
    __syncthreads();
    unsigned rank = pre[tid];
    double s = 0.0;
    for (int i = beg; i < end; ++i) {
        const float v = sv[i];
        bool sel = false;
        if (k > 0 && (!only_positive || v > 0.0f)) {
            const unsigned key = fkey(v);
            if (key < tau) sel = true;
            else if (key == tau) { sel = rank < r_eq; ++rank; }
        }
        mk[i] = sel ? 1 : 0;
        if (sel) s += (double)uv[i];
    }
    red[tid] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) { sums[seg] = red[0]; ks[seg] = k; }
}

// ---------------------------------------------------------------- dropimagedroppixel: pixel term on the dropped images
// v[m][p] = target * (KL(z1, z2) + CE(zA, target)) for the images idx[m]; which = 0: CE on z1, 1: CE on z2
__global__ __launch_bounds__(256) void droppixel_map_kernel(const float* __restrict__ z1, long b1,
                                                            const float* __restrict__ z2, long b2,
                                                            const long long* __restrict__ t, long tb,
                                                            const long long* __restrict__ idx, int HW, int which,
                                                            float* __restrict__ v) {
    const int m = blockIdx.y;
    const long n = idx[m];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const long long tg = t[n * tb + p];
        float r = 0.0f;
        if (tg != 0) {
            const float d1 = z1[n * b1 + HW + p] - z1[n * b1 + p], d2 = z2[n * b2 + HW + p] - z2[n * b2 + p];
            r = (kl2(d1, d2) + ce2(which ? d2 : d1, 1)) * (float)tg;
        }
        v[(long)m * HW + p] = r;
    }
}

// g1/g2 [N][2][HW] (pre-zeroed): += coeff * mask * d v / d logits on the images idx[m]
__global__ __launch_bounds__(256) void droppixel_bwd_kernel(const float* __restrict__ z1, long b1,
                                                            const float* __restrict__ z2, long b2,
                                                            const long long* __restrict__ t, long tb,
                                                            const long long* __restrict__ idx, int HW, int which,
                                                            const unsigned char* __restrict__ mask,
                                                            const float* __restrict__ coeff, float* __restrict__ g1,
                                                            float* __restrict__ g2) {
    const int m = blockIdx.y;
    const long n = idx[m];
    const float cf = coeff[0];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        if (!mask[(long)m * HW + p]) continue;
        const float tg = (float)t[n * tb + p];
        const float d1 = z1[n * b1 + HW + p] - z1[n * b1 + p], d2 = z2[n * b2 + HW + p] - z2[n * b2 + p];
        float a, b;
        kl2_grad(d1, d2, a, b);
        const float ce = sigmoidf_(which ? d2 : d1) - 1.0f;       // d softplus(-d) / dd
        if (which) b += ce; else a += ce;
        a *= cf * tg; b *= cf * tg;
        g1[(n * 2) * HW + p] = -a; g1[(n * 2 + 1) * HW + p] = a;
        g2[(n * 2) * HW + p] = -b; g2[(n * 2 + 1) * HW + p] = b;
    }
}

// ---------------------------------------------------------------- Pixelcoreg_Focalloss (utils/reg_loss.py:58-193)
// focal(gamma = 2) cross entropy of a logit difference d = z1 - z0: t = 1: (1 - s)^2 softplus(-d), t = 0: s^2 softplus(d)
__device__ __forceinline__ float focal2(float d, int t) {
    const float s = sigmoidf_(d), q = t ? 1.0f - s : s;
    return q * q * ce2(d, t);
}
__device__ __forceinline__ float focal2_grad(float d, int t) {
    const float s = sigmoidf_(d);
    if (t) { const float q = 1.0f - s; return -q * q * (2.0f * s * ce2(d, 1) + q); }
    return s * s * (2.0f * (1.0f - s) * ce2(d, 0) + s);
}

// key = (1 - kd) * (focal1 + focal2 [+ focal3]) + kd * KL(1, 2); val = focal3 (three nets) or key (two nets)
__global__ __launch_bounds__(256) void pixelcoreg_map_kernel(const float* __restrict__ z1, const float* __restrict__ z2,
                                                             const float* __restrict__ z3,
                                                             const long long* __restrict__ t, long tb, int HW,
                                                             float kd, long total, float* __restrict__ key,
                                                             float* __restrict__ val, float* __restrict__ tf) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, p = i - n * HW;
        const int tg = t[n * tb + p] != 0;
        const float d1 = z1[(n * 2 + 1) * HW + p] - z1[n * 2 * HW + p], d2 = z2[(n * 2 + 1) * HW + p] - z2[n * 2 * HW + p];
        float f = focal2(d1, tg) + focal2(d2, tg), f3 = 0.0f;
        if (z3) { f3 = focal2(z3[(n * 2 + 1) * HW + p] - z3[n * 2 * HW + p], tg); f += f3; }
        const float k = (1.0f - kd) * f + kd * kl2(d1, d2);
        key[i] = k;
        if (z3) val[i] = f3;
        tf[i] = (float)tg;
    }
}

// gradients of coeff * sum over the selection: two nets -> g1, g2 from the key itself; three nets -> g3 from focal3
__global__ __launch_bounds__(256) void pixelcoreg_bwd_kernel(const float* __restrict__ z1, const float* __restrict__ z2,
                                                             const float* __restrict__ z3,
                                                             const long long* __restrict__ t, long tb, int HW,
                                                             float kd, long total, const unsigned char* __restrict__ mask,
                                                             const float* __restrict__ coeff, float* __restrict__ g1,
                                                             float* __restrict__ g2, float* __restrict__ g3) {
    const float cf = coeff[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, p = i - n * HW;
        const int tg = t[n * tb + p] != 0;
        const float m = mask[i] ? cf : 0.0f;
        if (z3) {
            const float a = m * focal2_grad(z3[(n * 2 + 1) * HW + p] - z3[n * 2 * HW + p], tg);
            g3[n * 2 * HW + p] = -a; g3[(n * 2 + 1) * HW + p] = a;
        } else {
            const float d1 = z1[(n * 2 + 1) * HW + p] - z1[n * 2 * HW + p], d2 = z2[(n * 2 + 1) * HW + p] - z2[n * 2 * HW + p];
            float a, b;
            kl2_grad(d1, d2, a, b);
            a = m * ((1.0f - kd) * focal2_grad(d1, tg) + kd * a);
            b = m * ((1.0f - kd) * focal2_grad(d2, tg) + kd * b);
            g1[n * 2 * HW + p] = -a; g1[(n * 2 + 1) * HW + p] = a;
            g2[n * 2 * HW + p] = -b; g2[(n * 2 + 1) * HW + p] = b;
        }
    }
}

int grid1(long total) { return (int)max(1L, min((total + 255) / 256, 4096L)); }

}  // namespace

extern "C" {

int aide_kl_map(const float* z1, int64_t b1, const float* z2, int64_t b2, int N, int HW, float* out,
                const float* gout, float* g1, int64_t gb1, float* g2, int64_t gb2, hipStream_t stream) {
    if (!z1 || !z2 || N <= 0 || HW <= 0 || (!out && !gout) || (gout && (!g1 || !g2))) return AIDE_ERR_ARG;
    const long total = (long)N * HW;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, kl_map_kernel, dim3(grid1(total)), dim3(256), 0, stream, z1, (long)b1, z2, (long)b2, HW, total,
                       out, gout, g1, (long)gb1, g2, (long)gb2);
    return aide_launch_status();
}

int aide_region_ce_fwd(const float* z, int64_t zb, const long long* t, int64_t tb, int N, int H, int W,
                       int ignore_index, float* loss, unsigned char* aux, hipStream_t stream) {
    if (!z || !t || !loss || !aux || N <= 0 || H % 2 || W % 2) return AIDE_ERR_ARG;
    const long total = (long)N * (H / 2) * (W / 2);
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, region_ce_kernel, dim3(grid1(total)), dim3(256), 0, stream, z, (long)zb, t, (long)tb, H, W,
                       ignore_index, total, loss, aux);
    return aide_launch_status();
}

int aide_region_ce_bwd(const float* z, int64_t zb, const unsigned char* aux, const unsigned char* mask,
                       const float* coeff, int N, int H, int W, float* dz, int64_t db, hipStream_t stream) {
    if (!z || !aux || !mask || !coeff || !dz || N <= 0 || H % 2 || W % 2) return AIDE_ERR_ARG;
    const long total = (long)N * (H / 2) * (W / 2);
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, region_ce_bwd_kernel, dim3(grid1(total)), dim3(256), 0, stream, z, (long)zb, aux, mask, coeff,
                       H, W, total, dz, (long)db);
    return aide_launch_status();
}

int aide_region_ce_fwd_win(const float* z, int64_t zb, const long long* t, int64_t tb, int C, int N, int H, int W, int KH,
                           int KW, int ignore_index, float* loss, int* aux, hipStream_t stream) {
    if (!z || !t || !loss || !aux || N <= 0 || C < 2 || C > 8 || KH < 1 || KW < 1 || H < 1 || W < 1 || (long)H * W >= (1L << 31))
        return AIDE_ERR_ARG;
    const long total = (long)N * ((H + KH - 1) / KH) * ((W + KW - 1) / KW);
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, region_ce_win_kernel<8>, dim3(grid1(total)), dim3(256), 0, stream, z, (long)zb, t, (long)tb, C, H, W, KH,
                       KW, ignore_index, total, loss, aux);
    return aide_launch_status();
}

int aide_region_ce_bwd_win(const float* z, int64_t zb, const int* aux, const unsigned char* mask, const float* coeff, int C,
                           int N, int H, int W, int KH, int KW, float* dz, int64_t db, hipStream_t stream) {
    if (!z || !aux || !mask || !coeff || !dz || N <= 0 || C < 2 || C > 8 || KH < 1 || KW < 1) return AIDE_ERR_ARG;
    const long total = (long)N * ((H + KH - 1) / KH) * ((W + KW - 1) / KW);
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, region_ce_win_bwd_kernel<8>, dim3(grid1(total)), dim3(256), 0, stream, z, (long)zb, aux, mask, coeff, C,
                       H, W, KH, KW, total, dz, (long)db);
    return aide_launch_status();
}

int aide_select_smallest(const float* sel_vals, const float* sum_vals, int64_t seg_stride, int nseg, int M,
                         int64_t k_host, double rr, const long long* k_in, int only_positive, unsigned char* mask,
                         double* sums, long long* ks, hipStream_t stream) {
    if (!sel_vals || !sum_vals || !mask || !sums || !ks || nseg <= 0 || M <= 0) return AIDE_ERR_ARG;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, select_smallest_kernel, dim3(nseg), dim3(1024), 0, stream, sel_vals, sum_vals, (long)seg_stride,
                       M, (long)k_host, rr, k_in, only_positive, mask, sums, ks);
    return aide_launch_status();
}

// z1, z2 (, z3 or NULL): contiguous [N][2][HW] logits.  key/val/tf: [N][HW]; val is written only with three nets
int aide_pixelcoreg_map(const float* z1, const float* z2, const float* z3, const long long* t, int64_t tb, int N,
                        int HW, float kd, float* key, float* val, float* tf, hipStream_t stream) {
    if (!z1 || !z2 || !t || !key || !tf || (z3 && !val) || N <= 0 || HW <= 0) return AIDE_ERR_ARG;
    const long total = (long)N * HW;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, pixelcoreg_map_kernel, dim3(grid1(total)), dim3(256), 0, stream, z1, z2, z3, t, (long)tb, HW, kd,
                       total, key, val, tf);
    return aide_launch_status();
}

int aide_pixelcoreg_bwd(const float* z1, const float* z2, const float* z3, const long long* t, int64_t tb, int N,
                        int HW, float kd, const unsigned char* mask, const float* coeff, float* g1, float* g2,
                        float* g3, hipStream_t stream) {
    if (!z1 || !z2 || !t || !mask || !coeff || N <= 0 || HW <= 0 || (z3 ? !g3 : (!g1 || !g2))) return AIDE_ERR_ARG;
    const long total = (long)N * HW;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, pixelcoreg_bwd_kernel, dim3(grid1(total)), dim3(256), 0, stream, z1, z2, z3, t, (long)tb, HW, kd,
                       total, mask, coeff, g1, g2, g3);
    return aide_launch_status();
}

int aide_droppixel_map(const float* z1, int64_t b1, const float* z2, int64_t b2, const long long* t, int64_t tb,
                       const long long* idx, int ndrop, int HW, int which, float* v, hipStream_t stream) {
    if (!z1 || !z2 || !t || !idx || !v || ndrop <= 0 || HW <= 0) return AIDE_ERR_ARG;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, droppixel_map_kernel, dim3(min((HW + 255) / 256, 256), ndrop), dim3(256), 0, stream, z1, (long)b1,
                       z2, (long)b2, t, (long)tb, idx, HW, which, v);
    return aide_launch_status();
}

int aide_droppixel_bwd(const float* z1, int64_t b1, const float* z2, int64_t b2, const long long* t, int64_t tb,
                       const long long* idx, int ndrop, int HW, int which, const unsigned char* mask,
                       const float* coeff, float* g1, float* g2, hipStream_t stream) {
    if (!z1 || !z2 || !t || !idx || !mask || !coeff || !g1 || !g2 || ndrop <= 0 || HW <= 0) return AIDE_ERR_ARG;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, droppixel_bwd_kernel, dim3(min((HW + 255) / 256, 256), ndrop), dim3(256), 0, stream, z1, (long)b1,
                       z2, (long)b2, t, (long)tb, idx, HW, which, mask, coeff, g1, g2);
    return aide_launch_status();
}

}  // extern "C"
