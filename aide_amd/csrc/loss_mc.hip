// The fused segmentation-loss kernels of loss.hip for num_classes = 3 .. 8 (gfx950).
//
// The shipped scripts of the reference run num_classes = 2 (train_files/trainchaos_comparison_1case.py:121) and loss.hip
// is specialised for it (one sigmoid per pixel).  The reference's modules themselves are written for any C
// (`fuseunet(num_classes=...)`, models_twomodalinputs/fuseunet.py:7,41; `nn.CrossEntropyLoss(weight)`, utils/loss2d.py:8;
// `F.softmax(inputs, dim=1)[:, 1]`, utils/loss2d.py:44-46,65-66,96,106; `MulticlassMSELoss`, utils/loss2d.py:115-117;
// `sharpen`, train_files/trainchaos_proposed_30cases1labeled.py:97-101; `torch.argmax(softmax)`,
// trainchaos_comparison_1case.py:262-264).  These kernels are that general form:
//   p = softmax(z) over C planes, lse = log sum exp;  cross-entropy l = w_t (lse - z_t);
//   Dice and the hard-Dice metrics on p_1 against the target INDEX taken as a number (`tflat = target.float()`,
//   utils/loss2d.py:50 -- with C > 2 an index 2 counts twice, as in the reference);
//   consistency term sum_c wm (p_c - q_c)^2; weight map 1 - 4 q_0 q_1 (proposed loop :285-288).
// They write the SAME eight per-image statistics as seg_stats_kernel, so the finalize kernels of loss.hip (loss values,
// stable argsort, small-loss selection, backward coefficients) serve both; the backward is again one elementwise pass:
//   dz_c = g [ cce w_t (p_c - [c = t]) + cdice dDice/dp_1 p_1 ([c = 1] - p_c)
//              + cmse wm 2 p_c ((p_c - q_c) - sum_j (p_j - q_j) p_j) ].
// Partial sums are fp64 in a fixed order (no atomics): bit-reproducible like the two-class path.
#include "common.h"

extern "C" int aide_seg_loss_blocks(int HW);

namespace {

constexpr int NS = 8;   // statistics per image: the layout of loss.hip
enum { S_CE = 0, S_W, S_I, S_P, S_T, S_M, S_HP, S_HI };
constexpr int MAXC = 8;

struct ClassW { float w[MAXC]; };

template <int C>
__device__ __forceinline__ void soft_terms(const float* __restrict__ zp, int HW, float (&z)[C], float (&p)[C], float& lse) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) { z[c] = zp[(long)c * HW]; m = fmaxf(m, z[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = expf(z[c] - m); s += p[c]; }
    const float inv = 1.0f / s;
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] *= inv;
    lse = m + logf(s);
}

template <int C>
__device__ __forceinline__ float pick(const float (&v)[C], int k) {   // v[k] without dynamic register indexing
    float r = v[0];
#pragma unroll
    for (int c = 1; c < C; ++c) r = (k == c) ? v[c] : r;
    return r;
}

template <int C>
__global__ __launch_bounds__(256) void seg_stats_mc_kernel(
    const float* __restrict__ logits, long l_bs, const long long* __restrict__ targets, long t_bs, const ClassW cw,
    int ignore_index, const float* __restrict__ pseudo, long p_bs, const float* __restrict__ wmap, long w_bs, int HW,
    double* __restrict__ partials) {
    __shared__ double sm[NS * 4];
    const int n = blockIdx.y, b = blockIdx.x, bpi = gridDim.x;
    const float* zn = logits + (long)n * l_bs;
    const long long* t = targets + (long)n * t_bs;
    const float* qn = pseudo ? pseudo + (long)n * p_bs : nullptr;
    const float* wm = wmap ? wmap + (long)n * w_bs : nullptr;
    float w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = cw.w[c];
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    for (int i = b * 256 + threadIdx.x; i < HW; i += bpi * 256) {
        float z[C], p[C], lse;
        soft_terms<C>(zn + i, HW, z, p, lse);
        const long long tv = t[i];
        const float tf = (float)tv;
        if (tv != ignore_index && tv >= 0 && tv < C) {
            const float wt = pick<C>(w, (int)tv);
            acc[S_CE] += (double)(wt * (lse - pick<C>(z, (int)tv)));
            acc[S_W] += (double)wt;
        }
        const float p1 = p[1];
        acc[S_I] += (double)(p1 * tf);
        acc[S_P] += (double)p1;
        acc[S_T] += (double)tf;
        if (qn) {
            float e2 = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) { const float e = p[c] - qn[(long)c * HW + i]; e2 += e * e; }
            acc[S_M] += (double)((wm ? wm[i] : 1.0f) * e2);
        }
        if (p1 >= 0.5f) {
            acc[S_HP] += 1.0;
            acc[S_HI] += (double)tf;
        }
    }
    block_sum_d<NS>(acc, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) partials[((long)n * bpi + b) * NS + k] = acc[k];
    }
}

template <int C>
__global__ __launch_bounds__(256) void seg_loss_bwd_mc_kernel(
    const float* __restrict__ logits, long l_bs, const long long* __restrict__ targets, long t_bs, const ClassW cw,
    int ignore_index, const float* __restrict__ pseudo, long p_bs, const float* __restrict__ wmap, long w_bs, int HW,
    const double* __restrict__ stats, const float* __restrict__ coef, int N, float smooth,
    const float* __restrict__ gout, int g_stride, float* __restrict__ dlogits, long d_bs) {
    const int n = blockIdx.y;
    const float* zn = logits + (long)n * l_bs;
    const long long* t = targets + (long)n * t_bs;
    const float* qn = pseudo ? pseudo + (long)n * p_bs : nullptr;
    const float* wm = wmap ? wmap + (long)n * w_bs : nullptr;
    const double* S = stats + n * NS;
    const float g = gout[n * g_stride];
    const float cce = g * coef[n], cdice = g * coef[N + n], cmse = g * coef[2 * N + n];
    const float den = (float)(S[S_P] + S[S_T] + (double)smooth);
    const float num = (float)(2.0 * S[S_I] + (double)smooth);
    const float inv_den2 = 1.0f / (den * den);
    float w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = cw.w[c];
    float* on = dlogits + (long)n * d_bs;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float z[C], p[C], lse;
        soft_terms<C>(zn + i, HW, z, p, lse);
        const long long tv = t[i];
        const float tf = (float)tv;
        const bool on_ce = tv != ignore_index && tv >= 0 && tv < C;
        const float wce = on_ce ? cce * pick<C>(w, (int)tv) : 0.f;
        const float gd = cdice * (-(2.0f * tf * den - num) * inv_den2) * p[1];     // x ([c = 1] - p_c)
        float e[C], ep = 0.f, mw = 0.f;
        if (qn && cmse != 0.f) {
            mw = cmse * (wm ? wm[i] : 1.0f) * 2.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) { e[c] = p[c] - qn[(long)c * HW + i]; ep += e[c] * p[c]; }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) e[c] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float d = wce * (p[c] - ((int)tv == c ? 1.0f : 0.0f));
            d += gd * ((c == 1 ? 1.0f : 0.0f) - p[c]);
            d += mw * p[c] * (e[c] - ep);
            on[(long)c * HW + i] = d;
        }
    }
}

// reduction='none' cross-entropy map [N][HW] and its backward
template <int C>
__global__ __launch_bounds__(256) void ce_map_mc_kernel(const float* __restrict__ logits, long l_bs,
                                                        const long long* __restrict__ targets, long t_bs, const ClassW cw,
                                                        int ignore_index, int HW, float* __restrict__ out,
                                                        const float* __restrict__ gout, float* __restrict__ dlogits,
                                                        long d_bs) {
    const int n = blockIdx.y;
    const float* zn = logits + (long)n * l_bs;
    const long long* t = targets + (long)n * t_bs;
    float w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = cw.w[c];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float z[C], p[C], lse;
        soft_terms<C>(zn + i, HW, z, p, lse);
        const long long tv = t[i];
        const bool on = tv != ignore_index && tv >= 0 && tv < C;
        const float wt = on ? pick<C>(w, (int)tv) : 0.f;
        if (!gout) {
            out[(long)n * HW + i] = on ? wt * (lse - pick<C>(z, (int)tv)) : 0.f;
        } else {
            const float gw = gout[(long)n * HW + i] * wt;
#pragma unroll
            for (int c = 0; c < C; ++c)
                dlogits[(long)n * d_bs + (long)c * HW + i] = gw * (p[c] - ((int)tv == c ? 1.0f : 0.0f));
        }
    }
}

// MulticlassMSELoss(reduction='none'): out[n][c][i] = (softmax_c - target_c)^2 and its backward
template <int C>
__global__ __launch_bounds__(256) void mse_map_mc_kernel(const float* __restrict__ logits, long l_bs,
                                                         const float* __restrict__ target, long q_bs, int HW,
                                                         float* __restrict__ out, const float* __restrict__ gout,
                                                         float* __restrict__ dlogits, long d_bs) {
    const int n = blockIdx.y;
    const float* zn = logits + (long)n * l_bs;
    const float* q = target + (long)n * q_bs;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float z[C], p[C], lse;
        soft_terms<C>(zn + i, HW, z, p, lse);
        if (!gout) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float e = p[c] - q[(long)c * HW + i];
                out[(long)n * C * HW + (long)c * HW + i] = e * e;
            }
        } else {
            float ge[C], s = 0.f;                    // d/dz_c = 2 p_c (g_c e_c - sum_j g_j e_j p_j)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                ge[c] = gout[(long)n * C * HW + (long)c * HW + i] * (p[c] - q[(long)c * HW + i]);
                s += ge[c] * p[c];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) dlogits[(long)n * d_bs + (long)c * HW + i] = 2.0f * p[c] * (ge[c] - s);
        }
    }
}

struct PseudoMcArgs {
    const float* logits[8];
    int K, HW;
    long l_bs;
    float temperature;
    float* pl;      // [N][C][HW]
    float* wm;      // [N][HW]
};

// mean softmax over K passes -> sharpen (p^T / sum p^T) -> weight map 1 - 4 q_0 q_1
template <int C>
__global__ __launch_bounds__(256) void pseudo_label_mc_kernel(const PseudoMcArgs a) {
    const int n = blockIdx.y, HW = a.HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float s[C];
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] = 0.f;
        for (int k = 0; k < a.K; ++k) {
            float z[C], p[C], lse;
            soft_terms<C>(a.logits[k] + (long)n * a.l_bs + i, HW, z, p, lse);
#pragma unroll
            for (int c = 0; c < C; ++c) s[c] += p[c];
        }
        float tot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            s[c] /= (float)a.K;
            if (a.temperature != 1.0f) s[c] = powf(s[c], a.temperature);
            tot += s[c];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            s[c] /= tot;
            a.pl[(long)n * C * HW + (long)c * HW + i] = s[c];
        }
        a.wm[(long)n * HW + i] = 1.0f - 4.0f * s[0] * s[1];
    }
}

// argmax(softmax(z), dim=1): the first index of the largest PROBABILITY -- logits closer than the softmax rounding
// merge into equal probabilities and the earlier class wins, as with the two-class kernel
template <int C>
__global__ __launch_bounds__(256) void label_map_mc_kernel(const float* __restrict__ logits, long l_bs, int HW,
                                                           long long* __restrict__ labels, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, px = i - n * HW;
        float z[C], p[C], lse;
        soft_terms<C>(logits + n * l_bs + px, HW, z, p, lse);
        int best = 0;
        float pb = p[0];
#pragma unroll
        for (int c = 1; c < C; ++c)
            if (p[c] > pb) { pb = p[c]; best = c; }
        labels[i] = best;
    }
}

// MulticlassDiceLoss with one-hot targets [N][C][HW] (utils/loss2d.py:98-104): one Dice term per class on softmax_c,
// weighted.  partials[n][b][c][3] = { sum p_c t_c, sum p_c, sum t_c }
template <int C>
__global__ __launch_bounds__(256) void dice_terms_mc_stats_kernel(const float* __restrict__ x, long x_bs,
                                                                  const float* __restrict__ t, long t_bs, int HW, int bpi,
                                                                  double* __restrict__ partials) {
    __shared__ double sm[3 * C * 4];
    const int n = blockIdx.y, b = blockIdx.x;
    const float* xn = x + (long)n * x_bs;
    const float* tn = t + (long)n * t_bs;
    double acc[3 * C];
#pragma unroll
    for (int k = 0; k < 3 * C; ++k) acc[k] = 0.0;
    for (int i = b * 256 + threadIdx.x; i < HW; i += bpi * 256) {
        float z[C], p[C], lse;
        soft_terms<C>(xn + i, HW, z, p, lse);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float tv = tn[(long)c * HW + i];
            acc[3 * c] += (double)(p[c] * tv); acc[3 * c + 1] += (double)p[c]; acc[3 * c + 2] += (double)tv;
        }
    }
    block_sum_d<3 * C>(acc, sm);
    if (threadIdx.x == 0)
        for (int k = 0; k < 3 * C; ++k) partials[((long)n * bpi + b) * 3 * C + k] = acc[k];
}

__global__ void dice_terms_mc_finalize_kernel(const double* __restrict__ partials, int N, int bpi, int C, const ClassW cw,
                                              float smooth, int reduction, double* __restrict__ stats,
                                              float* __restrict__ per_image, float* __restrict__ out) {
    for (int e = threadIdx.x; e < N * 3 * C; e += blockDim.x) {
        const int n = e / (3 * C), k = e - n * 3 * C;
        double v = 0.0;
        for (int b = 0; b < bpi; ++b) v += partials[((long)n * bpi + b) * 3 * C + k];
        stats[e] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // the reference adds the per-class REDUCED terms: mean_c1 + mean_c2 + ... -- for 'none' the per-image vectors
        double total = 0.0;
        for (int n = 0; n < N; ++n) {
            float li = 0.f;
            for (int c = 0; c < C; ++c) {
                const double* S = stats + ((long)n * C + c) * 3;
                const float I = (float)S[0], P = (float)S[1], T = (float)S[2];
                li += cw.w[c] * (1.0f - (2.0f * I + smooth) / (P + T + smooth));
            }
            per_image[n] = li;
            total += (double)li;
            if (reduction == 2) out[n] = li;
        }
        if (reduction == 0) out[0] = (float)(total / N);
        else if (reduction == 1) out[0] = (float)total;
    }
}

template <int C>
__global__ __launch_bounds__(256) void dice_terms_mc_bwd_kernel(const float* __restrict__ x, long x_bs,
                                                                const float* __restrict__ t, long t_bs, int HW, int N,
                                                                const double* __restrict__ stats, const ClassW cw,
                                                                float smooth, int reduction, const float* __restrict__ g,
                                                                float* __restrict__ dx, long dx_bs) {
    const int n = blockIdx.y;
    const float gi = reduction == 2 ? g[n] : (reduction == 0 ? g[0] / (float)N : g[0]);
    float c2[C], c1[C];                         // d loss_c / d p_c = -w (2 t D - Nn) / D^2 = c1 - c2 t
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double* S = stats + ((long)n * C + c) * 3;
        const float D = (float)S[1] + (float)S[2] + smooth, Nn = 2.0f * (float)S[0] + smooth;
        const float w = cw.w[c] * gi;
        c2[c] = w * 2.0f / D;
        c1[c] = w * Nn / (D * D);
    }
    const float* xn = x + (long)n * x_bs;
    const float* tn = t + (long)n * t_bs;
    float* dn = dx + (long)n * dx_bs;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float z[C], p[C], lse;
        soft_terms<C>(xn + i, HW, z, p, lse);
        float gc[C], s = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { gc[c] = c1[c] - c2[c] * tn[(long)c * HW + i]; s += gc[c] * p[c]; }
#pragma unroll
        for (int c = 0; c < C; ++c) dn[(long)c * HW + i] = p[c] * (gc[c] - s);
    }
}

// ---- the generic co-teaching operators of utils/coteach_loss.py for C classes (the two-class forms: coteach_ext.hip) ----
// KLbidirection (:85-92): KL(p1||p2) + KL(p2||p1) = sum_c (p1_c - p2_c)(l1_c - l2_c), l = log-softmax.
//   d/dz1_c = p1_c ((l1_c - l2_c) - E1) + (p1_c - p2_c),  E1 = sum_j p1_j (l1_j - l2_j);  z2 by symmetry.
template <int C>
__device__ __forceinline__ float kl_terms(const float (&z1)[C], const float (&p1)[C], float lse1, const float (&z2)[C],
                                          const float (&p2)[C], float lse2, float (&dl)[C]) {
    float kl = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        dl[c] = (z1[c] - lse1) - (z2[c] - lse2);            // l1_c - l2_c
        kl += (p1[c] - p2[c]) * dl[c];
    }
    return kl;
}
template <int C>
__device__ __forceinline__ void kl_grads(const float (&p1)[C], const float (&p2)[C], const float (&dl)[C], float (&g1)[C],
                                         float (&g2)[C]) {
    float e1 = 0.f, e2 = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { e1 += p1[c] * dl[c]; e2 -= p2[c] * dl[c]; }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        g1[c] = p1[c] * (dl[c] - e1) + (p1[c] - p2[c]);
        g2[c] = p2[c] * (-dl[c] - e2) + (p2[c] - p1[c]);
    }
}

template <int C>
__global__ __launch_bounds__(256) void kl_map_mc_kernel(const float* __restrict__ z1, long b1, const float* __restrict__ z2,
                                                        long b2, int HW, long total, float* __restrict__ out,
                                                        const float* __restrict__ gout, float* __restrict__ g1, long gb1,
                                                        float* __restrict__ g2, long gb2) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, p = i - n * HW;
        float a[C], pa[C], la, b[C], pb[C], lb, dl[C];
        soft_terms<C>(z1 + n * b1 + p, HW, a, pa, la);
        soft_terms<C>(z2 + n * b2 + p, HW, b, pb, lb);
        const float kl = kl_terms<C>(a, pa, la, b, pb, lb, dl);
        if (out) out[i] = kl;
        if (gout) {
            float ga[C], gb[C];
            kl_grads<C>(pa, pb, dl, ga, gb);
            const float g = gout[i];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                g1[n * gb1 + (long)c * HW + p] = g * ga[c];
                g2[n * gb2 + (long)c * HW + p] = g * gb[c];
            }
        }
    }
}

// Coteachingloss_dropregionce (:163-196): cross entropy of the 2x2 max-pooled logits (per class plane) against the 2x2
// max-pooled target.  aux word: bits 2c..2c+1 the window index of class c's maximum (first maximum in scan order, like
// max_pool2d), bits 16-23 the pooled target, bit 24 ignored.
template <int C>
__global__ __launch_bounds__(256) void region_ce_mc_kernel(const float* __restrict__ z, long zb,
                                                           const long long* __restrict__ t, long tb, int H, int W,
                                                           int ignore_index, long total, float* __restrict__ loss,
                                                           unsigned* __restrict__ aux) {
    const int Wp = W / 2, P = (H / 2) * Wp, HW = H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / P;
        const int p = (int)(i - n * P), ph = p / Wp, pw = p - ph * Wp;
        const long o = (long)(2 * ph) * W + 2 * pw;
        float m[C];
        unsigned a = 0;
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float* q = z + n * zb + (long)c * HW + o;
            const float v0 = q[0], v1 = q[1], v2 = q[W], v3 = q[W + 1];
            m[c] = v0; unsigned am = 0;
            if (v1 > m[c]) { m[c] = v1; am = 1; }
            if (v2 > m[c]) { m[c] = v2; am = 2; }
            if (v3 > m[c]) { m[c] = v3; am = 3; }
            a |= am << (2 * c);
            mx = fmaxf(mx, m[c]);
        }
        const long long* tq = t + n * tb + o;
        long long tp = tq[0];
        tp = tq[1] > tp ? tq[1] : tp; tp = tq[W] > tp ? tq[W] : tp; tp = tq[W + 1] > tp ? tq[W + 1] : tp;
        const bool ign = tp == ignore_index || tp < 0 || tp >= C;
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) se += expf(m[c] - mx);
        loss[i] = ign ? 0.0f : (mx + logf(se)) - pick<C>(m, (int)tp);
        aux[i] = a | ((unsigned)(tp & 0xff) << 16) | ((unsigned)ign << 24);
    }
}

template <int C>
__global__ __launch_bounds__(256) void region_ce_bwd_mc_kernel(const float* __restrict__ z, long zb,
                                                               const unsigned* __restrict__ aux,
                                                               const unsigned char* __restrict__ mask,
                                                               const float* __restrict__ coeff, int H, int W, long total,
                                                               float* __restrict__ dz, long db) {
    const int Wp = W / 2, P = (H / 2) * Wp, HW = H * W;
    const float cf = coeff[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / P;
        const int p = (int)(i - n * P), ph = p / Wp, pw = p - ph * Wp;
        const long o = (long)(2 * ph) * W + 2 * pw;
        const unsigned a = aux[i];
        const int tp = (int)((a >> 16) & 0xff);
        const bool on = mask[i] && !((a >> 24) & 1);
        float m[C], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int am = (a >> (2 * c)) & 3;
            m[c] = z[n * zb + (long)c * HW + o + (am >> 1) * W + (am & 1)];
            mx = fmaxf(mx, m[c]);
        }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { m[c] = expf(m[c] - mx); se += m[c]; }
        const float inv = 1.0f / se;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int am = (a >> (2 * c)) & 3;
            const float g = on ? cf * (m[c] * inv - (c == tp ? 1.0f : 0.0f)) : 0.0f;
            float* d = dz + n * db + (long)c * HW + o;
#pragma unroll
            for (int k = 0; k < 4; ++k) d[(k >> 1) * W + (k & 1)] = (k == am) ? g : 0.0f;
        }
    }
}

// Coteachingloss_dropimagedroppixel, pixel term on the dropped images (:221-252):
// v[m][p] = target * (KL(z1, z2) + CE(z_which, target)) for the images idx[m], the target index taken as a number
template <int C>
__global__ __launch_bounds__(256) void droppixel_map_mc_kernel(const float* __restrict__ z1, long b1,
                                                               const float* __restrict__ z2, long b2,
                                                               const long long* __restrict__ t, long tb,
                                                               const long long* __restrict__ idx, int HW, int which,
                                                               int ignore_index, float* __restrict__ v) {
    const int m = blockIdx.y;
    const long n = idx[m];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const long long tg = t[n * tb + p];
        float r = 0.0f;
        if (tg != 0) {
            float a[C], pa[C], la, b[C], pb[C], lb, dl[C];
            soft_terms<C>(z1 + n * b1 + p, HW, a, pa, la);
            soft_terms<C>(z2 + n * b2 + p, HW, b, pb, lb);
            float ce = 0.f;
            if (tg != ignore_index && tg > 0 && tg < C) ce = which ? lb - pick<C>(b, (int)tg) : la - pick<C>(a, (int)tg);
            r = (kl_terms<C>(a, pa, la, b, pb, lb, dl) + ce) * (float)tg;
        }
        v[(long)m * HW + p] = r;
    }
}

// g1 / g2 [N][C][HW] (pre-zeroed): coeff * mask * d v / d logits on the images idx[m]
template <int C>
__global__ __launch_bounds__(256) void droppixel_bwd_mc_kernel(const float* __restrict__ z1, long b1,
                                                               const float* __restrict__ z2, long b2,
                                                               const long long* __restrict__ t, long tb,
                                                               const long long* __restrict__ idx, int HW, int which,
                                                               int ignore_index, const unsigned char* __restrict__ mask,
                                                               const float* __restrict__ coeff, float* __restrict__ g1,
                                                               float* __restrict__ g2) {
    const int m = blockIdx.y;
    const long n = idx[m];
    const float cf = coeff[0];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        if (!mask[(long)m * HW + p]) continue;
        const long long tg = t[n * tb + p];
        float a[C], pa[C], la, b[C], pb[C], lb, dl[C], ga[C], gb[C];
        soft_terms<C>(z1 + n * b1 + p, HW, a, pa, la);
        soft_terms<C>(z2 + n * b2 + p, HW, b, pb, lb);
        kl_terms<C>(a, pa, la, b, pb, lb, dl);
        kl_grads<C>(pa, pb, dl, ga, gb);
        const bool ce_on = tg != ignore_index && tg > 0 && tg < C;
        const float s = cf * (float)tg;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float hot = ((int)tg == c) ? 1.0f : 0.0f;
            float x1 = ga[c], x2 = gb[c];
            if (ce_on) { if (which) x2 += pb[c] - hot; else x1 += pa[c] - hot; }
            g1[(n * C + c) * HW + p] = s * x1;
            g2[(n * C + c) * HW + p] = s * x2;
        }
    }
}

bool load_w(const float* class_w, int C, ClassW& cw) {
    if (C < 3 || C > MAXC) return false;
    for (int c = 0; c < MAXC; ++c) cw.w[c] = (class_w && c < C) ? class_w[c] : 1.0f;
    return true;
}

}  // namespace

// one instantiation per class count
#define AIDE_MC_SWITCH(C, LAUNCH)                                                     \
    switch (C) {                                                                      \
        case 3: { LAUNCH(3); break; } case 4: { LAUNCH(4); break; } case 5: { LAUNCH(5); break; }  \
        case 6: { LAUNCH(6); break; } case 7: { LAUNCH(7); break; } default: { LAUNCH(8); break; } \
    }

extern "C" {

int aide_seg_max_classes(void) { return MAXC; }

// class_w: HOST array of C floats (NULL = all ones); partials as aide_seg_stats (aide_seg_loss_ws_bytes)
int aide_seg_stats_mc(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs, const float* class_w,
                      int C, int ignore_index, const float* pseudo, int64_t p_bs, const float* wmap, int64_t w_bs,
                      int N, int HW, double* partials, hipStream_t stream) {
    ClassW cw;
    if (!logits || !targets || !partials || N <= 0 || HW <= 0 || !load_w(class_w, C, cw)) return AIDE_ERR_ARG;
    const dim3 grid(aide_seg_loss_blocks(HW), N);
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, seg_stats_mc_kernel<CC>, grid, dim3(256), 0, stream, logits, (long)l_bs, targets, \
                                 (long)t_bs, cw, ignore_index, pseudo, (long)p_bs, wmap, (long)w_bs, HW, partials)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

int aide_seg_loss_bwd_mc(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs,
                         const float* class_w, int C, int ignore_index, const float* pseudo, int64_t p_bs,
                         const float* wmap, int64_t w_bs, int N, int HW, const double* stats, const float* coef,
                         float smooth, const float* gout, int g_stride, float* dlogits, int64_t d_bs,
                         hipStream_t stream) {
    ClassW cw;
    if (!logits || !targets || !stats || !coef || !gout || !dlogits || !load_w(class_w, C, cw)) return AIDE_ERR_ARG;
    const dim3 grid(aide_seg_loss_blocks(HW) * 4, N);
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, seg_loss_bwd_mc_kernel<CC>, grid, dim3(256), 0, stream, logits, (long)l_bs, targets, \
                                 (long)t_bs, cw, ignore_index, pseudo, (long)p_bs, wmap, (long)w_bs, HW, stats, coef, N, \
                                 smooth, gout, g_stride, dlogits, (long)d_bs)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

// gout == NULL: forward (writes out[N][HW]); else backward (writes dlogits[N][C][HW])
int aide_ce_map_mc(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs, const float* class_w,
                   int C, int ignore_index, int N, int HW, float* out, const float* gout, float* dlogits, int64_t d_bs,
                   hipStream_t stream) {
    ClassW cw;
    if (!logits || !targets || (!out && !gout) || !load_w(class_w, C, cw)) return AIDE_ERR_ARG;
    const dim3 grid(aide_seg_loss_blocks(HW) * 4, N);
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, ce_map_mc_kernel<CC>, grid, dim3(256), 0, stream, logits, (long)l_bs, targets, \
                                 (long)t_bs, cw, ignore_index, HW, out, gout, dlogits, (long)d_bs)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

int aide_mse_map_mc(const float* logits, int64_t l_bs, const float* target, int64_t q_bs, int C, int N, int HW,
                    float* out, const float* gout, float* dlogits, int64_t d_bs, hipStream_t stream) {
    if (!logits || !target || (!out && !gout) || C < 3 || C > MAXC) return AIDE_ERR_ARG;
    const dim3 grid(aide_seg_loss_blocks(HW) * 4, N);
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, mse_map_mc_kernel<CC>, grid, dim3(256), 0, stream, logits, (long)l_bs, target, \
                                 (long)q_bs, HW, out, gout, dlogits, (long)d_bs)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

int aide_label_map_mc(const float* logits, int64_t l_bs, int C, int N, int HW, long long* labels, hipStream_t stream) {
    if (!logits || !labels || N <= 0 || HW <= 0 || C < 3 || C > MAXC) return AIDE_ERR_ARG;
    const long total = (long)N * HW;
    const dim3 grid((unsigned)min((total + 255) / 256, 8192L));
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, label_map_mc_kernel<CC>, grid, dim3(256), 0, stream, logits, (long)l_bs, HW, labels, total)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

int aide_pseudo_label_mc(const float* const* logits, int K, int C, int64_t l_bs, int N, int HW, float temperature,
                         float* pl, float* wm, hipStream_t stream) {
    if (!logits || K < 1 || K > 8 || C < 3 || C > MAXC || !pl || !wm) return AIDE_ERR_ARG;
    PseudoMcArgs a;
    for (int k = 0; k < 8; ++k) a.logits[k] = k < K ? logits[k] : nullptr;
    a.K = K; a.HW = HW; a.l_bs = (long)l_bs; a.temperature = temperature; a.pl = pl; a.wm = wm;
    const dim3 grid(aide_seg_loss_blocks(HW) * 4, N);
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, pseudo_label_mc_kernel<CC>, grid, dim3(256), 0, stream, a)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

size_t aide_dice_terms_mc_ws_bytes(int N, int HW, int C) {
    return ((size_t)N * aide_seg_loss_blocks(HW) * 3 * C + (size_t)N * 3 * C) * sizeof(double);
}

// x = logits [N][C][HW], t = one-hot targets [N][C][HW] (fp32); ws: aide_dice_terms_mc_ws_bytes, its tail holds the
// per-image sums the backward needs
int aide_dice_terms_mc_fwd(const float* x, int64_t x_bs, const float* t, int64_t t_bs, int N, int HW, int C,
                           const float* class_w, float smooth, int reduction, double* ws, float* per_image, float* out,
                           hipStream_t stream) {
    ClassW cw;
    if (!x || !t || !ws || !per_image || !out || N <= 0 || HW <= 0 || reduction < 0 || reduction > 2 ||
        !load_w(class_w, C, cw))
        return AIDE_ERR_ARG;
    const int bpi = aide_seg_loss_blocks(HW);
    double* stats = ws + (size_t)N * bpi * 3 * C;
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, dice_terms_mc_stats_kernel<CC>, dim3(bpi, N), dim3(256), 0, stream, x, (long)x_bs, t, \
                                 (long)t_bs, HW, bpi, ws)
    AIDE_MC_SWITCH(C, L)
#undef L
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, dice_terms_mc_finalize_kernel, dim3(1), dim3(256), 0, stream, (const double*)ws, N, bpi, C, cw,
                       smooth, reduction, stats, per_image, out);
    return aide_launch_status();
}

int aide_dice_terms_mc_bwd(const float* x, int64_t x_bs, const float* t, int64_t t_bs, int N, int HW, int C,
                           const float* class_w, float smooth, int reduction, const double* ws, const float* g,
                           float* dx, int64_t dx_bs, hipStream_t stream) {
    ClassW cw;
    if (!x || !t || !ws || !g || !dx || !load_w(class_w, C, cw)) return AIDE_ERR_ARG;
    const int bpi = aide_seg_loss_blocks(HW);
    const double* stats = ws + (size_t)N * bpi * 3 * C;
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, dice_terms_mc_bwd_kernel<CC>, dim3(bpi * 4, N), dim3(256), 0, stream, x, (long)x_bs, t, \
                                 (long)t_bs, HW, N, stats, cw, smooth, reduction, g, dx, (long)dx_bs)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

int aide_kl_map_mc(const float* z1, int64_t b1, const float* z2, int64_t b2, int C, int N, int HW, float* out,
                   const float* gout, float* g1, int64_t gb1, float* g2, int64_t gb2, hipStream_t stream) {
    if (!z1 || !z2 || N <= 0 || HW <= 0 || C < 3 || C > MAXC || (!out && !gout) || (gout && (!g1 || !g2))) return AIDE_ERR_ARG;
    const long total = (long)N * HW;
    const dim3 grid((unsigned)max(1L, min((total + 255) / 256, 4096L)));
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, kl_map_mc_kernel<CC>, grid, dim3(256), 0, stream, z1, (long)b1, z2, (long)b2, HW, total, out, \
                                 gout, g1, (long)gb1, g2, (long)gb2)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

// aux: one 32-bit word per region (the two-class form packs into a byte)
int aide_region_ce_fwd_mc(const float* z, int64_t zb, const long long* t, int64_t tb, int C, int N, int H, int W,
                          int ignore_index, float* loss, unsigned* aux, hipStream_t stream) {
    if (!z || !t || !loss || !aux || N <= 0 || H % 2 || W % 2 || C < 3 || C > MAXC) return AIDE_ERR_ARG;
    const long total = (long)N * (H / 2) * (W / 2);
    const dim3 grid((unsigned)max(1L, min((total + 255) / 256, 4096L)));
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, region_ce_mc_kernel<CC>, grid, dim3(256), 0, stream, z, (long)zb, t, (long)tb, H, W, \
                                 ignore_index, total, loss, aux)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

int aide_region_ce_bwd_mc(const float* z, int64_t zb, const unsigned* aux, const unsigned char* mask, const float* coeff,
                          int C, int N, int H, int W, float* dz, int64_t db, hipStream_t stream) {
    if (!z || !aux || !mask || !coeff || !dz || N <= 0 || H % 2 || W % 2 || C < 3 || C > MAXC) return AIDE_ERR_ARG;
    const long total = (long)N * (H / 2) * (W / 2);
    const dim3 grid((unsigned)max(1L, min((total + 255) / 256, 4096L)));
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, region_ce_bwd_mc_kernel<CC>, grid, dim3(256), 0, stream, z, (long)zb, aux, mask, coeff, H, W, \
                                 total, dz, (long)db)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

int aide_droppixel_map_mc(const float* z1, int64_t b1, const float* z2, int64_t b2, const long long* t, int64_t tb,
                          const long long* idx, int ndrop, int C, int HW, int which, int ignore_index, float* v,
                          hipStream_t stream) {
    if (!z1 || !z2 || !t || !idx || !v || ndrop <= 0 || HW <= 0 || C < 3 || C > MAXC) return AIDE_ERR_ARG;
    const dim3 grid(min((HW + 255) / 256, 256), ndrop);
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, droppixel_map_mc_kernel<CC>, grid, dim3(256), 0, stream, z1, (long)b1, z2, (long)b2, t, \
                                 (long)tb, idx, HW, which, ignore_index, v)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

int aide_droppixel_bwd_mc(const float* z1, int64_t b1, const float* z2, int64_t b2, const long long* t, int64_t tb,
                          const long long* idx, int ndrop, int C, int HW, int which, int ignore_index,
                          const unsigned char* mask, const float* coeff, float* g1, float* g2, hipStream_t stream) {
    if (!z1 || !z2 || !t || !idx || !mask || !coeff || !g1 || !g2 || ndrop <= 0 || HW <= 0 || C < 3 || C > MAXC)
        return AIDE_ERR_ARG;
    const dim3 grid(min((HW + 255) / 256, 256), ndrop);
#define L(CC) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, droppixel_bwd_mc_kernel<CC>, grid, dim3(256), 0, stream, z1, (long)b1, z2, (long)b2, t, \
                                 (long)tb, idx, HW, which, ignore_index, mask, coeff, g1, g2)
    AIDE_MC_SWITCH(C, L)
#undef L
    return aide_launch_status();
}

}  // extern "C"
