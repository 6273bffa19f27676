// Fused 2-class segmentation losses and co-teaching small-loss selection (gfx950).
//
// Replaces (reference):
//   utils/loss2d.py:5-13 CrossEntropyLoss2d, :35-61 DiceLoss, :87-107 MulticlassDiceLoss,
//   :109-117 MulticlassMSELoss, :119-135 CEMDiceLoss, :137-154 CEMDiceLossImage,
//   utils/coteach_loss.py:94-161 Coteachingloss_dropimage / _weightimage,
//   utils/metrics2d.py:8-29 Dice_fn,
//   train_files/trainchaos_proposed_30cases1labeled.py:274-292 (pseudo-label ensemble, sharpen,
//   weightmap) and :303-321 (cross-scored small-loss selection and composite loss).
//
// Math, C = 2:  d = z1 - z0, p1 = sigmoid(d), p0 = 1 - p1,
//   l = softplus(-d) if t == 1 else softplus(d).  Per image i ONE pass produces (fp64)
//   S_i = { CE = sum w_t l, Wsum = sum w_t, I = sum p1 t, P = sum p1, T = sum t,
//           M = sum wm[(p1-q1)^2 + (p0-q0)^2], hP = #(p1 >= .5), hI = sum (p1 >= .5) t }
// Everything else (loss values, stable argsort, backward coefficients) is a function of S computed
// by a one-workgroup finalize kernel; the backward is one elementwise kernel driven by three
// per-image coefficient vectors (cross-entropy, Dice, consistency-MSE).  Partial sums are combined
// in a fixed order (no atomics) so the argsort mask is reproducible bit for bit.
#include "common.h"

namespace {

constexpr int NS = 8;   // statistics per image
enum { S_CE = 0, S_W, S_I, S_P, S_T, S_M, S_HP, S_HI };

__device__ __forceinline__ void pix_terms(float z0, float z1, float& p1, float& sp_pos, float& sp_neg) {
    const float d = z1 - z0;
    const float e = expf(-fabsf(d));
    const float l1p = log1pf(e);
    sp_pos = fmaxf(d, 0.f) + l1p;        // softplus(d)  = -log p0
    sp_neg = fmaxf(-d, 0.f) + l1p;       // softplus(-d) = -log p1
    const float inv = 1.0f / (1.0f + e);
    p1 = d >= 0.f ? inv : e * inv;
}

__global__ __launch_bounds__(256) void seg_stats_kernel(
    const float* __restrict__ logits, long l_bs, const long long* __restrict__ targets, long t_bs,
    float w0, float w1, int ignore_index, const float* __restrict__ pseudo, long p_bs,
    const float* __restrict__ wmap, long w_bs, int HW, double* __restrict__ partials) {
    __shared__ double sm[NS * 4];
    const int n = blockIdx.y, b = blockIdx.x, bpi = gridDim.x;
    const float* z0 = logits + (long)n * l_bs;
    const float* z1 = z0 + HW;
    const long long* t = targets + (long)n * t_bs;
    const float* q0 = pseudo ? pseudo + (long)n * p_bs : nullptr;
    const float* wm = wmap ? wmap + (long)n * w_bs : nullptr;
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    for (int i = b * 256 + threadIdx.x; i < HW; i += bpi * 256) {
        float p1, sp_pos, sp_neg;
        pix_terms(z0[i], z1[i], p1, sp_pos, sp_neg);
        const long long tv = t[i];
        const float tf = (float)tv;
        if (tv != ignore_index) {
            const float w = tv == 1 ? w1 : w0;
            acc[S_CE] += (double)(w * (tv == 1 ? sp_neg : sp_pos));
            acc[S_W] += (double)w;
        }
        acc[S_I] += (double)(p1 * tf);
        acc[S_P] += (double)p1;
        acc[S_T] += (double)tf;
        if (q0) {
            const float p0 = 1.0f - p1;
            const float e1 = p1 - q0[HW + i], e0 = p0 - q0[i];
            acc[S_M] += (double)((wm ? wm[i] : 1.0f) * (e1 * e1 + e0 * e0));
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

__device__ void reduce_partials(const double* __restrict__ partials, int N, int bpi, double* __restrict__ stats) {
    for (int k = threadIdx.x; k < N * NS; k += blockDim.x) {
        const int n = k / NS, s = k - n * NS;
        double v = 0.0;
        for (int b = 0; b < bpi; ++b) v += partials[((long)n * bpi + b) * NS + s];
        stats[k] = v;
    }
}

__device__ __forceinline__ double dice_of(const double* S, double smooth) {
    return 1.0 - (2.0 * S[S_I] + smooth) / (S[S_P] + S[S_T] + smooth);
}

// stable ascending argsort of the fp32 values v[0..N) by rank counting (N is a batch size).  NaN has a total order
// here: larger than every number (torch.sort / np.argsort place NaN last), NaNs among themselves by index -- so the
// ranks are always a permutation of 0..N-1 and every idx slot is written, also for a diverged step.
__device__ void argsort_stable(const float* v, int N, long long* idx) {
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        int rank = 0;
        const float vi = v[i];
        const bool ni = vi != vi;
        for (int j = 0; j < N; ++j) {
            const float vj = v[j];
            const bool nj = vj != vj;
            const bool eq = (ni && nj) || (vj == vi);
            rank += (vj < vi) || (ni && !nj) || (eq && j < i);
        }
        idx[rank] = i;
    }
}

__device__ double hard_dice_sum(const double* stats, int N) {
    double s = 0.0;
    for (int i = 0; i < N; ++i) {
        const double* S = stats + i * NS;
        if (S[S_T] == 0.0) s += (S[S_HP] == 0.0) ? 1.0 : 0.0;
        else s += 2.0 * S[S_HI] / (S[S_HP] + S[S_T]);
    }
    return s;
}

struct SegFinalizeArgs {
    const double* partials;
    int N, bpi, HW, reduction;          // reduction: 0 mean, 1 sum, 2 per-image (CEMDiceLossImage)
    float w_ce, w_dice, smooth;
    double* stats;                      // [N][NS]
    float* out;                         // [1] or [N]
    float* per_image;                   // [N]   w_ce*CE_i/HW + w_dice*D_i
    long long* idx;                     // [N]   stable argsort of per_image
    float* coef;                        // [3][N] backward coefficients (ce, dice, mse)
    float* hard_dice;                   // [1]   Dice_fn sum
};

__global__ void seg_finalize_kernel(const SegFinalizeArgs a) {
    extern __shared__ float shf[];      // N floats
    reduce_partials(a.partials, a.N, a.bpi, a.stats);
    __threadfence_block();
    __syncthreads();
    const int N = a.N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const double* S = a.stats + i * NS;
        const float li = (float)((double)a.w_ce * S[S_CE] / (double)a.HW + (double)a.w_dice * dice_of(S, a.smooth));
        shf[i] = li;
        a.per_image[i] = li;
    }
    __syncthreads();
    argsort_stable(shf, N, a.idx);
    if (threadIdx.x == 0) {
        double ce = 0.0, w = 0.0, dsum = 0.0;
        for (int i = 0; i < N; ++i) {
            const double* S = a.stats + i * NS;
            ce += S[S_CE]; w += S[S_W]; dsum += dice_of(S, a.smooth);
        }
        float cce, cdice;
        if (a.reduction == 0) {
            a.out[0] = (float)((double)a.w_ce * (ce / w) + (double)a.w_dice * (dsum / N));
            cce = (float)((double)a.w_ce / w); cdice = a.w_dice / (float)N;
        } else if (a.reduction == 1) {
            a.out[0] = (float)((double)a.w_ce * ce + (double)a.w_dice * dsum);
            cce = a.w_ce; cdice = a.w_dice;
        } else {
            for (int i = 0; i < N; ++i) a.out[i] = shf[i];
            cce = a.w_ce / (float)a.HW; cdice = a.w_dice;
        }
        for (int i = 0; i < N; ++i) { a.coef[i] = cce; a.coef[N + i] = cdice; a.coef[2 * N + i] = 0.f; }
        if (a.hard_dice) a.hard_dice[0] = (float)hard_dice_sum(a.stats, N);
    }
}

struct CoteachFinalizeArgs {
    const double* partials1; const double* partials2;
    int N, bpi, HW, variant, keep;      // variant: 0 proposed inline (:303-321), 1 dropimage, 2 weightimage
    int nclass;                         // channels of the consistency map (its `.mean()` runs over D x C x HW elements)
    float w_ce, w_dice, smooth, rate, w_seg, w_cor;
    double* stats1; double* stats2;
    float* loss;                        // [2]
    float* per_image1; float* per_image2;
    long long* idx1; long long* idx2;
    float* coef1; float* coef2;         // [3][N] each
    float* hard_dice;                   // [2]
};

__global__ void coteach_finalize_kernel(const CoteachFinalizeArgs a) {
    extern __shared__ float shf[];      // 2N floats
    const int N = a.N;
    reduce_partials(a.partials1, N, a.bpi, a.stats1);
    reduce_partials(a.partials2, N, a.bpi, a.stats2);
    __threadfence_block();
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * N; i += blockDim.x) {
        const int net = i / N, k = i - net * N;
        const double* S = (net ? a.stats2 : a.stats1) + k * NS;
        const float li = (float)((double)a.w_ce * S[S_CE] / (double)a.HW + (double)a.w_dice * dice_of(S, a.smooth));
        shf[i] = li;
        (net ? a.per_image2 : a.per_image1)[k] = li;
    }
    __syncthreads();
    argsort_stable(shf, N, a.idx1);
    argsort_stable(shf + N, N, a.idx2);
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 2) {
        // net `me` is trained on the samples the OTHER net ranks as small-loss
        const int me = threadIdx.x;
        const long long* sel = me ? a.idx1 : a.idx2;
        const float* L = shf + me * N;
        const double* stats = me ? a.stats2 : a.stats1;
        float* coef = me ? a.coef2 : a.coef1;
        const int R = a.keep < N ? a.keep : N, D = N - R;      // the reference slices indx[0:keep]: at most N images
        double keep_sum = 0.0, drop_sum = 0.0, mse_sum = 0.0;
        for (int r = 0; r < N; ++r) {
            const int i = (int)sel[r];
            if (r < R) keep_sum += (double)L[i];
            else { drop_sum += (double)L[i]; mse_sum += stats[i * NS + S_M]; }
        }
        // weights of the keep-set mean, the drop-set mean and the drop-set consistency term
        double wk, wd, wm;
        if (a.variant == 0) { wk = a.w_seg; wd = (double)a.w_seg * (1.0 - (double)a.rate); wm = (double)a.w_cor * a.rate; }
        else if (a.variant == 1) { wk = 1.0; wd = 0.0; wm = 0.0; }
        else { wk = 1.0; wd = (D > 0) ? 0.1 : 0.0; wm = 0.0; }
        double loss = wk * (keep_sum / R);
        if (wd != 0.0 || a.variant == 0) loss += wd * (drop_sum / D);
        if (wm != 0.0 || a.variant == 0) loss += wm * (mse_sum / ((double)D * (double)a.nclass * a.HW));
        a.loss[me] = (float)loss;
        for (int r = 0; r < N; ++r) {
            const int i = (int)sel[r];
            const double wi = (r < R) ? wk / R : wd / D;          // d loss / d L_i
            coef[i] = (float)(wi * a.w_ce / a.HW);
            coef[N + i] = (float)(wi * a.w_dice);
            coef[2 * N + i] = (r < R) ? 0.f : (float)(wm / ((double)D * (double)a.nclass * a.HW));
        }
        if (a.hard_dice) a.hard_dice[me] = (float)hard_dice_sum(stats, N);
    }
}

// dlogits = g_n * ( cce w_t (p1 - t) + cdice dDice/dd + cmse wm 2[(p1-q1)-(p0-q0)] p1 p0 )
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(
    const float* __restrict__ logits, long l_bs, const long long* __restrict__ targets, long t_bs,
    float w0, float w1, int ignore_index, const float* __restrict__ pseudo, long p_bs,
    const float* __restrict__ wmap, long w_bs, int HW, const double* __restrict__ stats,
    const float* __restrict__ coef, int N, float smooth, const float* __restrict__ gout, int g_stride,
    float* __restrict__ dlogits, long d_bs) {
    const int n = blockIdx.y;
    const float* z0 = logits + (long)n * l_bs;
    const float* z1 = z0 + HW;
    const long long* t = targets + (long)n * t_bs;
    const float* q0 = pseudo ? pseudo + (long)n * p_bs : nullptr;
    const float* wm = wmap ? wmap + (long)n * w_bs : nullptr;
    const double* S = stats + n * NS;
    const float g = gout[n * g_stride];
    const float cce = g * coef[n], cdice = g * coef[N + n], cmse = g * coef[2 * N + n];
    const float den = (float)(S[S_P] + S[S_T] + (double)smooth);
    const float num = (float)(2.0 * S[S_I] + (double)smooth);
    const float inv_den2 = 1.0f / (den * den);
    float* o0 = dlogits + (long)n * d_bs;
    float* o1 = o0 + HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float p1, sp, sn;
        pix_terms(z0[i], z1[i], p1, sp, sn);
        const float p0 = 1.0f - p1, pp = p1 * p0;
        const long long tv = t[i];
        const float tf = (float)tv;
        float dd = 0.f;
        if (tv != ignore_index) dd += cce * (tv == 1 ? w1 : w0) * (p1 - tf);
        dd += cdice * (-(2.0f * tf * den - num) * inv_den2) * pp;
        if (q0 && cmse != 0.f) {
            const float e1 = p1 - q0[HW + i], e0 = p0 - q0[i];
            dd += cmse * (wm ? wm[i] : 1.0f) * 2.0f * (e1 - e0) * pp;
        }
        o0[i] = -dd;
        o1[i] = dd;
    }
}

// reduction='none' cross-entropy map and its backward
__global__ __launch_bounds__(256) void ce_map_kernel(const float* __restrict__ logits, long l_bs,
                                                     const long long* __restrict__ targets, long t_bs, float w0,
                                                     float w1, int ignore_index, int HW,
                                                     float* __restrict__ out /* fwd: [N][HW] */,
                                                     const float* __restrict__ gout /* bwd: [N][HW] or null */,
                                                     float* __restrict__ dlogits, long d_bs) {
    const int n = blockIdx.y;
    const float* z0 = logits + (long)n * l_bs;
    const float* z1 = z0 + HW;
    const long long* t = targets + (long)n * t_bs;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float p1, sp, sn;
        pix_terms(z0[i], z1[i], p1, sp, sn);
        const long long tv = t[i];
        const bool on = tv != ignore_index;
        const float w = tv == 1 ? w1 : w0;
        if (!gout) {
            out[(long)n * HW + i] = on ? w * (tv == 1 ? sn : sp) : 0.f;
        } else {
            const float dd = on ? gout[(long)n * HW + i] * w * (p1 - (float)tv) : 0.f;
            dlogits[(long)n * d_bs + i] = -dd;
            dlogits[(long)n * d_bs + HW + i] = dd;
        }
    }
}

// MulticlassMSELoss(reduction='none'): out[n][c][i] = (softmax_c - target_c)^2 and its backward
__global__ __launch_bounds__(256) void mse_map_kernel(const float* __restrict__ logits, long l_bs,
                                                      const float* __restrict__ target, long q_bs, int HW,
                                                      float* __restrict__ out, const float* __restrict__ gout,
                                                      float* __restrict__ dlogits, long d_bs) {
    const int n = blockIdx.y;
    const float* z0 = logits + (long)n * l_bs;
    const float* q = target + (long)n * q_bs;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float p1, sp, sn;
        pix_terms(z0[i], z0[HW + i], p1, sp, sn);
        const float p0 = 1.0f - p1;
        const float e0 = p0 - q[i], e1 = p1 - q[HW + i];
        if (!gout) {
            out[(long)n * 2 * HW + i] = e0 * e0;
            out[(long)n * 2 * HW + HW + i] = e1 * e1;
        } else {
            const float g0 = gout[(long)n * 2 * HW + i], g1 = gout[(long)n * 2 * HW + HW + i];
            const float dd = 2.0f * (g1 * e1 - g0 * e0) * p1 * p0;
            dlogits[(long)n * d_bs + i] = -dd;
            dlogits[(long)n * d_bs + HW + i] = dd;
        }
    }
}

struct PseudoArgs {
    const float* logits[8];
    int K, HW;
    long l_bs;
    float temperature;
    float* pl;      // [N][2][HW]
    float* wm;      // [N][HW]
};

// mean softmax over K passes -> sharpen (p^T / sum p^T) -> weightmap 1 - 4 p0 p1
__global__ __launch_bounds__(256) void pseudo_label_kernel(const PseudoArgs a) {
    const int n = blockIdx.y, HW = a.HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        float s0 = 0.f, s1 = 0.f;
        for (int k = 0; k < a.K; ++k) {
            const float* z = a.logits[k] + (long)n * a.l_bs;
            float p1, sp, sn;
            pix_terms(z[i], z[HW + i], p1, sp, sn);
            s1 += p1;
            s0 += 1.0f - p1;
        }
        s0 /= (float)a.K;
        s1 /= (float)a.K;
        if (a.temperature != 1.0f) { s0 = powf(s0, a.temperature); s1 = powf(s1, a.temperature); }
        const float tot = s0 + s1;
        const float q0 = s0 / tot, q1 = s1 / tot;
        a.pl[(long)n * 2 * HW + i] = q0;
        a.pl[(long)n * 2 * HW + HW + i] = q1;
        a.wm[(long)n * HW + i] = 1.0f - 4.0f * q0 * q1;
    }
}

// label map of the per-case inference loop: argmax(softmax(z), dim=1) for two classes.  The softmax is
// monotonic, but its rounding merges logits closer than ~2^-25 into equal probabilities, and argmax then
// returns the FIRST index: label 1 needs z1 > z0 *and* exp(z0 - z1) < 1 in fp32.
__global__ __launch_bounds__(256) void label_map_kernel(const float* __restrict__ logits, long l_bs, int HW,
                                                        long long* __restrict__ labels, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, p = i - n * HW;
        const float z0 = logits[n * l_bs + p], z1 = logits[n * l_bs + HW + p];
        labels[i] = (z1 > z0 && expf(z0 - z1) < 1.0f) ? 1 : 0;
    }
}

int bpi_for(int HW) { return max(1, min((HW + 2047) / 2048, 64)); }

}  // namespace

extern "C" {

int aide_seg_loss_blocks(int HW) { return bpi_for(HW); }
size_t aide_seg_loss_ws_bytes(int N, int HW) { return (size_t)N * bpi_for(HW) * NS * sizeof(double); }

// One streaming pass over logits/targets -> per-(image, block) partial statistics in `partials`.
int aide_seg_stats(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs, float w0,
                   float w1, int ignore_index, const float* pseudo, int64_t p_bs, const float* wmap,
                   int64_t w_bs, int N, int HW, double* partials, hipStream_t stream) {
    if (!logits || !targets || !partials || N <= 0 || HW <= 0) return AIDE_ERR_ARG;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, seg_stats_kernel, dim3(bpi_for(HW), N), dim3(256), 0, stream, logits, (long)l_bs,
                       targets, (long)t_bs, w0, w1, ignore_index, pseudo, (long)p_bs, wmap, (long)w_bs, HW,
                       partials);
    return aide_launch_status();
}

int aide_seg_loss_finalize(const double* partials, int N, int HW, int reduction, float w_ce, float w_dice,
                           float smooth, double* stats, float* out, float* per_image, long long* idx,
                           float* coef, float* hard_dice, hipStream_t stream) {
    SegFinalizeArgs a;
    a.partials = partials; a.N = N; a.bpi = bpi_for(HW); a.HW = HW; a.reduction = reduction;
    a.w_ce = w_ce; a.w_dice = w_dice; a.smooth = smooth; a.stats = stats; a.out = out;
    a.per_image = per_image; a.idx = idx; a.coef = coef; a.hard_dice = hard_dice;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, seg_finalize_kernel, dim3(1), dim3(256), N * sizeof(float), stream, a);
    return aide_launch_status();
}

// C = classes of the logits the statistics came from (aide_seg_stats: 2, aide_seg_stats_mc: 3 .. 8)
int aide_coteach_finalize_mc(const double* partials1, const double* partials2, int N, int HW, int C, int variant,
                             int keep, float w_ce, float w_dice, float smooth, float rate, float w_seg,
                             float w_cor, double* stats1, double* stats2, float* loss, float* per_image1,
                             float* per_image2, long long* idx1, long long* idx2, float* coef1, float* coef2,
                             float* hard_dice, hipStream_t stream) {
    if (C < 2) return AIDE_ERR_ARG;
    CoteachFinalizeArgs a;
    a.partials1 = partials1; a.partials2 = partials2; a.N = N; a.bpi = bpi_for(HW); a.HW = HW; a.nclass = C;
    a.variant = variant; a.keep = keep; a.w_ce = w_ce; a.w_dice = w_dice; a.smooth = smooth; a.rate = rate;
    a.w_seg = w_seg; a.w_cor = w_cor; a.stats1 = stats1; a.stats2 = stats2; a.loss = loss;
    a.per_image1 = per_image1; a.per_image2 = per_image2; a.idx1 = idx1; a.idx2 = idx2;
    a.coef1 = coef1; a.coef2 = coef2; a.hard_dice = hard_dice;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, coteach_finalize_kernel, dim3(1), dim3(256), 2 * N * sizeof(float), stream, a);
    return aide_launch_status();
}

int aide_coteach_finalize(const double* partials1, const double* partials2, int N, int HW, int variant,
                          int keep, float w_ce, float w_dice, float smooth, float rate, float w_seg,
                          float w_cor, double* stats1, double* stats2, float* loss, float* per_image1,
                          float* per_image2, long long* idx1, long long* idx2, float* coef1, float* coef2,
                          float* hard_dice, hipStream_t stream) {
    return aide_coteach_finalize_mc(partials1, partials2, N, HW, 2, variant, keep, w_ce, w_dice, smooth, rate, w_seg,
                                    w_cor, stats1, stats2, loss, per_image1, per_image2, idx1, idx2, coef1, coef2,
                                    hard_dice, stream);
}

int aide_seg_loss_bwd(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs, float w0,
                      float w1, int ignore_index, const float* pseudo, int64_t p_bs, const float* wmap,
                      int64_t w_bs, int N, int HW, const double* stats, const float* coef, float smooth,
                      const float* gout, int g_stride, float* dlogits, int64_t d_bs, hipStream_t stream) {
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, seg_loss_bwd_kernel, dim3(bpi_for(HW) * 4, N), dim3(256), 0, stream, logits,
                       (long)l_bs, targets, (long)t_bs, w0, w1, ignore_index, pseudo, (long)p_bs, wmap,
                       (long)w_bs, HW, stats, coef, N, smooth, gout, g_stride, dlogits, (long)d_bs);
    return aide_launch_status();
}

// gout == NULL: forward (writes out[N][HW]); else backward (writes dlogits)
int aide_ce_map(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs, float w0, float w1,
                int ignore_index, int N, int HW, float* out, const float* gout, float* dlogits, int64_t d_bs,
                hipStream_t stream) {
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, ce_map_kernel, dim3(bpi_for(HW) * 4, N), dim3(256), 0, stream, logits, (long)l_bs,
                       targets, (long)t_bs, w0, w1, ignore_index, HW, out, gout, dlogits, (long)d_bs);
    return aide_launch_status();
}

int aide_mse_map(const float* logits, int64_t l_bs, const float* target, int64_t q_bs, int N, int HW,
                 float* out, const float* gout, float* dlogits, int64_t d_bs, hipStream_t stream) {
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, mse_map_kernel, dim3(bpi_for(HW) * 4, N), dim3(256), 0, stream, logits, (long)l_bs,
                       target, (long)q_bs, HW, out, gout, dlogits, (long)d_bs);
    return aide_launch_status();
}

int aide_label_map(const float* logits, int64_t l_bs, int N, int HW, long long* labels, hipStream_t stream) {
    if (!logits || !labels || N <= 0 || HW <= 0) return AIDE_ERR_ARG;
    const long total = (long)N * HW;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, label_map_kernel, dim3((unsigned)min((total + 255) / 256, 8192L)), dim3(256), 0, stream,
                       logits, (long)l_bs, HW, labels, total);
    return aide_launch_status();
}

int aide_pseudo_label(const float* const* logits, int K, int64_t l_bs, int N, int HW, float temperature,
                      float* pl, float* wm, hipStream_t stream) {
    if (K < 1 || K > 8) return AIDE_ERR_ARG;
    PseudoArgs a;
    for (int k = 0; k < 8; ++k) a.logits[k] = k < K ? logits[k] : nullptr;
    a.K = K; a.HW = HW; a.l_bs = (long)l_bs; a.temperature = temperature; a.pl = pl; a.wm = wm;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, pseudo_label_kernel, dim3(bpi_for(HW) * 4, N), dim3(256), 0, stream, a);
    return aide_launch_status();
}

}  // extern "C"

// ---- branches of the reference's loss modules that no shipped script reaches (utils/loss2d.py:11-12 one-hot targets ->
// arg-max, :44-61 DiceLoss on PROBABILITY input, :98-104 MulticlassDiceLoss with one-hot targets and class weights).
// Plain streaming kernels, fp64 fixed-order reductions (bit-reproducible), one launch per pass.
namespace {

// idx[n][p] = first arg-max over C channels of t[n][c][p]   (torch.argmax(targets.float(), dim=1))
__global__ __launch_bounds__(256) void onehot_argmax_kernel(const float* __restrict__ t, long t_bs, int C, int HW,
                                                            long long* __restrict__ idx, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, p = i - n * HW;
        const float* tp = t + n * t_bs + p;
        float best = tp[0];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            const float v = tp[(long)c * HW];
            if (v > best) { best = v; bi = c; }
        }
        idx[i] = bi;
    }
}

// terms per image: K = 1 (probability input p [N][HW], target t [N][HW]) or K = 2 (logits [N][2][HW], p0 = 1 - p1,
// one-hot target [N][2][HW]).  partials[n][b][k][3] = { sum p_k t_k, sum p_k, sum t_k }
template <int K>
__global__ __launch_bounds__(256) void dice_terms_stats_kernel(const float* __restrict__ x, long x_bs,
                                                               const float* __restrict__ t, long t_bs, int HW, int bpi,
                                                               double* __restrict__ partials) {
    __shared__ double sm[3 * K * 4];
    const int n = blockIdx.y, b = blockIdx.x;
    const float* xn = x + (long)n * x_bs;
    const float* tn = t + (long)n * t_bs;
    double acc[3 * K];
#pragma unroll
    for (int k = 0; k < 3 * K; ++k) acc[k] = 0.0;
    for (int i = b * 256 + threadIdx.x; i < HW; i += bpi * 256) {
        if (K == 1) {
            const float p = xn[i], tv = tn[i];
            acc[0] += (double)(p * tv); acc[1] += (double)p; acc[2] += (double)tv;
        } else {
            const float d = xn[HW + i] - xn[i];
            const float p1 = 1.0f / (1.0f + __expf(-d)), p0 = 1.0f - p1;
            const float t0 = tn[i], t1 = tn[HW + i];
            acc[0] += (double)(p0 * t0); acc[1] += (double)p0; acc[2] += (double)t0;
            acc[3] += (double)(p1 * t1); acc[4] += (double)p1; acc[5] += (double)t1;
        }
    }
    block_sum_d<3 * K>(acc, sm);
    if (threadIdx.x == 0)
        for (int k = 0; k < 3 * K; ++k) partials[((long)n * bpi + b) * 3 * K + k] = acc[k];
}

// stats[n][k][3] from the partials; per_image[n] = sum_k w_k (1 - (2 I + s) / (P + T + s)); out by reduction (0 mean / N,
// 1 sum, 2 none)
__global__ void dice_terms_finalize_kernel(const double* __restrict__ partials, int N, int bpi, int K, float w0, float w1,
                                           float smooth, int reduction, double* __restrict__ stats,
                                           float* __restrict__ per_image, float* __restrict__ out) {
    for (int e = threadIdx.x; e < N * 3 * K; e += blockDim.x) {
        const int n = e / (3 * K), k = e - n * 3 * K;
        double v = 0.0;
        for (int b = 0; b < bpi; ++b) v += partials[((long)n * bpi + b) * 3 * K + k];
        stats[e] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int n = 0; n < N; ++n) {
            float li = 0.f;
            for (int k = 0; k < K; ++k) {
                const double* S = stats + ((long)n * K + k) * 3;
                // fp32 like the reference's per-image expression on fp32 sums
                const float I = (float)S[0], P = (float)S[1], T = (float)S[2];
                const float d = 1.0f - (2.0f * I + smooth) / (P + T + smooth);
                li += (k == 0 ? w0 : w1) * d;
            }
            per_image[n] = li;
            total += (double)li;
            if (reduction == 2) out[n] = li;
        }
        if (reduction == 0) out[0] = (float)(total / N);
        else if (reduction == 1) out[0] = (float)total;
    }
}

// gradient wrt the input: probability (K = 1) or the two logit planes (K = 2).  g: [1] (mean / sum) or [N] (none)
template <int K>
__global__ __launch_bounds__(256) void dice_terms_bwd_kernel(const float* __restrict__ x, long x_bs,
                                                             const float* __restrict__ t, long t_bs, int HW, int N,
                                                             const double* __restrict__ stats, float w0, float w1,
                                                             float smooth, int reduction, const float* __restrict__ g,
                                                             float* __restrict__ dx, long dx_bs) {
    const int n = blockIdx.y;
    const float gi = reduction == 2 ? g[n] : (reduction == 0 ? g[0] / (float)N : g[0]);
    float c2[K], c1[K];                         // d loss_k / d p_k = -w (2 t D - Nn) / D^2 = c1 - c2 t
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double* S = stats + ((long)n * K + k) * 3;
        const float D = (float)S[1] + (float)S[2] + smooth, Nn = 2.0f * (float)S[0] + smooth;
        const float w = (k == 0 ? w0 : w1) * gi;
        c2[k] = w * 2.0f / D;
        c1[k] = w * Nn / (D * D);
    }
    const float* xn = x + (long)n * x_bs;
    const float* tn = t + (long)n * t_bs;
    float* dn = dx + (long)n * dx_bs;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        if (K == 1) {
            dn[i] = c1[0] - c2[0] * tn[i];
        } else {
            const float d = xn[HW + i] - xn[i];
            const float p1 = 1.0f / (1.0f + __expf(-d)), p0 = 1.0f - p1;
            const float g0 = c1[0] - c2[0] * tn[i], g1 = c1[1] - c2[1] * tn[HW + i];
            const float dz1 = (g1 - g0) * p1 * p0;          // d p1 / d z1 = p1 p0 = -d p0 / d z1
            dn[HW + i] = dz1;
            dn[i] = -dz1;
        }
    }
}

}  // namespace

extern "C" {

int aide_onehot_argmax(const float* t, int64_t t_bs, int N, int C, int HW, long long* idx, hipStream_t stream) {
    if (!t || !idx || N <= 0 || C <= 0 || HW <= 0) return AIDE_ERR_ARG;
    const long total = (long)N * HW;
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, onehot_argmax_kernel, dim3((unsigned)min((total + 255) / 256, 8192L)), dim3(256), 0, stream, t,
                       (long)t_bs, C, HW, idx, total);
    return aide_launch_status();
}

size_t aide_dice_terms_ws_bytes(int N, int HW) { return ((size_t)N * bpi_for(HW) * 6 + (size_t)N * 6) * sizeof(double); }

// K = 1: x = probabilities [N][HW] (batch stride x_bs), t [N][HW];  K = 2: x = logits [N][2][HW], t = one-hot [N][2][HW].
// ws: aide_dice_terms_ws_bytes; its tail (N * 3 K doubles at ws + N * bpi * 3 K) holds the per-image sums the backward needs.
int aide_dice_terms_fwd(const float* x, int64_t x_bs, const float* t, int64_t t_bs, int N, int HW, int K, float w0,
                        float w1, float smooth, int reduction, double* ws, float* per_image, float* out,
                        hipStream_t stream) {
    if (!x || !t || !ws || !per_image || !out || N <= 0 || HW <= 0 || (K != 1 && K != 2) || reduction < 0 || reduction > 2)
        return AIDE_ERR_ARG;
    const int bpi = bpi_for(HW);
    double* stats = ws + (size_t)N * bpi * 3 * K;
    if (K == 1) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, dice_terms_stats_kernel<1>, dim3(bpi, N), dim3(256), 0, stream, x, (long)x_bs, t, (long)t_bs, HW, bpi, ws);
    else AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, dice_terms_stats_kernel<2>, dim3(bpi, N), dim3(256), 0, stream, x, (long)x_bs, t, (long)t_bs, HW, bpi, ws);
    AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, dice_terms_finalize_kernel, dim3(1), dim3(256), 0, stream, (const double*)ws, N, bpi, K, w0, w1,
                       smooth, reduction, stats, per_image, out);
    return aide_launch_status();
}

int aide_dice_terms_bwd(const float* x, int64_t x_bs, const float* t, int64_t t_bs, int N, int HW, int K, float w0,
                        float w1, float smooth, int reduction, const double* ws, const float* g, float* dx,
                        int64_t dx_bs, hipStream_t stream) {
    if (!x || !t || !ws || !g || !dx || (K != 1 && K != 2)) return AIDE_ERR_ARG;
    const int bpi = bpi_for(HW);
    const double* stats = ws + (size_t)N * bpi * 3 * K;
    if (K == 1) AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, dice_terms_bwd_kernel<1>, dim3(bpi * 4, N), dim3(256), 0, stream, x, (long)x_bs, t, (long)t_bs, HW, N, stats, w0, w1, smooth, reduction, g, dx, (long)dx_bs);
    else AIDE_LAUNCH_TIMED(AIDE_KT_OTHER, 0.0, dice_terms_bwd_kernel<2>, dim3(bpi * 4, N), dim3(256), 0, stream, x, (long)x_bs, t, (long)t_bs, HW, N, stats, w0, w1, smooth, reduction, g, dx, (long)dx_bs);
    return aide_launch_status();
}

}  // extern "C"
