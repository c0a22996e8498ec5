// Interaction + loss stage of the sequence models, fused with their item lookups:
//   two-distance score  s = alpha * ||X_i - p_u||^2 + (1 - alpha) * ||o - Q_i||^2      (AttRec.py:190-232, PRME.py:104-131)
//     with the pairwise hinge  sum max(s_pos - s_neg + margin, 0)                       (AttRec.py:97)
//     or the pairwise BPR      -sum log sigmoid(-s_pos + s_neg)                         (PRME.py:73)
//   dot score           s = <x, W_i> + b_i,  loss = -mean log(sigmoid(s_pos) + 1e-10) - mean log(1 - sigmoid(s_neg) + 1e-10)
//                                                                                      (Caser.py:79-80,184-192)
// One sub-wave per sample: it holds the sample's user-side vectors in registers across its lanes, walks the sample's
// item rows twice (scores; then, with the loss coefficients known, the gradient rows), and writes per-(sample, item)
// gradient rows for the item tables (fed to the segment update like any lookup backward) plus the gradients of the
// user-side vectors.  Replaces four lookups and ~25 elementwise / reduction launches under autograd and their backward.
// HBM / L2 bound (item rows are read twice, the second time from cache); no MFMA: a handful of rows per sample.
#include "dr_common.h"

namespace dr {

constexpr int SEQ_MAX_SLOTS = 64;  // targets + negatives per sample

__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }

// sub-wave-local scratch for the sample's scores / coefficients (T + Tn <= SEQ_MAX_SLOTS)
template <int V, int LPR, int KIND>  // KIND 0: hinge, 1: BPR
__global__ __launch_bounds__(256) void seq_dist_loss_kernel(
    const float* __restrict__ P, int64_t nu, const float* __restrict__ X, const float* __restrict__ Q, int64_t ni, int k,
    const int32_t* __restrict__ users, const float* __restrict__ o, const int32_t* __restrict__ tgt,
    const int32_t* __restrict__ neg, int64_t B, int T, float alpha, float beta, float margin, float* __restrict__ loss_b,
    float* __restrict__ gX, float* __restrict__ gQ, float* __restrict__ gp, float* __restrict__ go) {
    constexpr int SPB = 256 / LPR;
    __shared__ float sc[SPB][SEQ_MAX_SLOTS];
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int cpr = k / V;
    const int S = 2 * T;
    for (int64_t b0 = (int64_t)blockIdx.x * SPB; b0 < B; b0 += (int64_t)gridDim.x * SPB) {
        const int64_t b = b0 + sub;
        const bool live = b < B;
        const int64_t u = live ? users[b] : -1;
        const bool vu = u >= 0 && u < nu;
        // pass 1: the 2T scores
        for (int s = 0; s < S; ++s) {
            int64_t it = -1;
            if (live) it = s < T ? tgt[b * T + s] : neg[b * T + (s - T)];
            const bool ok = vu && it >= 0 && it < ni;
            float d1 = 0.0f, d2 = 0.0f;
            if (ok)
                for (int c = lane; c < cpr; c += LPR) {
                    float p[V], ov[V], x[V], q[V];
                    vload<V>(p, P + u * k + c * V);
                    vload<V>(ov, o + b * k + c * V);
                    vload<V>(x, X + it * k + c * V);
                    vload<V>(q, Q + it * k + c * V);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const float a1 = x[e] - p[e], a2 = ov[e] - q[e];
                        d1 = d1 + a1 * a1;
                        d2 = d2 + a2 * a2;
                    }
                }
            d1 = subwave_sum<LPR>(d1);
            d2 = subwave_sum<LPR>(d2);
            if (lane == 0) sc[sub][s] = ok ? (alpha * d1 + beta * d2) : __builtin_nanf("");
        }
        __syncthreads();
        // coefficients dL/ds per slot (pair t: target t with negative t); an invalid id voids its pair
        float lsum = 0.0f;
        if (lane == 0 && live) {
            for (int t = 0; t < T; ++t) {
                const float sp = sc[sub][t], sn = sc[sub][T + t];
                float cp = 0.0f, l = 0.0f;
                if (sp == sp && sn == sn) {
                    if (KIND == 0) {
                        const float a = sp - sn + margin;
                        l = fmaxf(a, 0.0f);
                        cp = a >= 0.0f ? 1.0f : 0.0f;     // [TF-ext] tf.maximum passes the gradient to x where x >= y
                    } else {
                        const float sg = sigmoid_(-sp + sn);
                        l = -logf(sg);
                        cp = 1.0f - sg;                   // d(-log sigmoid(x))/dx = -(1 - sigmoid(x)), x = -sp + sn
                    }
                }
                lsum = lsum + l;
                sc[sub][t] = cp;
                sc[sub][T + t] = -cp;
            }
            loss_b[b] = lsum;
        }
        __syncthreads();
        // pass 2: gradient rows.  ds/dX = 2 alpha (X - p), ds/dp = -that; ds/dQ = -2 beta (o - Q), ds/do = +that
        if (live) {
            for (int c = lane; c < cpr; c += LPR) {
                float p[V], ov[V], ap[V], ao[V];
#pragma unroll
                for (int e = 0; e < V; ++e) p[e] = ov[e] = ap[e] = ao[e] = 0.0f;
                if (vu) vload<V>(p, P + u * k + c * V);
                vload<V>(ov, o + b * k + c * V);
                for (int s = 0; s < S; ++s) {
                    const float cf = sc[sub][s];
                    const int64_t it = s < T ? tgt[b * T + s] : neg[b * T + (s - T)];
                    float gx[V], gq[V];
#pragma unroll
                    for (int e = 0; e < V; ++e) gx[e] = gq[e] = 0.0f;
                    if (cf != 0.0f) {                      // (a void pair has cf == 0 for both of its slots)
                        float x[V], q[V];
                        vload<V>(x, X + it * k + c * V);
                        vload<V>(q, Q + it * k + c * V);
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            gx[e] = cf * ((2.0f * alpha) * (x[e] - p[e]));
                            gq[e] = -(cf * ((2.0f * beta) * (ov[e] - q[e])));
                            ap[e] = ap[e] - gx[e];
                            ao[e] = ao[e] - gq[e];
                        }
                    }
                    vstore<V>(gX + (b * S + s) * k + c * V, gx);
                    vstore<V>(gQ + (b * S + s) * k + c * V, gq);
                }
                vstore<V>(gp + b * k + c * V, ap);
                vstore<V>(go + b * k + c * V, ao);
            }
        }
        __syncthreads();
    }
}

// Caser: s = <x_b, W_i> + b_i over D = 2d columns; mean BCE with the reference's 1e-10 inside the logs.
// inv_t = scale / (B T), inv_n = scale / (B Tn) are the two means (scale = 1 / world for a data-parallel batch slice).
template <int V, int LPR>
__global__ __launch_bounds__(256) void seq_dot_loss_kernel(
    const float* __restrict__ W, const float* __restrict__ bias, int64_t ni, int D, const float* __restrict__ x,
    const int32_t* __restrict__ tgt, const int32_t* __restrict__ neg, int64_t B, int T, int Tn, float inv_t, float inv_n,
    float* __restrict__ loss_b, float* __restrict__ gW, float* __restrict__ gb, float* __restrict__ gx) {
    constexpr int SPB = 256 / LPR;
    __shared__ float sc[SPB][SEQ_MAX_SLOTS];
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int cpr = D / V;
    const int S = T + Tn;
    for (int64_t b0 = (int64_t)blockIdx.x * SPB; b0 < B; b0 += (int64_t)gridDim.x * SPB) {
        const int64_t b = b0 + sub;
        const bool live = b < B;
        for (int s = 0; s < S; ++s) {
            int64_t it = -1;
            if (live) it = s < T ? tgt[b * T + s] : neg[b * Tn + (s - T)];
            const bool ok = it >= 0 && it < ni;
            float d = 0.0f;
            if (ok)
                for (int c = lane; c < cpr; c += LPR) {
                    float xv[V], w[V];
                    vload<V>(xv, x + b * D + c * V);
                    vload<V>(w, W + it * D + c * V);
#pragma unroll
                    for (int e = 0; e < V; ++e) d = d + xv[e] * w[e];
                }
            d = subwave_sum<LPR>(d);
            if (lane == 0) sc[sub][s] = ok ? (d + bias[it]) : __builtin_nanf("");
        }
        __syncthreads();
        if (lane == 0 && live) {
            float lsum = 0.0f;
            for (int s = 0; s < S; ++s) {
                const float z = sc[sub][s];
                float cf = 0.0f;
                if (z == z) {
                    const float sg = sigmoid_(z);
                    const float ds = sg * (1.0f - sg);
                    if (s < T) {
                        lsum = lsum + (-logf(sg + 1e-10f)) * inv_t;
                        cf = -(ds / (sg + 1e-10f)) * inv_t;
                    } else {
                        lsum = lsum + (-logf(1.0f - sg + 1e-10f)) * inv_n;
                        cf = (ds / (1.0f - sg + 1e-10f)) * inv_n;
                    }
                }
                sc[sub][s] = cf;
                gb[b * S + s] = cf;
            }
            loss_b[b] = lsum;
        }
        __syncthreads();
        if (live) {
            for (int c = lane; c < cpr; c += LPR) {
                float xv[V], ax[V];
                vload<V>(xv, x + b * D + c * V);
#pragma unroll
                for (int e = 0; e < V; ++e) ax[e] = 0.0f;
                for (int s = 0; s < S; ++s) {
                    const float cf = sc[sub][s];
                    float g[V];
#pragma unroll
                    for (int e = 0; e < V; ++e) g[e] = 0.0f;
                    if (cf != 0.0f) {
                        const int64_t it = s < T ? tgt[b * T + s] : neg[b * Tn + (s - T)];
                        float w[V];
                        vload<V>(w, W + it * D + c * V);
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            g[e] = cf * xv[e];
                            ax[e] = ax[e] + cf * w[e];
                        }
                    }
                    vstore<V>(gW + (b * S + s) * D + c * V, g);
                }
                vstore<V>(gx + b * D + c * V, ax);
            }
        }
        __syncthreads();
    }
}

#define DR_SEQ_DISPATCH(MACRO)                                                                       \
    MACRO(4, 1) MACRO(4, 2) MACRO(4, 4) MACRO(4, 8) MACRO(4, 16) MACRO(4, 32) MACRO(4, 64)           \
    MACRO(2, 1) MACRO(2, 2) MACRO(2, 4) MACRO(2, 8) MACRO(2, 16) MACRO(2, 32) MACRO(2, 64)           \
    MACRO(1, 1) MACRO(1, 2) MACRO(1, 4) MACRO(1, 8) MACRO(1, 16) MACRO(1, 32) MACRO(1, 64)
}  // namespace dr

using namespace dr;

extern "C" int32_t dr_seq_dist_loss_f32(const float* P, int64_t num_user, const float* X, const float* Q, int64_t num_item,
                                        int32_t k, const int32_t* users, const float* o, const int32_t* tgt,
                                        const int32_t* neg, int64_t B, int32_t T, float alpha, float beta,
                                        int32_t loss_kind, float margin, float* loss_b, float* gX, float* gQ, float* gp,
                                        float* go, void* stream) {
    DR_REQUIRE(num_user >= 0 && num_item >= 0 && k >= 1 && B >= 0 && T >= 1 && 2 * T <= SEQ_MAX_SLOTS,
               "dr_seq_dist_loss_f32: bad sizes (2 T <= %d)", SEQ_MAX_SLOTS);
    DR_REQUIRE(loss_kind == 0 || loss_kind == 1, "dr_seq_dist_loss_f32: loss_kind must be 0 (hinge) or 1 (bpr)");
    if (B == 0) return DR_OK;
    DR_REQUIRE(P && X && Q && users && o && tgt && neg && loss_b && gX && gQ && gp && go, "dr_seq_dist_loss_f32: NULL pointer");
    RowShape s = row_shape(k);
    DR_REQUIRE(s.V == 1 || (((uintptr_t)P | (uintptr_t)X | (uintptr_t)Q | (uintptr_t)o | (uintptr_t)gX | (uintptr_t)gQ |
                             (uintptr_t)gp | (uintptr_t)go) % (4 * s.V) == 0),
               "dr_seq_dist_loss_f32: misaligned rows");
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
#define DR_F(V_, L_)                                                                                                     \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                               \
        const dim3 grid(grid_for(B, 1, 256 / L_));                                                                       \
        if (loss_kind == 0)                                                                                              \
            DR_LAUNCH((seq_dist_loss_kernel<V_, L_, 0>), grid, dim3(256), 0, st, P, num_user, X, Q, num_item, (int)k, users, \
                      o, tgt, neg, B, (int)T, alpha, beta, margin, loss_b, gX, gQ, gp, go);                             \
        else                                                                                                             \
            DR_LAUNCH((seq_dist_loss_kernel<V_, L_, 1>), grid, dim3(256), 0, st, P, num_user, X, Q, num_item, (int)k, users, \
                      o, tgt, neg, B, (int)T, alpha, beta, margin, loss_b, gX, gQ, gp, go);                             \
        ok = true;                                                                                                       \
    }
    DR_SEQ_DISPATCH(DR_F)
#undef DR_F
    DR_REQUIRE(ok, "dr_seq_dist_loss_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("dr_seq_dist_loss_f32");
    return DR_OK;
}

extern "C" int32_t dr_seq_dot_loss_f32(const float* W, const float* bias, int64_t num_item, int32_t D, const float* x,
                                       const int32_t* tgt, const int32_t* neg, int64_t B, int32_t T, int32_t Tn,
                                       float scale, float* loss_b, float* gW, float* gb, float* gx, void* stream) {
    DR_REQUIRE(num_item >= 0 && D >= 1 && B >= 0 && T >= 1 && Tn >= 1 && T + Tn <= SEQ_MAX_SLOTS,
               "dr_seq_dot_loss_f32: bad sizes (T + Tn <= %d)", SEQ_MAX_SLOTS);
    if (B == 0) return DR_OK;
    DR_REQUIRE(W && bias && x && tgt && neg && loss_b && gW && gb && gx, "dr_seq_dot_loss_f32: NULL pointer");
    RowShape s = row_shape(D);
    DR_REQUIRE(s.V == 1 || (((uintptr_t)W | (uintptr_t)x | (uintptr_t)gW | (uintptr_t)gx) % (4 * s.V) == 0),
               "dr_seq_dot_loss_f32: misaligned rows");
    const float inv_t = scale / ((float)B * (float)T), inv_n = scale / ((float)B * (float)Tn);
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
#define DR_F(V_, L_)                                                                                                   \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                             \
        DR_LAUNCH((seq_dot_loss_kernel<V_, L_>), dim3(grid_for(B, 1, 256 / L_)), dim3(256), 0, st, W, bias, num_item,  \
                  (int)D, x, tgt, neg, B, (int)T, (int)Tn, inv_t, inv_n, loss_b, gW, gb, gx);                          \
        ok = true;                                                                                                     \
    }
    DR_SEQ_DISPATCH(DR_F)
#undef DR_F
    DR_REQUIRE(ok, "dr_seq_dot_loss_f32: unsupported D=%d", D);
    DR_CHECK_LAUNCH("dr_seq_dot_loss_f32");
    return DR_OK;
}
