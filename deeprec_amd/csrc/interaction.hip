// NFM bi-interaction pooling (nfm.py:54-61) fused with its embedding lookup, and its backward.
// HBM/L2-bound: one sub-wave per sample walks the sample's F feature rows (sequential in f, the
// order oracle/ref_ops.py:bi_interaction_fwd uses) and never materialises the [B,F,k] gather.
#include "dr_common.h"

namespace dr {

template <int V, int LPR>
__global__ __launch_bounds__(256) void bi_fwd_kernel(const float* __restrict__ E, int64_t rows, int k,
                                                     const int32_t* __restrict__ feats, int64_t B, int F,
                                                     float* __restrict__ fm, float* __restrict__ sum_out) {
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int cpr = k / V;
    for (int64_t b = (int64_t)blockIdx.x * SPB + sub; b < B; b += (int64_t)gridDim.x * SPB) {
        for (int c = lane; c < cpr; c += LPR) {
            float s[V], q[V];
#pragma unroll
            for (int x = 0; x < V; ++x) s[x] = q[x] = 0.0f;
            for (int f = 0; f < F; ++f) {
                const int64_t id = feats[b * F + f];
                if (id < 0 || id >= rows) continue;  // zero row, like the lookup's DR_OOB_ZERO
                float v[V];
                vload<V>(v, E + id * k + c * V);
#pragma unroll
                for (int x = 0; x < V; ++x) {
                    s[x] = s[x] + v[x];
                    q[x] = q[x] + v[x] * v[x];
                }
            }
            float o[V];
#pragma unroll
            for (int x = 0; x < V; ++x) o[x] = 0.5f * (s[x] * s[x] - q[x]);
            vstore<V>(fm + b * k + c * V, o);
            if (sum_out) vstore<V>(sum_out + b * k + c * V, s);
        }
    }
}

template <int V, int LPR>
__global__ __launch_bounds__(256) void bi_bwd_kernel(const float* __restrict__ E, int64_t rows, int k,
                                                     const int32_t* __restrict__ feats, int64_t B, int F,
                                                     const float* __restrict__ gfm, const float* __restrict__ sum_in,
                                                     float* __restrict__ gv) {
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int cpr = k / V;
    for (int64_t b = (int64_t)blockIdx.x * SPB + sub; b < B; b += (int64_t)gridDim.x * SPB) {
        for (int c = lane; c < cpr; c += LPR) {
            float g[V], s[V];
            vload<V>(g, gfm + b * k + c * V);
            vload<V>(s, sum_in + b * k + c * V);
            for (int f = 0; f < F; ++f) {
                const int64_t id = feats[b * F + f];
                float v[V], o[V];
                if (id >= 0 && id < rows) {
                    vload<V>(v, E + id * k + c * V);
                } else {
#pragma unroll
                    for (int x = 0; x < V; ++x) v[x] = 0.0f;
                }
#pragma unroll
                for (int x = 0; x < V; ++x) o[x] = g[x] * (s[x] - v[x]);
                vstore<V>(gv + (b * F + f) * k + c * V, o);
            }
        }
    }
}
}  // namespace dr

using namespace dr;

#define DR_SHAPE_DISPATCH(MACRO)                                                                     \
    MACRO(4, 1) MACRO(4, 2) MACRO(4, 4) MACRO(4, 8) MACRO(4, 16) MACRO(4, 32) MACRO(4, 64)           \
    MACRO(2, 1) MACRO(2, 2) MACRO(2, 4) MACRO(2, 8) MACRO(2, 16) MACRO(2, 32) MACRO(2, 64)           \
    MACRO(1, 1) MACRO(1, 2) MACRO(1, 4) MACRO(1, 8) MACRO(1, 16) MACRO(1, 32) MACRO(1, 64)

extern "C" int32_t dr_bi_interaction_fwd_f32(const float* E, int64_t rows, int32_t k, const int32_t* feats, int64_t B,
                                             int32_t F, float* fm, float* sum_out, void* stream) {
    DR_REQUIRE(rows >= 0 && k >= 1 && B >= 0 && F >= 0, "dr_bi_interaction_fwd_f32: bad sizes");
    if (B == 0) return DR_OK;
    DR_REQUIRE(E && fm && (feats || F == 0), "dr_bi_interaction_fwd_f32: NULL pointer");
    RowShape s = row_shape(k);
    bool ok = false;
#define DR_F(V_, L_)                                                                                            \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                      \
        DR_LAUNCH((bi_fwd_kernel<V_, L_>), dim3(grid_for(B, 1, 256 / L_)), dim3(256), 0,               \
                           (hipStream_t)stream, E, rows, k, feats, B, F, fm, sum_out);                          \
        ok = true;                                                                                              \
    }
    DR_SHAPE_DISPATCH(DR_F)
#undef DR_F
    DR_REQUIRE(ok, "dr_bi_interaction_fwd_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("dr_bi_interaction_fwd_f32");
    return DR_OK;
}

extern "C" int32_t dr_bi_interaction_bwd_f32(const float* E, int64_t rows, int32_t k, const int32_t* feats, int64_t B,
                                             int32_t F, const float* gfm, const float* sum_in, float* gv,
                                             void* stream) {
    DR_REQUIRE(rows >= 0 && k >= 1 && B >= 0 && F >= 0, "dr_bi_interaction_bwd_f32: bad sizes");
    if (B == 0 || F == 0) return DR_OK;
    DR_REQUIRE(E && feats && gfm && sum_in && gv, "dr_bi_interaction_bwd_f32: NULL pointer");
    RowShape s = row_shape(k);
    bool ok = false;
#define DR_F(V_, L_)                                                                                            \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                      \
        DR_LAUNCH((bi_bwd_kernel<V_, L_>), dim3(grid_for(B, 1, 256 / L_)), dim3(256), 0,               \
                           (hipStream_t)stream, E, rows, k, feats, B, F, gfm, sum_in, gv);                      \
        ok = true;                                                                                              \
    }
    DR_SHAPE_DISPATCH(DR_F)
#undef DR_F
    DR_REQUIRE(ok, "dr_bi_interaction_bwd_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("dr_bi_interaction_bwd_f32");
    return DR_OK;
}
