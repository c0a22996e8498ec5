// AttRec's self-attention pooling (reference: models/seq_rec/AttRec.py:147-186) -- SURVEY 8f rank 3, row a15.
//
//   q = k = relu(pre)            pre = (Q[item_seq] + timing) W   [B, L, d]   (the projection GEMM stays a library GEMM)
//   s = softmax(q q^T / sqrt(d)) over the last axis                [B, L, L]
//   w = s with the diagonal zeroed AFTER the softmax (AttRec.py:178-180)
//   out = mean_l (w v)[l] = (1 / L) sum_m c[m] v[m],  c[m] = sum_{l != m} s[l][m],   v = Q[item_seq]   [B, L, d]
//
// One workgroup per sequence: q lives in LDS (row stride = 2 mod 4: the operand reads of an MFMA hit 64 different banks),
// q q^T and the backward's G q run on v_mfma_f32_32x32x2_f32 (one 32 x 32 tile per wavefront), the score matrix never
// leaves LDS except as the saved softmax the backward needs, and the [B, L, L] x [B, L, d] product collapses to one
// weighted sum of the L value rows.  Under autograd this was ~25 library
// launches (two batched GEMMs, softmax, mask, mean and their backward) over [B, L, d] temporaries.  Deterministic: every
// sum has a fixed order (the MFMA accumulates k in ascending order like an fmaf chain).
#include "dr_common.h"

namespace dr {

constexpr int ATT_MAX_L = 64;
constexpr int ATT_MAX_D = 256;

typedef float att_f32x16 __attribute__((ext_vector_type(16)));

struct AttLayout {
    int LP, DK, DS, SS;   // L rounded up to 4; d rounded up to 2; row strides of q and of the score matrix
    int q, S, c, dc, rd, go, part, total;   // offsets in floats
};

// Row strides are = 2 (mod 4): the 64 operand reads of one v_mfma_f32_32x32x2_f32 (32 rows x 2 adjacent k) then fall
// into 64 different LDS banks.  The forward needs c + part behind the matrices, the backward c, dc, rd, go.
static inline __host__ __device__ AttLayout att_layout(int L, int d, bool backward = false) {
    AttLayout a;
    a.LP = (L + 3) / 4 * 4;
    a.DK = (d + 1) / 2 * 2;
    a.DS = (a.DK % 4 == 0) ? a.DK + 2 : a.DK;
    a.SS = a.LP + 2;
    a.q = 0;
    a.S = a.q + a.LP * a.DS;
    a.c = a.S + a.LP * a.SS;
    a.dc = a.c + ATT_MAX_L;
    a.rd = a.dc + ATT_MAX_L;
    a.go = a.rd + ATT_MAX_L;
    a.part = a.c + ATT_MAX_L;                         // forward only: aliases dc / rd / go
    a.total = backward ? a.go + ATT_MAX_D : a.part + 256;
    return a;
}

// q[l][t] = relu(pre[b][l][t]) for l < L, 0 for the padding rows
template <int V>
__device__ __forceinline__ void att_stage_q(const float* __restrict__ pre_b, int L, int d, const AttLayout& A,
                                            float* __restrict__ q) {
    const int cpr = d / V;
    // four row pieces in flight per thread before the first LDS store
    for (int base = 0; base < A.LP * cpr; base += 256 * 4) {
        float r[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            const int l = idx / cpr, c = idx - l * cpr;
            if (l < L) {
                vload<V>(r[u], pre_b + (int64_t)l * d + c * V);
            } else {
#pragma unroll
                for (int x = 0; x < V; ++x) r[u][x] = 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            const int l = idx / cpr, c = idx - l * cpr;
            if (l < A.LP) {
#pragma unroll
                for (int x = 0; x < V; ++x) q[l * A.DS + c * V + x] = fmaxf(r[u][x], 0.0f);
            }
        }
    }
    if (A.DK != d)       // odd d: the k-pair of the last MFMA step reads one column past the row
        for (int l = threadIdx.x; l < A.LP; l += 256) q[l * A.DS + d] = 0.0f;
}

// c[m] = sum over l != m of S[l][m], l ascending
__device__ __forceinline__ void att_colsum(const float* __restrict__ S, int L, int SS, float* __restrict__ c) {
    const int m = threadIdx.x;
    if (m < L) {
        float acc = 0.0f;
        for (int l = 0; l < L; ++l)
            if (l != m) acc += S[l * SS + m];
        c[m] = acc;
    }
}

template <int V>
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ v,
                                                            int64_t B, int L, int d, float sqrt_d,
                                                            float* __restrict__ out, float* __restrict__ s_out) {
    extern __shared__ float lds[];
    const AttLayout A = att_layout(L, d);
    float* q = lds + A.q;
    float* S = lds + A.S;
    float* c = lds + A.c;
    float* part = lds + A.part;
    const int tid = threadIdx.x;
    const int nT = A.LP / 4;
    // the weighted row sum is split over G groups of dp threads (dp = d rounded up to a power of two)
    int dp = 1;
    while (dp < d) dp <<= 1;
    const int G = 256 / dp;

    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const float* pre_b = pre + b * L * d;
        const float* v_b = v + b * L * d;
        __syncthreads();   // the previous sequence's readers are done with q / S / c
        att_stage_q<V>(pre_b, L, d, A, q);
        __syncthreads();
        // ---- logits: q q^T on the matrix cores, one 32 x 32 tile per wavefront (L <= 64: at most 2 x 2 tiles).
        // v_mfma_f32_32x32x2_f32 accumulates in ascending k like an fp32 fmaf chain; lane l feeds row l % 32 of either
        // operand at k = 2 step + l / 32.  Rows beyond LP are clamped (their products are never stored).
        {
            const int wave = tid >> 6, lane = tid & 63;
            const int nMt = L > 32 ? 2 : 1;
            if (wave < nMt * nMt) {
                const int I = wave / nMt, J = wave - I * nMt;
                const int ra = min(32 * I + (lane & 31), A.LP - 1), rb = min(32 * J + (lane & 31), A.LP - 1);
                const float* pa = q + ra * A.DS + (lane >> 5);
                const float* pb = q + rb * A.DS + (lane >> 5);
                att_f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                for (int k = 0; k < A.DK; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k], acc, 0, 0, 0);
                const int col = 32 * J + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * I + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < A.LP && col < A.LP) S[row * A.SS + col] = acc[r] / sqrt_d;
                }
            }
        }
        __syncthreads();
        // ---- softmax over each row: four lanes per row, the lane's <= 16 logits stay in registers -----------------
        {
            const int row = tid >> 2, sub = tid & 3;
            const bool on = row < L;
            float e[ATT_MAX_L / 4];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < ATT_MAX_L / 4; ++i) {
                const int m = sub + 4 * i;
                e[i] = (on && m < L) ? S[row * A.SS + m] : -INFINITY;
                mx = fmaxf(mx, e[i]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < ATT_MAX_L / 4; ++i) {
                const int m = sub + 4 * i;
                if (on && m < L) {
                    e[i] = expf(e[i] - mx);
                    sum += e[i];
                }
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            float* so = s_out + (b * L + row) * L;
#pragma unroll
            for (int i = 0; i < ATT_MAX_L / 4; ++i) {
                const int m = sub + 4 * i;
                if (on && m < L) {
                    const float p = e[i] / sum;
                    S[row * A.SS + m] = p;
                    so[m] = p;
                }
            }
        }
        __syncthreads();
        att_colsum(S, L, A.SS, c);
        __syncthreads();
        // ---- out[col] = (sum_m c[m] v[m][col]) / L ----------------------------------------------------
        {
            const int col = tid & (dp - 1), g = tid / dp;
            float acc = 0.0f;
            if (col < d) {
                // sixteen value rows in flight per thread (a plain loop waits for every load in turn)
                for (int m0 = g; m0 < L; m0 += 16 * G) {
                    float x[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int m = m0 + u * G;
                        x[u] = m < L ? v_b[(int64_t)m * d + col] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int m = m0 + u * G;
                        if (m < L) acc = fmaf(c[m], x[u], acc);
                    }
                }
            }
            if (G > 1) {
                part[tid] = acc;
                __syncthreads();
                if (g == 0 && col < d) {
                    for (int x = 1; x < G; ++x) acc += part[x * dp + col];
                }
            }
            if (g == 0 && col < d) out[b * d + col] = acc / (float)L;
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ v,
                                                            const float* __restrict__ s_in,
                                                            const float* __restrict__ gout, int64_t B, int L, int d,
                                                            float sqrt_d, float* __restrict__ dpre,
                                                            float* __restrict__ dv) {
    extern __shared__ float lds[];
    const AttLayout A = att_layout(L, d, true);
    float* q = lds + A.q;
    float* S = lds + A.S;
    float* c = lds + A.c;
    float* dc = lds + A.dc;
    float* rd = lds + A.rd;
    float* go = lds + A.go;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int dp = 1;
    while (dp < d) dp <<= 1;
    const int G = 256 / dp;
    const float fL = (float)L;

    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const float* pre_b = pre + b * L * d;
        const float* v_b = v + b * L * d;
        __syncthreads();
        att_stage_q<V>(pre_b, L, d, A, q);
        for (int base = 0; base < A.LP * A.LP; base += 256 * 4) {
            float sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + tid;
                const int l = idx / A.LP, m = idx - l * A.LP;
                sv[u] = (l < L && m < L) ? s_in[(b * L + l) * L + m] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + tid;
                const int l = idx / A.LP, m = idx - l * A.LP;
                if (l < A.LP) S[l * A.SS + m] = sv[u];
            }
        }
        for (int t = tid; t < d; t += 256) go[t] = gout[b * d + t];
        __syncthreads();
        att_colsum(S, L, A.SS, c);
        // dc[m] = <gout, v[m]> / L: one wavefront per value row, columns strided over the lanes
        // (four rows x up to four column pieces in flight per lane)
        for (int m0 = wave; m0 < L; m0 += 16) {
            float x[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + 4 * u, t = lane + 64 * j;
                    x[u][j] = (m < L && t < d) ? v_b[(int64_t)m * d + t] : 0.0f;
                }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = m0 + 4 * u;
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = lane + 64 * j;
                    if (t < d) acc = fmaf(go[t], x[u][j], acc);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
                if (lane == 0 && m < L) dc[m] = acc / fL;
            }
        }
        __syncthreads();
        // direct part of the value gradient: dv[m] = (c[m] / L) gout
        for (int idx = tid; idx < L * d; idx += 256) {
            const int m = idx / d, t = idx - m * d;
            dv[(b * L + m) * d + t] = (c[m] / fL) * go[t];
        }
        // rd[l] = sum over n != l of s[l][n] dc[n] (softmax backward's row term; the diagonal's upstream gradient is 0)
        {
            const int row = tid >> 2, sub = tid & 3;
            float acc = 0.0f;
            if (row < L)
                for (int n = sub; n < L; n += 4)
                    if (n != row) acc = fmaf(S[row * A.SS + n], dc[n], acc);
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if (row < L && sub == 0) rd[row] = acc;
        }
        __syncthreads();
        // dlogits[l][m] = s[l][m] ((l != m ? dc[m] : 0) - rd[l]) / sqrt(d)
        for (int idx = tid; idx < L * L; idx += 256) {
            const int l = idx / L, m = idx - l * L;
            const float up = (l != m) ? dc[m] : 0.0f;
            S[l * A.SS + m] = (S[l * A.SS + m] * (up - rd[l])) / sqrt_d;
        }
        __syncthreads();
        // query and key are the same tensor: dq = (dlogits + dlogits^T) q
        for (int idx = tid; idx < L * L; idx += 256) {
            const int l = idx / L, m = idx - l * L;
            if (l <= m) {
                const float t = S[l * A.SS + m] + S[m * A.SS + l];
                S[l * A.SS + m] = t;
                S[m * A.SS + l] = t;
            }
        }
        __syncthreads();
        // dpre[l][col] = relu'(pre) * sum_m G[l][m] q[m][col] on the matrix cores: 32 x 32 output tiles (rows l, columns
        // col) dealt to the four wavefronts, k = m ascending in pairs (G's columns and q's rows beyond L are zero)
        {
            const int lane = tid & 63;
            const int nMt = L > 32 ? 2 : 1, nNt = (d + 31) / 32;
            const int LK = (L + 1) / 2 * 2;
            for (int tile = wave; tile < nMt * nNt; tile += 4) {
                const int I = tile / nNt, Jc = tile - I * nNt;
                const int ra = min(32 * I + (lane & 31), A.LP - 1), cb = min(32 * Jc + (lane & 31), d - 1);
                const float* pa = S + ra * A.SS + (lane >> 5);
                const float* pb = q + (lane >> 5) * A.DS + cb;
                att_f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                for (int k = 0; k < LK; k += 2)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k * A.DS], acc, 0, 0, 0);
                const int col = 32 * Jc + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * I + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < L && col < d)
                        dpre[(b * L + row) * d + col] = q[row * A.DS + col] > 0.0f ? acc[r] : 0.0f;
                }
            }
        }
    }
}

static int att_check(const char* who, int64_t B, int32_t L, int32_t d) {
    DR_REQUIRE(B >= 0 && L >= 1 && L <= ATT_MAX_L && d >= 1 && d <= ATT_MAX_D,
               "%s: need B >= 0, 1 <= L <= %d, 1 <= d <= %d (got B=%lld L=%d d=%d)", who, ATT_MAX_L, ATT_MAX_D,
               (long long)B, L, d);
    return DR_OK;
}

// grid = the blocks the device holds at once for this (kernel, LDS size); the answer is remembered per thread
template <typename KernelT>
static int att_grid(KernelT kernel, int64_t B, size_t lds_bytes, int* grid) {
    struct Seen {
        const void* fn;
        size_t lds;
        int64_t cap;
    };
    thread_local Seen seen[8];
    thread_local int n_seen = 0;
    const void* fn = reinterpret_cast<const void*>(kernel);
    int64_t cap = 0;
    for (int x = 0; x < n_seen; ++x)
        if (seen[x].fn == fn && seen[x].lds == lds_bytes) cap = seen[x].cap;
    if (cap == 0) {
        if (lds_bytes > 48 * 1024)
            DR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        int per_cu = 0;
        DR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds_bytes));
        if (per_cu < 1)
            DR_FAIL(DR_ERR_LAUNCH, "attention kernel does not fit a compute unit (%zu bytes of LDS)", lds_bytes);
        int dev = 0, cus = 256;
        DR_HIP(hipGetDevice(&dev));
        DR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        cap = (int64_t)per_cu * cus;
        seen[n_seen % 8] = Seen{fn, lds_bytes, cap};
        if (n_seen < 8) ++n_seen;
    }
    *grid = (int)(B < cap ? B : cap);
    return DR_OK;
}

}  // namespace dr

using namespace dr;

extern "C" int32_t dr_attn_pool_forward_f32(const float* pre, const float* v, int64_t B, int32_t L, int32_t d,
                                            float* out, float* s, void* stream) {
    if (int rc = att_check("dr_attn_pool_forward_f32", B, L, d)) return rc;
    if (B == 0) return DR_OK;
    DR_REQUIRE(pre && v && out && s, "dr_attn_pool_forward_f32: NULL pointer");
    const int V = (d % 4 == 0 && (uintptr_t)pre % 16 == 0) ? 4 : ((d % 2 == 0 && (uintptr_t)pre % 8 == 0) ? 2 : 1);
    const size_t lds = (size_t)att_layout(L, d).total * 4;
    const float sd = (float)sqrt((double)d);
    hipStream_t st = (hipStream_t)stream;
    int grid = 0;
#define DR_AF(V_)                                                                                                    \
    do {                                                                                                             \
        if (int rc = att_grid(attn_pool_fwd_kernel<V_>, B, lds, &grid)) return rc;                                   \
        DR_LAUNCH_NAMED("attn_pool_fwd_kernel", attn_pool_fwd_kernel<V_>, dim3(grid), dim3(256), lds, st, pre, v, B, \
                        (int)L, (int)d, sd, out, s);                                                                 \
    } while (0)
    if (V == 4)
        DR_AF(4);
    else if (V == 2)
        DR_AF(2);
    else
        DR_AF(1);
#undef DR_AF
    DR_CHECK_LAUNCH("attn_pool_fwd_kernel");
    return DR_OK;
}

extern "C" int32_t dr_attn_pool_backward_f32(const float* pre, const float* v, const float* s, const float* gout,
                                             int64_t B, int32_t L, int32_t d, float* dpre, float* dv, void* stream) {
    if (int rc = att_check("dr_attn_pool_backward_f32", B, L, d)) return rc;
    if (B == 0) return DR_OK;
    DR_REQUIRE(pre && v && s && gout && dpre && dv, "dr_attn_pool_backward_f32: NULL pointer");
    const int V = (d % 4 == 0 && (uintptr_t)pre % 16 == 0) ? 4 : ((d % 2 == 0 && (uintptr_t)pre % 8 == 0) ? 2 : 1);
    const size_t lds = (size_t)att_layout(L, d, true).total * 4;
    const float sd = (float)sqrt((double)d);
    hipStream_t st = (hipStream_t)stream;
    int grid = 0;
#define DR_AB(V_)                                                                                                      \
    do {                                                                                                               \
        if (int rc = att_grid(attn_pool_bwd_kernel<V_>, B, lds, &grid)) return rc;                                     \
        DR_LAUNCH_NAMED("attn_pool_bwd_kernel", attn_pool_bwd_kernel<V_>, dim3(grid), dim3(256), lds, st, pre, v, s,  \
                        gout, B, (int)L, (int)d, sd, dpre, dv);                                                        \
    } while (0)
    if (V == 4)
        DR_AB(4);
    else if (V == 2)
        DR_AB(2);
    else
        DR_AB(1);
#undef DR_AB
    DR_CHECK_LAUNCH("attn_pool_bwd_kernel");
    return DR_OK;
}
