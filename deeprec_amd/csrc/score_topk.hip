// K6: full-catalog scoring as an fp32 MFMA GEMM with a fused masked top-K / rank-of-target
// epilogue (scores never reach HBM), plus the same selection on a caller-provided score matrix.
//
// Every supported score is brought to the bilinear form
//     score[u,i] = cdot * <U'[u,:], I'[i,:]> + uadd[u] + iadd[i]
// (dot: cdot=1; -||p-q||^2 = 2<p,q> - ||p||^2 - ||q||^2; the AttRec/PRME two-distance score is
// the same with U' = [alpha*p, (1-alpha)*s], I' = [X, Q]).  <.,.> is accumulated by
// v_mfma_f32_32x32x2_f32, which is bit for bit the k-ordered fp32 fmaf chain, so the scores equal
// oracle/c/oracle.c:or_score_* exactly and the (score desc, item id asc) selection is exact.
//
// Block = 4 waves = 64 users x 64 items per tile step (each wave one 32x32 MFMA tile), user panel
// fixed per block, item tiles streamed in ascending order; operands are staged through LDS in
// k-chunks of KC (row stride KC+1 floats: conflict-free ds_read_b32 for the MFMA lane pattern).
#include "dr_common.h"

namespace dr {

constexpr int TU = 64;   // users per block tile
constexpr int TI = 64;   // items per tile step
constexpr int KC = 128;  // k-chunk staged in LDS
constexpr int LDK = KC + 1;
constexpr int LDS_S = TI + 1;
constexpr int MAXK_TOP = 100;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bool beats(float s, int c, float st, int ct) {
    // total order of RankingMetrics.py:98 (stable sort, score descending, candidates ascending id)
    return (s > st) || (s == st && c < ct);
}

struct ScoreArgs {
    const float* U;      // [Nu, k]
    const float* I;      // [Ni, k]
    const float* uadd;   // [Nu] or NULL
    const float* iadd;   // [Ni] or NULL
    float cdot;
    int64_t Nu, Ni;
    int k;
    int act;             // 0 none, 1 sigmoid (matrix output only)
    // selection
    const int32_t* train_ptr;
    const int32_t* train_idx;
    const int32_t* tgt_ptr;
    const int32_t* tgt_idx;
    const float* tgt_score;
    const int32_t* tgt_masked;
    int32_t* tgt_rank;
    int K;
    int ntiles, tiles_per_split;
    float* part_v;       // [S, Nu, K]
    int32_t* part_i;     // [S, Nu, K]
    float* out;          // [Nu, Ni] (matrix mode)
};

__device__ __forceinline__ float score_epilogue(float acc, float cdot, const float* uadd, const float* iadd, int64_t u,
                                                int64_t i) {
    float s = acc;
    if (cdot != 1.0f) s = cdot * s;
    if (uadd) s = s + uadd[u];
    if (iadd) s = s + iadd[i];
    return s;
}

constexpr int TCAP = 1024;  // held-out items of one 64-user panel cached in LDS (more: read from global)

// Stage a [64 rows x kce] operand chunk into LDS (row stride LDK).  Global reads are float4 along k when
// the row stride allows it; LDS writes are scalar because the padded stride breaks 16-byte alignment.
__device__ __forceinline__ void stage_chunk(float* __restrict__ dst, const float* __restrict__ src, int64_t row0,
                                            int64_t nrows, int k, int kc0, int kcn, int kce, int tid) {
    if ((k & 3) == 0 && (kcn & 3) == 0) {
        const int gpr = kcn >> 2;  // float4 groups per row
        for (int e = tid; e < 64 * gpr; e += 256) {
            const int r = e / gpr, g = e - r * gpr;
            const int64_t row = row0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < nrows) v = *reinterpret_cast<const float4*>(src + row * k + kc0 + g * 4);
            float* d = dst + r * LDK + g * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
        for (int e = tid; e < 64 * kce; e += 256) {
            const int r = e / kce, c = e - r * kce;
            const int64_t row = row0 + r;
            dst[r * LDK + c] = (row < nrows && c < kcn) ? src[row * k + kc0 + c] : 0.0f;
        }
    }
}

template <bool MATRIX>
__global__ __launch_bounds__(256) void score_kernel(ScoreArgs a) {
    extern __shared__ __align__(16) float smem[];
    float* sU = smem;                // [TU][LDK]
    float* sI = sU + TU * LDK;       // [TI][LDK]
    float* sS = sI + TI * LDK;       // [TU][LDS_S]
    float* topv = sS + TU * LDS_S;   // [TU][K]
    int* topi = (int*)(topv + TU * a.K);
    float* tau = (float*)(topi + TU * a.K);                           // [TU] K-th best so far (NaN: list not full)
    unsigned long long* maskw = (unsigned long long*)(tau + TU);      // [TU] masked (train) items of this tile
    unsigned long long* candw = maskw + TU;                           // [TU] columns that beat tau in this tile
    int* tptr = (int*)(candw + TU);                                   // [TU+1] panel-local target offsets
    float* tsc_l = (float*)(tptr + TU + 1);                           // [TCAP]
    int* tid_l = (int*)(tsc_l + TCAP);                                // [TCAP] item id, -1 = masked target
    int* tcnt_l = tid_l + TCAP;                                       // [TCAP]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wu = wave >> 1, wi = wave & 1;
    const int64_t u0 = (int64_t)blockIdx.x * TU;
    const int split = blockIdx.y;
    const int t_begin = split * a.tiles_per_split;
    int t_end = t_begin + a.tiles_per_split;
    if (t_end > a.ntiles) t_end = a.ntiles;
    const int K = a.K;

    // ---- selection state -------------------------------------------------------------------
    const int64_t myu = u0 + tid;
    const bool rowthread = (!MATRIX) && tid < TU && myu < a.Nu;
    int cur = 0, cur_end = 0, nxt = 0x7fffffff;   // train cursor of the row thread, next train item cached
    int tg0 = 0, ntg = 0;                          // this panel's slice of the target CSR
    bool tg_lds = false;
    if (!MATRIX) {
        for (int e = tid; e < TU * K; e += 256) {
            topv[e] = -INFINITY;
            topi[e] = 0x7fffffff;
        }
        if (tid < TU) {
            tau[tid] = __builtin_nanf("");
            candw[tid] = 0ull;
            maskw[tid] = 0ull;
        }
        if (rowthread && a.train_ptr) {
            cur = a.train_ptr[myu];
            cur_end = a.train_ptr[myu + 1];
            const int first = t_begin * TI;
            int lo = cur, hi = cur_end;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (a.train_idx[mid] < first) lo = mid + 1; else hi = mid;
            }
            cur = lo;
            nxt = cur < cur_end ? a.train_idx[cur] : 0x7fffffff;
        }
        if (a.tgt_ptr) {
            const int64_t ulast = (u0 + TU < a.Nu) ? u0 + TU : a.Nu;
            tg0 = a.tgt_ptr[u0];
            ntg = a.tgt_ptr[ulast] - tg0;
            tg_lds = ntg <= TCAP;
            if (tid <= TU) {
                const int64_t u = (u0 + tid < a.Nu) ? u0 + tid : a.Nu;
                tptr[tid] = a.tgt_ptr[u] - tg0;
            }
            if (tg_lds)
                for (int x = tid; x < ntg; x += 256) {
                    tsc_l[x] = a.tgt_score[tg0 + x];
                    tid_l[x] = a.tgt_masked[tg0 + x] ? -1 : a.tgt_idx[tg0 + x];
                    tcnt_l[x] = 0;
                }
        }
    }
    const bool one_chunk = a.k <= KC;

    for (int t = t_begin; t < t_end; ++t) {
        const int64_t i0 = (int64_t)t * TI;
        // row threads: train-item bitmap of this tile (visible after the staging barrier below)
        if (rowthread) {
            unsigned long long m = 0ull;
            const int hi = (int)i0 + TI;
            while (nxt < hi) {
                m |= 1ull << (nxt - (int)i0);
                ++cur;
                nxt = cur < cur_end ? a.train_idx[cur] : 0x7fffffff;
            }
            maskw[tid] = m;
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        for (int kc0 = 0; kc0 < a.k; kc0 += KC) {
            const int kcn = (a.k - kc0 < KC) ? (a.k - kc0) : KC;
            const int kce = (kcn + 1) & ~1;  // even, zero padded
            __syncthreads();
            if (!(one_chunk && t > t_begin))  // the user panel stays resident when k fits one chunk
                stage_chunk(sU, a.U, u0, a.Nu, a.k, kc0, kcn, kce, tid);
            stage_chunk(sI, a.I, i0, a.Ni, a.k, kc0, kcn, kce, tid);
            __syncthreads();
            const float* pa = sU + (wu * 32 + (lane & 31)) * LDK + (lane >> 5);
            const float* pb = sI + (wi * 32 + (lane & 31)) * LDK + (lane >> 5);
            if (kce == KC) {
#pragma unroll 16
                for (int kk = 0; kk < KC; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk], pb[kk], acc, 0, 0, 0);
            } else {
                for (int kk = 0; kk < kce; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk], pb[kk], acc, 0, 0, 0);
            }
        }
        // accumulator -> LDS score tile with the affine epilogue
        {
            const int col = wi * 32 + (lane & 31);
            const int64_t it = i0 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wu * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t u = u0 + row;
                float s = 0.0f;
                if (u < a.Nu && it < a.Ni) s = score_epilogue(acc[r], a.cdot, a.uadd, a.iadd, u, it);
                sS[row * LDS_S + col] = s;
            }
        }
        __syncthreads();
        if (MATRIX) {
            for (int e = tid; e < TU * TI; e += 256) {
                const int r = e / TI, c = e - r * TI;
                const int64_t u = u0 + r, i = i0 + c;
                if (u < a.Nu && i < a.Ni) {
                    float s = sS[r * LDS_S + c];
                    if (a.act == 1) s = 1.0f / (1.0f + expf(-s));
                    a.out[u * a.Ni + i] = s;
                }
            }
        } else {
            // (row, quarter) threads: threshold filter for the top-K and rank-of-target counting
            const int row = tid >> 2, q = tid & 3;
            const int64_t u = u0 + row;
            if (u < a.Nu) {
                const float* srow = sS + row * LDS_S + q * 16;
                const int nc = (a.Ni - i0 < TI) ? (int)(a.Ni - i0) : TI;
                unsigned live = 0u, cand = 0u;
                const unsigned mrow = (unsigned)((maskw[row] >> (q * 16)) & 0xffffull);
                const float th = tau[row];
                float sv[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    sv[c] = srow[c];
                    const bool ok = (q * 16 + c < nc) && !((mrow >> c) & 1u) && (sv[c] == sv[c]);
                    live |= ok ? (1u << c) : 0u;
                    cand |= (ok && !(sv[c] <= th)) ? (1u << c) : 0u;
                }
                if (cand) atomicOr(&candw[row], (unsigned long long)cand << (q * 16));
                const int tb = a.tgt_ptr ? tptr[row] : 0, te = a.tgt_ptr ? tptr[row + 1] : 0;
                for (int x = tb; x < te; ++x) {
                    const int ct = tg_lds ? tid_l[x] : (a.tgt_masked[tg0 + x] ? -1 : a.tgt_idx[tg0 + x]);
                    if (ct < 0) continue;
                    const float st = tg_lds ? tsc_l[x] : a.tgt_score[tg0 + x];
                    const int item0 = (int)i0 + q * 16;
                    int cnt = 0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) cnt += (((live >> c) & 1u) && beats(sv[c], item0 + c, st, ct)) ? 1 : 0;
                    if (cnt) {
                        if (tg_lds) atomicAdd(&tcnt_l[x], cnt);
                        else atomicAdd(a.tgt_rank + tg0 + x, cnt);
                    }
                }
            }
            __syncthreads();
            // row threads: fold this tile's candidates into the sorted top-K list
            if (rowthread) {
                unsigned long long m = candw[tid];
                if (m) {
                    candw[tid] = 0ull;
                    const float* srow = sS + tid * LDS_S;
                    float* tv = topv + tid * K;
                    int* ti = topi + tid * K;
                    while (m) {
                        const int c = __builtin_ctzll(m);
                        m &= m - 1;
                        const float s = srow[c];
                        const int item = (int)(i0 + c);
                        if (beats(s, item, tv[K - 1], ti[K - 1])) {
                            int p = K - 1;
                            while (p > 0 && beats(s, item, tv[p - 1], ti[p - 1])) {
                                tv[p] = tv[p - 1];
                                ti[p] = ti[p - 1];
                                --p;
                            }
                            tv[p] = s;
                            ti[p] = item;
                        }
                    }
                    if (ti[K - 1] != 0x7fffffff) tau[tid] = tv[K - 1];
                }
            }
        }
    }
    if (!MATRIX) {
        __syncthreads();
        for (int e = tid; e < TU * K; e += 256) {
            const int r = e / K, c = e - r * K;
            const int64_t u = u0 + r;
            if (u < a.Nu) {
                const int64_t o = ((int64_t)split * a.Nu + u) * K + c;
                a.part_v[o] = topv[e];
                a.part_i[o] = topi[e];
            }
        }
        if (a.tgt_ptr && tg_lds)
            for (int x = tid; x < ntg; x += 256)
                if (tcnt_l[x]) atomicAdd(a.tgt_rank + tg0 + x, tcnt_l[x]);
    }
}

// S-way merge of the per-split sorted candidate lists
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ part_v,
                                                         const int32_t* __restrict__ part_i, int S, int64_t Nu, int K,
                                                         int32_t* __restrict__ topk_idx, float* __restrict__ topk_score) {
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < Nu; u += (int64_t)gridDim.x * 256) {
        int head[64];
        for (int s = 0; s < S; ++s) head[s] = 0;
        for (int o = 0; o < K; ++o) {
            int best = -1;
            float bv = 0.0f;
            int bi = 0;
            for (int s = 0; s < S; ++s) {
                if (head[s] >= K) continue;
                const int64_t p = ((int64_t)s * Nu + u) * K + head[s];
                const float v = part_v[p];
                const int id = part_i[p];
                if (id == 0x7fffffff) continue;
                if (best < 0 || beats(v, id, bv, bi)) {
                    best = s;
                    bv = v;
                    bi = id;
                }
            }
            if (best < 0) {
                topk_idx[u * K + o] = -1;
                topk_score[u * K + o] = -INFINITY;
            } else {
                topk_idx[u * K + o] = bi;
                topk_score[u * K + o] = bv;
                ++head[best];
            }
        }
    }
}

// target prologue: canonical score of every held-out item + masked flag; rank counters zeroed
__global__ __launch_bounds__(256) void target_prologue_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                              const float* __restrict__ uadd,
                                                              const float* __restrict__ iadd, float cdot, int64_t Nu,
                                                              int64_t Ni, int k, const float* __restrict__ scores,
                                                              const int32_t* __restrict__ train_ptr,
                                                              const int32_t* __restrict__ train_idx,
                                                              const int32_t* __restrict__ tgt_ptr,
                                                              const int32_t* __restrict__ tgt_idx,
                                                              float* __restrict__ tgt_score,
                                                              int32_t* __restrict__ tgt_masked,
                                                              int32_t* __restrict__ tgt_rank) {
    // one thread per user walks its targets
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < Nu; u += (int64_t)gridDim.x * 256) {
        const int tb = tgt_ptr[u], te = tgt_ptr[u + 1];
        for (int x = tb; x < te; ++x) {
            const int64_t it = tgt_idx[x];
            bool masked = it < 0 || it >= Ni;
            if (!masked && train_ptr) {
                int lo = train_ptr[u], hi = train_ptr[u + 1];
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if (train_idx[mid] < it) lo = mid + 1; else hi = mid;
                }
                masked = lo < train_ptr[u + 1] && train_idx[lo] == it;
            }
            float s = 0.0f;
            if (!masked) {
                if (scores) {
                    s = scores[u * Ni + it];
                } else {
                    float acc = 0.0f;
                    const float* pu = U + u * k;
                    const float* pi = I + it * k;
                    for (int c = 0; c < k; ++c) acc = __builtin_fmaf(pu[c], pi[c], acc);
                    s = score_epilogue(acc, cdot, uadd, iadd, u, it);
                }
                if (s != s) masked = true;
            }
            tgt_score[x] = s;
            tgt_masked[x] = masked ? 1 : 0;
            tgt_rank[x] = 0;
        }
    }
}

__global__ __launch_bounds__(256) void target_finalize_kernel(const int32_t* __restrict__ tgt_masked, int64_t n,
                                                              int32_t* __restrict__ tgt_rank) {
    for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < n; x += (int64_t)gridDim.x * 256)
        if (tgt_masked[x]) tgt_rank[x] = -1;
}

// weighted squared norms as k-ordered fmaf chains:  out[r] = -sum_c (w_c * x[r,c]) * x[r,c]
__global__ __launch_bounds__(256) void neg_sqnorm_kernel(const float* __restrict__ X, const float* __restrict__ X2,
                                                         int64_t rows, int k, float w1, float w2,
                                                         float* __restrict__ out) {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
        float acc = 0.0f;
        const float* p = X + r * k;
        for (int c = 0; c < k; ++c) acc = __builtin_fmaf(w1 * p[c], p[c], acc);
        if (X2) {
            const float* q = X2 + r * k;
            for (int c = 0; c < k; ++c) acc = __builtin_fmaf(w2 * q[c], q[c], acc);
        }
        out[r] = -acc;
    }
}

// out[r, :] = [w1 * X[r,:], w2 * X2[r,:]]
__global__ __launch_bounds__(256) void concat_scale_kernel(const float* __restrict__ X, const float* __restrict__ X2,
                                                           int64_t rows, int k, float w1, float w2,
                                                           float* __restrict__ out) {
    const int64_t total = rows * 2 * k;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / (2 * k);
        const int c = (int)(e - r * 2 * k);
        out[e] = (c < k) ? w1 * X[r * k + c] : w2 * X2[r * k + (c - k)];
    }
}

__global__ __launch_bounds__(256) void add_vec_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) a[e] = a[e] + b[e];
}

// ---- selection on a given score matrix: one block per user, K rounds of block-wide "next best" ----
__global__ __launch_bounds__(256) void topk_rank_matrix_kernel(const float* __restrict__ scores, int64_t Nu, int64_t Ni,
                                                               const int32_t* __restrict__ train_ptr,
                                                               const int32_t* __restrict__ train_idx,
                                                               const int32_t* __restrict__ tgt_ptr,
                                                               const int32_t* __restrict__ tgt_idx,
                                                               const float* __restrict__ tgt_score,
                                                               const int32_t* __restrict__ tgt_masked, int K,
                                                               int32_t* __restrict__ topk_idx,
                                                               float* __restrict__ topk_score,
                                                               int32_t* __restrict__ tgt_rank) {
    __shared__ float rv[256];
    __shared__ int ri[256];
    __shared__ int rc[256];
    const int tid = threadIdx.x;
    for (int64_t u = blockIdx.x; u < Nu; u += gridDim.x) {
        const float* srow = scores + u * Ni;
        const int tb = train_ptr ? train_ptr[u] : 0, te = train_ptr ? train_ptr[u + 1] : 0;
        auto is_masked = [&](int item) {
            int lo = tb, hi = te;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (train_idx[mid] < item) lo = mid + 1; else hi = mid;
            }
            return lo < te && train_idx[lo] == item;
        };
        float lastv = INFINITY;
        int lasti = -1;  // everything is "after" (+inf, -1)
        for (int o = 0; o < K; ++o) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int64_t c = tid; c < Ni; c += 256) {
                const float s = srow[c];
                if (s != s) continue;
                if (!beats(lastv, lasti, s, (int)c)) continue;  // not strictly after the last pick
                if (!beats(s, (int)c, bv, bi)) continue;
                if (is_masked((int)c)) continue;
                bv = s;
                bi = (int)c;
            }
            rv[tid] = bv;
            ri[tid] = bi;
            __syncthreads();
            for (int st = 128; st > 0; st >>= 1) {
                if (tid < st && beats(rv[tid + st], ri[tid + st], rv[tid], ri[tid])) {
                    rv[tid] = rv[tid + st];
                    ri[tid] = ri[tid + st];
                }
                __syncthreads();
            }
            lastv = rv[0];
            lasti = ri[0];
            __syncthreads();
            if (lasti == 0x7fffffff) {  // candidates exhausted
                if (tid == 0)
                    for (int z = o; z < K; ++z) {
                        topk_idx[u * K + z] = -1;
                        topk_score[u * K + z] = -INFINITY;
                    }
                break;
            }
            if (tid == 0) {
                topk_idx[u * K + o] = lasti;
                topk_score[u * K + o] = lastv;
            }
        }
        if (tgt_ptr) {
            for (int x = tgt_ptr[u]; x < tgt_ptr[u + 1]; ++x) {
                if (tgt_masked[x]) continue;  // block-uniform
                const float st = tgt_score[x];
                const int ct = tgt_idx[x];
                int cnt = 0;
                for (int64_t c = tid; c < Ni; c += 256) {
                    const float s = srow[c];
                    if (s != s) continue;
                    if (beats(s, (int)c, st, ct) && !is_masked((int)c)) ++cnt;
                }
                rc[tid] = cnt;
                __syncthreads();
                for (int s2 = 128; s2 > 0; s2 >>= 1) {
                    if (tid < s2) rc[tid] += rc[tid + s2];
                    __syncthreads();
                }
                if (tid == 0) tgt_rank[x] = rc[0];
                __syncthreads();
            }
        }
    }
}

struct TopkLayout {
    int64_t uadd, iadd, ucat, icat, tgt_score, tgt_masked, part_v, part_i, total;
    int S, ntiles, tiles_per_split;
};

static void topk_layout(int64_t Nu, int64_t Ni, int k, int K, int64_t ntgt, int kind, TopkLayout* L) {
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += align_up(bytes > 0 ? bytes : 1, 256);
        return o;
    };
    L->ntiles = (int)((Ni + TI - 1) / TI);
    if (L->ntiles < 1) L->ntiles = 1;
    int64_t panels = (Nu + TU - 1) / TU;
    if (panels < 1) panels = 1;
    int S = (int)((1024 + panels - 1) / panels);
    if (S > L->ntiles) S = L->ntiles;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    L->tiles_per_split = (L->ntiles + S - 1) / S;
    L->S = (L->ntiles + L->tiles_per_split - 1) / L->tiles_per_split;
    L->uadd = take(Nu * 4);
    L->iadd = take(Ni * 4);
    const bool two = kind == DR_SCORE_TWO_DIST;
    L->ucat = take(two ? Nu * 2 * (int64_t)k * 4 : 0);
    L->icat = take(two ? Ni * 2 * (int64_t)k * 4 : 0);
    L->tgt_score = take(ntgt * 4);
    L->tgt_masked = take(ntgt * 4);
    L->part_v = take((int64_t)L->S * Nu * K * 4);
    L->part_i = take((int64_t)L->S * Nu * K * 4);
    L->total = off;
}

static size_t score_smem_bytes(int K) {
    return (size_t)(2 * TU * LDK + TU * LDS_S) * 4 + (size_t)TU * K * 8   // operands, score tile, top-K lists
           + (size_t)TU * 4 + (size_t)TU * 16 + (size_t)(TU + 1) * 4 + 4    // tau, maskw + candw, tptr (+pad)
           + (size_t)TCAP * 12;                                            // cached targets
}

// Prepare (U', I', uadd, iadd, cdot) for a score kind.  Returns DR_OK / error.
static int prepare_operands(const float* U, const float* U2, int64_t Nu, const float* I, const float* I2,
                            const float* item_bias, int64_t Ni, int k, int kind, float alpha, char* base,
                            const TopkLayout& L, hipStream_t st, ScoreArgs* a) {
    a->U = U;
    a->I = I;
    a->k = k;
    a->uadd = nullptr;
    a->iadd = item_bias;
    a->cdot = 1.0f;
    if (kind == DR_SCORE_DOT) return DR_OK;
    float* uadd = (float*)(base + L.uadd);
    float* iadd = (float*)(base + L.iadd);
    a->cdot = 2.0f;
    if (kind == DR_SCORE_NEG_SQDIST) {
        if (Nu > 0) DR_LAUNCH(neg_sqnorm_kernel, dim3(grid_for(Nu, 256)), dim3(256), 0, st, U, (const float*)nullptr, Nu, k, 1.0f, 0.0f, uadd);
        if (Ni > 0) DR_LAUNCH(neg_sqnorm_kernel, dim3(grid_for(Ni, 256)), dim3(256), 0, st, I, (const float*)nullptr, Ni, k, 1.0f, 0.0f, iadd);
    } else {
        const float w1 = alpha, w2 = 1.0f - alpha;
        float* ucat = (float*)(base + L.ucat);
        float* icat = (float*)(base + L.icat);
        if (Nu > 0) {
            DR_LAUNCH(neg_sqnorm_kernel, dim3(grid_for(Nu, 256)), dim3(256), 0, st, U, U2, Nu, k, w1, w2, uadd);
            DR_LAUNCH(concat_scale_kernel, dim3(grid_for(Nu * 2 * k, 256)), dim3(256), 0, st, U, U2, Nu, k, w1, w2, ucat);
        }
        if (Ni > 0) {
            DR_LAUNCH(neg_sqnorm_kernel, dim3(grid_for(Ni, 256)), dim3(256), 0, st, I, I2, Ni, k, w1, w2, iadd);
            DR_LAUNCH(concat_scale_kernel, dim3(grid_for(Ni * 2 * k, 256)), dim3(256), 0, st, I, I2, Ni, k, 1.0f, 1.0f, icat);
        }
        a->U = ucat;
        a->I = icat;
        a->k = 2 * k;
    }
    if (item_bias && Ni > 0)
        DR_LAUNCH(add_vec_kernel, dim3(grid_for(Ni, 256)), dim3(256), 0, st, iadd, item_bias, Ni);
    a->uadd = uadd;
    a->iadd = iadd;
    DR_CHECK_LAUNCH("score operand preparation");
    return DR_OK;
}

static int check_score_args(const float* U, const float* U2, int64_t Nu, const float* I, const float* I2, int64_t Ni,
                            int k, int kind, const char* who) {
    DR_REQUIRE(Nu >= 0 && Ni >= 0 && k >= 1, "%s: bad sizes", who);
    DR_REQUIRE(Ni < (1ll << 31) - 64, "%s: Ni too large for int32 item ids", who);
    DR_REQUIRE(kind >= DR_SCORE_DOT && kind <= DR_SCORE_TWO_DIST, "%s: bad score_kind %d", who, kind);
    DR_REQUIRE((U || Nu == 0) && (I || Ni == 0), "%s: U / I is NULL", who);
    DR_REQUIRE(kind != DR_SCORE_TWO_DIST || ((U2 || Nu == 0) && (I2 || Ni == 0)), "%s: two-distance score needs U2 / I2",
               who);
    return DR_OK;
}

}  // namespace dr

using namespace dr;

extern "C" int64_t dr_score_topk_bytes(int64_t Nu, int64_t Ni, int32_t k, int32_t K, int64_t n_targets,
                                       int32_t score_kind) {
    if (Nu < 0 || Ni < 0 || k < 1 || K < 1 || n_targets < 0) return DR_ERR_ARG;
    TopkLayout L;
    topk_layout(Nu, Ni, k, K, n_targets, score_kind, &L);
    return L.total;
}

extern "C" int32_t dr_score_topk_f32(const float* U, const float* U2, int64_t Nu, const float* I, const float* I2,
                                     const float* item_bias, int64_t Ni, int32_t k, int32_t score_kind, float alpha,
                                     const int32_t* train_ptr, const int32_t* train_idx, const int32_t* tgt_ptr,
                                     const int32_t* tgt_idx, int64_t n_targets, int32_t K, int32_t* topk_idx,
                                     float* topk_score, int32_t* tgt_rank, void* ws, int64_t ws_bytes, void* stream) {
    int rc = check_score_args(U, U2, Nu, I, I2, Ni, k, score_kind, "dr_score_topk_f32");
    if (rc != DR_OK) return rc;
    DR_REQUIRE(K >= 1 && K <= MAXK_TOP, "dr_score_topk_f32: K must be in [1,%d]", MAXK_TOP);
    DR_REQUIRE(n_targets >= 0 && (n_targets == 0 || (tgt_ptr && tgt_idx && tgt_rank)),
               "dr_score_topk_f32: target CSR / rank output is NULL");
    DR_REQUIRE((train_ptr == nullptr) == (train_idx == nullptr) || train_ptr, "dr_score_topk_f32: train CSR incomplete");
    DR_REQUIRE(topk_idx && topk_score, "dr_score_topk_f32: top-K outputs are NULL");
    DR_REQUIRE(ws && ((uintptr_t)ws % 256 == 0), "dr_score_topk_f32: workspace NULL / misaligned");
    if (Nu == 0) return DR_OK;
    TopkLayout L;
    topk_layout(Nu, Ni, k, K, n_targets, score_kind, &L);
    if (ws_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_score_topk_f32: workspace %lld < required %lld", (long long)ws_bytes,
                (long long)L.total);
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    ScoreArgs a;
    memset(&a, 0, sizeof(a));
    rc = prepare_operands(U, U2, Nu, I, I2, item_bias, Ni, k, score_kind, alpha, base, L, st, &a);
    if (rc != DR_OK) return rc;
    a.Nu = Nu;
    a.Ni = Ni;
    a.train_ptr = train_ptr;
    a.train_idx = train_idx;
    a.K = K;
    a.ntiles = L.ntiles;
    a.tiles_per_split = L.tiles_per_split;
    a.part_v = (float*)(base + L.part_v);
    a.part_i = (int32_t*)(base + L.part_i);
    if (n_targets > 0) {
        a.tgt_ptr = tgt_ptr;
        a.tgt_idx = tgt_idx;
        a.tgt_score = (float*)(base + L.tgt_score);
        a.tgt_masked = (int32_t*)(base + L.tgt_masked);
        a.tgt_rank = tgt_rank;
        DR_LAUNCH(target_prologue_kernel, dim3(grid_for(Nu, 256)), dim3(256), 0, st, a.U, a.I, a.uadd, a.iadd,
                           a.cdot, Nu, Ni, a.k, (const float*)nullptr, train_ptr, train_idx, tgt_ptr, tgt_idx,
                           (float*)(base + L.tgt_score), (int32_t*)(base + L.tgt_masked), tgt_rank);
        DR_CHECK_LAUNCH("target_prologue_kernel");
    }
    const size_t smem = score_smem_bytes(K);
    DR_HIP(hipFuncSetAttribute((const void*)score_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int64_t panels = (Nu + TU - 1) / TU;
    DR_REQUIRE(panels < (1ll << 31), "dr_score_topk_f32: too many user panels");
    DR_LAUNCH((score_kernel<false>), dim3((unsigned)panels, (unsigned)L.S), dim3(256), smem, st, a);
    DR_CHECK_LAUNCH("score_kernel");
    DR_LAUNCH(topk_merge_kernel, dim3(grid_for(Nu, 256)), dim3(256), 0, st, a.part_v, a.part_i, L.S, Nu, K,
                       topk_idx, topk_score);
    DR_CHECK_LAUNCH("topk_merge_kernel");
    if (n_targets > 0) {
        DR_LAUNCH(target_finalize_kernel, dim3(grid_for(n_targets, 256)), dim3(256), 0, st,
                           (const int32_t*)(base + L.tgt_masked), n_targets, tgt_rank);
        DR_CHECK_LAUNCH("target_finalize_kernel");
    }
    return DR_OK;
}

extern "C" int32_t dr_score_matrix_f32(const float* U, const float* U2, int64_t Nu, const float* I, const float* I2,
                                       const float* item_bias, int64_t Ni, int32_t k, int32_t score_kind, float alpha,
                                       int32_t act, float* out, void* ws, int64_t ws_bytes, void* stream) {
    int rc = check_score_args(U, U2, Nu, I, I2, Ni, k, score_kind, "dr_score_matrix_f32");
    if (rc != DR_OK) return rc;
    DR_REQUIRE(act == 0 || act == 1, "dr_score_matrix_f32: act must be 0 (none) or 1 (sigmoid)");
    if (Nu == 0 || Ni == 0) return DR_OK;
    DR_REQUIRE(out != nullptr, "dr_score_matrix_f32: out is NULL");
    TopkLayout L;
    topk_layout(Nu, Ni, k, 1, 0, score_kind, &L);
    DR_REQUIRE(score_kind == DR_SCORE_DOT || (ws && ((uintptr_t)ws % 256 == 0)),
               "dr_score_matrix_f32: distance kinds need a workspace (dr_score_topk_bytes)");
    if (score_kind != DR_SCORE_DOT && ws_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_score_matrix_f32: workspace %lld < required %lld", (long long)ws_bytes,
                (long long)L.total);
    hipStream_t st = (hipStream_t)stream;
    ScoreArgs a;
    memset(&a, 0, sizeof(a));
    rc = prepare_operands(U, U2, Nu, I, I2, item_bias, Ni, k, score_kind, alpha, (char*)ws, L, st, &a);
    if (rc != DR_OK) return rc;
    a.Nu = Nu;
    a.Ni = Ni;
    a.K = 1;
    a.act = act;
    a.out = out;
    a.ntiles = L.ntiles;
    // matrix mode: one block per (user panel, group of item tiles)
    int64_t panels = (Nu + TU - 1) / TU;
    int S = (int)((2048 + panels - 1) / panels);
    if (S > L.ntiles) S = L.ntiles;
    if (S < 1) S = 1;
    if (S > 65535) S = 65535;
    a.tiles_per_split = (L.ntiles + S - 1) / S;
    S = (L.ntiles + a.tiles_per_split - 1) / a.tiles_per_split;
    const size_t smem = score_smem_bytes(1);
    DR_HIP(hipFuncSetAttribute((const void*)score_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DR_LAUNCH((score_kernel<true>), dim3((unsigned)panels, (unsigned)S), dim3(256), smem, st, a);
    DR_CHECK_LAUNCH("score_kernel<matrix>");
    return DR_OK;
}

extern "C" int32_t dr_topk_rank_f32(const float* scores, int64_t Nu, int64_t Ni, const int32_t* train_ptr,
                                    const int32_t* train_idx, const int32_t* tgt_ptr, const int32_t* tgt_idx,
                                    int64_t n_targets, int32_t K, int32_t* topk_idx, float* topk_score,
                                    int32_t* tgt_rank, void* ws, int64_t ws_bytes, void* stream) {
    DR_REQUIRE(Nu >= 0 && Ni >= 0 && K >= 1, "dr_topk_rank_f32: bad sizes");
    DR_REQUIRE(Ni < (1ll << 31) - 64, "dr_topk_rank_f32: Ni too large for int32 item ids");
    DR_REQUIRE(n_targets >= 0 && (n_targets == 0 || (tgt_ptr && tgt_idx && tgt_rank && ws)),
               "dr_topk_rank_f32: target CSR / rank output / workspace is NULL");
    if (Nu == 0) return DR_OK;
    DR_REQUIRE(scores || Ni == 0, "dr_topk_rank_f32: scores is NULL");
    DR_REQUIRE(topk_idx && topk_score, "dr_topk_rank_f32: top-K outputs are NULL");
    hipStream_t st = (hipStream_t)stream;
    float* tgt_score = nullptr;
    int32_t* tgt_masked = nullptr;
    if (n_targets > 0) {
        const int64_t need = 2 * align_up(n_targets * 4, 256);
        if (ws_bytes < need)
            DR_FAIL(DR_ERR_WORKSPACE, "dr_topk_rank_f32: workspace %lld < required %lld", (long long)ws_bytes,
                    (long long)need);
        tgt_score = (float*)ws;
        tgt_masked = (int32_t*)((char*)ws + align_up(n_targets * 4, 256));
        DR_LAUNCH(target_prologue_kernel, dim3(grid_for(Nu, 256)), dim3(256), 0, st, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 1.0f, Nu, Ni, 0, scores,
                           train_ptr, train_idx, tgt_ptr, tgt_idx, tgt_score, tgt_masked, tgt_rank);
        DR_CHECK_LAUNCH("target_prologue_kernel");
    }
    int grid = (int)(Nu < 256 * 8 ? Nu : 256 * 8);
    DR_LAUNCH(topk_rank_matrix_kernel, dim3(grid), dim3(256), 0, st, scores, Nu, Ni, train_ptr, train_idx,
                       n_targets > 0 ? tgt_ptr : nullptr, tgt_idx, tgt_score, tgt_masked, K, topk_idx, topk_score,
                       tgt_rank);
    DR_CHECK_LAUNCH("topk_rank_matrix_kernel");
    if (n_targets > 0) {
        DR_LAUNCH(target_finalize_kernel, dim3(grid_for(n_targets, 256)), dim3(256), 0, st, tgt_masked,
                           n_targets, tgt_rank);
        DR_CHECK_LAUNCH("target_finalize_kernel");
    }
    return DR_OK;
}
