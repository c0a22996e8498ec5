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

#ifndef DR_ABL
#define DR_ABL 0   // timing ablations of score_kernel_t1 (tools/build_abl.sh); 0 = the product
#endif
constexpr int TU = 64;   // users per block tile
constexpr int TI = 64;   // items per tile step
constexpr int KC = 128;  // k-chunk staged in LDS
constexpr int LDK = KC + 1;
constexpr int LDS_S = TI + 4;  // score tile row stride: 16-byte aligned rows, conflict-free for both access patterns
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
    const int32_t* mode_flag;  // device word: 0 = every user has <= 1 held-out item (score_kernel_t1 runs), 1 = general
    // BAND kernels (dr_score_rank_exact_f32): tgt_score[x] / tgt_hi[x] = the held-out item's reference-order score
    // minus / plus a bound on |MFMA score - reference-order score|.  A candidate above the band counts, one below
    // does not, one inside is appended to band_list as (user row, target slot, item) for the exact re-score.
    const float* tgt_hi;
    int32_t* band_list;
    int32_t* band_count;       // [0] entries appended (may exceed band_cap: then the list overflowed)
    int band_cap;
};

__device__ __forceinline__ float score_epilogue(float acc, float cdot, const float* uadd, const float* iadd, int64_t u,
                                                int64_t i) {
    float s = acc;
    if (cdot != 1.0f) s = cdot * s;
    if (uadd) s = s + uadd[u];
    if (iadd) s = s + iadd[i];
    return s;
}

constexpr int TCAP = 512;  // held-out items of one 64-user panel cached in LDS (more: read from global)

// ---- selection state of one 256-thread group (one 64-user panel over a range of item tiles) -------------
// Thread th of the group looks at row th/4, columns (th%4)*16 .. +15 of each 64x64 score tile; the four
// threads of a row are adjacent lanes of one wave, and lane th%4 == 0 is the row's owner: it keeps the row's
// train-CSR cursor and K-th best score in registers and maintains the row's sorted top-K list in LDS.
// Quad-local exchange is DPP (no LDS, no barrier).
//
// LDS layout after the operand tiles (float offsets; everything but the top-K lists at compile-time offsets)
constexpr int SS_OFF = 0;                         // [TU][LDS_S] score tile
constexpr int TPTR_OFF = SS_OFF + TU * LDS_S;     // [TU+1] panel-local target offsets (+1 pad)
constexpr int TSC_OFF = TPTR_OFF + TU + 2;        // [TCAP] target scores
constexpr int TID_OFF = TSC_OFF + TCAP;           // [TCAP] target item id, -1 = masked target
constexpr int TCNT_OFF = TID_OFF + TCAP;          // [TCAP] rank counters
constexpr int THI_OFF = TCNT_OFF + TCAP;          // [TCAP] upper end of the target's band (BAND kernels only)
constexpr int TOPV_OFF = THI_OFF + TCAP + 2;      // [TU][K] scores, then [TU][K] item ids
static_assert((SS_OFF % 4) == 0 && (TOPV_OFF % 4) == 0, "LDS alignment");

struct Sel {
    float* base;
    int cur, cur_end, nxt, nxt2;  // row owner: train cursor, next two train items cached
    float tau;                    // row owner: K-th best so far (NaN: list not full, +inf: no such user)
    int tg0, ntg;                 // this panel's slice of the target CSR
    bool tg_lds, owner;
    __device__ __forceinline__ float* sS() const { return base + SS_OFF; }
    __device__ __forceinline__ int* tptr() const { return (int*)(base + TPTR_OFF); }
    __device__ __forceinline__ float* tsc() const { return base + TSC_OFF; }
    __device__ __forceinline__ int* tidl() const { return (int*)(base + TID_OFF); }
    __device__ __forceinline__ int* tcnt() const { return (int*)(base + TCNT_OFF); }
    __device__ __forceinline__ float* thi() const { return base + THI_OFF; }
    __device__ __forceinline__ float* topv() const { return base + TOPV_OFF; }
    __device__ __forceinline__ int* topi(int K) const { return (int*)(base + TOPV_OFF + TU * K); }
};

// DPP quad permutes: every lane reads lane (quad base + SRC) / lane ^ 1 / lane ^ 2 of its quad
template <int SRC>
__device__ __forceinline__ int quad_get(int v) {
    return __builtin_amdgcn_mov_dpp(v, SRC * 0x55, 0xf, 0xf, true);
}
__device__ __forceinline__ int quad_sum(int v) {
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    return v;
}

// th = thread index within the 256-thread group; a barrier must follow before the first sel_tile
__device__ __forceinline__ void sel_init(Sel& z, const ScoreArgs& a, int64_t u0, int t_begin, int th) {
    const int K = a.K;
    const int64_t myu = u0 + (th >> 2);
    z.owner = (th & 3) == 0 && myu < a.Nu;
    z.cur = 0;
    z.cur_end = 0;
    z.nxt = 0x7fffffff;
    z.nxt2 = 0x7fffffff;
    z.tau = (myu < a.Nu) ? __builtin_nanf("") : INFINITY;
    z.tg0 = 0;
    z.ntg = 0;
    z.tg_lds = false;
    for (int e = th; e < TU * K; e += 256) {
        z.topv()[e] = -INFINITY;
        z.topi(K)[e] = 0x7fffffff;
    }
    if (z.owner && a.train_ptr) {
        z.cur = a.train_ptr[myu];
        z.cur_end = a.train_ptr[myu + 1];
        const int first = t_begin * TI;
        int lo = z.cur, hi = z.cur_end;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (a.train_idx[mid] < first) lo = mid + 1; else hi = mid;
        }
        z.cur = lo;
        z.nxt = z.cur < z.cur_end ? a.train_idx[z.cur] : 0x7fffffff;
        z.nxt2 = z.cur + 1 < z.cur_end ? a.train_idx[z.cur + 1] : 0x7fffffff;
    }
    if (th <= TU) z.tptr()[th] = 0;
    if (a.tgt_ptr) {
        const int64_t ulast = (u0 + TU < a.Nu) ? u0 + TU : a.Nu;
        const int64_t ufirst = u0 < a.Nu ? u0 : a.Nu;
        z.tg0 = a.tgt_ptr[ufirst];
        z.ntg = a.tgt_ptr[ulast > ufirst ? ulast : ufirst] - z.tg0;
        z.tg_lds = z.ntg <= TCAP;
        if (th <= TU) {
            const int64_t u = (u0 + th < a.Nu) ? u0 + th : a.Nu;
            z.tptr()[th] = a.tgt_ptr[u] - z.tg0;
        }
        if (z.tg_lds)
            for (int x = th; x < z.ntg; x += 256) {
                z.tsc()[x] = a.tgt_score[z.tg0 + x];
                z.tidl()[x] = a.tgt_masked[z.tg0 + x] ? -1 : a.tgt_idx[z.tg0 + x];
                z.tcnt()[x] = 0;
                if (a.tgt_hi) z.thi()[x] = a.tgt_hi[z.tg0 + x];
            }
    }
}

// this lane's 16 uadd values (its rows are fixed for the whole item stream); -0.0f is the exact neutral element
__device__ __forceinline__ void load_uadd(float (&ua)[16], const ScoreArgs& a, int64_t u0, int wu, int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t u = u0 + wu * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        ua[r] = (a.uadd && u < a.Nu) ? a.uadd[u] : -0.0f;
    }
}
__device__ __forceinline__ float load_iadd(const ScoreArgs& a, int64_t it) {
    return (a.iadd && it < a.Ni) ? a.iadd[it] : -0.0f;
}

// 32x32 accumulator tile of wave (wu, wi) -> LDS score tile with the affine epilogue
//   s = cdot * acc + uadd[u] + iadd[i];  ua[r] = uadd of this lane's r-th row, ia = iadd of its column.
// Rows / columns beyond Nu / Ni hold don't-care values: the selection never looks at them.
__device__ __forceinline__ void sel_store_scores(const Sel& z, const ScoreArgs& a, const f32x16& acc,
                                                 const float (&ua)[16], float ia, bool affine, int wu, int wi,
                                                 int lane) {
    float* d = z.sS() + (wu * 32 + 4 * (lane >> 5)) * LDS_S + wi * 32 + (lane & 31);
    if (affine) {
        const float cd = a.cdot;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = cd * acc[r];
            s = s + ua[r];
            s = s + ia;
            d[((r & 3) + 8 * (r >> 2)) * LDS_S] = s;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2)) * LDS_S] = acc[r];
    }
}

// The score tile of items [i0, i0+64) is in sS (barrier before).  MATRIX: write it out.  Otherwise the
// threshold filter for the top-K, the rank-of-target counting and the fold into the row's sorted list.
// Common case first: no masked item in this quarter and a full tile -> one group maximum against the K-th
// best, and a plain count of scores above the target (ties and masked / ragged quarters take the exact
// per-score path).  Must be called by whole waves (DPP).
template <bool MATRIX, bool TG_LDS, bool BAND>
__device__ __forceinline__ void sel_tile_impl(Sel& z, const ScoreArgs& a, int64_t u0, int64_t i0, int th) {
    if (MATRIX) {
        for (int e = th; e < TU * TI; e += 256) {
            const int r = e / TI, c = e - r * TI;
            const int64_t u = u0 + r, i = i0 + c;
            if (u < a.Nu && i < a.Ni) {
                float s = z.sS()[r * LDS_S + c];
                if (a.act == 1) s = 1.0f / (1.0f + expf(-s));
                a.out[u * a.Ni + i] = s;
            }
        }
        return;
    }
    const int row = th >> 2, q = th & 3;
    const float4* s4 = reinterpret_cast<const float4*>(z.sS() + row * LDS_S + q * 16);
    float sv[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 v = s4[j];
        sv[4 * j] = v.x; sv[4 * j + 1] = v.y; sv[4 * j + 2] = v.z; sv[4 * j + 3] = v.w;
    }
    // row owner: train-item bitmap of this tile, then its state goes to the quad
    unsigned mlo = 0u, mhi = 0u;
    if (z.owner) {
        // The item after next is always already loaded, so the usual single advance of a tile does not
        // wait on memory (and not on the tile prefetch in flight); only a second advance in one tile does.
        const int hi = (int)i0 + TI;
        if (z.nxt < hi) {
            int b = z.nxt - (int)i0;
            if (b < 32) mlo |= 1u << b; else mhi |= 1u << (b - 32);
            ++z.cur;
            z.nxt = z.nxt2;
            z.nxt2 = z.cur + 1 < z.cur_end ? a.train_idx[z.cur + 1] : 0x7fffffff;
            while (z.nxt < hi) {
                b = z.nxt - (int)i0;
                if (b < 32) mlo |= 1u << b; else mhi |= 1u << (b - 32);
                ++z.cur;
                z.nxt = z.nxt2;
                z.nxt2 = z.cur + 1 < z.cur_end ? a.train_idx[z.cur + 1] : 0x7fffffff;
            }
        }
    }
    const unsigned qlo = (unsigned)quad_get<0>((int)mlo), qhi = (unsigned)quad_get<0>((int)mhi);  // all lanes active
    const unsigned wsel = (q & 2) ? qhi : qlo;
    const unsigned mrow = (wsel >> ((q & 1) * 16)) & 0xffffu;
    const float th_k = __int_as_float(quad_get<0>(__float_as_int(z.tau)));
    const int nc = (a.Ni - i0 < TI) ? (int)(a.Ni - i0) : TI;
    const bool plain = (mrow == 0u) && (nc == TI);
    unsigned live = 0xffffu;
    if (!plain) {
        live = 0u;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const bool ok = (q * 16 + c < nc) && !((mrow >> c) & 1u) && (sv[c] == sv[c]);
            live |= ok ? (1u << c) : 0u;
        }
    }
    float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
    mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(sv[8], sv[9]), fmaxf(sv[10], sv[11])),
                         fmaxf(fmaxf(sv[12], sv[13]), fmaxf(sv[14], sv[15]))));
    unsigned cand = 0u;
    if (!(mx <= th_k)) {  // also taken while the list is not full (th_k = NaN)
#pragma unroll
        for (int c = 0; c < 16; ++c)
            cand |= (((live >> c) & 1u) && (sv[c] == sv[c]) && !(sv[c] <= th_k)) ? (1u << c) : 0u;
    }
    // rank-of-target counting: the quad's four partial counts are summed and the owner adds them up
    const int tb = z.tptr()[row], te = z.tptr()[row + 1];
    for (int x = tb; x < te; ++x) {  // same trip count on the four lanes of a row
        const int ct = TG_LDS ? z.tidl()[x] : (a.tgt_masked[z.tg0 + x] ? -1 : a.tgt_idx[z.tg0 + x]);
        if (ct < 0) continue;
        const float st = TG_LDS ? z.tsc()[x] : a.tgt_score[z.tg0 + x];
        int cnt = 0;
        bool tie = false;
        if (BAND) {
            // st = lower end of the band, hi = upper end: above counts, inside goes to the exact re-score
            const float hi = TG_LDS ? z.thi()[x] : a.tgt_hi[z.tg0 + x];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                cnt += (sv[c] > hi) ? 1 : 0;
                tie |= (sv[c] >= st) && (sv[c] <= hi);
            }
            if (tie || !plain) {
                const int item0 = (int)i0 + q * 16;
                cnt = 0;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (!((live >> c) & 1u)) continue;
                    if (sv[c] > hi) {
                        ++cnt;
                    } else if (sv[c] >= st && item0 + c != ct) {
                        const int p = atomicAdd(a.band_count, 1);
                        if (p < a.band_cap) {
                            a.band_list[3 * (int64_t)p] = (int32_t)(u0 + row);
                            a.band_list[3 * (int64_t)p + 1] = z.tg0 + x;
                            a.band_list[3 * (int64_t)p + 2] = item0 + c;
                        }
                    }
                }
            }
        } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            cnt += (sv[c] > st) ? 1 : 0;
            tie |= (sv[c] == st);
        }
        }
        if (!BAND && (tie || !plain)) {
            const int item0 = (int)i0 + q * 16;
            cnt = 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) cnt += (((live >> c) & 1u) && beats(sv[c], item0 + c, st, ct)) ? 1 : 0;
        }
        cnt = quad_sum(cnt);
        if (q == 0 && cnt) {
            if (TG_LDS) z.tcnt()[x] += cnt;  // only this lane ever touches entry x
            else atomicAdd(a.tgt_rank + z.tg0 + x, cnt);
        }
    }
    // candidates of the row -> its owner, which folds them into the sorted top-K list
    const unsigned c1 = (unsigned)quad_get<1>((int)cand), c2 = (unsigned)quad_get<2>((int)cand),
                   c3 = (unsigned)quad_get<3>((int)cand);
    if (z.owner) {
        unsigned long long m = (unsigned long long)(cand | (c1 << 16)) | ((unsigned long long)(c2 | (c3 << 16)) << 32);
        if (m) {
            const int K = a.K;
            const float* srow = z.sS() + row * LDS_S;
            float* tv = z.topv() + row * K;
            int* ti = z.topi(K) + row * K;
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
            if (ti[K - 1] != 0x7fffffff) z.tau = tv[K - 1];
        }
    }
}

// two instantiations, so that the usual one (targets cached in LDS) has no global access in its target loop:
// a wait on one would also wait for the item-tile prefetch in flight
template <bool MATRIX, bool BAND = false>
__device__ __forceinline__ void sel_tile(Sel& z, const ScoreArgs& a, int64_t u0, int64_t i0, int th) {
    if (MATRIX || z.tg_lds) sel_tile_impl<MATRIX, true, BAND && !MATRIX>(z, a, u0, i0, th);
    else sel_tile_impl<MATRIX, false, BAND && !MATRIX>(z, a, u0, i0, th);
}

// a barrier must precede (top-K lists and counters are written by other threads of the group)
__device__ __forceinline__ void sel_finish(const Sel& z, const ScoreArgs& a, int64_t u0, int split, int th) {
    const int K = a.K;
    for (int e = th; e < TU * K; e += 256) {
        const int r = e / K, c = e - r * K;
        const int64_t u = u0 + r;
        if (u < a.Nu) {
            const int64_t o = ((int64_t)split * a.Nu + u) * K + c;
            a.part_v[o] = z.topv()[e];
            a.part_i[o] = z.topi(K)[e];
        }
    }
    if (a.tgt_ptr && z.tg_lds)
        for (int x = th; x < z.ntg; x += 256)
            if (z.tcnt()[x]) atomicAdd(a.tgt_rank + z.tg0 + x, z.tcnt()[x]);
}

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

// ---- any k: user panel and item tile both staged through LDS in k-chunks (one block per CU) -----------
template <bool MATRIX, bool BAND = false>
__global__ __launch_bounds__(256) void score_kernel_anyk(ScoreArgs a) {
    extern __shared__ __align__(16) float smem[];
    float* sU = smem;                // [TU][LDK]
    float* sI = sU + TU * LDK;       // [TI][LDK]
    Sel z;
    z.base = sI + TI * LDK;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wu = wave >> 1, wi = wave & 1;
    const int64_t u0 = (int64_t)blockIdx.x * TU;
    const int split = blockIdx.y;
    const int t_begin = split * a.tiles_per_split;
    int t_end = t_begin + a.tiles_per_split;
    if (t_end > a.ntiles) t_end = a.ntiles;
    if (!MATRIX) sel_init(z, a, u0, t_begin, tid);
    const bool one_chunk = a.k <= KC;
    const bool affine = a.uadd || a.iadd || a.cdot != 1.0f;
    float ua[16];
    load_uadd(ua, a, u0, wu, lane);

    for (int t = t_begin; t < t_end; ++t) {
        const int64_t i0 = (int64_t)t * TI;
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
        sel_store_scores(z, a, acc, ua, load_iadd(a, i0 + wi * 32 + (lane & 31)), affine, wu, wi, lane);
        __syncthreads();
        sel_tile<MATRIX, BAND>(z, a, u0, i0, tid);  // the next write of sS is two barriers away
    }
    if (!MATRIX) {
        __syncthreads();
        sel_finish(z, a, u0, split, tid);
    }
}

// ---- k <= 2*KS (the usual case): a group's user panel lives in REGISTERS as the MFMA A operand for the
// whole item stream (lane l holds U[row l%32][2j + l/32], j < KS); only item tiles go through LDS, and the
// next tile's global loads are issued a whole phase before they are needed. ----
//
// Item tile in LDS for the register kernels: row stride LDP floats (16-byte aligned rows), and inside each group of
// 8 k the even k first, then the odd k.  MFMA lane (n, h) needs B[n][2j + h] for consecutive j: with this layout
// four consecutive k-steps are ONE ds_read_b128 at slot 8*(j/4) + 4h, and a thread that fetched 8 consecutive k
// of a row (two float4) stores them as two ds_write_b128.  The permutation only moves values inside LDS: every
// MFMA still receives k = 2j + h, so the accumulation order -- and every bit of the result -- is unchanged.
constexpr int LDP = KC + 4;
// Every second run of four 8-k groups has its even / odd halves swapped (bit 5 of k): a thread that stores the 8 k of
// group g as two ds_write_b128 then hits 16-byte slot 2g + ((g >> 2) & 1) with the first and the other slot with the
// second, so the eight lanes that one LDS cycle serves cover all 32 banks (unswizzled, lanes g and g + 4 collided: every
// tile store ran at half rate, in the way of the co-resident block's operand reads).  Readers: mfma_chunk's pbx.
__device__ __forceinline__ constexpr int kpos(int c) {
    return (c & ~7) + (((c & 1) ^ ((c >> 5) & 1)) << 2) + ((c & 7) >> 1);
}

// Thread -> chunk element mapping, all compile-time: W floats per element (8 when k % 8 == 0, else 1), CPT
// elements per row (power of two covering the widest chunk of this instantiation), thread th owns column
// element th % CPT of rows th / CPT + q * RPP.  Global addresses are (uniform tile base) + (per-thread
// constant 32-bit offset); LDS addresses are (per-thread constant) + immediates.
template <int KS, bool VEC, int NT = 256>
struct Geo {
    static constexpr int W = VEC ? 8 : 1;
    static constexpr int CPT = VEC ? (KS >= 64 ? 16 : 8) : (KS >= 64 ? 128 : 64);
    static constexpr int RPP = NT / CPT;
    static constexpr int PASSES = TI / RPP;
    static constexpr int NPRE = PASSES * W;
};

template <int KS, bool VEC, int N, int NT = 256>
__device__ __forceinline__ void prefetch_chunk(float (&pre)[N], const float* __restrict__ I, int64_t i0, int64_t Ni,
                                               int k, int kc0, int r0, int gc) {
    using G = Geo<KS, VEC, NT>;
    const char* tile = reinterpret_cast<const char*>(I) + (i0 * k + kc0) * 4;  // uniform
    const bool full = i0 + TI <= Ni;
    const int rmax = full ? TI : (int)(Ni - 1 - i0);   // ragged last tile: rows beyond Ni re-read the last valid
    const uint32_t voff = ((uint32_t)r0 * (uint32_t)k + (uint32_t)gc * G::W) * 4u;  // row (never looked at)
#pragma unroll
    for (int q = 0; q < G::PASSES; ++q) {
        const char* p;
        if (full) {
            p = tile + (int64_t)(q * G::RPP) * k * 4 + voff;
        } else {
            int r = r0 + q * G::RPP;
            r = r < rmax ? r : rmax;
            p = tile + ((int64_t)r * k + gc * G::W) * 4;
        }
        if (VEC) {
            const float4 v0 = *reinterpret_cast<const float4*>(p);
            const float4 v1 = *reinterpret_cast<const float4*>(p + 16);
            pre[8 * q] = v0.x; pre[8 * q + 1] = v0.y; pre[8 * q + 2] = v0.z; pre[8 * q + 3] = v0.w;
            pre[8 * q + 4] = v1.x; pre[8 * q + 5] = v1.y; pre[8 * q + 6] = v1.z; pre[8 * q + 7] = v1.w;
        } else {
            pre[q] = *reinterpret_cast<const float*>(p);
        }
    }
}

// The values a prefetch left in `pre` become opaque here: the even / odd shuffle of store_chunk (and the s_waitcnt vmcnt
// it needs) cannot be hoisted from the store to right behind the loads -- where the compiler otherwise puts it, so that
// the wave waited for the tile it had just requested BEFORE its MFMA chain instead of after it.
template <int N>
__device__ __forceinline__ void pin_prefetched(float (&pre)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(pre[i]));
}

// d = sI + r0 * LDP + (VEC ? g * 8 + 4 * ((g >> 2) & 1) : kpos(g)): where the even k of the thread's group go (VEC: the
// odd k go to the other half of the group, d ^ 16 bytes)
template <int KS, bool VEC, int N, int NT = 256>
__device__ __forceinline__ void store_chunk(float* __restrict__ d, float* __restrict__ dodd, const float (&pre)[N]) {
    using G = Geo<KS, VEC, NT>;
#pragma unroll
    for (int q = 0; q < G::PASSES; ++q) {
        if (VEC) {
            *reinterpret_cast<float4*>(d + q * G::RPP * LDP) =
                make_float4(pre[8 * q], pre[8 * q + 2], pre[8 * q + 4], pre[8 * q + 6]);          // even k
            *reinterpret_cast<float4*>(dodd + q * G::RPP * LDP) =
                make_float4(pre[8 * q + 1], pre[8 * q + 3], pre[8 * q + 5], pre[8 * q + 7]);      // odd k
        } else {
            d[q * G::RPP * LDP] = pre[q];
        }
    }
}

// 8 * NG MFMAs of one k-chunk.  pb = sI + column * LDP + 4 * (lane >> 5); B operands arrive as one float4 per four
// k-steps, each group of 8 k-steps read one group ahead of its MFMAs.  ng = number of groups that hold data.
template <int NG>
__device__ __forceinline__ void mfma_chunk(f32x16& acc, const float* areg, const float* __restrict__ pb,
                                           const float* __restrict__ pbx, int ng) {
    // B operands are requested one group (8 MFMAs, 512 cycles) ahead of their use (two groups ahead: no faster for
    // k = 128, and the k = 256 instantiation spills more)
    float4 b[2][2];
    // pb = sI + column * LDP + 4 * h, pbx = the same with the halves swapped (8-k groups 4..7, 12..15: see kpos)
#define DR_PB(g_) ((((g_) >> 1) & 1) ? pbx : pb)   /* MFMA group g_ = 8-k groups 2 g_, 2 g_ + 1 */
    b[0][0] = *reinterpret_cast<const float4*>(pb);
    b[0][1] = *reinterpret_cast<const float4*>(pb + 8);
#define DR_MFMA_GROUP(gq)                                                                                          \
    {                                                                                                              \
        if ((gq) + 1 < NG) {                                                                                       \
            b[((gq) + 1) & 1][0] = *reinterpret_cast<const float4*>(DR_PB((gq) + 1) + ((gq) + 1) * 16);            \
            b[((gq) + 1) & 1][1] = *reinterpret_cast<const float4*>(DR_PB((gq) + 1) + ((gq) + 1) * 16 + 8);        \
        }                                                                                                          \
        __builtin_amdgcn_sched_barrier(0); /* keep the next group's reads ahead of this group's MFMAs */          \
        const float4 x = b[(gq) & 1][0], y = b[(gq) & 1][1];                                                       \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[(gq) * 8 + 0], x.x, acc, 0, 0, 0);                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[(gq) * 8 + 1], x.y, acc, 0, 0, 0);                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[(gq) * 8 + 2], x.z, acc, 0, 0, 0);                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[(gq) * 8 + 3], x.w, acc, 0, 0, 0);                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[(gq) * 8 + 4], y.x, acc, 0, 0, 0);                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[(gq) * 8 + 5], y.y, acc, 0, 0, 0);                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[(gq) * 8 + 6], y.z, acc, 0, 0, 0);                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[(gq) * 8 + 7], y.w, acc, 0, 0, 0);                         \
    }
    if (ng == NG) {  // full chunk: one straight-line block
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) DR_MFMA_GROUP(gq)
    } else {
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if (gq < ng) DR_MFMA_GROUP(gq)  // uniform
        }
    }
#undef DR_MFMA_GROUP
#undef DR_PB
}

// mfma_chunk with the A operands in LDS too (second half of a k > 128 user panel, stored like an item tile: row =
// the lane's user, same k permutation): pa / pax = sP + row * LDP + 4 * h and the same with the halves swapped.
template <int NG>
__device__ __forceinline__ void mfma_chunk_pl(f32x16& acc, const float* __restrict__ pa, const float* __restrict__ pax,
                                              const float* __restrict__ pb, const float* __restrict__ pbx, int ng) {
    float4 b[2][2], av[2][2];
#define DR_PB(g_) ((((g_) >> 1) & 1) ? pbx : pb)
#define DR_PA(g_) ((((g_) >> 1) & 1) ? pax : pa)
    b[0][0] = *reinterpret_cast<const float4*>(pb);
    b[0][1] = *reinterpret_cast<const float4*>(pb + 8);
    av[0][0] = *reinterpret_cast<const float4*>(pa);
    av[0][1] = *reinterpret_cast<const float4*>(pa + 8);
#define DR_MFMA_GROUP(gq)                                                                                          \
    {                                                                                                              \
        if ((gq) + 1 < NG) {                                                                                       \
            b[((gq) + 1) & 1][0] = *reinterpret_cast<const float4*>(DR_PB((gq) + 1) + ((gq) + 1) * 16);            \
            b[((gq) + 1) & 1][1] = *reinterpret_cast<const float4*>(DR_PB((gq) + 1) + ((gq) + 1) * 16 + 8);        \
            av[((gq) + 1) & 1][0] = *reinterpret_cast<const float4*>(DR_PA((gq) + 1) + ((gq) + 1) * 16);           \
            av[((gq) + 1) & 1][1] = *reinterpret_cast<const float4*>(DR_PA((gq) + 1) + ((gq) + 1) * 16 + 8);       \
        }                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        const float4 x = b[(gq) & 1][0], y = b[(gq) & 1][1];                                                       \
        const float4 p = av[(gq) & 1][0], q = av[(gq) & 1][1];                                                     \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p.x, x.x, acc, 0, 0, 0);                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p.y, x.y, acc, 0, 0, 0);                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p.z, x.z, acc, 0, 0, 0);                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p.w, x.w, acc, 0, 0, 0);                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.x, y.x, acc, 0, 0, 0);                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.y, y.y, acc, 0, 0, 0);                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.z, y.z, acc, 0, 0, 0);                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.w, y.w, acc, 0, 0, 0);                                        \
    }
    if (ng == NG) {
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) DR_MFMA_GROUP(gq)
    } else {
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if (gq < ng) DR_MFMA_GROUP(gq)  // uniform
        }
    }
#undef DR_MFMA_GROUP
#undef DR_PB
#undef DR_PA
}

template <int KS>
__device__ __forceinline__ void load_areg(float (&areg)[KS], const ScoreArgs& a, int64_t u0, int wu, int lane) {
    const int64_t urow = u0 + wu * 32 + (lane & 31);
    const float* up = a.U + urow * a.k + (lane >> 5);
#pragma unroll
    for (int j = 0; j < KS; ++j) areg[j] = (urow < a.Nu && 2 * j + (lane >> 5) < a.k) ? up[2 * j] : 0.0f;
}

// Up to two k-chunks per tile (k <= 256).  LDS per block is small enough for two blocks per CU (K <= ~15), so
// one block's MFMA stream covers the other's staging and selection stalls.  fp32 MFMA shares the SIMD's vector
// pipe with ordinary VALU work (measured: a barrier-locked two-group ping-pong form ran no faster), so what
// counts is the number of non-MFMA instructions per tile.
#ifndef DR_SCORE_LB128
#define DR_SCORE_LB128 1   // k in (128, 256]: the 128-register user panel leaves no room for two workgroups per CU without
                           // spilling 76-115 registers to memory; one per CU spills to AGPRs only: C5 evaluation 59.2 -> 47.0 ms
#endif
#define SCORE_LB(KS_) ((KS_) >= 128 ? DR_SCORE_LB128 : 2)
template <bool MATRIX, int KS, bool VEC, bool BAND = false>
__global__ __launch_bounds__(256, SCORE_LB(KS)) void score_kernel(ScoreArgs a) {
    using G = Geo<KS, VEC>;
    extern __shared__ __align__(16) float smem[];
    float* sI = smem;  // [TI][LDP], k-permuted (see LDP)
    Sel z;
    z.base = smem + TI * LDP;
    constexpr int NCH = (2 * KS + KC - 1) / KC;      // item chunks per tile (compile-time upper bound)
    constexpr int JPC = KS < KC / 2 ? KS : KC / 2;   // k-steps per chunk

    if (!MATRIX && a.mode_flag && *a.mode_flag == 0) return;  // score_kernel_t1 handles this call
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wu = wave >> 1, wi = wave & 1;
    const int64_t u0 = (int64_t)blockIdx.x * TU;
    const int split = blockIdx.y;
    const int t_begin = split * a.tiles_per_split;
    int t_end = t_begin + a.tiles_per_split;
    if (t_end > a.ntiles) t_end = a.ntiles;
    const int k = a.k;

    float areg[KS];
    load_areg<KS>(areg, a, u0, wu, lane);
    // columns that no chunk store touches must read as zero in the MFMA loop (groups of 16 k)
    for (int e = tid; e < TI * LDP; e += 256) sI[e] = 0.0f;
    if (!MATRIX) sel_init(z, a, u0, t_begin, tid);
    const bool affine = a.uadd || a.iadd || a.cdot != 1.0f;
    float ua[16];
    load_uadd(ua, a, u0, wu, lane);
    const int mycol = wi * 32 + (lane & 31);

    // chunk geometry of this thread: chunk 0 is min(k, KC) wide, chunk 1 (k > KC) the rest
    const int g = tid % G::CPT, r0 = tid / G::CPT;
    const int kc_first = k < KC ? k : KC;
    const bool live0 = g * G::W < kc_first;
    const bool live1 = NCH > 1 && g * G::W < k - KC;
    const int gc0 = live0 ? g : 0, gc1 = live1 ? g : 0;
    float* dst = sI + r0 * LDP + (VEC ? g * 8 + 4 * ((g >> 2) & 1) : kpos(g));
    float* dst_odd = sI + r0 * LDP + (VEC ? g * 8 + 4 * (1 - ((g >> 2) & 1)) : kpos(g));
    const float* pb = sI + mycol * LDP + 4 * (lane >> 5);
    const float* pbx = sI + mycol * LDP + 4 * (1 - (lane >> 5));

    float pre[G::NPRE];
    float ia_next = -0.0f;
    if (t_begin < t_end) {
        prefetch_chunk<KS, VEC, G::NPRE>(pre, a.I, (int64_t)t_begin * TI, a.Ni, k, 0, r0, gc0);
        ia_next = load_iadd(a, (int64_t)t_begin * TI + mycol);
    }
    // every prologue load has landed: inside the loop only the tile prefetch is ever outstanding, so no wait
    // in the loop body is inherited from the prologue
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

    for (int t = t_begin; t < t_end; ++t) {
        const int64_t i0 = (int64_t)t * TI;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float ia = ia_next;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int kc0 = c * KC;
            if (kc0 < k) {  // block-uniform
                const int kcn = (k - kc0 < KC) ? (k - kc0) : KC;
                __syncthreads();  // every wave is done with the previous chunk in sI
                if (c == 0 ? live0 : live1) store_chunk<KS, VEC, G::NPRE>(dst, dst_odd, pre);
                __syncthreads();
                // next chunk (this tile's or the next tile's first) -> registers, in flight during the MFMAs
                if (kc0 + KC < k) {
                    prefetch_chunk<KS, VEC, G::NPRE>(pre, a.I, i0, a.Ni, k, kc0 + KC, r0, gc1);
                } else if (t + 1 < t_end) {
                    prefetch_chunk<KS, VEC, G::NPRE>(pre, a.I, i0 + TI, a.Ni, k, 0, r0, gc0);
                    ia_next = load_iadd(a, i0 + TI + mycol);
                }
                mfma_chunk<JPC / 8>(acc, areg + c * (KC / 2), pb, pbx, (kcn + 15) >> 4);
            }
        }
        sel_store_scores(z, a, acc, ua, ia, affine, wu, wi, lane);
        __syncthreads();
        sel_tile<MATRIX, BAND>(z, a, u0, i0, tid);  // the next write of sS is two barriers away
    }
    if (!MATRIX) {
        __syncthreads();
        sel_finish(z, a, u0, split, tid);
    }
}

// ---- k <= 128 and at most ONE held-out item per user (leave-one-out evaluation, the C4 shape): the selection
// runs on the accumulators in registers.  fp32 MFMA shares the SIMD's vector pipe with ordinary VALU work, so
// what the selection costs is its instruction count: here a tile costs, per lane, 16 compares against the row's
// K-th best, 16 compare-and-count against the row's held-out score and 16 tie tests -- no score tile, no LDS
// round trip, two barriers per tile.  A score that beats the K-th best (rare after the first tiles) is
// written to the LDS score tile with its bit in the row's candidate word, and the row's owner lane folds it
// into the sorted list at the next barrier.  Masked (train) items of a tile are published by the owners as
// 64-bit row words with a block-wide "any" flag, so tiles without masked items (most) skip the patch.
// LDS (floats, after the item tile):
constexpr int T1_SS = 0;                          // [TU][LDS_S] candidate scores (sparse)
constexpr int T1_TAU = T1_SS + TU * LDS_S;        // [TU]
constexpr int T1_CANDW = T1_TAU + TU;             // [TU] u64
constexpr int T1_MASKW = T1_CANDW + 2 * TU;       // [TU] u64
constexpr int T1_TSC = T1_MASKW + 2 * TU;         // [TU] held-out score (+inf: none)
constexpr int T1_TID = T1_TSC + TU;               // [TU] held-out item id (-1: none)
constexpr int T1_TCNT = T1_TID + TU;              // [TU] rank counters
constexpr int T1_MISC = T1_TCNT + TU;             // [0] epoch of tau, [1..2] masked entries of the tile (tile parity)
constexpr int T1_MCAP = 32;                       // masked (row, column) entries listed per tile; more: the row words
constexpr int T1_MLIST = T1_MISC + 4;             // [2][T1_MCAP] row << 8 | column
constexpr int T1_TOPV = T1_MLIST + 2 * T1_MCAP;   // [TU][K] scores, then [TU][K] ids
static_assert((T1_CANDW % 2) == 0 && (T1_MASKW % 2) == 0 && (T1_TOPV % 4) == 0 && (T1_MISC % 4) == 0, "LDS alignment");

#if DR_ABL == 9   // phase trace of score_kernel_t1 (tools/score_trace.py): wave 0 of every block, first DBG_T tiles
constexpr int DBG_T = 96, DBG_STRIDE = 4 + 8 * DBG_T;
__device__ long long dbg_trace[4096 * DBG_STRIDE];
#define DR_TRACE(slot) if (tid == 0 && t - t_begin >= 2000 && t - t_begin < 2000 + DBG_T) dbg_row[4 + 8 * (t - t_begin - 2000) + (slot)] = (long long)__builtin_readcyclecounter()
#else
#define DR_TRACE(slot)
#endif
// (Round 3, measured and removed: the rank-only kernel's pipelined epilogue -- tile t - 1's compares and candidate stores
// behind tile t's MFMA chain on a second accumulator, thresholds re-read from LDS, candidates folded one barrier later,
// row words double-buffered -- is bit-identical here too (69 scoring tests) but SLOWER at C4 shape: 43.4 ms against
// 42.2; as two renamed copies of the loop body 44.2 ms, and that structure without the pipelining 48.6 ms.  The two
// co-resident workgroups of this kernel already run in anti-phase, so the drain it hides is covered by the other
// workgroup's chain, and the loop is sensitive to code size.)
template <int KS, bool VEC, bool AFFINE>
__global__ __launch_bounds__(256, 2) void score_kernel_t1(ScoreArgs a) {
    using G = Geo<KS, VEC>;
    static_assert(KS <= KC / 2, "one k-chunk per tile");
    if (*a.mode_flag != 0) return;  // some user has several held-out items: score_kernel handles this call
#if DR_ABL == 7
    if (blockIdx.y & 1) __builtin_amdgcn_s_sleep(55);   // experiment: co-resident blocks out of phase
#endif
    extern __shared__ __align__(16) float smem[];
    float* sI = smem;  // [TI][LDP], k-permuted (see LDP)
    float* base = smem + TI * LDP;
    float* sS = base + T1_SS;
    float* tau = base + T1_TAU;
    unsigned long long* candw = reinterpret_cast<unsigned long long*>(base + T1_CANDW);
    unsigned long long* maskw = reinterpret_cast<unsigned long long*>(base + T1_MASKW);
    float* tsc = base + T1_TSC;
    int* tidl = reinterpret_cast<int*>(base + T1_TID);
    int* tcnt = reinterpret_cast<int*>(base + T1_TCNT);
    int* misc = reinterpret_cast<int*>(base + T1_MISC);
    int* mlist = reinterpret_cast<int*>(base + T1_MLIST);
    const int K = a.K;
    float* topv = base + T1_TOPV;
    int* topi = reinterpret_cast<int*>(base + T1_TOPV + TU * K);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wu = wave >> 1, wi = wave & 1;
    const int64_t u0 = (int64_t)blockIdx.x * TU;
    const int split = blockIdx.y;
    const int t_begin = split * a.tiles_per_split;
    int t_end = t_begin + a.tiles_per_split;
    if (t_end > a.ntiles) t_end = a.ntiles;
    const int k = a.k;

#if DR_ABL == 9
    long long* dbg_row = dbg_trace + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) % 4096) * DBG_STRIDE;
    if (tid == 0) {
        dbg_row[0] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        dbg_row[1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        dbg_row[2] = blockIdx.x;
        dbg_row[3] = blockIdx.y;
    }
#endif
    float areg[KS];
    load_areg<KS>(areg, a, u0, wu, lane);
    for (int e = tid; e < TI * LDP; e += 256) sI[e] = 0.0f;
    for (int e = tid; e < TU * K; e += 256) {
        topv[e] = -INFINITY;
        topi[e] = 0x7fffffff;
    }
    // row owner = lane th % 4 == 0 of row th / 4: train cursor (two items cached ahead) in registers
    const int orow = tid >> 2;
    const int64_t ou = u0 + orow;
    const bool owner = (tid & 3) == 0 && ou < a.Nu;
    int cur = 0, cur_end = 0, nxt = 0x7fffffff, nxt2 = 0x7fffffff;
    if (tid < TU) {
        const int64_t u = u0 + tid;
        tau[tid] = (u < a.Nu) ? -INFINITY : INFINITY;   // -inf: list not full yet
        candw[tid] = 0ull;
        maskw[tid] = 0ull;
        tcnt[tid] = 0;
        float sc = INFINITY;
        int id = -1;
        if (a.tgt_ptr && u < a.Nu) {
            const int p = a.tgt_ptr[u];
            if (a.tgt_ptr[u + 1] > p && !a.tgt_masked[p]) {
                sc = a.tgt_score[p];
                id = a.tgt_idx[p];
            }
        }
        tsc[tid] = sc;
        tidl[tid] = id;
    }
    if (tid < 4) misc[tid] = 0;
    if (owner && a.train_ptr) {
        cur = a.train_ptr[ou];
        cur_end = a.train_ptr[ou + 1];
        const int first = t_begin * TI;
        int lo = cur, hi = cur_end;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (a.train_idx[mid] < first) lo = mid + 1; else hi = mid;
        }
        cur = lo;
        nxt = cur < cur_end ? a.train_idx[cur] : 0x7fffffff;
        nxt2 = cur + 1 < cur_end ? a.train_idx[cur + 1] : 0x7fffffff;
    }
    float ua[16];
    if (AFFINE) load_uadd(ua, a, u0, wu, lane);
    const int mycol = wi * 32 + (lane & 31);
    const int rbase = wu * 32 + 4 * (lane >> 5);   // row of accumulator element r = rbase + (r&3) + 8*(r>>2)

    const int g = tid % G::CPT, r0 = tid / G::CPT;
    const bool live0 = g * G::W < k;
    const int gc0 = live0 ? g : 0;
    float* dst = sI + r0 * LDP + (VEC ? g * 8 + 4 * ((g >> 2) & 1) : kpos(g));
    float* dst_odd = sI + r0 * LDP + (VEC ? g * 8 + 4 * (1 - ((g >> 2) & 1)) : kpos(g));
    const float* pb = sI + mycol * LDP + 4 * (lane >> 5);
    const float* pbx = sI + mycol * LDP + 4 * (1 - (lane >> 5));

    float pre[G::NPRE];
    float ia_next = -0.0f;
    if (t_begin < t_end) {
        prefetch_chunk<KS, VEC, G::NPRE>(pre, a.I, (int64_t)t_begin * TI, a.Ni, k, 0, r0, gc0);
        ia_next = load_iadd(a, (int64_t)t_begin * TI + mycol);
    }
    __syncthreads();
    float tauR[16], stR[16];
    int cnt[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        tauR[r] = tau[row];
        stR[r] = tsc[row];
        cnt[r] = 0;
    }
    int my_epoch = 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no wait inside the loop is inherited from the prologue

    for (int t = t_begin; t <= t_end; ++t) {   // iteration t_end only folds the last tile's candidates
        const int64_t i0 = (int64_t)t * TI;
#if DR_ABL != 4
        // Explicit lgkmcnt(0): on the loop's back edge the compiler emitted s_barrier with this wave's last candidate
        // store + ds_or still in flight (no s_waitcnt in the latch), and the row's owner -- another wave -- then read the
        // candidate word without the bit: lost top-K entries for the rows written last (found with tools/_dbg, 6-12 of
        // 300 users wrong, different ones every run).
        // Round 3, looked at again in the ISA of all 29 instantiations (hipcc of ROCm 7.2, tools/check_barriers.py): the
        // compiler now puts `s_waitcnt lgkmcnt(0)` in front of EVERY s_barrier of these kernels with or without this
        // intrinsic (the generated code is identical, instruction for instruction, when both intrinsics are deleted; so is
        // it without the vmcnt(0) intrinsic above and without the s_setprio pair) -- the latch that lacked the wait belonged
        // to an earlier shape of this loop and cannot be reproduced from the present source.  The intrinsic stays as the
        // statement of intent; what GUARDS the property is `make check` (every path into every s_barrier of every
        // instantiation must pass an lgkmcnt(0) wait after its last LDS store / atomic) and
        // test_score_kernel_instantiations_repeatable_and_exact.
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __syncthreads();  // A: tile t-1's candidates are visible, every wave is done with the item tile in LDS
#endif
        DR_TRACE(0);
        if (owner) {
            unsigned long long m = candw[orow];
            if (m) {  // fold tile t-1's candidates of this row into its sorted list
                candw[orow] = 0ull;
                const float* srow = sS + orow * LDS_S;
                float* tv = topv + orow * K;
                int* ti = topi + orow * K;
                while (m) {
                    const int c = __builtin_ctzll(m);
                    m &= m - 1;
                    const float sc = srow[c];
                    const int item = (int)(i0 - TI + c);
                    if (beats(sc, item, tv[K - 1], ti[K - 1])) {
                        int p = K - 1;
                        while (p > 0 && beats(sc, item, tv[p - 1], ti[p - 1])) {
                            tv[p] = tv[p - 1];
                            ti[p] = ti[p - 1];
                            --p;
                        }
                        tv[p] = sc;
                        ti[p] = item;
                    }
                }
                if (ti[K - 1] != 0x7fffffff && !(tv[K - 1] == tau[orow])) {
                    tau[orow] = tv[K - 1];
                    atomicAdd(&misc[0], 1);
                }
            }
        }
        if (t == t_end) break;
        if (owner) {  // masked (train) items of tile t for this row
            unsigned long long m = 0ull;
            const int hi = (int)i0 + TI;
            while (nxt < hi) {   // rare: ~ one masked item per 64-row tile
                const int c = nxt - (int)i0;
                m |= 1ull << c;
                const int e = atomicAdd(&misc[1 + (t & 1)], 1);
                if (e < T1_MCAP) mlist[(t & 1) * T1_MCAP + e] = (orow << 8) | c;
                ++cur;
                nxt = nxt2;
                nxt2 = cur + 1 < cur_end ? a.train_idx[cur + 1] : 0x7fffffff;
            }
            maskw[orow] = m;
        }
        DR_TRACE(1);
        if (tid == 0) misc[1 + ((t + 1) & 1)] = 0;
#if DR_ABL != 3
        if (live0) store_chunk<KS, VEC, G::NPRE>(dst, dst_odd, pre);
#endif
        DR_TRACE(2);
        const float ia = ia_next;
#if DR_ABL != 4
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): tile, mask list and tau stores of this wave have landed
        __syncthreads();  // B
#endif
        DR_TRACE(3);
        if (t + 1 < t_end) {
#if DR_ABL != 3
            prefetch_chunk<KS, VEC, G::NPRE>(pre, a.I, i0 + TI, a.Ni, k, 0, r0, gc0);
#endif
            if (AFFINE) ia_next = load_iadd(a, i0 + TI + mycol);
        }
        // everything the selection needs from LDS is requested BEFORE the MFMA chain (it was published before barrier
        // B): the epoch of tau, the number of masked entries of this tile and the first four of them, refreshed taus
        const int4 mi = *reinterpret_cast<const int4*>(misc);
        const int nm = (t & 1) ? mi.z : mi.y;
        int4 me = make_int4(-1, -1, -1, -1);
        if (nm) me = *reinterpret_cast<const int4*>(mlist + (t & 1) * T1_MCAP);
        if (mi.x != my_epoch) {
            my_epoch = mi.x;
#pragma unroll
            for (int r = 0; r < 16; ++r) tauR[r] = tau[rbase + (r & 3) + 8 * (r >> 2)];
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#if DR_ABL == 2
        {
            const int ngr = (k + 15) >> 4;
            for (int gq = 0; gq < KS / 8; ++gq)
                if (gq < ngr) {
                    const float4 x = *reinterpret_cast<const float4*>(pb + gq * 16), y = *reinterpret_cast<const float4*>(pb + gq * 16 + 8);
                    acc[gq & 15] += areg[gq * 8] * x.x + areg[gq * 8 + 1] * x.y + areg[gq * 8 + 4] * y.x + areg[gq * 8 + 7] * y.w;
                }
        }
#else
        // The other block's wave on this SIMD does its selection / staging (plain VALU work) while this wave streams
        // MFMAs through the same pipe: at equal priority every one of its VALU instructions queues behind a 64-cycle
        // MFMA.  The non-MFMA part of a tile runs at raised priority so that it slips in after the MFMA in flight.
#if DR_ABL != 12
        __builtin_amdgcn_s_setprio(0);
#endif
        mfma_chunk<KS / 8>(acc, areg, pb, pbx, (k + 15) >> 4);
#if DR_ABL != 12
        __builtin_amdgcn_s_setprio(3);
#endif
#endif
        DR_TRACE(4);
#if DR_ABL == 8   // experiment: what does one more plain VALU instruction cost? (64 dependent v_add per tile)
        {
            float dmy = acc[0];
#pragma unroll
            for (int z = 0; z < 64; ++z) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dmy));
            if (dmy == 123.456f) cnt[0] += 1;
        }
#endif
        if (AFFINE) {
            const float cd = a.cdot;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sc = cd * acc[r];
                sc = sc + ua[r];
                acc[r] = sc + ia;
            }
        }
        // entries that must be looked at by nobody (masked train items, columns beyond the catalog) become -inf: they
        // beat nothing, count for nothing and tie with nothing, so every tile takes the same straight-line path below
        const bool ragged = (t == a.ntiles - 1) && ((int)a.Ni & (TI - 1)) != 0;
        if (DR_ABL != 5 && (nm != 0 || ragged)) {                      // uniform
            if (DR_ABL != 10 && nm <= T1_MCAP && !ragged) {
                unsigned deadbits = 0u;
                for (int e = 0; e < nm; ++e) {
                    const int ent = e == 0 ? me.x : (e == 1 ? me.y : (e == 2 ? me.z : (e == 3 ? me.w : mlist[(t & 1) * T1_MCAP + e])));
                    const int rr = (ent >> 8) - rbase;                 // this lane's rows: rr in {0..3, 8..11, 16..19, 24..27}
                    if ((ent & 255) == mycol && (unsigned)rr < 32u && !(rr & 4)) deadbits |= 1u << ((rr & 3) | ((rr >> 3) << 2));
                }
                if (deadbits) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((deadbits >> r) & 1u) acc[r] = -INFINITY;
                }
            } else {                                                   // many masked items in one tile / the last tile
                const int nc = ragged ? ((int)a.Ni & (TI - 1)) : TI;
                const bool beyond = mycol >= nc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned long long m = maskw[rbase + (r & 3) + 8 * (r >> 2)];
                    if (beyond || ((m >> mycol) & 1ull)) acc[r] = -INFINITY;
                }
            }
        }
        DR_TRACE(5);
        // wave-uniform "something to do" mask built from compare results in scalar registers (ballots): written as a
        // per-lane bool the compiler assembled it with ~60 extra vector instructions per tile
        unsigned long long any = 0ull;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sc = acc[r];
            cnt[r] += (sc > stR[r]) ? 1 : 0;
#if DR_ABL != 1
            any |= __builtin_amdgcn_ballot_w64(sc > tauR[r]);   // beats the K-th best (tau = -inf: list not full)
            any |= __builtin_amdgcn_ballot_w64(sc == stR[r]);   // tie with the held-out item: decided by item id
#endif
        }
        if (any) {                        // rare after the first tiles: only the lanes / entries concerned do anything
            const int item = (int)i0 + mycol;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sc = acc[r];
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (sc == stR[r]) {
                    const int ct = tidl[row];
                    if (ct >= 0 && item < ct) ++cnt[r];
                }
                if (sc > tauR[r]) {
                    sS[row * LDS_S + mycol] = sc;
                    atomicOr(&candw[row], 1ull << mycol);
                }
            }
        }
        DR_TRACE(6);
    }
    // per-lane rank counts -> rows -> global
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (cnt[r]) atomicAdd(&tcnt[rbase + (r & 3) + 8 * (r >> 2)], cnt[r]);
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
    if (a.tgt_ptr && tid < TU && u0 + tid < a.Nu && tcnt[tid] != 0 && tidl[tid] >= 0)
        atomicAdd(a.tgt_rank + a.tgt_ptr[u0 + tid], tcnt[tid]);
}

// ---- rank only (round 3): k <= 256, at most ONE held-out item per user, no top-K list ---------------------------
// What the reference's evaluators consume is the RANK of the held-out item among the candidates (SeqRecMetrics.py:33,59,
// 75-80: HR@50 / MRR / AUC; RankingMetrics.py:31,52 likewise) -- a count of the scores that beat the held-out item's.
// Without a list to maintain, a tile costs per lane 16 compares whose results are counted by SCALAR popcounts of the
// ballots (32 row counters per wave in scalar registers; fp32 MFMA shares the vector pipe with ordinary vector work,
// the scalar unit is free) and 16 tie tests; there is no score tile, no candidate word, no row owner.  Train items are
// not looked at in the stream at all: every score is counted, and afterwards the panel's train items are scored ONCE MORE
// as gathered "correction" tiles through the same MFMA sequence (so each (user, item) score has the same bits as in the
// stream) and the ones that had beaten the held-out item are taken off the count -- 64 train entries per extra tile,
// ~1 % of the stream at C4 / C5.  The item tile is double-buffered in LDS (2 x 33 KB, two workgroups per CU): ONE
// barrier per k-chunk, and the store of the next chunk is followed by the MFMA chain instead of a barrier.
// LDS after the two item-tile buffers, for a panel of TUK users (floats / ints):
//   [TUK] held-out score (+inf: none) | [TUK] held-out item (-1: none / masked) | [TUK] rank counters |
//   [TUK + 4] train_ptr of the panel's users | [TI] correction tile: train item of column c (-1: none) | [TI] its panel row
// Panel height (template WU = waves of 32 users; two waves of 32 items each): k <= 128 runs 128-user panels (ONE
// workgroup of 8 waves per CU: every staged item tile serves twice the MFMA work; C4 shape 0.655 of peak against 0.634
// with two 64-user workgroups per CU) once there are enough users to fill the chip that way, 128 < k <= 256 stays at 64
// users (the 128-register user panel + 8 waves per CU leaves 256 registers per wave: 40-60 spill to memory, 0.46
// against 0.58).
#ifndef DR_RK_PIPE
#define DR_RK_PIPE 1   // the epilogue of tile t runs behind the MFMA chain of tile t + 1 (second accumulator)
#endif
// (requesting a stage's first B operands right behind the barrier, ahead of the stage's stores and loads: no change, 38.0 ms)
// (stages of TWO item tiles -- one barrier per 128 MFMAs of a wave, four 33 KB tile buffers, both epilogues behind the second
// chain: bit-identical, 36.1 ms against 34.8 at C4 shape; one loop body with a 16-register accumulator copy instead of two
// renamed copies: 34.8 against 34.9 -- neither kept)
#ifndef DR_RK_PIN
#define DR_RK_PIN 1
#endif
#ifndef RK_CNT
#define RK_CNT 1   // 1: per-lane vector counters; 0: row counters in scalar registers (ballot + popcount: the 32 counters do
                   // not fit the scalar file beside the kernel's other state -- 80 v_readlane / v_writelane per two tiles)
#endif
// prefetch_chunk for a gathered tile: row r of the tile is item rows[r] (LDS; < 0: any valid row, never looked at)
template <int KS, bool VEC, int N, int NT>
__device__ __forceinline__ void prefetch_rows(float (&pre)[N], const float* __restrict__ I, const int* rows, int k, int kc0,
                                              int r0, int gc) {
    using G = Geo<KS, VEC, NT>;
#pragma unroll
    for (int q = 0; q < G::PASSES; ++q) {
        int m = rows[r0 + q * G::RPP];
        m = m < 0 ? 0 : m;
        const float* p = I + (int64_t)m * k + kc0 + gc * G::W;
        if (VEC) {
            const float4 v0 = *reinterpret_cast<const float4*>(p);
            const float4 v1 = *reinterpret_cast<const float4*>(p + 4);
            pre[8 * q] = v0.x; pre[8 * q + 1] = v0.y; pre[8 * q + 2] = v0.z; pre[8 * q + 3] = v0.w;
            pre[8 * q + 4] = v1.x; pre[8 * q + 5] = v1.y; pre[8 * q + 6] = v1.z; pre[8 * q + 7] = v1.w;
        } else {
            pre[q] = *p;
        }
    }
}

// PL (k in (128, 256], 128-user panels): k-steps 0..63 of the user panel in registers, 64..127 in LDS beside the item
// tiles (stored like an item tile), so that 8 waves per CU fit their registers without spilling to memory.
template <int KS, bool VEC, bool AFFINE, int WU, bool BAND, bool PL = false>
__global__ __launch_bounds__(128 * WU, (WU == 2 && KS < 128 && !(BAND && KS >= 64)) ? 2 : 1) void score_rank_kernel(ScoreArgs a) {
    static_assert(!PL || (KS == 128 && WU == 4 && !BAND), "panel-in-LDS form: k > 128, 128-user panels");
    constexpr int NT = 128 * WU, TUK = 32 * WU;     // WU waves of 32 users x 2 waves of 32 items
    constexpr int KR = PL ? KS / 2 : KS;            // k-steps of the user panel held in registers
    // PIPE: the epilogue of tile t (compare / count) runs after the MFMA chain of tile t + 1 has been issued, on a second
    // accumulator, with the rows' thresholds (and affine terms) read from LDS: its vector instructions and LDS reads
    // overlap that chain instead of waiting for the pipeline to drain
    // (not where the second accumulator would spill to memory inside the stream: the BAND forms, the panel-in-LDS form and
    // 64-user panels with k > 64 keep the epilogue behind its own chain)
    constexpr bool PIPE = DR_RK_PIPE && !BAND && !PL && (WU == 4 || KS == 32);
    using G = Geo<KS, VEC, NT>;
    constexpr int NCH = (2 * KS + KC - 1) / KC;      // k-chunks per tile: 1 (k <= 128) or 2 (this instantiation: k > 128)
    constexpr int JPC = KS < KC / 2 ? KS : KC / 2;
    constexpr int BUF = TI * LDP;
    if (*a.mode_flag != 0) return;  // some user has several held-out items: score_kernel handles this call
    extern __shared__ __align__(16) float smem[];
    float* sI = smem;               // [2][TI][LDP], k-permuted (see LDP); with two chunks buffer c always holds chunk c
    float* sP = smem + 2 * BUF;     // PL: [TUK][LDP], k = KC.. of the panel's users, same permutation
    float* base = smem + 2 * BUF + (PL ? TUK * LDP : 0);
    float* tsc = base;
    int* tidl = reinterpret_cast<int*>(base + TUK);
    int* tcnt = reinterpret_cast<int*>(base + 2 * TUK);
    int* tptr = reinterpret_cast<int*>(base + 3 * TUK);
    int* mitem = reinterpret_cast<int*>(base + 4 * TUK + 4);
    int* mrow = mitem + TI;
    float* tlo = reinterpret_cast<float*>(mrow + TI);      // BAND: lower end of the held-out item's band
    int* tslot = reinterpret_cast<int*>(tlo + TUK);        // BAND: its slot in the target CSR
    int* wmax = tslot + TUK;                               // BAND: widest band of the panel (float bits)
    float* tua = reinterpret_cast<float*>(wmax + 4);       // PIPE: the rows' affine terms
    if (threadIdx.x == 0) *wmax = 0;
    __builtin_amdgcn_s_waitcnt(0xC07F);   // no LDS store / atomic of this wave in flight at a barrier (make check)
    __syncthreads();

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wu = wave >> 1, wi = wave & 1;
    const int64_t u0 = (int64_t)blockIdx.x * TUK;
    const int t_begin = blockIdx.y * a.tiles_per_split;
    int t_end = t_begin + a.tiles_per_split;
    if (t_end > a.ntiles) t_end = a.ntiles;
    const int k = a.k;

    float areg[KR];
    load_areg<KR>(areg, a, u0, wu, lane);
    for (int e = tid; e < 2 * BUF; e += NT) sI[e] = 0.0f;   // columns that no chunk store touches read as zero
    if (PL) {
        for (int e = tid; e < TUK * KC; e += NT) {
            const int row = e / KC, kk = e % KC;
            const int64_t u = u0 + row;
            sP[row * LDP + kpos(kk)] = (u < a.Nu && KC + kk < k) ? a.U[u * k + KC + kk] : 0.0f;
        }
    }
    const float* pa = sP + (wu * 32 + (lane & 31)) * LDP + 4 * (lane >> 5);
    const float* pax = sP + (wu * 32 + (lane & 31)) * LDP + 4 * (1 - (lane >> 5));
    if (tid < TUK) {
        const int64_t u = u0 + tid;
        tcnt[tid] = 0;
        float sc = INFINITY, lo = INFINITY;
        int id = -1, slot = 0;
        if (a.tgt_ptr && u < a.Nu) {
            const int p = a.tgt_ptr[u];
            if (a.tgt_ptr[u + 1] > p && !a.tgt_masked[p]) {
                // BAND (dr_score_rank_exact_f32): a score above tgt_hi counts, one inside [tgt_score, tgt_hi] goes to the
                // exact re-score; otherwise tgt_score is the held-out item's own score
                sc = BAND ? a.tgt_hi[p] : a.tgt_score[p];
                lo = a.tgt_score[p];
                id = a.tgt_idx[p];
                slot = p;
            }
        }
        tsc[tid] = sc;
        tidl[tid] = id;
        tlo[tid] = lo;
        tslot[tid] = slot;
        if (BAND && id >= 0) atomicMax(wmax, __float_as_int(sc - lo));   // widths are >= 0: their bit patterns order like ints
    }
    if (tid <= TUK) tptr[tid] = a.train_ptr ? a.train_ptr[(u0 + tid < a.Nu) ? u0 + tid : a.Nu] : 0;
    float ua[16];
    if (AFFINE && !PIPE) load_uadd(ua, a, u0, wu, lane);
    if (AFFINE && PIPE && tid < TUK) tua[tid] = (a.uadd && u0 + tid < a.Nu) ? a.uadd[u0 + tid] : -0.0f;
    const int mycol = wi * 32 + (lane & 31);
    const int rbase = wu * 32 + 4 * (lane >> 5);   // row of accumulator element r = rbase + (r&3) + 8*(r>>2)

    const int g = tid % G::CPT, r0 = tid / G::CPT;
    const int kc_first = k < KC ? k : KC;
    const bool live0 = g * G::W < kc_first;
    const bool live1 = NCH > 1 && g * G::W < k - KC;
    const int gc0 = live0 ? g : 0, gc1 = live1 ? g : 0;
    float* dst = sI + r0 * LDP + (VEC ? g * 8 + 4 * ((g >> 2) & 1) : kpos(g));
    float* dst_odd = sI + r0 * LDP + (VEC ? g * 8 + 4 * (1 - ((g >> 2) & 1)) : kpos(g));
    const float* pb = sI + mycol * LDP + 4 * (lane >> 5);
    const float* pbx = sI + mycol * LDP + 4 * (1 - (lane >> 5));
    const int ng0 = (kc_first + 15) >> 4, ng1 = NCH > 1 ? ((k - KC + 15) >> 4) : 0;

    __builtin_amdgcn_s_waitcnt(0xC07F);   // no LDS store / atomic of this wave in flight at a barrier (make check)
    __syncthreads();
    float stR[16];
    if (!PIPE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stR[r] = tsc[rbase + (r & 3) + 8 * (r >> 2)];
    }
    // BAND: the stream only asks "within the panel's WIDEST band below hi?" (one scalar, no second threshold per row in
    // registers; padded so that rounding of the sum cannot lose a score on a band's lower end); the row's own lower end is
    // read from LDS on the rare path
    const float wpad = BAND ? __int_as_float(*wmax) * 1.000002f + 1e-37f : 0.0f;
    auto band_append = [&](int row, int item_word) {
        const int p = atomicAdd(a.band_count, 1);
        if (p < a.band_cap) {
            a.band_list[3 * (int64_t)p] = (int32_t)(u0 + row);
            a.band_list[3 * (int64_t)p + 1] = tslot[row];
            a.band_list[3 * (int64_t)p + 2] = item_word;
        }
    };
    int cA[16], cB[16];                            // wave-uniform: rows wu*32 + (r&3) + 8*(r>>2) and those + 4
#pragma unroll
    for (int r = 0; r < 16; ++r) cA[r] = cB[r] = 0;

    // stage s = (tile t_begin + s / NCH, chunk s % NCH) lives in buffer s & 1; `pre` holds stage s + 1 while s computes
    const int nst = (t_end - t_begin) * NCH;
    float pre[G::NPRE];
    float ia_next = -0.0f;
    if (nst > 0) {
        prefetch_chunk<KS, VEC, G::NPRE, NT>(pre, a.I, (int64_t)t_begin * TI, a.Ni, k, 0, r0, gc0);
        if (AFFINE) ia_next = load_iadd(a, (int64_t)t_begin * TI + mycol);
        if (live0) store_chunk<KS, VEC, G::NPRE, NT>(dst, dst_odd, pre);
        if (nst > 1) {
            if (NCH > 1) prefetch_chunk<KS, VEC, G::NPRE, NT>(pre, a.I, (int64_t)t_begin * TI, a.Ni, k, KC, r0, gc1);
            else prefetch_chunk<KS, VEC, G::NPRE, NT>(pre, a.I, (int64_t)(t_begin + 1) * TI, a.Ni, k, 0, r0, gc0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __syncthreads();

    // The epilogue of tile t (compare / count) runs after the MFMA chain of tile t + 1 has been issued, on a second
    // accumulator: its vector instructions overlap that chain instead of waiting for the pipeline to drain (DR_RK_PIPE 0:
    // the epilogue follows its own chain).
    auto epilogue = [&](f32x16& acc, int t, float ia) {
        const int64_t i0 = (int64_t)t * TI;
        float st[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (PIPE) {
                const float4 v = *reinterpret_cast<const float4*>(tsc + rbase + 8 * q);
                st[4 * q] = v.x; st[4 * q + 1] = v.y; st[4 * q + 2] = v.z; st[4 * q + 3] = v.w;
            } else {
                st[4 * q] = stR[4 * q]; st[4 * q + 1] = stR[4 * q + 1]; st[4 * q + 2] = stR[4 * q + 2]; st[4 * q + 3] = stR[4 * q + 3];
            }
        }
        if (AFFINE) {
            const float cd = a.cdot;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 w;
                if (PIPE) w = *reinterpret_cast<const float4*>(tua + rbase + 8 * q);
                else w = make_float4(ua[4 * q], ua[4 * q + 1], ua[4 * q + 2], ua[4 * q + 3]);
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float sc = cd * acc[4 * q + e];
                    sc = sc + wv[e];
                    acc[4 * q + e] = sc + ia;
                }
            }
        }
        if (t == a.ntiles - 1 && ((int)a.Ni & (TI - 1)) != 0 && mycol >= ((int)a.Ni & (TI - 1))) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = -INFINITY;   // columns beyond the catalog beat nothing
        }
        unsigned long long eq = 0ull;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#if RK_CNT == 0
            const unsigned long long m = __builtin_amdgcn_ballot_w64(acc[r] > st[r]);
            cA[r] += __builtin_popcount((unsigned)m);
            cB[r] += __builtin_popcount((unsigned)(m >> 32));
#else
            cA[r] += (acc[r] > st[r]) ? 1 : 0;
#endif
            if (BAND) eq |= __builtin_amdgcn_ballot_w64(acc[r] <= st[r] && acc[r] + wpad >= st[r]);
            else eq |= __builtin_amdgcn_ballot_w64(acc[r] == st[r]);
        }
        if (eq) {   // a tie with the held-out item (or the item itself): decided by item id; BAND: inside the band
            const int item = (int)i0 + mycol;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (BAND) {
                    const bool near = acc[r] <= st[r] && acc[r] + wpad >= st[r];
                    if (__builtin_amdgcn_ballot_w64(near) == 0ull) continue;      // uniform: nothing for this r
                    if (near && acc[r] >= tlo[row]) {
                        const int ct = tidl[row];
                        if (ct >= 0 && item != ct) band_append(row, item);
                    }
                } else if (acc[r] == st[r]) {
                    const int ct = tidl[row];
                    if (ct >= 0 && item < ct) atomicAdd(&tcnt[row], 1);
                }
            }
        }
    };
    int s = 0;
    float ia_prev = -0.0f;
    // one tile: its chunks' stores / prefetches / MFMA chains into `acc`; `accp` = the previous tile's scores
    auto tile = [&](int t, f32x16& acc, f32x16& accp, bool havep) {
        const int64_t i0 = (int64_t)t * TI;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float ia = ia_next;
#pragma unroll
        for (int c = 0; c < NCH; ++c, ++s) {
            const int cur = (NCH > 1) ? c : (s & 1);
            const int oth = cur ^ 1;
            // stage s + 1 (in registers since the previous stage's MFMA chain) -> the other buffer: its last readers
            // passed the barrier that ended stage s - 1
            if (DR_RK_PIN) pin_prefetched(pre);
            if (s + 1 < nst) {
                const bool lv = (NCH > 1 && c == 0) ? live1 : live0;
                if (lv) store_chunk<KS, VEC, G::NPRE, NT>(dst + oth * BUF, dst_odd + oth * BUF, pre);
            }
            if (s + 2 < nst) {   // stage s + 2 -> registers, in flight during this stage's MFMAs
                if (NCH > 1) prefetch_chunk<KS, VEC, G::NPRE, NT>(pre, a.I, i0 + TI, a.Ni, k, c * KC, r0, c == 0 ? gc0 : gc1);
                else prefetch_chunk<KS, VEC, G::NPRE, NT>(pre, a.I, i0 + 2 * TI, a.Ni, k, 0, r0, gc0);
            }
            if (AFFINE && c == NCH - 1 && t + 1 < t_end) ia_next = load_iadd(a, i0 + TI + mycol);
            if (PL && c == 1) mfma_chunk_pl<JPC / 8>(acc, pa, pax, pb + cur * BUF, pbx + cur * BUF, ng1);
            else mfma_chunk<JPC / 8>(acc, areg + (PL ? 0 : c * (KC / 2)), pb + cur * BUF, pbx + cur * BUF, c == 0 ? ng0 : ng1);
            if (PIPE) {
                if (c == 0 && havep) epilogue(accp, t - 1, ia_prev);
            } else if (c == NCH - 1) epilogue(acc, t, ia);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __syncthreads();
        }
        ia_prev = ia;
    };
    if (!PIPE) {
        f32x16 acc;
        for (int t = t_begin; t < t_end; ++t) tile(t, acc, acc, false);
    } else {
        f32x16 acc0, acc1;
        int t = t_begin;
        for (; t + 1 < t_end; t += 2) {
            tile(t, acc0, acc1, t > t_begin);
            tile(t + 1, acc1, acc0, true);
        }
        if (t < t_end) {
            tile(t, acc0, acc1, t > t_begin);
            if (PIPE) epilogue(acc0, t, ia_prev);
        } else if (t_end > t_begin) {
            if (PIPE) epilogue(acc1, t_end - 1, ia_prev);
        }
    }
    if (PIPE) {   // for the correction tiles below
#pragma unroll
        for (int r = 0; r < 16; ++r) stR[r] = tsc[rbase + (r & 3) + 8 * (r >> 2)];
        if (AFFINE) load_uadd(ua, a, u0, wu, lane);
    }
#if RK_CNT == 0
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wu * 32 + (r & 3) + 8 * (r >> 2);
            if (cA[r]) atomicAdd(&tcnt[row], cA[r]);
            if (cB[r]) atomicAdd(&tcnt[row + 4], cB[r]);
        }
    }
#else
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (cA[r]) atomicAdd(&tcnt[rbase + (r & 3) + 8 * (r >> 2)], cA[r]);
#endif
    // ---- correction tiles: the panel's train items inside this split's item range, 64 entries per tile
    if (a.train_ptr) {
        const int E0 = tptr[0], E1 = tptr[TUK];
        const int lo_item = t_begin * TI;
        const int64_t hi64 = (int64_t)t_end * TI < a.Ni ? (int64_t)t_end * TI : a.Ni;
        const int hi_item = (int)hi64;
        for (int e0 = E0; e0 < E1; e0 += TI) {
            __builtin_amdgcn_s_waitcnt(0xC07F);   // no LDS store / atomic of this wave in flight at a barrier (make check)
            __syncthreads();                   // the previous tile's readers of mitem / mrow / sI are done
            int item = -1;
            if (tid < TI) {
                const int e = e0 + tid;
                int row = 0;
                if (e < E1) {
                    const int m = a.train_idx[e];
                    if (m >= lo_item && m < hi_item) {
                        item = m;
                        int lo = 0, hi = TUK;  // the row whose train range holds entry e: last x with tptr[x] <= e
                        while (lo < hi) {
                            const int mid = (lo + hi + 1) >> 1;
                            if (tptr[mid] <= e) lo = mid; else hi = mid - 1;
                        }
                        row = lo;
                        if (tidl[row] < 0) item = -1;   // no held-out item in that row: nothing to correct
                        if (e > tptr[row] && a.train_idx[e - 1] == m) item = -1;   // a repeated entry masks only once
                    }
                }
                mitem[tid] = item;
                mrow[tid] = row;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (!__syncthreads_or(item >= 0)) continue;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            float prc[G::NPRE];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                prefetch_rows<KS, VEC, G::NPRE, NT>(prc, a.I, mitem, k, c * KC, r0, c == 0 ? gc0 : gc1);
                if (c == 0 ? live0 : live1) store_chunk<KS, VEC, G::NPRE, NT>(dst + c * BUF, dst_odd + c * BUF, prc);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (PL && c == 1) mfma_chunk_pl<JPC / 8>(acc, pa, pax, pb + c * BUF, pbx + c * BUF, ng1);
                else mfma_chunk<JPC / 8>(acc, areg + (PL ? 0 : c * (KC / 2)), pb + c * BUF, pbx + c * BUF, c == 0 ? ng0 : ng1);
            }
            const int it = mitem[mycol], rw = mrow[mycol];
            if (it >= 0) {
                const float ia2 = AFFINE ? load_iadd(a, it) : -0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row == rw) {
                        float sc = acc[r];
                        if (AFFINE) {
                            sc = a.cdot * sc;
                            sc = sc + ua[r];
                            sc = sc + ia2;
                        }
                        const int ct = tidl[row];
                        if (BAND) {
                            // counted in the stream -> taken off; listed in the stream -> a cancelling entry (bit 31)
                            if (ct >= 0 && sc > stR[r]) atomicSub(&tcnt[row], 1);
                            else if (ct >= 0 && sc >= tlo[row] && sc <= stR[r] && it != ct) band_append(row, it | (int)0x80000000);
                        } else if (ct >= 0 && beats(sc, it, stR[r], ct)) atomicSub(&tcnt[row], 1);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // no LDS store / atomic of this wave in flight at a barrier (make check)
    __syncthreads();
    if (a.tgt_ptr && tid < TUK && u0 + tid < a.Nu && tcnt[tid] != 0 && tidl[tid] >= 0)
        atomicAdd(a.tgt_rank + a.tgt_ptr[u0 + tid], tcnt[tid]);
}

// 1 in *flag when some user has more than one held-out item (flag is zeroed before the launch)
__global__ __launch_bounds__(256) void target_mode_kernel(const int32_t* __restrict__ tgt_ptr, int64_t Nu,
                                                          int32_t* __restrict__ flag) {
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < Nu; u += (int64_t)gridDim.x * 256)
        if (tgt_ptr[u + 1] - tgt_ptr[u] > 1) *flag = 1;
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
    int64_t uadd, iadd, ucat, icat, tgt_score, tgt_masked, part_v, part_i, mode, total;
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
    L->mode = take(256);
    L->total = off;
}

static size_t sel_area_floats(int K) { return (size_t)TOPV_OFF + (size_t)2 * TU * K; }
static size_t score_smem_bytes(int K, bool anyk) {
    return ((anyk ? (size_t)2 * TI * LDK : (size_t)TI * LDP) + sel_area_floats(K)) * 4;
}

template <bool MATRIX, int KS, bool VEC, bool BAND = false>
static int launch_score_regs(const ScoreArgs& a, dim3 grid, hipStream_t st) {
    const size_t smem = score_smem_bytes(a.K, false);
    DR_HIP(hipFuncSetAttribute((const void*)score_kernel<MATRIX, KS, VEC, BAND>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DR_LAUNCH((score_kernel<MATRIX, KS, VEC, BAND>), grid, dim3(256), smem, st, a);
    DR_CHECK_LAUNCH("score_kernel");
    return DR_OK;
}

static size_t t1_smem_bytes(int K) { return ((size_t)TI * LDP + T1_TOPV + (size_t)2 * TU * K) * 4; }

template <int KS, bool VEC, bool AFFINE>
static int launch_score_t1_ka(const ScoreArgs& a, dim3 grid, hipStream_t st) {
    const size_t smem = t1_smem_bytes(a.K);
    DR_HIP(hipFuncSetAttribute((const void*)score_kernel_t1<KS, VEC, AFFINE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
    DR_LAUNCH((score_kernel_t1<KS, VEC, AFFINE>), grid, dim3(256), smem, st, a);
    DR_CHECK_LAUNCH("score_kernel_t1");
    return DR_OK;
}
template <int KS, bool VEC>
static int launch_score_t1_k(const ScoreArgs& a, dim3 grid, hipStream_t st) {
    // plain dot scores (the common evaluation) skip the affine epilogue and its 16 per-row registers altogether
    const bool affine = DR_ABL == 11 || a.uadd || a.iadd || a.cdot != 1.0f;
    return affine ? launch_score_t1_ka<KS, VEC, true>(a, grid, st) : launch_score_t1_ka<KS, VEC, false>(a, grid, st);
}
static int launch_score_t1(const ScoreArgs& a, dim3 grid, hipStream_t st) {
    const bool vec = (a.k & 7) == 0;  // rows fetched as pairs of float4
    if (a.k <= 64) return vec ? launch_score_t1_k<32, true>(a, grid, st) : launch_score_t1_k<32, false>(a, grid, st);
    return vec ? launch_score_t1_k<64, true>(a, grid, st) : launch_score_t1_k<64, false>(a, grid, st);
}

// grid.y = item splits (as for the other kernels); grid.x is computed here from the kernel's own panel height
template <int KS, bool VEC, bool AFFINE, int WU, bool BAND, bool PL = false>
static int launch_score_rank_kaw(const ScoreArgs& a, unsigned splits, hipStream_t st) {
    const size_t smem = ((size_t)2 * TI * LDP + (PL ? (size_t)32 * WU * LDP : 0) + 7 * 32 * WU + 12 + 2 * TI) * 4;
    const int64_t panels = (a.Nu + 32 * WU - 1) / (32 * WU);
    DR_HIP(hipFuncSetAttribute((const void*)score_rank_kernel<KS, VEC, AFFINE, WU, BAND, PL>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DR_LAUNCH_NAMED("score_rank_kernel", (score_rank_kernel<KS, VEC, AFFINE, WU, BAND, PL>), dim3((unsigned)panels, splits),
                    dim3(128 * WU), smem, st, a);
    DR_CHECK_LAUNCH("score_rank_kernel");
    return DR_OK;
}
template <int KS, bool VEC, int WU, bool BAND, bool PL = false>
static int launch_score_rank_k(const ScoreArgs& a, unsigned splits, hipStream_t st) {
    const bool affine = a.uadd || a.iadd || a.cdot != 1.0f;
    return affine ? launch_score_rank_kaw<KS, VEC, true, WU, BAND, PL>(a, splits, st)
                  : launch_score_rank_kaw<KS, VEC, false, WU, BAND, PL>(a, splits, st);
}
template <bool BAND>
static int launch_score_rank(const ScoreArgs& a, unsigned splits, hipStream_t st) {   // k <= 256
    const bool vec = (a.k & 7) == 0;
    const bool tall = !BAND && vec && a.Nu >= 128 * 128;      // 128-user panels: >= 128 workgroups before the item splits
    if (a.k <= 64) {
        if (tall) return launch_score_rank_k<32, true, 4, false>(a, splits, st);
        return vec ? launch_score_rank_k<32, true, 2, BAND>(a, splits, st) : launch_score_rank_k<32, false, 2, BAND>(a, splits, st);
    }
    if (a.k <= 128) {
        if (tall) return launch_score_rank_k<64, true, 4, false>(a, splits, st);
        return vec ? launch_score_rank_k<64, true, 2, BAND>(a, splits, st) : launch_score_rank_k<64, false, 2, BAND>(a, splits, st);
    }
    if (tall) return launch_score_rank_k<128, true, 4, false, true>(a, splits, st);   // second half of the panel in LDS
    return vec ? launch_score_rank_k<128, true, 2, BAND>(a, splits, st) : launch_score_rank_k<128, false, 2, BAND>(a, splits, st);
}

// grid = (64-user panels, item splits)
template <bool MATRIX>
static int launch_score(const ScoreArgs& a, dim3 grid, hipStream_t st) {
    const bool vec = (a.k & 7) == 0;  // rows fetched as pairs of float4
    if (a.k <= 64) return vec ? launch_score_regs<MATRIX, 32, true>(a, grid, st) : launch_score_regs<MATRIX, 32, false>(a, grid, st);
    if (a.k <= 128) return vec ? launch_score_regs<MATRIX, 64, true>(a, grid, st) : launch_score_regs<MATRIX, 64, false>(a, grid, st);
    if (a.k <= 256) return vec ? launch_score_regs<MATRIX, 128, true>(a, grid, st) : launch_score_regs<MATRIX, 128, false>(a, grid, st);
    const size_t smem = score_smem_bytes(a.K, true);
    DR_HIP(hipFuncSetAttribute((const void*)score_kernel_anyk<MATRIX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DR_LAUNCH((score_kernel_anyk<MATRIX>), grid, dim3(256), smem, st, a);
    DR_CHECK_LAUNCH("score_kernel_anyk");
    return DR_OK;
}

// the BAND form of launch_score<false> (dr_score_rank_exact_f32)
static int launch_score_band(const ScoreArgs& a, dim3 grid, hipStream_t st) {
    const bool vec = (a.k & 7) == 0;
    if (a.k <= 64) return vec ? launch_score_regs<false, 32, true, true>(a, grid, st) : launch_score_regs<false, 32, false, true>(a, grid, st);
    if (a.k <= 128) return vec ? launch_score_regs<false, 64, true, true>(a, grid, st) : launch_score_regs<false, 64, false, true>(a, grid, st);
    if (a.k <= 256) return vec ? launch_score_regs<false, 128, true, true>(a, grid, st) : launch_score_regs<false, 128, false, true>(a, grid, st);
    const size_t smem = score_smem_bytes(a.K, true);
    DR_HIP(hipFuncSetAttribute((const void*)score_kernel_anyk<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DR_LAUNCH((score_kernel_anyk<false, true>), grid, dim3(256), smem, st, a);
    DR_CHECK_LAUNCH("score_kernel_anyk");
    return DR_OK;
}

// Prepare (U', I', uadd, iadd, cdot) for a score kind.  Returns DR_OK / error.
static int prepare_operands(const float* U, const float* U2, int64_t Nu, const float* I, const float* I2,
                            const float* item_bias, int64_t Ni, int k, int kind, float w1, float w2, char* base,
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

// ---------------------------------------------------------------------------------------------------------------
// dr_score_rank_exact_f32: ranks in the REFERENCE's float32 operation order (numpy), SURVEY H2.
// The reference evaluators score in numpy: direct squared differences, np.sum over the last axis
// (SeqRecMetrics.py:122-123) -- numpy's pairwise summation, restated below -- while the MFMA pass computes the
// expanded form as fmaf chains.  The two differ in the last bits, so near-ties could order differently.  Here the
// MFMA pass only decides what is certain: a candidate whose MFMA score lies outside [s_t - delta, s_t + delta]
// (s_t = the held-out item's reference-order score, delta = a bound on |MFMA score - reference-order score|) is
// counted or not; the few inside are re-scored in the reference's order and compared exactly.
// ---------------------------------------------------------------------------------------------------------------

// numpy's float32 pairwise summation (numpy/_core/src/umath/loops_utils.h.src, *_pairwise_sum): n < 8 sequential from
// 0; n <= 128: eight strided accumulators, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n % 8 rest; above: split at
// n/2 rounded down to a multiple of 8.  term(c) = the c-th float32 summand.  Checked against numpy in
// tests/test_golden_cpu.py through the same restatement in oracle/ref_ops.py.
template <class Term>
__device__ __forceinline__ float np_block_sum(const Term& term, int lo, int n) {
    if (n < 8) {
        float res = 0.0f;
        for (int c = 0; c < n; ++c) res = res + term(lo + c);
        return res;
    }
    float r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = term(lo + q);
    int c = 8;
    for (; c < n - (n % 8); c += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = r[q] + term(lo + c + q);
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; c < n; ++c) res = res + term(lo + c);
    return res;
}
template <class Term>
__device__ __forceinline__ float np_pairwise_sum(const Term& term, int n) {
    if (n <= 128) return np_block_sum(term, 0, n);
    // post-order walk of numpy's recursion with an explicit stack (n <= 128 * 2^7)
    int lo_s[8], n_s[8], stage[8];
    float left[8];
    int sp = 0;
    lo_s[0] = 0; n_s[0] = n; stage[0] = 0; left[0] = 0.0f;
    float ret = 0.0f;
    while (sp >= 0) {
        if (stage[sp] == 0) {
            if (n_s[sp] <= 128) {
                ret = np_block_sum(term, lo_s[sp], n_s[sp]);
                --sp;
            } else {
                int n2 = n_s[sp] / 2;
                n2 -= n2 % 8;
                stage[sp] = 1;
                lo_s[sp + 1] = lo_s[sp]; n_s[sp + 1] = n2; stage[sp + 1] = 0;
                ++sp;
            }
        } else if (stage[sp] == 1) {
            left[sp] = ret;
            int n2 = n_s[sp] / 2;
            n2 -= n2 % 8;
            stage[sp] = 2;
            lo_s[sp + 1] = lo_s[sp] + n2; n_s[sp + 1] = n_s[sp] - n2; stage[sp + 1] = 0;
            ++sp;
        } else {
            ret = left[sp] + ret;
            --sp;
        }
    }
    return ret;
}

struct ExactArgs {
    const float* U; const float* U2; const float* I; const float* I2; const float* bias;
    int64_t Nu, Ni;
    int k, kind;
    float w1, w2;
};

// the reference's float32 arithmetic for one (user row, item) pair:
//   DR_SCORE_DOT         (p * q).sum()                         [+ bias]   (the evaluators' predict stand-in)
//   DR_SCORE_NEG_SQDIST  -((p - q) ** 2).sum()
//   DR_SCORE_TWO_DIST    (-w1) * ((p - X) ** 2).sum() - w2 * ((s - Q) ** 2).sum()   (SeqRecMetrics.py:122-123)
__device__ __forceinline__ float ref_order_score(const ExactArgs& e, int64_t u, int64_t it) {
    const float* pu = e.U + u * e.k;
    const float* pi = e.I + it * e.k;
    float s;
    if (e.kind == DR_SCORE_DOT) {
        s = np_pairwise_sum([&](int c) { return pu[c] * pi[c]; }, e.k);
    } else {
        const float d1 = np_pairwise_sum([&](int c) { const float d = pu[c] - pi[c]; return d * d; }, e.k);
        if (e.kind == DR_SCORE_NEG_SQDIST) {
            s = -d1;
        } else {
            const float* su = e.U2 + u * e.k;
            const float* qi = e.I2 + it * e.k;
            const float d2 = np_pairwise_sum([&](int c) { const float d = su[c] - qi[c]; return d * d; }, e.k);
            const float t1 = (-e.w1) * d1;
            const float t2 = e.w2 * d2;
            s = t1 - t2;
        }
    }
    if (e.bias) s = s + e.bias[it];
    return s;
}

// max |x[i]| over i < n into out[0] (one block; n is a table's row count)
__global__ __launch_bounds__(1024) void max_abs_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float red[1024];
    float m = 0.0f;
    if (x)
        for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(x[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// target prologue of the exact path: reference-order score of every held-out item, its band, masked flag, counters
// un[u] = -|u'|^2 (neg_sqnorm_kernel; weighted for the two-distance score), in_max[0] = max_i |i'|^2,
// bmax[0] = max |bias|.  d_eff = length of the MFMA dot product (2k for the two-distance score).
// delta bounds |MFMA score - reference-order score| for every item: both are fp32 evaluations of the same real
// number whose summands are bounded by (|u'|^2 + |i'|^2) in absolute value; (2 D + 2 d + 16) roundings at most.
__global__ __launch_bounds__(256) void exact_prologue_kernel(ExactArgs e, int d_eff, const float* __restrict__ un,
                                                             const float* __restrict__ in_max,
                                                             const float* __restrict__ bmax,
                                                             const int32_t* __restrict__ train_ptr,
                                                             const int32_t* __restrict__ train_idx,
                                                             const int32_t* __restrict__ tgt_ptr,
                                                             const int32_t* __restrict__ tgt_idx,
                                                             float* __restrict__ tgt_ref, float* __restrict__ tgt_lo,
                                                             float* __restrict__ tgt_hi,
                                                             int32_t* __restrict__ tgt_masked,
                                                             int32_t* __restrict__ tgt_rank,
                                                             int32_t* __restrict__ band_count) {
    if (blockIdx.x == 0 && threadIdx.x == 0) band_count[0] = 0;
    const float c1 = 4.0f * (float)(d_eff + 8) * 5.9604645e-08f * 1.01f;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < e.Nu; u += (int64_t)gridDim.x * 256) {
        const int tb = tgt_ptr[u], te = tgt_ptr[u + 1];
        const float delta = c1 * ((-un[u]) + in_max[0]) + 8.0f * 5.9604645e-08f * bmax[0];
        for (int x = tb; x < te; ++x) {
            const int64_t it = tgt_idx[x];
            bool masked = it < 0 || it >= e.Ni;
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
                s = ref_order_score(e, u, it);
                if (s != s) masked = true;
            }
            tgt_ref[x] = s;
            tgt_lo[x] = s - delta;
            tgt_hi[x] = s + delta;
            tgt_masked[x] = masked ? 1 : 0;
            tgt_rank[x] = 0;
        }
    }
}

// exact re-score of the band entries: one thread per (user row, target slot, item)
__global__ __launch_bounds__(256) void exact_refine_kernel(ExactArgs e, const int32_t* __restrict__ band_list,
                                                           const int32_t* __restrict__ band_count, int band_cap,
                                                           const int32_t* __restrict__ tgt_idx,
                                                           const float* __restrict__ tgt_ref,
                                                           int32_t* __restrict__ tgt_rank) {
    const int n = band_count[0];
    if (n > band_cap) return;  // overflowed: exact_fallback_kernel recounts everything
    for (int64_t y = (int64_t)blockIdx.x * 256 + threadIdx.x; y < n; y += (int64_t)gridDim.x * 256) {
        const int64_t u = band_list[3 * y];
        const int x = band_list[3 * y + 1];
        const int word = band_list[3 * y + 2];
        const int item = word & 0x7fffffff;      // bit 31: a train item that the rank-only kernel listed in its stream --
        const float s = ref_order_score(e, u, item);                      // the entry cancels the listing
        if (s == s && beats(s, item, tgt_ref[x], tgt_idx[x])) atomicAdd(tgt_rank + x, word < 0 ? -1 : 1);
    }
}

// band list overflowed (degenerate inputs: a large part of the catalog within rounding distance of a held-out item):
// every rank is recounted from reference-order scores of the whole catalog, one block per user
__global__ __launch_bounds__(256) void exact_fallback_kernel(ExactArgs e, const int32_t* __restrict__ band_count,
                                                             int band_cap, const int32_t* __restrict__ train_ptr,
                                                             const int32_t* __restrict__ train_idx,
                                                             const int32_t* __restrict__ tgt_ptr,
                                                             const int32_t* __restrict__ tgt_idx,
                                                             const float* __restrict__ tgt_ref,
                                                             const int32_t* __restrict__ tgt_masked,
                                                             int32_t* __restrict__ tgt_rank) {
    if (band_count[0] <= band_cap) return;
    __shared__ int rc[256];
    const int tid = threadIdx.x;
    for (int64_t u = blockIdx.x; u < e.Nu; u += gridDim.x) {
        const int tb = train_ptr ? train_ptr[u] : 0, te = train_ptr ? train_ptr[u + 1] : 0;
        for (int x = tgt_ptr[u]; x < tgt_ptr[u + 1]; ++x) {
            if (tgt_masked[x]) continue;  // block-uniform
            const float st = tgt_ref[x];
            const int ct = tgt_idx[x];
            int cnt = 0;
            for (int64_t c = tid; c < e.Ni; c += 256) {
                int lo = tb, hi = te;
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if (train_idx[mid] < c) lo = mid + 1; else hi = mid;
                }
                if (lo < te && train_idx[lo] == c) continue;
                const float s = ref_order_score(e, u, c);
                if (s == s && beats(s, (int)c, st, ct)) ++cnt;
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

struct ExactLayout {
    int64_t topk, tgt_ref, tgt_hi, un, in, scal, band_count, band_list, total;
    int64_t band_cap;
    TopkLayout T;
};
static void exact_layout(int64_t Nu, int64_t Ni, int k, int64_t ntgt, int kind, ExactLayout* E) {
    topk_layout(Nu, Ni, k, 1, ntgt, kind, &E->T);
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += align_up(bytes > 0 ? bytes : 1, 256);
        return o;
    };
    E->topk = take(E->T.total);
    E->tgt_ref = take(ntgt * 4);
    E->tgt_hi = take(ntgt * 4);
    E->un = take(Nu * 4);
    E->in = take(Ni * 4);
    E->scal = take(256);
    E->band_count = take(256);
    E->band_cap = 64 * ntgt + 65536;
    if (E->band_cap > (1ll << 30)) E->band_cap = 1ll << 30;
    E->band_list = take(E->band_cap * 12);
    E->total = off;
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
    rc = prepare_operands(U, U2, Nu, I, I2, item_bias, Ni, k, score_kind, alpha, 1.0f - alpha, base, L, st, &a);
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
    const int64_t panels = (Nu + TU - 1) / TU;
    DR_REQUIRE(panels < (1ll << 31), "dr_score_topk_f32: too many user panels");
    const dim3 grid((unsigned)panels, (unsigned)L.S);
    // k <= 128: the register-selection kernel serves calls in which no user has more than one held-out item; the
    // general kernel the others.  Which it is is a device-side fact (the target CSR), so both are enqueued and
    // the one not in charge returns at once.
    const bool t1 = a.k <= 128 && n_targets <= Nu && DR_EXP_ENV("DR_SCORE_NO_T1") == nullptr;
    if (t1) {
        int32_t* flag = (int32_t*)(base + L.mode);
        DR_HIP(hipMemsetAsync(flag, 0, 4, st));
        a.mode_flag = flag;
        if (n_targets > 0) {
            DR_LAUNCH(target_mode_kernel, dim3(grid_for(Nu, 256)), dim3(256), 0, st, tgt_ptr, Nu, flag);
            DR_CHECK_LAUNCH("target_mode_kernel");
        }
        rc = launch_score_t1(a, grid, st);
        if (rc != DR_OK) return rc;
    }
    if (!t1 || n_targets > 0) {   // (with t1 and no targets at all the register kernel is certainly in charge)
        rc = launch_score<false>(a, grid, st);
        if (rc != DR_OK) return rc;
    }
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

extern "C" int32_t dr_score_rank_f32(const float* U, const float* U2, int64_t Nu, const float* I, const float* I2,
                                     const float* item_bias, int64_t Ni, int32_t k, int32_t score_kind, float alpha,
                                     const int32_t* train_ptr, const int32_t* train_idx, const int32_t* tgt_ptr,
                                     const int32_t* tgt_idx, int64_t n_targets, int32_t* tgt_rank, void* ws,
                                     int64_t ws_bytes, void* stream) {
    int rc = check_score_args(U, U2, Nu, I, I2, Ni, k, score_kind, "dr_score_rank_f32");
    if (rc != DR_OK) return rc;
    DR_REQUIRE(n_targets >= 0 && (n_targets == 0 || (tgt_ptr && tgt_idx && tgt_rank)),
               "dr_score_rank_f32: target CSR / rank output is NULL");
    DR_REQUIRE((train_ptr == nullptr) == (train_idx == nullptr) || train_ptr, "dr_score_rank_f32: train CSR incomplete");
    DR_REQUIRE(ws && ((uintptr_t)ws % 256 == 0), "dr_score_rank_f32: workspace NULL / misaligned");
    if (Nu == 0 || n_targets == 0) return DR_OK;
    TopkLayout L;
    topk_layout(Nu, Ni, k, 1, n_targets, score_kind, &L);
    if (ws_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_score_rank_f32: workspace %lld < required %lld", (long long)ws_bytes,
                (long long)L.total);
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    ScoreArgs a;
    memset(&a, 0, sizeof(a));
    rc = prepare_operands(U, U2, Nu, I, I2, item_bias, Ni, k, score_kind, alpha, 1.0f - alpha, base, L, st, &a);
    if (rc != DR_OK) return rc;
    a.Nu = Nu;
    a.Ni = Ni;
    a.train_ptr = train_ptr;
    a.train_idx = train_idx;
    a.K = 1;                                   // the general kernel (several held-out items per user) keeps a 1-list
    a.ntiles = L.ntiles;
    a.tiles_per_split = L.tiles_per_split;
    a.part_v = (float*)(base + L.part_v);
    a.part_i = (int32_t*)(base + L.part_i);
    a.tgt_ptr = tgt_ptr;
    a.tgt_idx = tgt_idx;
    a.tgt_score = (float*)(base + L.tgt_score);
    a.tgt_masked = (int32_t*)(base + L.tgt_masked);
    a.tgt_rank = tgt_rank;
    DR_LAUNCH(target_prologue_kernel, dim3(grid_for(Nu, 256)), dim3(256), 0, st, a.U, a.I, a.uadd, a.iadd, a.cdot, Nu, Ni,
              a.k, (const float*)nullptr, train_ptr, train_idx, tgt_ptr, tgt_idx, (float*)(base + L.tgt_score),
              (int32_t*)(base + L.tgt_masked), tgt_rank);
    DR_CHECK_LAUNCH("target_prologue_kernel");
    if (Ni > 0) {
        const int64_t panels = (Nu + TU - 1) / TU;
        DR_REQUIRE(panels < (1ll << 31), "dr_score_rank_f32: too many user panels");
        const dim3 grid((unsigned)panels, (unsigned)L.S);
        // the rank-only kernel serves calls in which no user has more than one held-out item, the general kernel the
        // others: a device-side fact (the target CSR), so both are enqueued and the one not in charge returns at once
        const bool fast = a.k <= 256 && n_targets <= Nu;
        if (fast) {
            int32_t* flag = (int32_t*)(base + L.mode);
            DR_HIP(hipMemsetAsync(flag, 0, 4, st));
            a.mode_flag = flag;
            DR_LAUNCH(target_mode_kernel, dim3(grid_for(Nu, 256)), dim3(256), 0, st, tgt_ptr, Nu, flag);
            DR_CHECK_LAUNCH("target_mode_kernel");
            rc = launch_score_rank<false>(a, (unsigned)L.S, st);
            if (rc != DR_OK) return rc;
        }
        rc = launch_score<false>(a, grid, st);
        if (rc != DR_OK) return rc;
    }
    DR_LAUNCH(target_finalize_kernel, dim3(grid_for(n_targets, 256)), dim3(256), 0, st,
              (const int32_t*)(base + L.tgt_masked), n_targets, tgt_rank);
    DR_CHECK_LAUNCH("target_finalize_kernel");
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
    rc = prepare_operands(U, U2, Nu, I, I2, item_bias, Ni, k, score_kind, alpha, 1.0f - alpha, (char*)ws, L, st, &a);
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
    return launch_score<true>(a, dim3((unsigned)panels, (unsigned)S), st);
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

extern "C" int64_t dr_score_rank_exact_bytes(int64_t Nu, int64_t Ni, int32_t k, int64_t n_targets, int32_t score_kind) {
    if (Nu < 0 || Ni < 0 || k < 1 || n_targets < 0) return DR_ERR_ARG;
    ExactLayout E;
    exact_layout(Nu, Ni, k, n_targets, score_kind, &E);
    return E.total;
}

extern "C" int32_t dr_score_rank_exact_f32(const float* U, const float* U2, int64_t Nu, const float* I, const float* I2,
                                           const float* item_bias, int64_t Ni, int32_t k, int32_t score_kind, float w1,
                                           float w2, const int32_t* train_ptr, const int32_t* train_idx,
                                           const int32_t* tgt_ptr, const int32_t* tgt_idx, int64_t n_targets,
                                           int32_t* tgt_rank, void* ws, int64_t ws_bytes, void* stream) {
    int rc = check_score_args(U, U2, Nu, I, I2, Ni, k, score_kind, "dr_score_rank_exact_f32");
    if (rc != DR_OK) return rc;
    DR_REQUIRE(k <= 8192, "dr_score_rank_exact_f32: k must be <= 8192");
    DR_REQUIRE(n_targets >= 0 && (n_targets == 0 || (tgt_ptr && tgt_idx && tgt_rank)),
               "dr_score_rank_exact_f32: target CSR / rank output is NULL");
    DR_REQUIRE((train_ptr == nullptr) == (train_idx == nullptr) || train_ptr, "dr_score_rank_exact_f32: train CSR incomplete");
    DR_REQUIRE(ws && ((uintptr_t)ws % 256 == 0), "dr_score_rank_exact_f32: workspace NULL / misaligned");
    if (Nu == 0 || n_targets == 0) return DR_OK;
    ExactLayout E;
    exact_layout(Nu, Ni, k, n_targets, score_kind, &E);
    if (ws_bytes < E.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_score_rank_exact_f32: workspace %lld < required %lld", (long long)ws_bytes,
                (long long)E.total);
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws + E.topk;
    const TopkLayout& L = E.T;
    ScoreArgs a;
    memset(&a, 0, sizeof(a));
    rc = prepare_operands(U, U2, Nu, I, I2, item_bias, Ni, k, score_kind, w1, w2, base, L, st, &a);
    if (rc != DR_OK) return rc;
    a.Nu = Nu;
    a.Ni = Ni;
    a.train_ptr = train_ptr;
    a.train_idx = train_idx;
    a.K = 1;
    a.ntiles = L.ntiles;
    a.tiles_per_split = L.tiles_per_split;
    a.part_v = (float*)(base + L.part_v);
    a.part_i = (int32_t*)(base + L.part_i);
    a.tgt_ptr = tgt_ptr;
    a.tgt_idx = tgt_idx;
    a.tgt_score = (float*)(base + L.tgt_score);  // lower end of the band
    a.tgt_masked = (int32_t*)(base + L.tgt_masked);
    a.tgt_rank = tgt_rank;
    char* xb = (char*)ws;
    float* tgt_ref = (float*)(xb + E.tgt_ref);
    float* tgt_hi = (float*)(xb + E.tgt_hi);
    float* un = (float*)(xb + E.un);
    float* in = (float*)(xb + E.in);
    float* scal = (float*)(xb + E.scal);
    a.tgt_hi = tgt_hi;
    a.band_list = (int32_t*)(xb + E.band_list);
    a.band_count = (int32_t*)(xb + E.band_count);
    a.band_cap = (int)E.band_cap;
    // (weighted) squared norms of both sides, as the MFMA operands see them; max over the catalog; max |bias|
    const bool two = score_kind == DR_SCORE_TWO_DIST;
    const float n1 = two ? w1 : 1.0f, n2 = two ? w2 : 0.0f;
    DR_LAUNCH(neg_sqnorm_kernel, dim3(grid_for(Nu, 256)), dim3(256), 0, st, U, two ? U2 : (const float*)nullptr, Nu, k, n1, n2, un);
    if (Ni > 0)
        DR_LAUNCH(neg_sqnorm_kernel, dim3(grid_for(Ni, 256)), dim3(256), 0, st, I, two ? I2 : (const float*)nullptr, Ni, k, n1, n2, in);
    DR_LAUNCH(max_abs_kernel, dim3(1), dim3(1024), 0, st, (const float*)in, Ni, scal);
    DR_LAUNCH(max_abs_kernel, dim3(1), dim3(1024), 0, st, item_bias, Ni, scal + 1);
    DR_CHECK_LAUNCH("dr_score_rank_exact_f32: norms");
    ExactArgs e;
    e.U = U; e.U2 = U2; e.I = I; e.I2 = I2; e.bias = item_bias;
    e.Nu = Nu; e.Ni = Ni; e.k = k; e.kind = score_kind; e.w1 = w1; e.w2 = w2;
    DR_LAUNCH(exact_prologue_kernel, dim3(grid_for(Nu, 256)), dim3(256), 0, st, e, a.k, (const float*)un,
              (const float*)scal, (const float*)(scal + 1), train_ptr, train_idx, tgt_ptr, tgt_idx, tgt_ref,
              (float*)(base + L.tgt_score), tgt_hi, (int32_t*)(base + L.tgt_masked), tgt_rank, a.band_count);
    DR_CHECK_LAUNCH("exact_prologue_kernel");
    const int64_t panels = (Nu + TU - 1) / TU;
    DR_REQUIRE(panels < (1ll << 31), "dr_score_rank_exact_f32: too many user panels");
    if (a.k <= 256 && n_targets <= Nu && Ni > 0) {
        // no user with more than one held-out item (a device-side fact): the rank-only kernel is in charge and the general
        // one returns at once
        int32_t* flag = (int32_t*)(base + L.mode);
        DR_HIP(hipMemsetAsync(flag, 0, 4, st));
        a.mode_flag = flag;
        DR_LAUNCH(target_mode_kernel, dim3(grid_for(Nu, 256)), dim3(256), 0, st, tgt_ptr, Nu, flag);
        DR_CHECK_LAUNCH("target_mode_kernel");
        rc = launch_score_rank<true>(a, (unsigned)L.S, st);
        if (rc != DR_OK) return rc;
    }
    rc = launch_score_band(a, dim3((unsigned)panels, (unsigned)L.S), st);
    if (rc != DR_OK) return rc;
    DR_LAUNCH(exact_refine_kernel, dim3(grid_for(E.band_cap < (1 << 22) ? E.band_cap : (1 << 22), 256)), dim3(256), 0, st, e,
              (const int32_t*)a.band_list, (const int32_t*)a.band_count, a.band_cap, tgt_idx, (const float*)tgt_ref,
              tgt_rank);
    DR_CHECK_LAUNCH("exact_refine_kernel");
    DR_LAUNCH(exact_fallback_kernel, dim3((unsigned)(Nu < 2048 ? Nu : 2048)), dim3(256), 0, st, e,
              (const int32_t*)a.band_count, a.band_cap, train_ptr, train_idx, tgt_ptr, tgt_idx, (const float*)tgt_ref,
              (const int32_t*)(base + L.tgt_masked), tgt_rank);
    DR_CHECK_LAUNCH("exact_fallback_kernel");
    DR_LAUNCH(target_finalize_kernel, dim3(grid_for(n_targets, 256)), dim3(256), 0, st,
              (const int32_t*)(base + L.tgt_masked), n_targets, tgt_rank);
    DR_CHECK_LAUNCH("target_finalize_kernel");
    return DR_OK;
}

#if DR_ABL == 9
extern "C" int32_t dr_debug_trace(void* dst, int64_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(dr::dbg_trace), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
