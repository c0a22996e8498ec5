// K1: embedding lookup  out[n,:] = table[ids[n],:]   (HBM-bound)
//
// Work decomposition: a row of k floats is moved by LPR lanes as V-float vectors, so one
// wavefront moves 64/LPR rows per pass (k=128: 32 lanes x float4, 2 rows per pass).  Each
// wave keeps U passes in flight (ids first, then the U independent row loads) so that a CU
// has >= 64 KB outstanding, which is what hides an HBM miss (MI355X_MICROARCH.md, "Indexed
// rows").  Output rows are written with streaming (non-temporal) stores: they are never
// re-read by this kernel and must not evict table rows that repeat (popular items) from L2.
#include "dr_common.h"

namespace dr {

template <int V, int LPR, int U, typename IdT>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ table, int64_t rows,
                                                     int k, const IdT* __restrict__ ids, int64_t n,
                                                     float* __restrict__ out,
                                                     int* __restrict__ oob_count) {
    constexpr int RPW = 64 / LPR;          // rows per wave pass
    constexpr int RPB = RPW * 4;           // rows per block pass (4 waves)
    const int tid = threadIdx.x;
    const int sub = tid / LPR;             // which row of the block pass
    const int lane = tid % LPR;
    const int cpr = k / V;                 // vectors per row
    int bad = 0;
    // block-cyclic over groups of RPB*U rows
    for (int64_t base = (int64_t)blockIdx.x * (RPB * U); base < n; base += (int64_t)gridDim.x * (RPB * U)) {
        int64_t id[U];
        int64_t row[U];
#pragma unroll
        // Loads are UNCONDITIONAL (a position past the end re-reads the last id, an id out of range reads row 0 and is
        // zeroed by a select afterwards): written as `x = 0; if (ok) x = load(..)` every load was followed by its own
        // s_waitcnt vmcnt(0) -- the U ids of a pass were U round trips in a row, not one.
        for (int p = 0; p < U; ++p) {
            row[p] = base + p * RPB + sub;
            id[p] = (int64_t)ids[row[p] < n ? row[p] : n - 1];
        }
        for (int c = lane; c < cpr; c += LPR) {
            float v[U][V];
            if (rows > 0) {
#pragma unroll
                for (int p = 0; p < U; ++p) {
                    const bool ok = (id[p] >= 0) && (id[p] < rows);
                    vload<V>(v[p], table + (ok ? id[p] : 0) * k + c * V);
                }
            }
#pragma unroll
            for (int p = 0; p < U; ++p) {
                const bool ok = (id[p] >= 0) && (id[p] < rows);
#pragma unroll
                for (int x = 0; x < V; ++x) v[p][x] = ok ? v[p][x] : 0.0f;
            }
#pragma unroll
            for (int p = 0; p < U; ++p) {
                if (row[p] < n) {
                    float* dst = out + row[p] * k + c * V;
                    if constexpr (V == 4) {
                        typedef float f4 __attribute__((ext_vector_type(4)));
                        f4 t = {v[p][0], v[p][1], v[p][2], v[p][3]};
                        __builtin_nontemporal_store(t, reinterpret_cast<f4*>(dst));
                    } else if constexpr (V == 2) {
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        f2 t = {v[p][0], v[p][1]};
                        __builtin_nontemporal_store(t, reinterpret_cast<f2*>(dst));
                    } else {
                        __builtin_nontemporal_store(v[p][0], dst);
                    }
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int p = 0; p < U; ++p)
                if (row[p] < n && (id[p] < 0 || id[p] >= rows)) ++bad;
        }
    }
    if (oob_count != nullptr && bad) atomicAdd(oob_count, bad);
}

template <int V, int LPR, typename IdT>
static int launch_gather(const float* table, int64_t rows, int k, const IdT* ids, int64_t n, float* out,
                         int* oob_count, hipStream_t st) {
    constexpr int U = 4;
    constexpr int RPB = (64 / LPR) * 4;
    int grid = grid_for(n, 1, RPB * U);
    DR_LAUNCH_NAMED("gather_kernel", (gather_kernel<V, LPR, U, IdT>), dim3(grid), dim3(256), 0, st, table, rows, k, ids,
                    n, out, oob_count);
    return 0;
}

template <typename IdT>
static int dispatch_gather(const float* table, int64_t rows, int k, const IdT* ids, int64_t n, float* out,
                           int* oob, hipStream_t st) {
    RowShape s = row_shape(k);
#define DR_G(V_, L_) \
    if (s.V == V_ && s.LPR == L_) return launch_gather<V_, L_, IdT>(table, rows, k, ids, n, out, oob, st);
    DR_G(4, 1) DR_G(4, 2) DR_G(4, 4) DR_G(4, 8) DR_G(4, 16) DR_G(4, 32) DR_G(4, 64)
    DR_G(2, 1) DR_G(2, 2) DR_G(2, 4) DR_G(2, 8) DR_G(2, 16) DR_G(2, 32) DR_G(2, 64)
    DR_G(1, 1) DR_G(1, 2) DR_G(1, 4) DR_G(1, 8) DR_G(1, 16) DR_G(1, 32) DR_G(1, 64)
#undef DR_G
    return -1;
}

// out[e,:] = coef[e] * src[idx[e],:]   (idx NULL = e).  Materialises per-contribution gradient rows
// (g_b * p_u) for the row-sharded path, where they travel to the owning rank before being summed.
template <int V, int LPR>
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ src, int64_t src_rows, int k,
                                                         const int32_t* __restrict__ idx,
                                                         const float* __restrict__ coef, int64_t n,
                                                         float* __restrict__ out) {
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int cpr = k / V;
    for (int64_t e = (int64_t)blockIdx.x * RPB + sub; e < n; e += (int64_t)gridDim.x * RPB) {
        const int64_t r = idx ? (int64_t)idx[e] : e;
        const float cf = coef[e];
        const bool ok = r >= 0 && r < src_rows;
        for (int c = lane; c < cpr; c += LPR) {
            float v[V];
            if (ok) {
                vload<V>(v, src + r * k + c * V);
#pragma unroll
                for (int x = 0; x < V; ++x) v[x] = cf * v[x];
            } else {
#pragma unroll
                for (int x = 0; x < V; ++x) v[x] = 0.0f;
            }
            vstore<V>(out + e * k + c * V, v);
        }
    }
}

}  // namespace dr

extern "C" int32_t dr_scale_rows_f32(const float* src, int64_t src_rows, int32_t k, const int32_t* idx,
                                     const float* coef, int64_t n, float* out, void* stream) {
    DR_REQUIRE(src_rows >= 0 && k >= 1 && n >= 0, "dr_scale_rows_f32: bad sizes");
    if (n == 0) return DR_OK;
    DR_REQUIRE(src && coef && out, "dr_scale_rows_f32: NULL pointer");
    dr::RowShape s = dr::row_shape(k);
    bool ok = false;
#define DR_F(V_, L_)                                                                                           \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                     \
        DR_LAUNCH((dr::scale_rows_kernel<V_, L_>), dim3(dr::grid_for(n, 1, 256 / L_)), dim3(256), 0,  \
                           (hipStream_t)stream, src, src_rows, k, idx, coef, n, out);                          \
        ok = true;                                                                                             \
    }
    DR_F(4, 1) DR_F(4, 2) DR_F(4, 4) DR_F(4, 8) DR_F(4, 16) DR_F(4, 32) DR_F(4, 64)
    DR_F(2, 1) DR_F(2, 2) DR_F(2, 4) DR_F(2, 8) DR_F(2, 16) DR_F(2, 32) DR_F(2, 64)
    DR_F(1, 1) DR_F(1, 2) DR_F(1, 4) DR_F(1, 8) DR_F(1, 16) DR_F(1, 32) DR_F(1, 64)
#undef DR_F
    DR_REQUIRE(ok, "dr_scale_rows_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("dr_scale_rows_f32");
    return DR_OK;
}

extern "C" int32_t dr_gather_f32(const float* table, int64_t rows, int32_t k, const void* ids, int64_t n,
                                 int32_t id_bits, float* out, int32_t oob_policy, int32_t* oob_count,
                                 void* stream) {
    DR_REQUIRE(rows >= 0 && k >= 1 && n >= 0, "dr_gather_f32: bad sizes rows=%lld k=%d n=%lld",
               (long long)rows, k, (long long)n);
    DR_REQUIRE(id_bits == 32 || id_bits == 64, "dr_gather_f32: id_bits must be 32 or 64, got %d", id_bits);
    DR_REQUIRE(oob_policy == DR_OOB_ZERO || oob_policy == DR_OOB_ERROR, "dr_gather_f32: bad oob_policy %d",
               oob_policy);
    DR_REQUIRE(oob_policy != DR_OOB_ERROR || oob_count != nullptr,
               "dr_gather_f32: DR_OOB_ERROR needs an oob_count word");
    if (n == 0) return DR_OK;
    DR_REQUIRE(table != nullptr || rows == 0, "dr_gather_f32: table is NULL");
    DR_REQUIRE(ids != nullptr && out != nullptr, "dr_gather_f32: ids/out is NULL");
    DR_REQUIRE(((uintptr_t)table % 16 == 0) && ((uintptr_t)out % 16 == 0),
               "dr_gather_f32: table/out must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int rc = (id_bits == 32)
                 ? dr::dispatch_gather<int32_t>(table, rows, k, (const int32_t*)ids, n, out, oob_count, st)
                 : dr::dispatch_gather<int64_t>(table, rows, k, (const int64_t*)ids, n, out, oob_count, st);
    DR_REQUIRE(rc == 0, "dr_gather_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("dr_gather_f32");
    return DR_OK;
}

// ---------------------------------------------------------------------------------------
// on-device negative sampler (bprmf.py:126-133,175-181)
// ---------------------------------------------------------------------------------------
namespace dr {
// 32 uniform bits of the counter-based generator shared with dr_init_normal_f32 (splitmix64 finaliser)
__host__ __device__ inline uint32_t counter_u32(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

// out[e*n_neg + d] = neg_items[u][s], s = floor(r * |neg_items[u]| / 2^32): the s-th item (ascending) that is
// NOT in the user's sorted train list -- found by a binary search for the largest prefix of the train list
// that lies before the answer (train[p] - p <= s), no rejection loop and no U x I structure.
__global__ __launch_bounds__(256) void sample_neg_kernel(const int32_t* __restrict__ train_ptr,
                                                         const int32_t* __restrict__ train_idx,
                                                         const int32_t* __restrict__ users, int64_t n, int n_neg,
                                                         int64_t num_user, int64_t num_item, int64_t first_item,
                                                         uint64_t seed, uint64_t offset, int32_t* __restrict__ out) {
    const int64_t total = n * n_neg;
    for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < total; x += (int64_t)gridDim.x * 256) {
        const int64_t e = x / n_neg;
        const int64_t u = users[e];
        int32_t res = -1;
        if (u >= 0 && u < num_user) {
            int lo0 = train_ptr[u];
            const int hi0 = train_ptr[u + 1];
            // train items below first_item are outside the candidate range (1-based item ids of the seq loaders)
            {
                int a = lo0, b = hi0;
                while (a < b) {
                    const int m = (a + b) >> 1;
                    if (train_idx[m] < first_item) a = m + 1; else b = m;
                }
                lo0 = a;
            }
            const int64_t cnt = (num_item - first_item) - (hi0 - lo0);
            if (cnt > 0) {
                const uint64_t r = counter_u32(seed, offset + (uint64_t)x);
                const int64_t sidx = (int64_t)((r * (uint64_t)cnt) >> 32);
                int lo = 0, hi = hi0 - lo0;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((int64_t)train_idx[lo0 + mid] - first_item - mid <= sidx) lo = mid + 1; else hi = mid;
                }
                res = (int32_t)(sidx + lo + first_item);
            }
        }
        out[x] = res;
    }
}

// The sequence models' sampler (AttRec.py:448-466, _get_neg_items_sbpr): the pool of sample e is every item
// first_item .. num_item-1 EXCEPT the sample's own T targets; out[e*n_neg + d] = pool[s], s = floor(r * |pool| / 2^32).
// T is small (the reference uses 3): the distinct in-range targets are insertion-sorted in registers and the s-th
// item of the complement is found by walking them (target[p] - p <= s), like the host code in _seq_common.py.
constexpr int EXCL_MAX_T = 16;
__global__ __launch_bounds__(256) void sample_neg_excl_kernel(const int32_t* __restrict__ targets, int64_t n, int T,
                                                              int n_neg, int64_t num_item, int64_t first_item,
                                                              uint64_t seed, uint64_t offset, int32_t* __restrict__ out) {
    const int64_t total = n * n_neg;
    for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < total; x += (int64_t)gridDim.x * 256) {
        const int64_t e = x / n_neg;
        int64_t t[EXCL_MAX_T];
        int m = 0;
        for (int q = 0; q < T; ++q) {
            const int64_t v = targets[e * T + q];
            if (v < first_item || v >= num_item) continue;
            int p = m;
            bool dup = false;
            while (p > 0 && t[p - 1] >= v) {
                if (t[p - 1] == v) { dup = true; break; }
                --p;
            }
            if (dup) continue;
            for (int z = m; z > p; --z) t[z] = t[z - 1];
            t[p] = v;
            ++m;
        }
        const int64_t cnt = (num_item - first_item) - m;
        int32_t res = -1;
        if (cnt > 0) {
            const uint64_t r = counter_u32(seed, offset + (uint64_t)x);
            const int64_t sidx = (int64_t)((r * (uint64_t)cnt) >> 32);
            int lo = 0;
            while (lo < m && t[lo] - first_item - lo <= sidx) ++lo;
            res = (int32_t)(sidx + lo + first_item);
        }
        out[x] = res;
    }
}
}  // namespace dr

extern "C" int32_t dr_sample_negatives_excl_i32(const int32_t* targets, int64_t n, int32_t T, int32_t n_neg,
                                                int64_t num_item, int64_t first_item, uint64_t seed, uint64_t offset,
                                                int32_t* out, void* stream) {
    DR_REQUIRE(n >= 0 && n_neg >= 1 && T >= 0 && T <= dr::EXCL_MAX_T && num_item >= 0 && first_item >= 0 &&
                   first_item <= num_item,
               "dr_sample_negatives_excl_i32: bad sizes (T <= %d)", dr::EXCL_MAX_T);
    DR_REQUIRE(num_item < (1ll << 31), "dr_sample_negatives_excl_i32: num_item too large for int32 item ids");
    if (n == 0) return DR_OK;
    DR_REQUIRE(out && (targets || T == 0), "dr_sample_negatives_excl_i32: NULL pointer");
    DR_LAUNCH(dr::sample_neg_excl_kernel, dim3(dr::grid_for(n * n_neg, 256)), dim3(256), 0, (hipStream_t)stream, targets,
              n, (int)T, (int)n_neg, num_item, first_item, seed, offset, out);
    DR_CHECK_LAUNCH("dr_sample_negatives_excl_i32");
    return DR_OK;
}

extern "C" int32_t dr_sample_negatives_i32(const int32_t* train_ptr, const int32_t* train_idx, const int32_t* users,
                                           int64_t n, int32_t n_neg, int64_t num_user, int64_t num_item,
                                           int64_t first_item, uint64_t seed, uint64_t offset, int32_t* out,
                                           void* stream) {
    DR_REQUIRE(n >= 0 && n_neg >= 1 && num_user >= 0 && num_item >= 0 && first_item >= 0 && first_item <= num_item,
               "dr_sample_negatives_i32: bad sizes");
    DR_REQUIRE(num_item < (1ll << 31), "dr_sample_negatives_i32: num_item too large for int32 item ids");
    if (n == 0) return DR_OK;
    DR_REQUIRE(train_ptr && users && out && (train_idx || num_user == 0), "dr_sample_negatives_i32: NULL pointer");
    DR_LAUNCH(dr::sample_neg_kernel, dim3(dr::grid_for(n * n_neg, 256)), dim3(256), 0, (hipStream_t)stream, train_ptr,
              train_idx, users, n, (int)n_neg, num_user, num_item, first_item, seed, offset, out);
    DR_CHECK_LAUNCH("dr_sample_negatives_i32");
    return DR_OK;
}
