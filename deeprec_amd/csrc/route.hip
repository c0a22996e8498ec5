// K7 routing helpers of the row-sharded BPR step (deeprec_amd/sharding.py; the reference is single-device).
//
// Integer-only work on a few hundred thousand keys per step: HBM / latency bound, no MFMA.  Every entry point is one
// host call that enqueues a handful of short kernels (keys -> rocPRIM radix sort -> pack), so that the Python side
// of a step is a few calls instead of dozens of tensor ops.  All outputs are pure functions of the inputs
// (stable sorts, no atomics on anything that is ordered), restated in tests/_oracle_ops.py.
#include "dr_common.h"

#include <rocprim/rocprim.hpp>

namespace dr {

// The view sort (4 * world * B keys, 262 144 at world 1 with the C4 batch) with rocPRIM's default configuration is a block
// sort plus SIXTEEN merge launches (8 passes of partition + merge-path above 200 k keys).  4096 keys per block sort and
// the one-kernel odd-even merge up to 1 M keys: 6 passes, 7 launches -- the sharded step is bound by its launch count.
using RouteSortConfig = rocprim::radix_sort_config<rocprim::default_config,
                                                   rocprim::merge_sort_config<512, 1024, 4, 128, 128, 4, (1u << 20)>,
                                                   rocprim::default_config, 1024 * 1024>;

struct RouteLayout {
    int64_t keys_in, keys_out, vals_in, vals_out, flags, segidx, temp, temp_bytes, total;
};

static int route_layout_compute(int64_t n, RouteLayout* L);
static int route_layout(int64_t n, RouteLayout* L) {   // pure function of n; the rocPRIM size queries are not free
    static thread_local int64_t memo_n[4] = {-1, -1, -1, -1};
    static thread_local RouteLayout memo_L[4];
    static thread_local int next = 0;
    for (int i = 0; i < 4; ++i)
        if (memo_n[i] == n) {
            *L = memo_L[i];
            return 0;
        }
    const int rc = route_layout_compute(n, L);
    if (rc == 0) {
        memo_n[next] = n;
        memo_L[next] = *L;
        next = (next + 1) % 4;
    }
    return rc;
}

static int route_layout_compute(int64_t n, RouteLayout* L) {
    size_t t_sort = 0, t_scan = 0;
    const int64_t nn = n > 0 ? n : 1;
    hipError_t e = rocprim::radix_sort_pairs<RouteSortConfig, uint32_t*, uint32_t*, int32_t*, int32_t*>(
        nullptr, t_sort, nullptr, nullptr, nullptr, nullptr, (size_t)nn, 0, 32, 0, false);
    if (e != hipSuccess) return -1;
    e = rocprim::exclusive_scan<rocprim::default_config, int32_t*, int32_t*, int32_t, rocprim::plus<int32_t>>(
        nullptr, t_scan, nullptr, nullptr, 0, (size_t)nn, rocprim::plus<int32_t>(), 0, false);
    if (e != hipSuccess) return -1;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    L->keys_in = take(nn * 4);
    L->keys_out = take(nn * 4);
    L->vals_in = take(nn * 4);
    L->vals_out = take(nn * 4);
    L->flags = take(nn * 4);
    L->segidx = take(nn * 4);
    L->temp_bytes = (int64_t)(t_sort > t_scan ? t_sort : t_scan) + 256;
    L->temp = take(L->temp_bytes);
    L->total = off;
    return 0;
}

static unsigned bits_for(uint64_t max_value) {  // radix bits that cover 0 .. max_value
    unsigned b = 1;
    while (b < 32 && (max_value >> b) != 0) ++b;
    return b;
}

constexpr int MAX_WORLD = 64;

// ---- both roles of a rank, derived from the replicated id lists ----------------------------------------------
// ids = [W][3][B]: rank s's (u, i, j) lists (one fixed-size all-gather).  The global batch is the concatenation in rank
// order; global sample g = s * B + position.  List entry e in [0, 2 W B): part = e / (W B) (0: item i, 1: item j),
// sample g = e % (W B) -- ascending e IS the canonical contribution order of a single process (i-part before j-part,
// then global batch position).  Two views of the same list, sorted in one pass (view bit above the key):
//   view 0 (processing rank): entries of samples with u % W == me, key = owner(item) * rpo + item / W
//   view 1 (row owner):       entries with item % W == me,         key = (u % W)     * rpo + item / W
// Entries that do not belong to a view get the view's sentinel key (sorts behind the view's real keys).  A stable sort
// then leaves, per view and per major (owner resp. processing rank), rows ascending and equal rows in canonical order:
// exactly the order the sender packs and the receiver expects, so no ids, requests or metadata ever travel.
struct ViewArgs {
    const int32_t* ids;
    int64_t B, WB;
    int W, me;
    int64_t nu, ni;
    uint32_t rpo, sentinel;  // sentinel = W * rpo; view 1 keys are offset by sentinel + 1
};

__device__ __forceinline__ bool view_entry(const ViewArgs& a, int64_t x, uint32_t* key, int32_t* u_out) {
    // x in [0, 4 W B): view = x / (2 W B), e = x % (2 W B)
    const int view = x >= 2 * a.WB;
    const int64_t e = x - (view ? 2 * a.WB : 0);
    const int part = e >= a.WB;
    const int64_t g = e - (part ? a.WB : 0);
    const int64_t s = g / a.B, pos = g - s * a.B;
    const int32_t* base = a.ids + s * 3 * a.B;
    const int64_t uu = base[pos], ii = base[a.B + pos], jj = base[2 * a.B + pos];
    const bool ok = uu >= 0 && uu < a.nu && ii >= 0 && ii < a.ni && jj >= 0 && jj < a.ni;
    const uint32_t item = (uint32_t)(part ? jj : ii);
    const uint32_t r = ok ? (uint32_t)(uu % a.W) : 0u, o = ok ? item % (uint32_t)a.W : 0u;
    const bool mine = ok && (view ? o == (uint32_t)a.me : r == (uint32_t)a.me);
    const uint32_t major = view ? r : o;
    *key = (view ? a.sentinel + 1u : 0u) + (mine ? major * a.rpo + item / (uint32_t)a.W : a.sentinel);
    *u_out = (int32_t)uu;
    return mine;
}

__global__ void view_zero_kernel(unsigned long long* __restrict__ counts, int n, int32_t* __restrict__ last_flag) {
    for (int t = threadIdx.x; t < n; t += blockDim.x) counts[t] = 0ull;
    if (threadIdx.x == 0) *last_flag = 0;
}

// keys + values of the 4 W B list entries; counts[4 W + 2] = samples (of the GLOBAL batch) with an id out of range
__global__ __launch_bounds__(256) void view_keys_kernel(ViewArgs a, uint32_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                        int32_t* __restrict__ flags,
                                                        unsigned long long* __restrict__ counts) {
    int bad = 0;
    for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < 4 * a.WB; x += (int64_t)gridDim.x * 256) {
        uint32_t key;
        int32_t uu;
        const bool mine = view_entry(a, x, &key, &uu);
        keys[x] = key;
        vals[x] = (int32_t)x;
        flags[x] = mine ? 1 : 0;
        if (x < a.WB) {  // validity once per sample
            const int64_t s = x / a.B, pos = x - s * a.B;
            const int32_t* base = a.ids + s * 3 * a.B;
            const int64_t ii = base[a.B + pos], jj = base[2 * a.B + pos];
            if (!(uu >= 0 && uu < a.nu && ii >= 0 && ii < a.ni && jj >= 0 && jj < a.ni)) ++bad;
        }
    }
    if (bad) atomicAdd(&counts[4 * a.W + 2], (unsigned long long)bad);  // rare: a refused batch
}

// heads of the distinct keys of the sorted list (sentinels are never heads)
__global__ __launch_bounds__(256) void view_heads_kernel(const uint32_t* __restrict__ sk, int64_t n, uint32_t sentinel,
                                                         int32_t* __restrict__ flags) {
    for (int64_t y = (int64_t)blockIdx.x * 256 + threadIdx.x; y < n; y += (int64_t)gridDim.x * 256) {
        const uint32_t k = sk[y];
        const bool real = k != sentinel && k != 2u * sentinel + 1u;
        flags[y] = (real && (y == 0 || k != sk[y - 1])) ? 1 : 0;
    }
}

// per view v (outputs at offset v * 2 W B): c = compacted canonical index of the entry, y = its position in the view's
// sorted list, seg = index of its distinct (major, row):
//   rows[c] = local row, inv[c] = seg, order[y] = c, perm[c] = y, uniq[seg] = row, u_loc[c] = u / W (view 0, part 0)
// counts[2 (v W + major)] += distinct, counts[2 (v W + major) + 1] += entries; counts[4 W + v] = entries of the view
__global__ __launch_bounds__(256) void view_pack_kernel(ViewArgs a, const uint32_t* __restrict__ sk,
                                                        const int32_t* __restrict__ sv, const int32_t* __restrict__ cidx,
                                                        const int32_t* __restrict__ heads, const int32_t* __restrict__ segidx,
                                                        int32_t* __restrict__ u_loc, int32_t* __restrict__ rows,
                                                        int32_t* __restrict__ inv, int32_t* __restrict__ order,
                                                        int32_t* __restrict__ perm, int32_t* __restrict__ uniq,
                                                        unsigned long long* __restrict__ counts) {
    __shared__ int hist[4 * MAX_WORLD];
    for (int b = threadIdx.x; b < 4 * a.W; b += 256) hist[b] = 0;
    __syncthreads();
    const int64_t half = 2 * a.WB;
    const int32_t c1 = cidx[half], s1 = segidx[half];  // first compacted index / segment of view 1
    for (int64_t ys = (int64_t)blockIdx.x * 256 + threadIdx.x; ys < 2 * half; ys += (int64_t)gridDim.x * 256) {
        uint32_t key = sk[ys];
        const int view = ys >= half;
        if (view) key -= a.sentinel + 1u;
        if (key == a.sentinel) continue;
        const int64_t x = sv[ys];
        const int64_t off = view ? half : 0;
        const int32_t c = cidx[x] - (view ? c1 : 0);
        const int32_t y = (int32_t)(ys - off);
        const int head = heads[ys];
        const int32_t seg = segidx[ys] - (head ? 0 : 1) - (view ? s1 : 0);
        const uint32_t major = key / a.rpo;
        const int32_t row = (int32_t)(key - major * a.rpo);
        rows[off + c] = row;
        inv[off + c] = seg;
        order[off + y] = c;
        perm[off + c] = y;
        if (head) {
            uniq[off + seg] = row;
            atomicAdd(&hist[2 * (view * a.W + (int)major)], 1);
        }
        atomicAdd(&hist[2 * (view * a.W + (int)major) + 1], 1);
        if (!view) {
            const int64_t e = x;
            if (e < a.WB) {
                const int64_t s = e / a.B, pos = e - s * a.B;
                u_loc[c] = a.ids[s * 3 * a.B + pos] / a.W;
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 4 * a.W; b += 256)
        if (hist[b]) atomicAdd(&counts[b], (unsigned long long)hist[b]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counts[4 * a.W] = (unsigned long long)c1;
        counts[4 * a.W + 1] = (unsigned long long)(cidx[2 * half] - c1);
    }
}

// ---- item-side gradient rows in wire order: out[y] = (+-g[s]) * P[u_loc[s]] ----------------------------------
template <int V, int LPR>
__global__ __launch_bounds__(256) void bpr_item_grads_kernel(const float* __restrict__ P, int64_t rows, int k,
                                                             const int32_t* __restrict__ u_loc,
                                                             const float* __restrict__ g,
                                                             const int32_t* __restrict__ order, int64_t n,
                                                             float* __restrict__ out) {
    constexpr int RPB = 256 / LPR;
    const int lane = threadIdx.x % LPR;
    const int cpr = k / V;
    for (int64_t y = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR; y < 2 * n; y += (int64_t)gridDim.x * RPB) {
        const int e = order[y];
        const int64_t s = e < n ? e : e - n;
        const float gs = g[s];
        const float cf = e < n ? gs : -gs;
        const int64_t r = u_loc[s];
        const bool ok = r >= 0 && r < rows;
        for (int c = lane; c < cpr; c += LPR) {
            float v[V];
            if (ok) {
                vload<V>(v, P + r * k + c * V);
#pragma unroll
                for (int x = 0; x < V; ++x) v[x] = cf * v[x];
            } else {
#pragma unroll
                for (int x = 0; x < V; ++x) v[x] = 0.0f;
            }
            vstore<V>(out + y * k + c * V, v);
        }
    }
}

}  // namespace dr

using namespace dr;

extern "C" int64_t dr_route_bytes(int64_t n) {
    if (n < 0) return DR_ERR_ARG;
    RouteLayout L;
    if (route_layout(n, &L) != 0) return DR_ERR_LAUNCH;
    return L.total;
}

extern "C" int32_t dr_route_views_i32(const int32_t* ids, int32_t world, int64_t B, int32_t me, int64_t num_user,
                                      int64_t num_item, int64_t rows_per_owner, int32_t* u_loc, int32_t* rows, int32_t* inv,
                                      int32_t* order, int32_t* perm, int32_t* uniq, int64_t* counts, void* ws,
                                      int64_t ws_bytes, void* stream) {
    DR_REQUIRE(world >= 1 && world <= MAX_WORLD && me >= 0 && me < world && B >= 0 && num_user >= 0 && num_item >= 0 &&
                   rows_per_owner >= 1 && (uint64_t)world * (uint64_t)rows_per_owner < (1ull << 31) - 1 &&
                   4 * (int64_t)world * B + 1 < (1ll << 31) - 1,
               "dr_route_views_i32: bad sizes (world in [1,%d], world * rows_per_owner < 2^31, 4 * world * B < 2^31)",
               MAX_WORLD);
    DR_REQUIRE(counts != nullptr, "dr_route_views_i32: counts is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        DR_HIP(hipMemsetAsync(counts, 0, (size_t)(4 * world + 3) * 8, st));
        return DR_OK;
    }
    DR_REQUIRE(ids && u_loc && rows && inv && order && perm && uniq, "dr_route_views_i32: NULL pointer");
    const int64_t WB = (int64_t)world * B, m = 4 * WB;
    RouteLayout L;
    if (route_layout(m + 1, &L) != 0) DR_FAIL(DR_ERR_LAUNCH, "dr_route_views_i32: rocprim size query failed");
    DR_REQUIRE(ws != nullptr && ((uintptr_t)ws % 256 == 0), "dr_route_views_i32: ws is NULL / not 256-byte aligned");
    if (ws_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_route_views_i32: workspace %lld < dr_route_bytes(%lld) = %lld", (long long)ws_bytes,
                (long long)(m + 1), (long long)L.total);
    char* base = (char*)ws;
    uint32_t* kin = (uint32_t*)(base + L.keys_in);
    uint32_t* kout = (uint32_t*)(base + L.keys_out);
    int32_t* vin = (int32_t*)(base + L.vals_in);
    int32_t* vout = (int32_t*)(base + L.vals_out);
    int32_t* flags = (int32_t*)(base + L.flags);    // m + 1 entries; reused as the scan outputs' input twice
    int32_t* scan = (int32_t*)(base + L.segidx);    // m + 1 entries
    size_t tb = (size_t)L.temp_bytes;
    ViewArgs a;
    a.ids = ids;
    a.B = B;
    a.WB = WB;
    a.W = world;
    a.me = me;
    a.nu = num_user;
    a.ni = num_item;
    a.rpo = (uint32_t)rows_per_owner;
    a.sentinel = (uint32_t)world * (uint32_t)rows_per_owner;
    const int grid = grid_for(m, 256);
    // counts and the scan's extra entry start at zero: one small launch (two hipMemsetAsync calls cost more on the host)
    DR_LAUNCH(view_zero_kernel, dim3(1), dim3(256), 0, st, (unsigned long long*)counts, 4 * world + 3, flags + m);
    DR_CHECK_LAUNCH("view_zero_kernel");
    DR_LAUNCH(view_keys_kernel, dim3(grid), dim3(256), 0, st, a, kin, vin, flags, (unsigned long long*)counts);
    DR_CHECK_LAUNCH("view_keys_kernel");
    DR_HIP((rocprim::radix_sort_pairs<RouteSortConfig>(base + L.temp, tb, kin, kout, vin, vout, (size_t)m, 0u,
                                                               bits_for(2ull * a.sentinel + 1ull), st, false)));
    // the compacted canonical index needs its own buffer next to the segment index: it lives in keys_in after the sort
    int32_t* cidx = (int32_t*)kin;                  // keys_in is free now (m * 4 bytes + the 256-byte pad holds entry m)
    tb = (size_t)L.temp_bytes;
    DR_HIP((rocprim::exclusive_scan<rocprim::default_config>(base + L.temp, tb, flags, cidx, (int32_t)0, (size_t)(m + 1),
                                                             rocprim::plus<int32_t>(), st, false)));
    DR_LAUNCH(view_heads_kernel, dim3(grid), dim3(256), 0, st, (const uint32_t*)kout, m, a.sentinel, flags);
    DR_CHECK_LAUNCH("view_heads_kernel");
    tb = (size_t)L.temp_bytes;
    DR_HIP((rocprim::exclusive_scan<rocprim::default_config>(base + L.temp, tb, flags, scan, (int32_t)0, (size_t)(m + 1),
                                                             rocprim::plus<int32_t>(), st, false)));
    DR_LAUNCH(view_pack_kernel, dim3(grid), dim3(256), 0, st, a, (const uint32_t*)kout, (const int32_t*)vout,
              (const int32_t*)cidx, (const int32_t*)flags, (const int32_t*)scan, u_loc, rows, inv, order, perm, uniq,
              (unsigned long long*)counts);
    DR_CHECK_LAUNCH("view_pack_kernel");
    return DR_OK;
}

extern "C" int32_t dr_bpr_item_grads_f32(const float* P, int64_t rows, int32_t k, const int32_t* u_loc, const float* g,
                                         const int32_t* order, int64_t n, float* out, void* stream) {
    DR_REQUIRE(rows >= 0 && k >= 1 && n >= 0 && 2 * n < (1ll << 31) - 1, "dr_bpr_item_grads_f32: bad sizes");
    if (n == 0) return DR_OK;
    DR_REQUIRE(P && u_loc && g && order && out, "dr_bpr_item_grads_f32: NULL pointer");
    RowShape s = row_shape(k);
    DR_REQUIRE(s.V == 1 || (((uintptr_t)P | (uintptr_t)out) % (4 * s.V) == 0), "dr_bpr_item_grads_f32: misaligned rows");
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
#define DR_C(V_, L_)                                                                                                 \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                           \
        DR_LAUNCH((bpr_item_grads_kernel<V_, L_>), dim3(grid_for(2 * n, 1, 256 / L_)), dim3(256), 0, st, P, rows,    \
                  (int)k, u_loc, g, order, n, out);                                                                  \
        ok = true;                                                                                                   \
    }
    DR_C(4, 1) DR_C(4, 2) DR_C(4, 4) DR_C(4, 8) DR_C(4, 16) DR_C(4, 32) DR_C(4, 64)
    DR_C(2, 1) DR_C(2, 2) DR_C(2, 4) DR_C(2, 8) DR_C(2, 16) DR_C(2, 32) DR_C(2, 64)
    DR_C(1, 1) DR_C(1, 2) DR_C(1, 4) DR_C(1, 8) DR_C(1, 16) DR_C(1, 32) DR_C(1, 64)
#undef DR_C
    DR_REQUIRE(ok, "dr_bpr_item_grads_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("bpr_item_grads_kernel");
    return DR_OK;
}
