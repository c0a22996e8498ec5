// K7 routing helpers of the row-sharded BPR step (deeprec_amd/sharding.py; the reference is single-device).
//
// Integer-only work on a few hundred thousand keys per step: HBM / latency bound, no MFMA.  Every entry point is one
// host call that enqueues a handful of short kernels (keys -> rocPRIM radix sort -> pack), so that the Python side
// of a step is a few calls instead of dozens of tensor ops.  All outputs are pure functions of the inputs
// (stable sorts, no atomics on anything that is ordered), restated in tests/_oracle_ops.py.
#include "dr_common.h"

#include <rocprim/rocprim.hpp>

namespace dr {

struct RouteLayout {
    int64_t keys_in, keys_out, vals_in, vals_out, flags, segidx, temp, temp_bytes, total;
};

static int route_layout(int64_t n, RouteLayout* L) {
    size_t t_sort = 0, t_scan = 0;
    const int64_t nn = n > 0 ? n : 1;
    hipError_t e = rocprim::radix_sort_pairs<rocprim::default_config, uint32_t*, uint32_t*, int32_t*, int32_t*>(
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

// ---- samples -> the rank that owns their user row -------------------------------------------------------
// key = u % W (bucket W: a sample with any id out of range; sorts last, counted apart, never sent)
__global__ __launch_bounds__(256) void route_sample_keys_kernel(const int32_t* __restrict__ u,
                                                                const int32_t* __restrict__ i,
                                                                const int32_t* __restrict__ j, int64_t B, int W,
                                                                int64_t nu, int64_t ni, uint32_t* __restrict__ keys,
                                                                int32_t* __restrict__ vals,
                                                                unsigned long long* __restrict__ counts) {
    __shared__ int hist[MAX_WORLD + 1];
    for (int b = threadIdx.x; b <= W; b += 256) hist[b] = 0;
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < B; e += (int64_t)gridDim.x * 256) {
        const int64_t uu = u[e], ii = i[e], jj = j[e];
        const bool ok = uu >= 0 && uu < nu && ii >= 0 && ii < ni && jj >= 0 && jj < ni;
        const int b = ok ? (int)(uu % W) : W;
        keys[e] = (uint32_t)b;
        vals[e] = (int32_t)e;
        atomicAdd(&hist[b], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= W; b += 256)
        if (hist[b]) atomicAdd(&counts[b], (unsigned long long)hist[b]);
}

// tuples in wire order: (local user row, item i, item j, tag + batch position)
__global__ __launch_bounds__(256) void route_sample_pack_kernel(const int32_t* __restrict__ order,
                                                                const int32_t* __restrict__ u,
                                                                const int32_t* __restrict__ i,
                                                                const int32_t* __restrict__ j, int64_t B, int W,
                                                                int32_t tag, int4* __restrict__ tuples) {
    for (int64_t y = (int64_t)blockIdx.x * 256 + threadIdx.x; y < B; y += (int64_t)gridDim.x * 256) {
        const int o = order[y];
        tuples[y] = make_int4(u[o] / W, i[o], j[o], tag + o);
    }
}

// ---- the item rows a processing rank needs, and where its gradient contributions go -------------------
// list entry e < n: item i of sample e; e >= n: item j of sample e - n.  okey = owner * rpo + local row.
__global__ __launch_bounds__(256) void route_item_keys_kernel(const int4* __restrict__ smp, int64_t n, int W,
                                                              uint32_t rpo, uint32_t* __restrict__ keys,
                                                              int32_t* __restrict__ vals, int32_t* __restrict__ u_loc,
                                                              unsigned long long* __restrict__ counts) {
    __shared__ int hist[MAX_WORLD];
    for (int b = threadIdx.x; b < W; b += 256) hist[b] = 0;
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < 2 * n; e += (int64_t)gridDim.x * 256) {
        const int4 t = smp[e < n ? e : e - n];
        const uint32_t id = (uint32_t)(e < n ? t.y : t.z);
        const uint32_t o = id % (uint32_t)W;
        keys[e] = o * rpo + id / (uint32_t)W;
        vals[e] = (int32_t)e;
        if (e < n) u_loc[e] = t.x;
        atomicAdd(&hist[o], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < W; b += 256)
        if (hist[b]) atomicAdd(&counts[2 * b + 1], (unsigned long long)hist[b]);
}

__global__ __launch_bounds__(256) void route_heads_kernel(const uint32_t* __restrict__ sk, int64_t n,
                                                          int32_t* __restrict__ flags) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        flags[e] = (e == 0 || sk[e] != sk[e - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void route_item_pack_kernel(const uint32_t* __restrict__ sk,
                                                              const int32_t* __restrict__ order,
                                                              const int32_t* __restrict__ flags,
                                                              const int32_t* __restrict__ segidx,
                                                              const int4* __restrict__ smp, int64_t n, int W,
                                                              uint32_t rpo, int part_bit, int32_t* __restrict__ inv,
                                                              int32_t* __restrict__ uniq_rows,
                                                              int2* __restrict__ meta,
                                                              unsigned long long* __restrict__ counts) {
    __shared__ int hist[MAX_WORLD];
    for (int b = threadIdx.x; b < W; b += 256) hist[b] = 0;
    __syncthreads();
    for (int64_t y = (int64_t)blockIdx.x * 256 + threadIdx.x; y < 2 * n; y += (int64_t)gridDim.x * 256) {
        const uint32_t key = sk[y];
        const int e = order[y];
        const int head = flags[y];
        const int seg = segidx[y] - (head ? 0 : 1);
        const uint32_t o = key / rpo;
        const int32_t row = (int32_t)(key - o * rpo);
        inv[e] = seg;
        const int gpos = e < n ? smp[e].w : (smp[e - n].w | (1 << part_bit));
        meta[y] = make_int2(row, gpos);
        if (head) {
            uniq_rows[seg] = row;
            atomicAdd(&hist[o], 1);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < W; b += 256)
        if (hist[b]) atomicAdd(&counts[2 * b], (unsigned long long)hist[b]);
}

// ---- owner side: received contributions -> canonical order (ascending global key) ------------------------
__global__ __launch_bounds__(256) void canon_keys_kernel(const int2* __restrict__ meta, int64_t n,
                                                         uint32_t* __restrict__ keys, int32_t* __restrict__ vals) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        keys[e] = (uint32_t)meta[e].y;
        vals[e] = (int32_t)e;
    }
}

__global__ __launch_bounds__(256) void canon_ids_kernel(const int2* __restrict__ meta, const int32_t* __restrict__ perm,
                                                        int64_t n, int32_t* __restrict__ ids) {
    for (int64_t y = (int64_t)blockIdx.x * 256 + threadIdx.x; y < n; y += (int64_t)gridDim.x * 256)
        ids[y] = meta[perm[y]].x;
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

#define DR_ROUTE_WS(n_, who)                                                                                         \
    RouteLayout L;                                                                                                   \
    if (route_layout((n_), &L) != 0) DR_FAIL(DR_ERR_LAUNCH, who ": rocprim size query failed");                      \
    DR_REQUIRE(ws != nullptr && ((uintptr_t)ws % 256 == 0), who ": ws is NULL / not 256-byte aligned");              \
    if (ws_bytes < L.total)                                                                                          \
        DR_FAIL(DR_ERR_WORKSPACE, who ": workspace %lld < dr_route_bytes(%lld) = %lld", (long long)ws_bytes,         \
                (long long)(n_), (long long)L.total);                                                                \
    char* base = (char*)ws;                                                                                          \
    uint32_t* kin = (uint32_t*)(base + L.keys_in);                                                                   \
    uint32_t* kout = (uint32_t*)(base + L.keys_out);                                                                 \
    int32_t* vin = (int32_t*)(base + L.vals_in);                                                                     \
    size_t tb = (size_t)L.temp_bytes

extern "C" int32_t dr_route_samples_i32(const int32_t* u, const int32_t* i, const int32_t* j, int64_t B, int32_t world,
                                        int64_t num_user, int64_t num_item, int32_t tag, int32_t* tuples,
                                        int64_t* counts, void* ws, int64_t ws_bytes, void* stream) {
    DR_REQUIRE(B >= 0 && B < (1ll << 31) - 1 && world >= 1 && world <= MAX_WORLD && num_user >= 0 && num_item >= 0,
               "dr_route_samples_i32: bad sizes (world in [1,%d])", MAX_WORLD);
    DR_REQUIRE(counts != nullptr, "dr_route_samples_i32: counts is NULL");
    hipStream_t st = (hipStream_t)stream;
    DR_HIP(hipMemsetAsync(counts, 0, (size_t)(world + 1) * 8, st));
    if (B == 0) return DR_OK;
    DR_REQUIRE(u && i && j && tuples && ((uintptr_t)tuples % 16 == 0), "dr_route_samples_i32: NULL / misaligned pointer");
    DR_ROUTE_WS(B, "dr_route_samples_i32");
    int32_t* order = (int32_t*)(base + L.vals_out);
    const int grid = grid_for(B, 256);
    DR_LAUNCH(route_sample_keys_kernel, dim3(grid), dim3(256), 0, st, u, i, j, B, (int)world, num_user, num_item, kin,
              vin, (unsigned long long*)counts);
    DR_CHECK_LAUNCH("route_sample_keys_kernel");
    DR_HIP((rocprim::radix_sort_pairs<rocprim::default_config>(base + L.temp, tb, kin, kout, vin, order, (size_t)B, 0u,
                                                               bits_for((uint64_t)world), st, false)));
    DR_LAUNCH(route_sample_pack_kernel, dim3(grid), dim3(256), 0, st, (const int32_t*)order, u, i, j, B, (int)world, tag,
              (int4*)tuples);
    DR_CHECK_LAUNCH("route_sample_pack_kernel");
    return DR_OK;
}

extern "C" int32_t dr_route_items_i32(const int32_t* samples, int64_t n, int32_t world, int64_t rows_per_owner,
                                      int32_t part_bit, int32_t* u_loc, int32_t* inv, int32_t* uniq_rows, int32_t* meta,
                                      int32_t* order, int64_t* counts, void* ws, int64_t ws_bytes, void* stream) {
    DR_REQUIRE(n >= 0 && 2 * n < (1ll << 31) - 1 && world >= 1 && world <= MAX_WORLD && rows_per_owner >= 1 &&
                   (uint64_t)world * (uint64_t)rows_per_owner < (1ull << 32) && part_bit >= 1 && part_bit <= 30,
               "dr_route_items_i32: bad sizes");
    DR_REQUIRE(counts != nullptr, "dr_route_items_i32: counts is NULL");
    hipStream_t st = (hipStream_t)stream;
    DR_HIP(hipMemsetAsync(counts, 0, (size_t)world * 16, st));
    if (n == 0) return DR_OK;
    DR_REQUIRE(samples && u_loc && inv && uniq_rows && meta && order && ((uintptr_t)samples % 16 == 0) &&
                   ((uintptr_t)meta % 8 == 0),
               "dr_route_items_i32: NULL / misaligned pointer");
    const int64_t m = 2 * n;
    DR_ROUTE_WS(m, "dr_route_items_i32");
    int32_t* flags = (int32_t*)(base + L.flags);
    int32_t* segidx = (int32_t*)(base + L.segidx);
    const int grid = grid_for(m, 256);
    const uint32_t rpo = (uint32_t)rows_per_owner;
    DR_LAUNCH(route_item_keys_kernel, dim3(grid), dim3(256), 0, st, (const int4*)samples, n, (int)world, rpo, kin, vin,
              u_loc, (unsigned long long*)counts);
    DR_CHECK_LAUNCH("route_item_keys_kernel");
    DR_HIP((rocprim::radix_sort_pairs<rocprim::default_config>(
        base + L.temp, tb, kin, kout, vin, order, (size_t)m, 0u,
        bits_for((uint64_t)world * (uint64_t)rows_per_owner - 1), st, false)));
    DR_LAUNCH(route_heads_kernel, dim3(grid), dim3(256), 0, st, (const uint32_t*)kout, m, flags);
    DR_CHECK_LAUNCH("route_heads_kernel");
    tb = (size_t)L.temp_bytes;
    DR_HIP((rocprim::exclusive_scan<rocprim::default_config>(base + L.temp, tb, flags, segidx, (int32_t)0, (size_t)m,
                                                             rocprim::plus<int32_t>(), st, false)));
    DR_LAUNCH(route_item_pack_kernel, dim3(grid), dim3(256), 0, st, (const uint32_t*)kout, (const int32_t*)order,
              (const int32_t*)flags, (const int32_t*)segidx, (const int4*)samples, n, (int)world, rpo, (int)part_bit, inv,
              uniq_rows, (int2*)meta, (unsigned long long*)counts);
    DR_CHECK_LAUNCH("route_item_pack_kernel");
    return DR_OK;
}

extern "C" int32_t dr_canon_contributions_i32(const int32_t* meta, int64_t n, int32_t key_bits, int32_t* ids_canon,
                                              int32_t* perm, void* ws, int64_t ws_bytes, void* stream) {
    DR_REQUIRE(n >= 0 && n < (1ll << 31) - 1 && key_bits >= 1 && key_bits <= 31, "dr_canon_contributions_i32: bad sizes");
    if (n == 0) return DR_OK;
    DR_REQUIRE(meta && ids_canon && perm && ((uintptr_t)meta % 8 == 0), "dr_canon_contributions_i32: NULL / misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    DR_ROUTE_WS(n, "dr_canon_contributions_i32");
    const int grid = grid_for(n, 256);
    DR_LAUNCH(canon_keys_kernel, dim3(grid), dim3(256), 0, st, (const int2*)meta, n, kin, vin);
    DR_CHECK_LAUNCH("canon_keys_kernel");
    DR_HIP((rocprim::radix_sort_pairs<rocprim::default_config>(base + L.temp, tb, kin, kout, vin, perm, (size_t)n, 0u,
                                                               (unsigned)key_bits, st, false)));
    DR_LAUNCH(canon_ids_kernel, dim3(grid), dim3(256), 0, st, (const int2*)meta, (const int32_t*)perm, n, ids_canon);
    DR_CHECK_LAUNCH("canon_ids_kernel");
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
