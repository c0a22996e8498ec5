// Wire packing for the data-parallel sequence models (the reference is single-process and has no
// counterpart): ONE route per training step.  Every rank sees every rank's id lists (one all-gather), derives both of
// its roles from them (dr_step_route_keys_i32 + two stable bucketings), and then all tables' rows travel in ONE
// all-to-all and all gradient rows in ONE all-to-all back.  A wire buffer is a flat float array; the block of a peer
// holds one segment per (table, id list), each a run of whole rows:
//
//   pack:    flat[seg_off[s] + (e - seg_start[s]) k ..] = src[idx[e] - idx_bias]      (a row out of range packs zeros)
//   unpack:  dst[idx[e] - idx_bias] = flat[seg_off[s] + (e - seg_start[s]) k ..]      (idx NULL: row e)
//
// for entry e of segment s (seg_start ascending, seg_start[nseg] = n).  HBM-bound row copies: LPR lanes move a row as
// V-float vectors exactly like the gather kernel; the segment of an entry is found by a binary search over <= 64
// boundaries held in LDS.
#include "dr_common.h"

namespace dr {

constexpr int WIRE_MAX_SEG = DR_WIRE_MAX_SEG;
struct WireSegs {
    int nseg;
    long long start[WIRE_MAX_SEG + 1];
    long long off[WIRE_MAX_SEG];
};

template <int V, int LPR, bool PACK>
__global__ __launch_bounds__(256) void wire_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        int64_t rows, int k, const int32_t* __restrict__ idx,
                                                        int idx_bias, int64_t n, WireSegs sg) {
    __shared__ long long s_start[WIRE_MAX_SEG + 1];
    __shared__ long long s_off[WIRE_MAX_SEG];
    for (int t = threadIdx.x; t <= sg.nseg; t += 256) s_start[t] = sg.start[t];
    for (int t = threadIdx.x; t < sg.nseg; t += 256) s_off[t] = sg.off[t];
    __syncthreads();
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int cpr = k / V;
    for (int64_t e = (int64_t)blockIdx.x * RPB + sub; e < n; e += (int64_t)gridDim.x * RPB) {
        int lo = 0, hi = sg.nseg - 1;
        while (lo < hi) {                       // last segment that starts at or before e (empty ones are skipped)
            const int mid = (lo + hi + 1) >> 1;
            if (s_start[mid] <= e) lo = mid; else hi = mid - 1;
        }
        const int64_t w = s_off[lo] + (e - s_start[lo]) * k;
        const int64_t r = idx ? (int64_t)idx[e] - idx_bias : e;
        const bool ok = r >= 0 && r < rows;
        for (int c = lane; c < cpr; c += LPR) {
            float v[V];
            if constexpr (PACK) {
                if (ok) {
                    vload<V>(v, src + r * k + c * V);
                } else {
#pragma unroll
                    for (int x = 0; x < V; ++x) v[x] = 0.0f;
                }
                vstore<V>(dst + w + c * V, v);
            } else if (ok) {
                vload<V>(v, src + w + c * V);
                vstore<V>(dst + r * k + c * V, v);
            }
        }
    }
}

struct RouteLists {
    int n;
    long long start[DR_ROUTE_MAX_LISTS + 1];
};

// x = s * n_tot + p: entry p of rank s's concatenated id lists.  owner = id mod W (non-negative), local row = floor(id / W).
__global__ __launch_bounds__(256) void step_route_keys_kernel(const int32_t* __restrict__ ids_all, int W, int me,
                                                              int64_t n_tot, RouteLists ls,
                                                              int32_t* __restrict__ keys_mine,
                                                              int32_t* __restrict__ keys_owned,
                                                              int32_t* __restrict__ local_rows) {
    const int64_t total = (int64_t)W * n_tot;
    for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < total; x += (int64_t)gridDim.x * 256) {
        const int s = (int)(x / n_tot);
        const int64_t p = x - (int64_t)s * n_tot;
        int l = 0;
        while (l + 1 < ls.n && ls.start[l + 1] <= p) ++l;
        const int id = ids_all[x];
        int owner = id % W;
        if (owner < 0) owner += W;
        local_rows[x] = (id - owner) / W;
        keys_owned[x] = owner == me ? l * W + s : ls.n * W;      // not mine: the bucketing's overflow bucket
        if (s == me) keys_mine[p] = l * W + owner;
    }
}

template <bool PACK>
static int launch_wire(const float* src, float* dst, int64_t rows, int k, const int32_t* idx, int idx_bias, int64_t n,
                       const WireSegs& sg, hipStream_t st) {
    RowShape s = row_shape(k);
#define DR_W(V_, L_)                                                                                                  \
    if (s.V == V_ && s.LPR == L_) {                                                                                   \
        DR_LAUNCH_NAMED(PACK ? "wire_pack_kernel" : "wire_unpack_kernel", (wire_rows_kernel<V_, L_, PACK>),           \
                        dim3(grid_for(n, 1, 256 / L_)), dim3(256), 0, st, src, dst, rows, k, idx, idx_bias, n, sg);   \
        return 0;                                                                                                     \
    }
    DR_W(4, 1) DR_W(4, 2) DR_W(4, 4) DR_W(4, 8) DR_W(4, 16) DR_W(4, 32) DR_W(4, 64)
    DR_W(2, 1) DR_W(2, 2) DR_W(2, 4) DR_W(2, 8) DR_W(2, 16) DR_W(2, 32) DR_W(2, 64)
    DR_W(1, 1) DR_W(1, 2) DR_W(1, 4) DR_W(1, 8) DR_W(1, 16) DR_W(1, 32) DR_W(1, 64)
#undef DR_W
    return -1;
}

// Host-side check of a segment table against the sizes the kernel will index with; fills sg.
static int32_t wire_segments(const char* who, int32_t k, int64_t n, const int64_t* seg_start, const int64_t* seg_off,
                             int32_t nseg, int64_t flat_floats, WireSegs* sg) {
    DR_REQUIRE(k >= 1 && n >= 0 && flat_floats >= 0, "%s: bad sizes", who);
    DR_REQUIRE(nseg >= 1 && nseg <= WIRE_MAX_SEG, "%s: nseg must be in [1, %d]", who, WIRE_MAX_SEG);
    DR_REQUIRE(seg_start && seg_off, "%s: NULL segment table", who);
    DR_REQUIRE(seg_start[0] == 0 && seg_start[nseg] == n, "%s: segments must cover entries 0 .. n", who);
    const int V = row_shape(k).V;
    sg->nseg = nseg;
    for (int s = 0; s < nseg; ++s) {
        const int64_t cnt = seg_start[s + 1] - seg_start[s];
        DR_REQUIRE(cnt >= 0, "%s: seg_start must ascend (segment %d)", who, s);
        DR_REQUIRE(seg_off[s] >= 0 && seg_off[s] % V == 0, "%s: offset of segment %d must be a multiple of %d floats", who,
                   s, V);
        DR_REQUIRE(seg_off[s] + cnt * (int64_t)k <= flat_floats, "%s: segment %d runs past the wire buffer", who, s);
        sg->start[s] = seg_start[s];
        sg->off[s] = seg_off[s];
    }
    sg->start[nseg] = n;
    return DR_OK;
}

}  // namespace dr

extern "C" int32_t dr_rows_pack_f32(const float* src, int64_t src_rows, int32_t k, const int32_t* idx, int32_t idx_bias,
                                    int64_t n, const int64_t* seg_start, const int64_t* seg_off, int32_t nseg,
                                    float* flat, int64_t flat_floats, void* stream) {
    dr::WireSegs sg;
    int32_t rc = dr::wire_segments("dr_rows_pack_f32", k, n, seg_start, seg_off, nseg, flat_floats, &sg);
    if (rc != DR_OK) return rc;
    DR_REQUIRE(src_rows >= 0, "dr_rows_pack_f32: bad src_rows");
    if (n == 0) return DR_OK;
    DR_REQUIRE(flat && (src || src_rows == 0), "dr_rows_pack_f32: NULL pointer");
    DR_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)flat % 16 == 0), "dr_rows_pack_f32: src / flat must be 16-byte aligned");
    DR_REQUIRE(dr::launch_wire<true>(src, flat, src_rows, k, idx, idx_bias, n, sg, (hipStream_t)stream) == 0,
               "dr_rows_pack_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("dr_rows_pack_f32");
    return DR_OK;
}

extern "C" int32_t dr_rows_unpack_f32(const float* flat, int64_t flat_floats, int32_t k, const int32_t* idx,
                                      int32_t idx_bias, int64_t n, const int64_t* seg_start, const int64_t* seg_off,
                                      int32_t nseg, float* dst, int64_t dst_rows, void* stream) {
    dr::WireSegs sg;
    int32_t rc = dr::wire_segments("dr_rows_unpack_f32", k, n, seg_start, seg_off, nseg, flat_floats, &sg);
    if (rc != DR_OK) return rc;
    DR_REQUIRE(dst_rows >= 0 && (idx != nullptr || dst_rows >= n), "dr_rows_unpack_f32: dst has fewer rows than entries");
    if (n == 0) return DR_OK;
    DR_REQUIRE(flat && dst, "dr_rows_unpack_f32: NULL pointer");
    DR_REQUIRE(((uintptr_t)dst % 16 == 0) && ((uintptr_t)flat % 16 == 0), "dr_rows_unpack_f32: dst / flat must be 16-byte aligned");
    DR_REQUIRE(dr::launch_wire<false>(flat, dst, dst_rows, k, idx, idx_bias, n, sg, (hipStream_t)stream) == 0,
               "dr_rows_unpack_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("dr_rows_unpack_f32");
    return DR_OK;
}

extern "C" int32_t dr_step_route_keys_i32(const int32_t* ids_all, int32_t world, int32_t me, int64_t n_tot,
                                          const int64_t* list_start, int32_t n_lists, int32_t* keys_mine,
                                          int32_t* keys_owned, int32_t* local_rows, void* stream) {
    DR_REQUIRE(world >= 1 && me >= 0 && me < world && n_tot >= 0, "dr_step_route_keys_i32: bad world / rank / size");
    DR_REQUIRE(n_lists >= 1 && n_lists <= DR_ROUTE_MAX_LISTS && (int64_t)n_lists * world <= 256,
               "dr_step_route_keys_i32: n_lists in [1, %d] and n_lists * world <= 256", DR_ROUTE_MAX_LISTS);
    DR_REQUIRE((int64_t)world * n_tot < (1ll << 31) - 1, "dr_step_route_keys_i32: world * n_tot exceeds the 31-bit position");
    DR_REQUIRE(list_start && list_start[0] == 0 && list_start[n_lists] == n_tot,
               "dr_step_route_keys_i32: the lists must cover entries 0 .. n_tot");
    dr::RouteLists ls;
    ls.n = n_lists;
    for (int l = 0; l <= n_lists; ++l) {
        DR_REQUIRE(l == 0 || list_start[l] >= list_start[l - 1], "dr_step_route_keys_i32: list_start must ascend");
        ls.start[l] = list_start[l];
    }
    if (n_tot == 0) return DR_OK;
    DR_REQUIRE(ids_all && keys_mine && keys_owned && local_rows, "dr_step_route_keys_i32: NULL pointer");
    DR_LAUNCH_NAMED("step_route_keys_kernel", dr::step_route_keys_kernel, dim3(dr::grid_for((int64_t)world * n_tot, 256)),
                    dim3(256), 0, (hipStream_t)stream, ids_all, (int)world, (int)me, n_tot, ls, keys_mine, keys_owned,
                    local_rows);
    DR_CHECK_LAUNCH("dr_step_route_keys_i32");
    return DR_OK;
}
