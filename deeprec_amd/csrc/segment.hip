// K2 / K5: segment plan (stable sort of an id list + run boundaries), deterministic
// sorted-segment scatter-add, and the fused "segment sum -> optimizer row update" kernel.
//
// HBM-bound integer/float streaming work: no MFMA.  One sub-wave of LPR lanes owns one
// destination row (k=128: 32 lanes x float4, two rows per wavefront); contributions of a row
// are summed in blocked position order (DR_SEG_CHUNK) so results are bit-repeatable and equal
// to oracle/ref_ops.py:segment_sums.  The sort / scan of the plan use rocPRIM device primitives
// (radix_sort_pairs, exclusive_scan); everything that touches table rows is hand-written.
#include "dr_common.h"
#include "dr_lsd_sort.h"
#include "dr_segment.h"

#include <cstring>
#include <stdlib.h>
#include <rocprim/rocprim.hpp>

namespace dr {

// ---------------------------------------------------------------------------------------
// plan layout inside the caller's workspace
// ---------------------------------------------------------------------------------------
struct PlanLayout {
    int64_t hdr, keys_in, keys_out, vals_in, vals_out, flags, segidx, uniq, seg_start, temp, temp_bytes, total;
    int64_t long_seg, long_cb, part_owner, partials, partial_rows, msrc, mcoef, solo, multi, long_cnt, lsd;
};

// Segments with more than LONG_SEG contributions ("long": anything beyond the few-duplicates common case) are
// summed in two passes so that neither a popular row nor a moderately repeated one serialises the launch: seg_long_partial_kernel gives every DR_SEG_CHUNK-block of every long segment its
// own sub-wave (block partials parked in the plan's scratch), seg_long_combine_kernel adds a segment's
// partials in block order and applies the update -- the same blocked position order, in parallel.
constexpr int LONG_SEG = 4;
constexpr int PLAN_TILE = 4096;          // run boundaries: 256 threads x 16 consecutive keys per workgroup
constexpr int PLAN_MAX_TILES = 1024;     // above 4 M keys: the rocPRIM scan

// Lists up to DR_LSD_MAX_N keys are sorted by the library's own LSD sort (dr_lsd_sort.h: 2 launches per 8-bit digit);
// rocPRIM's device sort (Onesweep at that size) stays for longer ones: 4 M keys 0.28 ms against 0.34 - 0.48 ms.
// (Rounds 1 - 2 used rocPRIM throughout: a 4096-key block sort plus 6 merge launches for the 196 608 keys of a C4
// batch, 56 us of a 76 - 100 us plan.)
#ifndef DR_LSD_MAX_N
#define DR_LSD_MAX_N (1ll << 20)
#endif

static int plan_layout_compute(int64_t n, int k_max, PlanLayout* L, char* err);
// The layout is a pure function of (n, k_max), and every plan / update / size call asks for it: the two rocPRIM
// temporary-storage queries inside were a measurable part of the host time of a step that issues ~50 short launches
// (sharded step: 8 queries per step).  The last few answers are kept per host thread.
static int plan_layout(int64_t n, int k_max, PlanLayout* L, char* err) {
    struct Memo {
        int64_t n = -1;
        int k = 0;
        PlanLayout L;
    };
    static thread_local Memo memo[8];
    static thread_local int next = 0;
    for (int i = 0; i < 8; ++i)
        if (memo[i].n == n && memo[i].k == k_max) {
            *L = memo[i].L;
            return 0;
        }
    const int rc = plan_layout_compute(n, k_max, L, err);
    if (rc == 0) {
        memo[next].n = n;
        memo[next].k = k_max;
        memo[next].L = *L;
        next = (next + 1) % 8;
    }
    return rc;
}

static int plan_layout_compute(int64_t n, int k_max, PlanLayout* L, char* err) {
    size_t t_sort = 0, t_scan = 0;
    int64_t nn = n > 0 ? n : 1;
    hipError_t e = hipSuccess;
    if (nn > DR_LSD_MAX_N) {
        e = rocprim::radix_sort_pairs<rocprim::default_config, uint32_t*, uint32_t*, int32_t*, int32_t*>(
            nullptr, t_sort, nullptr, nullptr, nullptr, nullptr, (size_t)nn, 0, 32, 0, false);
        if (e != hipSuccess) {
            snprintf(err, 512, "rocprim radix_sort size query: %s", hipGetErrorString(e));
            return -1;
        }
    }
    if ((nn + PLAN_TILE - 1) / PLAN_TILE > PLAN_MAX_TILES) {
        e = rocprim::exclusive_scan<rocprim::default_config, int32_t*, int32_t*, int32_t, rocprim::plus<int32_t>>(
            nullptr, t_scan, nullptr, nullptr, 0, (size_t)nn, rocprim::plus<int32_t>(), 0, false);
        if (e != hipSuccess) {
            snprintf(err, 512, "rocprim exclusive_scan size query: %s", hipGetErrorString(e));
            return -1;
        }
    }
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    L->hdr = take(256);
    L->keys_in = take(nn * 4);
    L->keys_out = take(nn * 4);
    L->vals_in = take(nn * 4);
    L->vals_out = take(nn * 4);
    L->flags = take(nn * 4);
    L->segidx = take(nn * 4);
    L->uniq = take(nn * 4);
    L->seg_start = take((nn + 1) * 4);
    L->temp_bytes = (int64_t)(t_sort > t_scan ? t_sort : t_scan) + 256;
    L->temp = take(L->temp_bytes);
    const int64_t max_long = nn / LONG_SEG + 1;
    L->long_seg = take(max_long * 4);
    L->long_cb = take(max_long * 4);
    L->partial_rows = nn / DR_SEG_CHUNK + max_long + 1;
    L->part_owner = take(L->partial_rows * 4);
    L->msrc = take(nn * 4);
    L->mcoef = take(nn * 4);
    L->solo = take(nn);  // one byte per ORIGINAL position: its segment has exactly one contribution (BPR plans)
    L->multi = take((nn + 1) * 16);  // int4 {segment, key, first sorted position, length} of every segment with
                                     // 2 .. LONG_SEG contributions (sharded step: also a single one that arrives from a
                                     // peer), arbitrary order, count in hdr[4]
    L->long_cnt = take(max_long * 4);    // blocks of a long segment summed so far (bpr_tail_kernel), zero between uses
    L->lsd = take(lsd_scratch_bytes(nn));  // tile histograms (+ column prefixes) of the plan's LSD sort
    L->partials = take(L->partial_rows * (int64_t)(k_max > 0 ? k_max : 1) * 4);
    L->total = off;
    return 0;
}

// key / position lists of a key source, materialised (only the rocPRIM path above LSD_MAX_N keys needs them)
template <class Src>
__global__ __launch_bounds__(256) void plan_keys_kernel(Src src, int64_t n, uint32_t* __restrict__ keys,
                                                        int32_t* __restrict__ vals, int32_t* __restrict__ hdr) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr[1] = 0;  // number of long segments
        hdr[2] = 0;  // partial rows handed out
        hdr[4] = 0;  // number of segments with 2 .. LONG_SEG contributions
    }
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        keys[e] = src.key(e);
        vals[e] = src.val(e);
    }
}

__global__ __launch_bounds__(256) void plan_heads_kernel(const uint32_t* __restrict__ sk, int64_t n,
                                                         int32_t* __restrict__ flags) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        flags[e] = (e == 0 || sk[e] != sk[e - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void plan_finalize_kernel(const uint32_t* __restrict__ sk,
                                                            const int32_t* __restrict__ flags,
                                                            const int32_t* __restrict__ segidx, int64_t n,
                                                            int32_t* __restrict__ uniq, int32_t* __restrict__ seg_start,
                                                            int32_t* __restrict__ hdr, uint32_t split_key) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        if (flags[e]) {
            uniq[segidx[e]] = (int32_t)sk[e];
            seg_start[segidx[e]] = (int32_t)e;
            // hdr[3]: first segment whose key is >= split_key (two tables sorted as one list)
            if (sk[e] >= split_key && (e == 0 || sk[e - 1] < split_key)) hdr[3] = segidx[e];
        }
        if (e == n - 1) {
            int32_t ns = segidx[e] + flags[e];
            hdr[0] = ns;
            seg_start[ns] = (int32_t)n;
            if (sk[e] < split_key) hdr[3] = ns;
        }
    }
}

// ---- run boundaries of the sorted keys in TWO launches (heads + rocPRIM scan [2 - 3 launches] + finalize were five) ----
// plan_count_kernel: heads per tile of PLAN_TILE keys; plan_segments_kernel: every tile adds the counts of the tiles
// before it (at most PLAN_MAX_TILES ints), scans its own keys in order and writes uniq / seg_start / hdr -- the same
// values as plan_heads_kernel + exclusive_scan + plan_finalize_kernel.  No workgroup waits for another one.

__global__ __launch_bounds__(256) void plan_count_kernel(const uint32_t* __restrict__ sk, int64_t n,
                                                         int32_t* __restrict__ tile_heads) {
    __shared__ int32_t red[4];
    const int64_t base = (int64_t)blockIdx.x * PLAN_TILE;
    int c = 0;
#pragma unroll
    for (int q = 0; q < PLAN_TILE / 256; ++q) {
        const int64_t e = base + q * 256 + threadIdx.x;
        if (e < n) c += (e == 0 || sk[e] != sk[e - 1]) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_heads[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void plan_segments_kernel(const uint32_t* __restrict__ sk, int64_t n,
                                                            const int32_t* __restrict__ tile_heads,
                                                            int32_t* __restrict__ uniq, int32_t* __restrict__ seg_start,
                                                            int32_t* __restrict__ hdr, uint32_t split_key) {
    __shared__ int32_t red[4];
    __shared__ int32_t wtot[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // segments that start in earlier tiles
    int before = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256) before += tile_heads[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    if (lane == 0) red[wave] = before;
    // this thread's 16 consecutive keys (and the key before them)
    constexpr int PER = PLAN_TILE / 256;
    const int64_t e0 = (int64_t)blockIdx.x * PLAN_TILE + (int64_t)tid * PER;
    uint32_t key[PER + 1];
    key[0] = (e0 > 0 && e0 - 1 < n) ? sk[e0 - 1] : 0u;
#pragma unroll
    for (int q = 0; q < PER; ++q) key[q + 1] = (e0 + q < n) ? sk[e0 + q] : 0u;
    int mine = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (e0 + q < n && (e0 + q == 0 || key[q + 1] != key[q])) ++mine;
    // exclusive scan of `mine` over the 256 threads
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int seg = red[0] + red[1] + red[2] + red[3] + incl - mine;
    for (int w = 0; w < wave; ++w) seg += wtot[w];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int64_t e = e0 + q;
        if (e >= n) break;
        const bool head = (e == 0 || key[q + 1] != key[q]);
        if (head) {
            uniq[seg] = (int32_t)key[q + 1];
            seg_start[seg] = (int32_t)e;
            // hdr[3]: first segment whose key is >= split_key (two tables sorted as one list)
            if (key[q + 1] >= split_key && (e == 0 || key[q] < split_key)) hdr[3] = seg;
            ++seg;
        }
        if (e == n - 1) {      // seg = number of segments now
            hdr[0] = seg;
            seg_start[seg] = (int32_t)n;
            if (key[q + 1] < split_key) hdr[3] = seg;
        }
    }
}

// source rows of the sorted contributions of a BPR batch: position p < B reads stash row p (P side), position
// B + x reads the user row u[x mod B] (Q side: g_b * p_u).
// fused (the planned step in a sparse regulariser mode): rows whose segment holds ONE contribution are updated by
// the forward kernel itself.  solo[p] marks those positions; a Q-side contribution must then not read P in place (the
// forward kernel may already have updated the row) but the pre-step copy the forward kernel leaves in stash row
// B + x: source index nu + B + x (SegArgs.src2).
__device__ __forceinline__ int32_t bpr_msrc(int p, const int32_t* __restrict__ u, int64_t B, int64_t nu, int fused) {
    if (p < B) return p;
    const int64_t x = (p - B < B) ? p - B : p - 2 * B;
    const int64_t id = u[x];
    if (id < 0 || id >= nu) return -1;
    return fused ? (int32_t)(nu + B + x) : (int32_t)id;
}

__global__ __launch_bounds__(256) void bpr_meta_kernel(const int32_t* __restrict__ sorted_pos,
                                                       const uint32_t* __restrict__ sk,
                                                       const int32_t* __restrict__ u, int64_t B, int64_t nu, int64_t ni,
                                                       int fused, int32_t* __restrict__ msrc,
                                                       unsigned char* __restrict__ solo) {
    const int64_t n = 3 * B;
    const uint32_t sentinel = (uint32_t)(nu + ni);
    for (int64_t y = (int64_t)blockIdx.x * 256 + threadIdx.x; y < n; y += (int64_t)gridDim.x * 256) {
        const int p = sorted_pos[y];
        if (fused) {
            const uint32_t key = sk[y];
            solo[p] = (key != sentinel && (y == 0 || sk[y - 1] != key) && (y == n - 1 || sk[y + 1] != key)) ? 1 : 0;
        }
        msrc[y] = bpr_msrc(p, u, B, nu, fused);
    }
}

// ---- everything the fused BPR step needs from the sorted keys, in ONE launch --------------------------------------
// (was: run heads per tile, segment table, long / multi lists, source rows, solo flags, solo fix-up = six launches.)
// The fused step never walks a segment table: the forward kernel wants a solo flag per position, the tail kernel a
// descriptor per row with 2 .. LONG_SEG contributions and a block list per longer row.  All of that follows from a
// key's neighbours in the sorted list, so no global scan is needed: the "segment index" of a row is simply the sorted
// position of its first contribution (uniq / seg_start are written at that index for the long rows only, and hdr[3],
// the first item-side segment, is the number of user-side keys).  A workgroup owns 1024 consecutive sorted positions
// (keys in LDS); a long run that leaves the workgroup's range finds its end by a galloping search in memory.
constexpr int FIN_TILE = 1024;
// What differs between the single-GPU step and the row-sharded one (shard.hip) is where a contribution's source row
// lives and who applies a row with ONE contribution:
struct FinBpr {   // positions [0, B): user rows (stash row p), [B, 3B): item rows of sample x = (p - B) mod B
    const int32_t* u;
    int64_t B, nu, ni;
    unsigned char* solo;
    __device__ __forceinline__ int64_t length(int64_t n) const { return n; }
    __device__ __forceinline__ uint32_t split() const { return (uint32_t)nu; }
    __device__ __forceinline__ uint32_t sentinel() const { return (uint32_t)(nu + ni); }
    __device__ __forceinline__ int32_t msrc(int p) const { return bpr_msrc(p, u, B, nu, 1); }
    __device__ __forceinline__ bool local(int) const { return true; }
    __device__ __forceinline__ void solo_store(int p, bool single) const { solo[p] = single ? 1 : 0; }
};
// Row-sharded step: the list is [u_loc (n_s entries) ; nuL + local item row (n_g entries, canonical order)], lengths on
// the device.  osrc[p] of an item entry: 2 s + slot >= 0 when this rank processes the sample (the contribution is
// (+-g) * the pre-step user row in stash row cap_s + s), -1 - y when the gradient row arrives from a peer (row y of the
// receive buffer, source index 2 cap_s + y).  A single contribution that arrives from a peer is the tail's, not the
// forward kernel's.
struct FinShard {
    const int32_t* n_dev;       // [0] n_s, [1] n_s + n_g
    const int32_t* osrc;
    int64_t cap_s, nuL, niL;
    unsigned char* solo;        // [3][cap_s]: user, item i, item j of local sample s
    __device__ __forceinline__ int64_t length(int64_t) const { return n_dev[1]; }
    __device__ __forceinline__ uint32_t split() const { return (uint32_t)nuL; }
    __device__ __forceinline__ uint32_t sentinel() const { return 0xFFFFFFFEu; }   // the sharded route refuses invalid ids
    __device__ __forceinline__ int32_t msrc(int p) const {
        if (p < n_dev[0]) return p;
        const int o = osrc[p];
        return o >= 0 ? (int32_t)(cap_s + (o >> 1)) : (int32_t)(2 * cap_s + (-1 - o));
    }
    __device__ __forceinline__ bool local(int p) const { return p < n_dev[0] || osrc[p] >= 0; }
    __device__ __forceinline__ void solo_store(int p, bool single) const {
        if (p < n_dev[0]) {
            solo[p] = single ? 1 : 0;
        } else {
            const int o = osrc[p];
            if (o >= 0) solo[(int64_t)(1 + (o & 1)) * cap_s + (o >> 1)] = single ? 1 : 0;
        }
    }
};

template <class Fin>
__global__ __launch_bounds__(256) void bpr_finish_kernel(const uint32_t* __restrict__ sk, const int32_t* __restrict__ spos,
                                                         int64_t n, Fin fin, int32_t* __restrict__ hdr,
                                                         int32_t* __restrict__ uniq,
                                                         int32_t* __restrict__ seg_start, int32_t* __restrict__ long_seg,
                                                         int32_t* __restrict__ long_cb, int32_t* __restrict__ long_cnt,
                                                         int32_t* __restrict__ part_owner, int4* __restrict__ multi,
                                                         int32_t* __restrict__ msrc) {
    n = fin.length(n);
    __shared__ uint32_t lk[FIN_TILE];
    __shared__ int wcnt[4];
    __shared__ int blk_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * FIN_TILE;
    if (base >= n) return;   // block-uniform (the launch may be sized for a capacity beyond the list)
    const int64_t end = (base + FIN_TILE < n) ? base + FIN_TILE : n;
    constexpr uint32_t NONE = 0xFFFFFFFFu;  // never a key (keys <= nu + ni < 2^32 - 2)
#pragma unroll
    for (int q = 0; q < FIN_TILE / 256; ++q) {
        const int64_t e = base + q * 256 + tid;
        lk[q * 256 + tid] = e < n ? sk[e] : NONE;
    }
    __syncthreads();
    auto keyat = [&](int64_t y) -> uint32_t {
        if (y >= base && y < base + FIN_TILE) return lk[y - base];
        return (y >= 0 && y < n) ? sk[y] : NONE;
    };
    const uint32_t sentinel = fin.sentinel(), split = fin.split();
    // this thread's four consecutive sorted positions
    int n_multi = 0, m_len[4] = {0, 0, 0, 0};
    int64_t m_y[4] = {0, 0, 0, 0};
    uint32_t m_key[4] = {0u, 0u, 0u, 0u};
    int64_t l_y = -1, l_end = 0;
    uint32_t l_key = 0u;
    const int64_t y0 = base + (int64_t)tid * 4;
    uint32_t prev = keyat(y0 - 1);
    int4 ms = make_int4(-1, -1, -1, -1);
    int32_t pp[4] = {0, 0, 0, 0};
    if (y0 + 3 < n) {
        const int4 t = *reinterpret_cast<const int4*>(spos + y0);
        pp[0] = t.x; pp[1] = t.y; pp[2] = t.z; pp[3] = t.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (y0 + q < n) pp[q] = spos[y0 + q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t y = y0 + q;
        if (y >= n) break;
        const uint32_t key = lk[tid * 4 + q];
        const uint32_t next = keyat(y + 1);
        const bool head = prev != key, last = next != key;
        const int p = pp[q];
        const bool single = key != sentinel && head && last;
        const bool mine = fin.local(p);   // a single contribution from a peer is the tail's: a one-entry descriptor
        fin.solo_store(p, single && mine);
        (&ms.x)[q] = fin.msrc(p);
        if (single && !mine) {
            m_y[n_multi] = y;
            m_key[n_multi] = key;
            m_len[n_multi] = 1;
            ++n_multi;
        }
        // hdr[3] = number of user-side keys = index of the first item-side segment (prev = NONE at y = 0)
        if (key >= split && (prev < split || y == 0)) hdr[3] = (int32_t)y;
        if (y == n - 1 && key < split) hdr[3] = (int32_t)n;
        if (head && !last && key != sentinel) {
            int len = 2;
            if (keyat(y + 2) == key) {
                len = 3;
                if (keyat(y + 3) == key) {
                    len = 4;
                    if (keyat(y + 4) == key) len = 5;
                }
            }
            if (len <= LONG_SEG) {
                m_y[n_multi] = y;
                m_key[n_multi] = key;
                m_len[n_multi] = len;
                ++n_multi;
            } else {
                // the run's end: inside this workgroup's range (binary search in LDS) or beyond it (gallop in memory)
                int64_t lo, hi;
                if (keyat(end - 1) != key) {
                    lo = y + 5;
                    hi = end - 1;
                } else {
                    lo = end;
                    int64_t step = 32;
                    while (lo + step <= n && sk[lo + step - 1] == key) {
                        lo += step;
                        step *= 2;
                    }
                    hi = (lo + step < n) ? lo + step : n;
                }
                while (lo < hi) {   // first index in [lo, hi] whose key differs (keys are sorted: it is larger)
                    const int64_t mid = (lo + hi) >> 1;
                    if (keyat(mid) == key) lo = mid + 1;
                    else hi = mid;
                }
                l_y = y;
                l_end = lo;
                l_key = key;
            }
        }
        prev = key;
    }
    if (y0 + 3 < n) {
        *reinterpret_cast<int4*>(msrc + y0) = ms;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (y0 + q < n) msrc[y0 + q] = (&ms.x)[q];
    }
    // descriptors of the rows with 2 .. LONG_SEG contributions: one reservation per workgroup
    int incl = n_multi;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wcnt[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        const int t = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        blk_base = t ? atomicAdd(&hdr[4], t) : 0;
    }
    __syncthreads();
    int slot = blk_base + incl - n_multi;
    for (int w = 0; w < wave; ++w) slot += wcnt[w];
    for (int x = 0; x < n_multi; ++x)
        multi[slot + x] = make_int4((int)m_y[x], (int)m_key[x], (int)m_y[x], m_len[x]);
    // long rows: a block list each; one reservation pair per wavefront that has any
    const bool is_long = l_y >= 0;
    const unsigned long long ml = __ballot(is_long);
    if (ml) {
        const int len = is_long ? (int)(l_end - l_y) : 0;
        const int nch = is_long ? (len + DR_SEG_CHUNK - 1) / DR_SEG_CHUNK : 0;
        int inc2 = nch;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc2, o, 64);
            if (lane >= o) inc2 += v;
        }
        const int total = __shfl(inc2, 63, 64);
        int l0 = 0, c0 = 0;
        if (lane == 0) {
            l0 = atomicAdd(&hdr[1], __popcll(ml));
            c0 = atomicAdd(&hdr[2], total);
        }
        l0 = __shfl(l0, 0, 64);
        c0 = __shfl(c0, 0, 64);
        if (is_long) {
            const int l = l0 + __popcll(ml & ((1ull << lane) - 1ull));
            const int cb = c0 + inc2 - nch;
            long_seg[l] = (int32_t)l_y;          // "segment index" = sorted position of the first contribution
            long_cb[l] = cb;
            long_cnt[l] = 0;
            uniq[l_y] = (int32_t)l_key;
            seg_start[l_y] = (int32_t)l_y;
            seg_start[l_y + 1] = (int32_t)l_end; // l_y + 1 lies inside the run: never another row's index
            for (int c = 0; c < nch; ++c) part_owner[cb + c] = l;
        }
    }
}

__global__ void plan_empty_kernel(int32_t* hdr, int32_t* seg_start) {
    hdr[0] = 0;
    hdr[1] = 0;
    hdr[2] = 0;
    hdr[3] = 0;
    hdr[4] = 0;
    seg_start[0] = 0;
}

// list the long segments and hand each a contiguous range of partial rows (allocation order is
// arbitrary; it never influences a result)
__global__ __launch_bounds__(256) void plan_long_kernel(const int32_t* __restrict__ seg_start,
                                                        const int32_t* __restrict__ uniq, int32_t* __restrict__ hdr,
                                                        int32_t* __restrict__ long_seg, int32_t* __restrict__ long_cb,
                                                        int32_t* __restrict__ part_owner, int4* __restrict__ multi,
                                                        int32_t* __restrict__ long_cnt, int64_t n) {
    // The three counters live in one cache line and device-scope atomics on one line are served one at a time by the
    // memory side of this multi-XCD part (~8 ns each: 20 k per-row atomics made this kernel 22 us).  So: ONE atomic per
    // workgroup and pass for the (many) rows with 2 .. LONG_SEG contributions, one per wavefront for the (few) long rows.
    __shared__ int wcnt[4], wbase[4];
    __shared__ int blk_base;
    const int nseg = hdr[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a workgroup owns a CONTIGUOUS range of 256-segment passes: with many passes per workgroup (4 M ids: 13 k passes)
    // it first counts its rows with 2 .. LONG_SEG contributions and reserves their descriptors with ONE atomic (one per
    // pass made this kernel 173 us of the 400 us plan at 4 M ids), then writes them in a second walk
    const int64_t passes = ((int64_t)nseg + 255) / 256;
    const int64_t ppb = (passes + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * ppb, p1 = (p0 + ppb < passes) ? p0 + ppb : passes;
    int run = 0;                        // next free descriptor of this workgroup (block-uniform)
    if (ppb > 1) {
        int mine = 0;
        for (int64_t ps = p0; ps < p1; ++ps) {
            const int64_t s = ps * 256 + threadIdx.x;
            const int len = s < nseg ? seg_start[s + 1] - seg_start[s] : 0;
            mine += (len >= 2 && len <= LONG_SEG) ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
        if (lane == 0) wcnt[wave] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            blk_base = t ? atomicAdd(&hdr[4], t) : 0;
        }
        __syncthreads();
        run = blk_base;
        __syncthreads();
    }
    for (int64_t ps = p0; ps < p1; ++ps) {   // block-uniform trip count
        const int64_t s = ps * 256 + threadIdx.x;
        const int first = s < nseg ? seg_start[s] : 0;
        const int len = s < nseg ? seg_start[s + 1] - first : 0;
        // long rows: count and block total of this wavefront, one atomic each
        const bool is_long = len > LONG_SEG;
        const unsigned long long ml = __ballot(is_long);
        if (ml) {
            const int nch = is_long ? (len + DR_SEG_CHUNK - 1) / DR_SEG_CHUNK : 0;
            int incl = nch;                                   // inclusive prefix sum of nch over the wavefront
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            const int total = __shfl(incl, 63, 64);
            int l0 = 0, c0 = 0;
            if (lane == 0) {
                l0 = atomicAdd(&hdr[1], __popcll(ml));
                c0 = atomicAdd(&hdr[2], total);
            }
            l0 = __shfl(l0, 0, 64);
            c0 = __shfl(c0, 0, 64);
            if (is_long) {
                const int l = l0 + __popcll(ml & ((1ull << lane) - 1ull));
                const int cb = c0 + incl - nch;
                long_seg[l] = (int32_t)s;
                long_cb[l] = cb;
                long_cnt[l] = 0;
                for (int c = 0; c < nch; ++c) part_owner[cb + c] = l;
            }
        }
        const bool is_multi = len >= 2 && len <= LONG_SEG;
        const unsigned long long m = __ballot(is_multi);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            int base = (ppb > 1) ? run : (t ? atomicAdd(&hdr[4], t) : 0);
            for (int w = 0; w < 4; ++w) {
                wbase[w] = base;
                base += wcnt[w];
            }
        }
        __syncthreads();
        // one 16-byte descriptor per row: the tail kernel then needs ONE load where it walked multi -> uniq / seg_start
        if (is_multi) multi[wbase[wave] + __popcll(m & ((1ull << lane) - 1ull))] = make_int4((int)s, uniq[s], first, len);
        run += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
}

// Whole plan tail in ONE workgroup for small lists (the reference's batch sizes: B = 128 .. 1024, i.e. a few hundred
// to a few thousand keys): block radix sort of (key, position), run heads, their scan, segment table, split index
// and the long-segment list.  The general path is ~15 launches of 3-8 us each -- for 3072 keys all of it latency.
// Same outputs as plan_finish (sorted positions, uniq, seg_start, hdr, long lists); the sort is stable, so they
// are identical to the general path's.
constexpr int SMALL_IPT = 4, SMALL_N = 1024 * SMALL_IPT;
template <int SMALL_T, class Src>  // 256 threads serve lists of <= 1024 keys (fewer, cheaper barriers), 1024 threads <= 4096
__global__ __launch_bounds__(SMALL_T) void plan_small_kernel(Src src, int n,
                                                              unsigned end_bit, uint32_t split_key,
                                                              uint32_t* __restrict__ keys_out,
                                                              int32_t* __restrict__ vals_out,
                                                              int32_t* __restrict__ uniq, int32_t* __restrict__ seg_start,
                                                              int32_t* __restrict__ hdr, int32_t* __restrict__ long_seg,
                                                              int32_t* __restrict__ long_cb,
                                                              int32_t* __restrict__ part_owner,
                                                              int4* __restrict__ multi,
                                                              int32_t* __restrict__ long_cnt) {
    using Sort = rocprim::block_radix_sort<uint32_t, SMALL_T, SMALL_IPT, int32_t>;
    using Scan = rocprim::block_scan<int, SMALL_T>;
    __shared__ union {
        typename Sort::storage_type sort;
        typename Scan::storage_type scan;
    } tmp;
    __shared__ uint32_t sk[SMALL_T * SMALL_IPT];
    __shared__ int sh_split, sh_nlong, sh_parts, sh_multi;
    const int tid = threadIdx.x;
    // padding keys carry the largest value the sorted bits can hold and positions >= n: the sort is stable, so
    // they end up behind every real entry
    const uint32_t pad = end_bit >= 32 ? 0xFFFFFFFFu : ((1u << end_bit) - 1u);
    uint32_t key[SMALL_IPT];
    int32_t val[SMALL_IPT];
#pragma unroll
    for (int q = 0; q < SMALL_IPT; ++q) {
        const int e = tid * SMALL_IPT + q;
        key[q] = e < n ? src.key(e) : pad;
        val[q] = e;
    }
    Sort().sort(key, val, tmp.sort, 0u, end_bit);
#pragma unroll
    for (int q = 0; q < SMALL_IPT; ++q) {
        const int e = tid * SMALL_IPT + q;
        sk[e] = key[q];
        if (e < n) {
            vals_out[e] = val[q];
            keys_out[e] = key[q];
        }
    }
    if (tid == 0) {
        sh_nlong = 0;
        sh_parts = 0;
        sh_multi = 0;
    }
    __syncthreads();
    int flag[SMALL_IPT], cnt = 0;
#pragma unroll
    for (int q = 0; q < SMALL_IPT; ++q) {
        const int e = tid * SMALL_IPT + q;
        flag[q] = (e < n && (e == 0 || sk[e] != sk[e - 1])) ? 1 : 0;
        cnt += flag[q];
    }
    int base = 0, nseg = 0;
    Scan().exclusive_scan(cnt, base, 0, nseg, tmp.scan);
    if (tid == 0) sh_split = nseg;  // no segment at / above the split unless one is found below
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SMALL_IPT; ++q) {
        const int e = tid * SMALL_IPT + q;
        if (flag[q]) {
            uniq[base] = (int32_t)sk[e];
            seg_start[base] = e;
            if (sk[e] >= split_key && (e == 0 || sk[e - 1] < split_key)) sh_split = base;
            ++base;
        }
    }
    if (tid == 0) seg_start[nseg] = n;
    __threadfence_block();
    __syncthreads();
    for (int s = tid; s < nseg; s += SMALL_T) {
        const int len = seg_start[s + 1] - seg_start[s];
        if (len > LONG_SEG) {
            const int l = atomicAdd(&sh_nlong, 1);
            const int nch = (len + DR_SEG_CHUNK - 1) / DR_SEG_CHUNK;
            const int cb = atomicAdd(&sh_parts, nch);
            long_seg[l] = s;
            long_cb[l] = cb;
            long_cnt[l] = 0;
            for (int c = 0; c < nch; ++c) part_owner[cb + c] = l;
        } else if (len >= 2) {
            multi[atomicAdd(&sh_multi, 1)] = make_int4(s, uniq[s], seg_start[s], len);
        }
    }
    __syncthreads();
    if (tid == 0) {
        hdr[0] = nseg;
        hdr[1] = sh_nlong;
        hdr[2] = sh_parts;
        hdr[3] = sh_split;
        hdr[4] = sh_multi;
    }
}

// ---------------------------------------------------------------------------------------
// segment sum (+ optimizer application)
// ---------------------------------------------------------------------------------------
// helper stream (dr_init()'s, dr_common.h: helpers()) for the long-segment passes; returns `main` when there is none
hipStream_t seg_side_stream(hipStream_t main, hipEvent_t* fork, hipEvent_t* join) {
    static const bool off = DR_EXP_ENV("DR_NO_OVERLAP") != nullptr;
    static const bool always = DR_EXP_ENV("DR_OVERLAP") != nullptr;
    Helpers* h = off ? nullptr : helpers();
    if (!h) return main;
    if (!always) {
        // Launched eagerly the fork / join costs four event calls (~45 us of host time per fused step, measured)
        // and buys no GPU time; as a parallel branch of a captured graph it is free on the host and worth ~5 %.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(main, &cs) != hipSuccess || cs != hipStreamCaptureStatusActive) return main;
    }
    if (hipEventRecord(h->seg_fork, main) != hipSuccess || hipStreamWaitEvent(h->seg, h->seg_fork, 0) != hipSuccess) return main;
    *fork = h->seg_fork;
    *join = h->seg_join;
    return h->seg;
}

// per-contribution metadata in SORTED order, so that the segment kernels read it with plain sequential loads
// instead of a position -> index -> row chain of dependent loads
__global__ __launch_bounds__(256) void seg_meta_kernel(const int32_t* __restrict__ sorted_pos,
                                                       const int32_t* __restrict__ src_idx, int64_t n,
                                                       int64_t src_rows, int32_t* __restrict__ msrc) {
    for (int64_t y = (int64_t)blockIdx.x * 256 + threadIdx.x; y < n; y += (int64_t)gridDim.x * 256) {
        const int p = sorted_pos[y];
        const int64_t si = src_idx ? (int64_t)src_idx[p] : (int64_t)p;
        msrc[y] = (si >= 0 && si < src_rows) ? (int32_t)si : -1;
    }
}

// sum of contributions [y0, y1) (one DR_SEG_CHUNK block) for vector column c, sequential in y;
// the row loads of a block are independent of the running sum, so they are issued UNR at a time (the ragged
// end of a block is predicated, not serialised: a long-segment pass is a chain of dependent memory round trips
// and nothing else)
template <int V, int UNR>
__device__ __forceinline__ void block_sum(const SegArgs& a, int y0, int y1, int c, float (&acc)[V]) {
#pragma unroll
    for (int x = 0; x < V; ++x) acc[x] = 0.0f;
    for (int y = y0; y < y1; y += UNR) {
        const int n = y1 - y;  // >= 1; entries q >= n do not exist
        float v[UNR][V];
        float cf[UNR];
        int si[UNR], sp[UNR];
        // Index loads are unconditional (entries past the end re-read the last one and are dropped by a select) and go
        // level by level: written per entry as `if (q < n) { si = msrc[..]; cf = coef[spos[..]]; }` the compiler waited
        // (vmcnt(0)) inside every entry for its position before it could request the coefficient -- 16 dependent round
        // trips per pass where three levels suffice: {msrc, spos} -> coef -> rows.
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int yy = (q < n) ? y + q : y1 - 1;
            si[q] = a.msrc[yy];
            sp[q] = a.coef ? a.spos[yy] : 0;
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) cf[q] = a.coef ? a.coef[sp[q] - a.pos_lo] : 1.0f;
#pragma unroll
        for (int q = 0; q < UNR; ++q) si[q] = (q < n) ? si[q] : -1;
#pragma unroll
        for (int q = 0; q < UNR; ++q)
            if (si[q] >= 0) vload<V>(v[q], seg_src_row(a, si[q]) + c * V);
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            if (si[q] >= 0) {
#pragma unroll
                for (int x = 0; x < V; ++x) acc[x] = acc[x] + (a.coef ? cf[q] * v[q][x] : v[q][x]);
            }
        }
    }
}

// short segments (<= LONG_SEG contributions, the common case): one sub-wave per segment
// PART (compile-time copy of a.part) also gives the two launches of a combined two-table plan distinct kernel
// names in profiler output: <.., 2> is the item-table side of the fused BPR step, <.., 1> the user-table side.
template <int V, int LPR, int PART, int KIND>
__global__ __launch_bounds__(256) void seg_update_kernel(SegArgs a) {
    a.opt = resolve(a.opt);
    constexpr int SPB = 256 / LPR;  // segments per block pass
    const int sub = threadIdx.x / LPR;
    const int lane = threadIdx.x % LPR;
    const int cpr = a.k / V;
    const int s_lo = (PART == 2) ? a.hdr[3] : 0;
    const int s_hi = (PART == 1) ? a.hdr[3] : a.hdr[0];
    // Two-stage software pipeline over the segments of this sub-wave: a segment's rows are three dependent loads
    // away from its index (segment table -> contribution list -> rows), so while segment i moves its rows, the
    // contribution list of segment i+1 and the table entry of segment i+2 are already in flight.
    const int stride = gridDim.x * SPB;
    struct Seg {
        uint32_t r;
        int b, len;
        bool go;
    };
    // Every load of the pipeline is UNCONDITIONAL (an index past the end re-reads a valid entry, the value is dropped
    // by a select): written as `x = 0; if (ok) x = load(..)` each load was followed by its own s_waitcnt vmcnt(0), which
    // also drained the row loads of the segment in flight -- the three stages ran one after the other.
    if (s_hi <= s_lo) return;
    auto stage_a = [&](int s) {
        Seg g;
        g.go = s < s_hi;
        const int sc = g.go ? s : s_hi - 1;
        const uint32_t r = (uint32_t)a.uniq[sc] - a.id_lo;
        const int b = a.seg_start[sc], e = a.seg_start[sc + 1];
        g.r = g.go ? r : 0u;
        g.b = g.go ? b : 0;
        g.len = g.go ? e - b : 0;
        return g;
    };
    // sentinel segment of out-of-range ids / the other table's rows; long segments are seg_long_kernel's
    auto live = [&](const Seg& g) {
        return g.go && (int64_t)g.r < a.rows && g.len <= LONG_SEG && !(a.skip_single && g.len == 1);
    };
    const int32_t* sposp = a.coef ? a.spos : a.msrc;   // without coefficients the positions are never used
    auto load_idx = [&](const Seg& g, bool lv, int (&si)[4], int (&sp)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = g.b + ((lv && q < g.len) ? q : 0);   // g.b is a valid entry for every segment (and 0 for none)
            si[q] = a.msrc[yy];
            sp[q] = sposp[yy];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            si[q] = (lv && q < g.len) ? si[q] : -1;
            sp[q] = (lv && q < g.len) ? sp[q] : 0;
        }
    };
    int s = s_lo + blockIdx.x * SPB + sub;
    Seg g0 = stage_a(s), g1 = stage_a(s + stride);
    int si0[4], sp0[4];
    load_idx(g0, live(g0), si0, sp0);
    while (g0.go) {
        const Seg g2 = stage_a(s + 2 * stride);
        int si1[4], sp1[4];
        load_idx(g1, live(g1), si1, sp1);
        if (live(g0)) {
            // the <= 4 contributions are resolved; every vector column issues its coefficient, source-row and
            // destination-row loads together (blocked order: a single block)
            float cf[4];
            if (a.coef) {
#pragma unroll
                for (int q = 0; q < 4; ++q) cf[q] = a.coef[si0[q] >= 0 ? sp0[q] - a.pos_lo : 0];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) cf[q] = 1.0f;
            }
            for (int c = lane; c < cpr; c += LPR) {
                float v[4][V], tot[V];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (si0[q] >= 0) vload<V>(v[q], seg_src_row(a, si0[q]) + c * V);
#pragma unroll
                for (int x = 0; x < V; ++x) tot[x] = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (si0[q] >= 0) {
#pragma unroll
                        for (int x = 0; x < V; ++x) tot[x] = tot[x] + (a.coef ? cf[q] * v[q][x] : v[q][x]);
                    }
#pragma unroll
                for (int x = 0; x < V; ++x) tot[x] = 0.0f + tot[x];
                apply_row<V, KIND>(a, g0.r, c, tot);
            }
        }
        g0 = g1;
        g1 = g2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            si0[q] = si1[q];
            sp0[q] = sp1[q];
        }
        s += stride;
    }
}

// long segments, pass 1: one sub-wave per DR_SEG_CHUNK-block of a long segment
template <int V, int LPR>
__global__ __launch_bounds__(256) void seg_long_partial_kernel(SegArgs a) {
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR;
    const int lane = threadIdx.x % LPR;
    const int cpr = a.k / V;
    const int nparts = a.hdr[2];
    for (int pr = blockIdx.x * SPB + sub; pr < nparts; pr += gridDim.x * SPB) {
        const int l = a.part_owner[pr];
        const int s = a.long_seg[l];
        if ((int64_t)((uint32_t)a.uniq[s] - a.id_lo) >= a.rows) continue;  // not this table's segment
        const int ch = pr - a.long_cb[l];
        const int b = a.seg_start[s], e = a.seg_start[s + 1];
        const int c0 = b + ch * DR_SEG_CHUNK;
        const int c1 = (c0 + DR_SEG_CHUNK < e) ? c0 + DR_SEG_CHUNK : e;
        for (int c = lane; c < cpr; c += LPR) {
            float acc[V];
            block_sum<V, 16>(a, c0, c1, c, acc);
            vstore<V>(a.partials + (int64_t)pr * a.k + c * V, acc);
        }
    }
}

// long segments, pass 2: one sub-wave per long segment adds its block partials in block order
template <int V, int LPR>
__global__ __launch_bounds__(256) void seg_long_combine_kernel(SegArgs a) {
    a.opt = resolve(a.opt);
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR;
    const int lane = threadIdx.x % LPR;
    const int cpr = a.k / V;
    const int nlong = a.hdr[1];
    for (int l = blockIdx.x * SPB + sub; l < nlong; l += gridDim.x * SPB) {
        const int s = a.long_seg[l];
        const uint32_t r = (uint32_t)a.uniq[s] - a.id_lo;
        if ((int64_t)r >= a.rows) continue;
        const int nch = (a.seg_start[s + 1] - a.seg_start[s] + DR_SEG_CHUNK - 1) / DR_SEG_CHUNK;
        const float* part = a.partials + (int64_t)a.long_cb[l] * a.k;
        for (int c = lane; c < cpr; c += LPR) {
            float tot[V];
#pragma unroll
            for (int x = 0; x < V; ++x) tot[x] = 0.0f;
            int ch = 0;
            // same sequence of additions; 32 (then 8) partial rows requested per round trip, the last < 8 as one
            // predicated batch instead of a load-add chain per row
            for (; ch + 32 <= nch; ch += 32) {
                float v[32][V];
#pragma unroll
                for (int q = 0; q < 32; ++q) vload<V>(v[q], part + (int64_t)(ch + q) * a.k + c * V);
#pragma unroll
                for (int q = 0; q < 32; ++q) {
#pragma unroll
                    for (int x = 0; x < V; ++x) tot[x] = tot[x] + v[q][x];
                }
            }
            for (; ch + 8 <= nch; ch += 8) {
                float v[8][V];
#pragma unroll
                for (int q = 0; q < 8; ++q) vload<V>(v[q], part + (int64_t)(ch + q) * a.k + c * V);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int x = 0; x < V; ++x) tot[x] = tot[x] + v[q][x];
                }
            }
            if (ch < nch) {
                float v[8][V];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    vload<V>(v[q], part + (int64_t)(ch + q < nch ? ch + q : nch - 1) * a.k + c * V);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (ch + q < nch) {
#pragma unroll
                        for (int x = 0; x < V; ++x) tot[x] = tot[x] + v[q][x];
                    }
                }
            }
            apply_row<V>(a, r, c, tot);
        }
    }
}


// ---- the fused BPR step's whole duplicate-row tail in ONE launch --------------------------------------------------
// After the fused forward kernel has applied every single-contribution row, what is left of a batch is (C4: per 65 536
// samples) ~20 k rows with 2 .. LONG_SEG contributions and a few hundred Zipf-head rows with more, spread over both
// tables.  As separate launches (short / long partial / long combine, per table) that was six kernels of 8 - 23 us,
// each mostly a walk over the 140 k-entry segment table and launch ramp; here one grid handles
//   work ids [0, hdr[2]):           one DR_SEG_CHUNK block of a long segment -> its partial row
//   work ids [hdr[2], + hdr[4]):    one row of the plan's `multi` list (2 .. LONG_SEG contributions)
// of both tables (segments below hdr[3] are user rows, the others item rows); bpr_tail_combine_kernel then adds each
// long segment's partial rows in block order and applies the row, for both tables.  (Combining inside the first launch
// -- last finished block of a segment, device-scope counter -- was measured: the two device-scope fences per block
// write back and invalidate a whole L2 each on this multi-XCD part, 101 us for the launch.)  Same blocked position
// order, the same operations per row as seg_update_kernel / seg_long_*_kernel: bit-identical results.
struct TailArgs {
    SegArgs p, q;
    const int4* multi;
    int32_t* long_cnt;
    const float* loss_b;   // per-sample losses of the step and where their sum goes (bpr_tail_combine_kernel's block 0);
    float* loss_out;       // NULL: no sum
    int64_t loss_n;
};

template <int V, int LPR, int KIND>
__global__ __launch_bounds__(256) void bpr_tail_kernel(TailArgs t) {
    t.p.opt = resolve(t.p.opt);
    t.q.opt = t.p.opt;
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR;
    const int lane = threadIdx.x % LPR;
    const int cpr = t.p.k / V;
    const int k = t.p.k;
    const int nparts = t.p.hdr[2], nmulti = t.p.hdr[4], split = t.p.hdr[3];
    const int total = nparts + nmulti;
    for (int w = blockIdx.x * SPB + sub; w < total; w += gridDim.x * SPB) {
        if (w < nparts) {
            const int l = t.p.part_owner[w];
            const int s = t.p.long_seg[l];
            const SegArgs& a = (s < split) ? t.p : t.q;
            const uint32_t r = (uint32_t)a.uniq[s] - a.id_lo;
            if ((int64_t)r >= a.rows) continue;  // the sentinel segment of out-of-range ids
            const int cb = a.long_cb[l];
            const int b = a.seg_start[s], e = a.seg_start[s + 1];
            const int c0 = b + (w - cb) * DR_SEG_CHUNK;
            const int c1 = (c0 + DR_SEG_CHUNK < e) ? c0 + DR_SEG_CHUNK : e;
            for (int c = lane; c < cpr; c += LPR) {
                float acc[V];
                block_sum<V, 16>(a, c0, c1, c, acc);
                vstore<V>(a.partials + (int64_t)w * k + c * V, acc);
            }
        } else {
            const int4 dsc = t.multi[w - nparts];    // {segment, key, first sorted position, length}
            const SegArgs& a = (dsc.x < split) ? t.p : t.q;
            const uint32_t r = (uint32_t)dsc.y - a.id_lo;
            if ((int64_t)r >= a.rows) continue;
            const int b = dsc.z;
            const int len = dsc.w;  // 2 .. LONG_SEG
            int si[4], sp[4];
            float cf[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                si[q] = -1;
                sp[q] = 0;
                if (q < len) {
                    si[q] = a.msrc[b + q];
                    if (a.coef) sp[q] = a.spos[b + q];
                }
            }
            const int ns = KIND == DR_OPT_ADAM ? 2 : kind_slots(a.kind);
            for (int c = lane; c < cpr; c += LPR) {
                // the destination row's (w, m, v) are requested BEFORE the source-row indices above are waited for:
                // descriptor -> {indices, destination row} -> {source rows, coefficients} = three round trips per row
                const int64_t off = (int64_t)r * k + c * V;
                float wv[V], s0[V], s1[V];
                vload<V>(wv, a.table + off);
                if (ns >= 1) vload<V>(s0, a.slot0 + off);
                if (ns >= 2) vload<V>(s1, a.slot1 + off);
#pragma unroll
                for (int q = 0; q < 4; ++q) cf[q] = (a.coef && si[q] >= 0) ? a.coef[sp[q] - a.pos_lo] : 1.0f;
                float v[4][V], tot[V];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (si[q] >= 0) vload<V>(v[q], seg_src_row(a, si[q]) + c * V);
#pragma unroll
                for (int x = 0; x < V; ++x) tot[x] = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (si[q] >= 0) {
#pragma unroll
                        for (int x = 0; x < V; ++x) tot[x] = tot[x] + (a.coef ? cf[q] * v[q][x] : v[q][x]);
                    }
#pragma unroll
                for (int x = 0; x < V; ++x) tot[x] = 0.0f + tot[x];
                apply_row_loaded<V, KIND>(a, r, c, tot, wv, s0, s1);
            }
        }
    }
}

// kept out of line: inlined, its 32 loads in flight change how the compiler schedules the combine loop below (the
// 32-row batch became three drained groups instead of a sliding window of 12; 10 -> 20 us for the launch)
__device__ __attribute__((noinline)) void tail_loss_sum(const float* x, int64_t n, float* out, float* red) {
    fixed_sum<256>(x, n, out, red);
}

template <int V, int LPR, int KIND>
__global__ __launch_bounds__(256) void bpr_tail_combine_kernel(TailArgs t) {
    t.p.opt = resolve(t.p.opt);
    t.q.opt = t.p.opt;
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR;
    const int lane = threadIdx.x % LPR;
    const int k = t.p.k;
    const int cpr = k / V;
    const int nlong = t.p.hdr[1], split = t.p.hdr[3];
    // the step's loss (dr_sum_f32's order and bits) rides along as block 0 instead of a launch of its own after this one
    __shared__ float red[1024];
    const int first = t.loss_out ? 1 : 0;
    if (first && blockIdx.x == 0) {
        tail_loss_sum(t.loss_b, t.loss_n, t.loss_out, red);
        return;
    }
    const int nblk = gridDim.x - first;
    for (int l = (blockIdx.x - first) * SPB + sub; l < nlong; l += nblk * SPB) {
        const int s = t.p.long_seg[l];
        const SegArgs& a = (s < split) ? t.p : t.q;
        const uint32_t r = (uint32_t)a.uniq[s] - a.id_lo;
        if ((int64_t)r >= a.rows) continue;
        const int nch = (a.seg_start[s + 1] - a.seg_start[s] + DR_SEG_CHUNK - 1) / DR_SEG_CHUNK;
        const float* part = a.partials + (int64_t)a.long_cb[l] * k;
        const int ns = KIND == DR_OPT_ADAM ? 2 : kind_slots(a.kind);
        for (int c = lane; c < cpr; c += LPR) {
            // the destination row's (w, m, v) are requested before the partial rows: one round trip less per row
            const int64_t off = (int64_t)r * k + c * V;
            float wv[V], s0[V], s1[V];
            vload<V>(wv, a.table + off);
            if (ns >= 1) vload<V>(s0, a.slot0 + off);
            if (ns >= 2) vload<V>(s1, a.slot1 + off);
            float tot[V];
#pragma unroll
            for (int x = 0; x < V; ++x) tot[x] = 0.0f;
            int ch = 0;
            // the Zipf-head rows have hundreds of partial rows and ONE sub-wave adds them in order: 32 loads in flight per
            // round trip (same sequence of additions), or the longest row alone takes 20 round trips
            for (; ch + 32 <= nch; ch += 32) {
                float v[32][V];
#pragma unroll
                for (int q = 0; q < 32; ++q) vload<V>(v[q], part + (int64_t)(ch + q) * k + c * V);
#pragma unroll
                for (int q = 0; q < 32; ++q) {
#pragma unroll
                    for (int x = 0; x < V; ++x) tot[x] = tot[x] + v[q][x];
                }
            }
            for (; ch + 8 <= nch; ch += 8) {
                float v[8][V];
#pragma unroll
                for (int q = 0; q < 8; ++q) vload<V>(v[q], part + (int64_t)(ch + q) * k + c * V);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int x = 0; x < V; ++x) tot[x] = tot[x] + v[q][x];
                }
            }
            if (ch < nch) {   // the last < 8 partial rows: one predicated batch, not a load-add chain per row
                float v[8][V];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    vload<V>(v[q], part + (int64_t)(ch + q < nch ? ch + q : nch - 1) * k + c * V);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (ch + q < nch) {
#pragma unroll
                        for (int x = 0; x < V; ++x) tot[x] = tot[x] + v[q][x];
                    }
                }
            }
            apply_row_loaded<V, KIND>(a, r, c, tot, wv, s0, s1);
        }
    }
}

template <int V, int LPR>
static void launch_tail(const TailArgs& t, int64_t n, hipStream_t st) {
    constexpr int SPB = 256 / LPR;
    // work items <= n / DR_SEG_CHUNK + n / LONG_SEG (long blocks) + n (multi rows; n / 2 in the single-GPU step)
    const int64_t items = n / DR_SEG_CHUNK + n / LONG_SEG + n + 1;
    const int64_t blocks = (items + SPB - 1) / SPB;
    const int cgrid = grid_for(n / LONG_SEG + 1, 1, SPB) + (t.loss_out ? 1 : 0);
    // Twice the resident blocks: a sub-wave then takes ONE work item (a 32-contribution block sum is 2 - 3 times a
    // short row) and the hardware hands the remaining blocks to whichever CU is free; with exactly the resident count
    // the sub-waves that start with a block sum also get a row each: 21.0 -> 18.9 us (x4 and x16: 18.8, 18.7).  Fetching
    // a block's 32 indices / coefficients one per lane and passing them across lanes was measured too: + 2.7 us.
    static const int mult = DR_EXP_ENV("DR_TAIL_GRID_MULT") ? atoi(DR_EXP_ENV("DR_TAIL_GRID_MULT")) : 2;
    auto tail_grid = [&](int resident) {
        const int64_t g = (int64_t)resident * mult;
        return (int)(g < blocks ? g : blocks);
    };
    if (t.p.kind == DR_OPT_ADAM) {
        DR_LAUNCH_NAMED("bpr_tail_kernel", (bpr_tail_kernel<V, LPR, DR_OPT_ADAM>),
                        dim3(tail_grid(grid_resident(bpr_tail_kernel<V, LPR, DR_OPT_ADAM>, blocks))), dim3(256), 0, st, t);
        if (n > LONG_SEG || t.loss_out)
            DR_LAUNCH_NAMED("bpr_tail_combine_kernel", (bpr_tail_combine_kernel<V, LPR, DR_OPT_ADAM>), dim3(cgrid), dim3(256),
                            0, st, t);
    } else {
        DR_LAUNCH_NAMED("bpr_tail_kernel", (bpr_tail_kernel<V, LPR, KIND_ANY>),
                        dim3(tail_grid(grid_resident(bpr_tail_kernel<V, LPR, KIND_ANY>, blocks))), dim3(256), 0, st, t);
        if (n > LONG_SEG || t.loss_out)
            DR_LAUNCH_NAMED("bpr_tail_combine_kernel", (bpr_tail_combine_kernel<V, LPR, KIND_ANY>), dim3(cgrid), dim3(256), 0,
                            st, t);
    }
}

template <int V, int LPR, int KIND>
static void launch_short_k(const SegArgs& a, int grid, hipStream_t st) {
    // grid: blocks of work; launched with at most the number of blocks the device holds at once (no partial last wave)
    if (a.part == 2) grid = grid_resident(seg_update_kernel<V, LPR, 2, KIND>, grid);
    else if (a.part == 1) grid = grid_resident(seg_update_kernel<V, LPR, 1, KIND>, grid);
    else grid = grid_resident(seg_update_kernel<V, LPR, 0, KIND>, grid);
    if (a.part == 2)
        DR_LAUNCH_NAMED("seg_update_kernel<item side>", (seg_update_kernel<V, LPR, 2, KIND>), dim3(grid), dim3(256), 0, st, a);
    else if (a.part == 1)
        DR_LAUNCH_NAMED("seg_update_kernel<user side>", (seg_update_kernel<V, LPR, 1, KIND>), dim3(grid), dim3(256), 0, st, a);
    else
        DR_LAUNCH_NAMED("seg_update_kernel", (seg_update_kernel<V, LPR, 0, KIND>), dim3(grid), dim3(256), 0, st, a);
}
template <int V, int LPR>
static void launch_short(const SegArgs& a, int grid, hipStream_t st) {
    if (a.kind == DR_OPT_ADAM) launch_short_k<V, LPR, DR_OPT_ADAM>(a, grid, st);  // fewer registers, one more wave/SIMD
    else if (a.kind == KIND_ACC) launch_short_k<V, LPR, KIND_ACC>(a, grid, st);
    else launch_short_k<V, LPR, KIND_ANY>(a, grid, st);
}

template <int V, int LPR>
static void launch_seg(const SegArgs& a, int64_t n, hipStream_t main) {
    constexpr int SPB = 256 / LPR;
    int grid = (int)((n + SPB - 1) / SPB > (1 << 30) ? (1 << 30) : (n + SPB - 1) / SPB);  // capped by launch_short_k
    if (grid < 1) grid = 1;
    // short and long segments own disjoint destination rows: the two long-segment passes run on a helper
    // stream beside the short-segment kernel (fork / join with events; also valid under graph capture)
    hipEvent_t fork = nullptr, join = nullptr;
    hipStream_t st = main;
    if (n > LONG_SEG) st = seg_side_stream(main, &fork, &join);
    if (st == main) launch_short<V, LPR>(a, grid, main);
    if (n > LONG_SEG) {
        // partial rows <= n / DR_SEG_CHUNK + n / LONG_SEG ; long segments <= n / LONG_SEG
        DR_LAUNCH_NAMED("seg_long_partial_kernel", (seg_long_partial_kernel<V, LPR>),
                        dim3(grid_for(n / DR_SEG_CHUNK + n / LONG_SEG + 1, 1, SPB)), dim3(256), 0, st, a);
        DR_LAUNCH_NAMED("seg_long_combine_kernel", (seg_long_combine_kernel<V, LPR>),
                        dim3(grid_for(n / LONG_SEG + 1, 1, SPB)), dim3(256), 0, st, a);
        if (st != main) {  // the (small) long-segment passes were enqueued first so that they are not starved
            (void)hipEventRecord(join, st);
            launch_short<V, LPR>(a, grid, main);
            (void)hipStreamWaitEvent(main, join, 0);
        }
    }
}

static int dispatch_seg(const SegArgs& a, int64_t n, hipStream_t st) {
    RowShape s = row_shape(a.k);
#define DR_S(V_, L_)                     \
    if (s.V == V_ && s.LPR == L_) {      \
        launch_seg<V_, L_>(a, n, st);    \
        return 0;                        \
    }
    DR_S(4, 1) DR_S(4, 2) DR_S(4, 4) DR_S(4, 8) DR_S(4, 16) DR_S(4, 32) DR_S(4, 64)
    DR_S(2, 1) DR_S(2, 2) DR_S(2, 4) DR_S(2, 8) DR_S(2, 16) DR_S(2, 32) DR_S(2, 64)
    DR_S(1, 1) DR_S(1, 2) DR_S(1, 4) DR_S(1, 8) DR_S(1, 16) DR_S(1, 32) DR_S(1, 64)
#undef DR_S
    return -1;
}

// ---------------------------------------------------------------------------------------
// dense (whole tensor) optimizer application and row-norm clip
// ---------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void dense_update_kernel(float* __restrict__ w, float* __restrict__ s0p,
                                                           float* __restrict__ s1p, const float* __restrict__ grad,
                                                           int64_t count, OptP o, const uint8_t* __restrict__ skip,
                                                           int k) {
    o = resolve(o);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        if (skip && skip[count <= 0xffffffffll ? (int64_t)((uint32_t)i / (uint32_t)k) : i / k]) continue;   // rows another kernel updates
        float wi = w[i];
        float g = grad ? grad[i] : 0.0f;
        if (o.reg_mode != DR_REG_NONE) g = g + o.reg * wi;
        float a = 0.0f, b = 0.0f;
        if constexpr (KIND != DR_OPT_SGD) a = s0p[i];
        if constexpr (KIND == DR_OPT_ADAM || KIND == DR_OPT_RMSPROP) b = s1p[i];
        opt_apply<KIND>(wi, a, b, g, o);
        w[i] = wi;
        if constexpr (KIND != DR_OPT_SGD) s0p[i] = a;
        if constexpr (KIND == DR_OPT_ADAM || KIND == DR_OPT_RMSPROP) s1p[i] = b;
    }
}

// the same update on float4 elements (count % 4 == 0, 16-byte aligned tensors): four times fewer, wider memory
// instructions on what is a pure streaming pass (Caser's [U, d L] table: 25.6 GB x 6 streams per step)
// NT: tables larger than the caches (>= DR_DENSE_NT_BYTES per stream) are streamed with the non-temporal hint, so that a
// pass over a 0.5 - 25 GB table does not evict the small tables and activations the step's other kernels re-read
template <int KIND, bool NT = false>
__global__ __launch_bounds__(256) void dense_update4_kernel(float4* __restrict__ w, float4* __restrict__ s0p,
                                                            float4* __restrict__ s1p, const float4* __restrict__ grad,
                                                            int64_t count4, OptP o, const uint8_t* __restrict__ skip,
                                                            int k4) {
    o = resolve(o);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (int64_t)gridDim.x * 256) {
        // k % 4 == 0: a float4 never straddles two rows.  32-bit division where the index allows it (a 64-bit one is
        // ~100 instructions on this pass of 4 loads and 3 stores)
        if (skip && skip[count4 <= 0xffffffffll ? (int64_t)((uint32_t)i / (uint32_t)k4) : i / k4]) continue;
        float wv[4], gv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, av[4] = {0.0f, 0.0f, 0.0f, 0.0f}, bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (NT) {
            vload_nt<4>(wv, reinterpret_cast<const float*>(w + i));
            if (grad) vload_nt<4>(gv, reinterpret_cast<const float*>(grad + i));
            if constexpr (KIND != DR_OPT_SGD) vload_nt<4>(av, reinterpret_cast<const float*>(s0p + i));
            if constexpr (KIND == DR_OPT_ADAM || KIND == DR_OPT_RMSPROP) vload_nt<4>(bv, reinterpret_cast<const float*>(s1p + i));
        } else {
            vload<4>(wv, reinterpret_cast<const float*>(w + i));
            if (grad) vload<4>(gv, reinterpret_cast<const float*>(grad + i));
            if constexpr (KIND != DR_OPT_SGD) vload<4>(av, reinterpret_cast<const float*>(s0p + i));
            if constexpr (KIND == DR_OPT_ADAM || KIND == DR_OPT_RMSPROP) vload<4>(bv, reinterpret_cast<const float*>(s1p + i));
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            float g = gv[x];
            if (o.reg_mode != DR_REG_NONE) g = g + o.reg * wv[x];
            opt_apply<KIND>(wv[x], av[x], bv[x], g, o);
        }
        if (NT) {
            vstore_nt<4>(reinterpret_cast<float*>(w + i), wv);
            if constexpr (KIND != DR_OPT_SGD) vstore_nt<4>(reinterpret_cast<float*>(s0p + i), av);
            if constexpr (KIND == DR_OPT_ADAM || KIND == DR_OPT_RMSPROP) vstore_nt<4>(reinterpret_cast<float*>(s1p + i), bv);
        } else {
            vstore<4>(reinterpret_cast<float*>(w + i), wv);
            if constexpr (KIND != DR_OPT_SGD) vstore<4>(reinterpret_cast<float*>(s0p + i), av);
            if constexpr (KIND == DR_OPT_ADAM || KIND == DR_OPT_RMSPROP) vstore<4>(reinterpret_cast<float*>(s1p + i), bv);
        }
    }
}

// one sub-wave per row: sequential-in-lane partial sums of squares, butterfly, scale
template <int V, int LPR>
__global__ __launch_bounds__(256) void rownorm_clip_kernel(float* __restrict__ t, int64_t rows, int k, float clip) {
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int cpr = k / V;
    const int64_t iters = (rows + RPB - 1) / RPB;
    for (int64_t it = blockIdx.x; it < iters; it += gridDim.x) {
        const int64_t r = it * RPB + sub;
        float ss = 0.0f;
        if (r < rows)
            for (int c = lane; c < cpr; c += LPR) {
                float v[V];
                vload<V>(v, t + r * k + c * V);
#pragma unroll
                for (int x = 0; x < V; ++x) ss = ss + v[x] * v[x];
            }
        ss = subwave_sum<LPR>(ss);
        const float nrm = sqrtf(ss);
        // [TF-ext] clip_by_norm: (t * clip) / max(norm, clip); rows with norm <= clip are left
        // untouched, which is exact for clip = 1 (the only value the reference uses, cml.py:56).
        if (r < rows && nrm > clip) {
            for (int c = lane; c < cpr; c += LPR) {
                float v[V];
                vload<V>(v, t + r * k + c * V);
#pragma unroll
                for (int x = 0; x < V; ++x) v[x] = (v[x] * clip) / nrm;
                vstore<V>(t + r * k + c * V, v);
            }
        }
    }
}

}  // namespace dr

using namespace dr;

namespace dr {
// shared tail of the plan builders: stable sort of (key, position) and -- with `tables` -- run boundaries (uniq /
// seg_start / hdr) and the long / multi lists; the fused BPR plan builds its own lists (bpr_finish_kernel)
template <class Src>
static int plan_finish(Src src, char* base, const PlanLayout& L, int64_t n, unsigned end_bit, uint32_t split_key,
                       hipStream_t st, bool tables) {
    int32_t* hdr = (int32_t*)(base + L.hdr);
    uint32_t* keys_in = (uint32_t*)(base + L.keys_in);
    uint32_t* keys_out = (uint32_t*)(base + L.keys_out);
    int32_t* vals_in = (int32_t*)(base + L.vals_in);
    int32_t* vals_out = (int32_t*)(base + L.vals_out);
    int32_t* flags = (int32_t*)(base + L.flags);
    int32_t* segidx = (int32_t*)(base + L.segidx);
    int32_t* uniq = (int32_t*)(base + L.uniq);
    int32_t* seg_start = (int32_t*)(base + L.seg_start);
    if (n <= SMALL_N) {
        if (n <= 256 * SMALL_IPT)
            DR_LAUNCH((plan_small_kernel<256, Src>), dim3(1), dim3(256), 0, st, src, (int)n, end_bit,
                      split_key, keys_out, vals_out, uniq, seg_start, hdr, (int32_t*)(base + L.long_seg),
                      (int32_t*)(base + L.long_cb), (int32_t*)(base + L.part_owner), (int4*)(base + L.multi),
                      (int32_t*)(base + L.long_cnt));
        else
            DR_LAUNCH((plan_small_kernel<1024, Src>), dim3(1), dim3(1024), 0, st, src, (int)n, end_bit,
                      split_key, keys_out, vals_out, uniq, seg_start, hdr, (int32_t*)(base + L.long_seg),
                      (int32_t*)(base + L.long_cb), (int32_t*)(base + L.part_owner), (int4*)(base + L.multi),
                      (int32_t*)(base + L.long_cnt));
        DR_CHECK_LAUNCH("plan_small_kernel");
        return DR_OK;
    }
    const int grid = grid_for(n, 256);
    if (n <= DR_LSD_MAX_N) {
        int rc = lsd_sort(src, n, end_bit, keys_in, vals_in, keys_out, vals_out, base + L.lsd, hdr, st);
        if (rc != DR_OK) return rc;
    } else {
        DR_LAUNCH((plan_keys_kernel<Src>), dim3(grid), dim3(256), 0, st, src, n, keys_in, vals_in, hdr);
        DR_CHECK_LAUNCH("plan_keys_kernel");
        size_t tb = (size_t)L.temp_bytes;
        DR_HIP((rocprim::radix_sort_pairs<rocprim::default_config>(base + L.temp, tb, keys_in, keys_out, vals_in, vals_out,
                                                                   (size_t)n, 0u, end_bit, st, false)));
    }
    if (!tables) return DR_OK;
    const int64_t tiles = (n + PLAN_TILE - 1) / PLAN_TILE;
    if (tiles <= PLAN_MAX_TILES) {
        DR_LAUNCH(plan_count_kernel, dim3((unsigned)tiles), dim3(256), 0, st, keys_out, n, flags);
        DR_CHECK_LAUNCH("plan_count_kernel");
        DR_LAUNCH(plan_segments_kernel, dim3((unsigned)tiles), dim3(256), 0, st, keys_out, n, (const int32_t*)flags, uniq,
                  seg_start, hdr, split_key);
        DR_CHECK_LAUNCH("plan_segments_kernel");
    } else {
        DR_LAUNCH(plan_heads_kernel, dim3(grid), dim3(256), 0, st, keys_out, n, flags);
        DR_CHECK_LAUNCH("plan_heads_kernel");
        size_t tb = (size_t)L.temp_bytes;
        DR_HIP((rocprim::exclusive_scan<rocprim::default_config>(base + L.temp, tb, flags, segidx, (int32_t)0, (size_t)n,
                                                                 rocprim::plus<int32_t>(), st, false)));
        DR_LAUNCH(plan_finalize_kernel, dim3(grid), dim3(256), 0, st, keys_out, flags, segidx, n, uniq, seg_start, hdr,
                  split_key);
        DR_CHECK_LAUNCH("plan_finalize_kernel");
    }
    if (n >= 2) {
        DR_LAUNCH(plan_long_kernel, dim3(grid), dim3(256), 0, st, seg_start, (const int32_t*)uniq, hdr, (int32_t*)(base + L.long_seg),
                  (int32_t*)(base + L.long_cb), (int32_t*)(base + L.part_owner), (int4*)(base + L.multi),
                  (int32_t*)(base + L.long_cnt), n);
        DR_CHECK_LAUNCH("plan_long_kernel");
    }
    return DR_OK;
}

// One plan for both tables of a BPR batch (3B contributions: [u ; i ; j], Q keys offset by num_user) plus the
// source-row list of the sorted contributions.  fused (the planned step): sort (2 x 2 launches) + bpr_finish_kernel;
// lists of <= 4096 keys: the one-workgroup plan + bpr_meta_kernel.
int plan_build_bpr(const int32_t* u, const int32_t* i, const int32_t* j, int64_t B, int64_t nu, int64_t ni, void* plan,
                   int64_t plan_bytes, int k, hipStream_t st, bool fused) {
    const int64_t n = 3 * B;
    DR_REQUIRE(n < (1ll << 31) - 1 && nu + ni < (1ll << 32) - 2, "bpr plan: batch / tables too large for 32-bit keys");
    DR_REQUIRE(!fused || nu + 2 * B < (1ll << 31) - 1, "bpr plan: num_user + 2 B exceeds 31 bits");
    PlanLayout L;
    if (plan_layout(n, k, &L, last_error_buf()) != 0) return DR_ERR_LAUNCH;
    if (plan_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "bpr plan: workspace %lld < required %lld", (long long)plan_bytes, (long long)L.total);
    char* base = (char*)plan;
    int32_t* hdr = (int32_t*)(base + L.hdr);
    if (n == 0) {
        DR_LAUNCH(plan_empty_kernel, dim3(1), dim3(1), 0, st, hdr, (int32_t*)(base + L.seg_start));
        DR_CHECK_LAUNCH("bpr plan (empty)");
        return DR_OK;
    }
    unsigned end_bit = 1;
    while (end_bit < 32 && ((uint64_t)(nu + ni) >> end_bit) != 0) ++end_bit;
    const bool finish = fused && n > SMALL_N;
    int rc = plan_finish(LsdBpr{u, i, j, B, nu, ni}, base, L, n, end_bit, (uint32_t)nu, st, !finish);
    if (rc != DR_OK) return rc;
    if (finish) {
        FinBpr fin{u, B, nu, ni, (unsigned char*)(base + L.solo)};
        DR_LAUNCH_NAMED("bpr_finish_kernel", (bpr_finish_kernel<FinBpr>), dim3((unsigned)((n + FIN_TILE - 1) / FIN_TILE)),
                        dim3(256), 0, st, (const uint32_t*)(base + L.keys_out), (const int32_t*)(base + L.vals_out), n, fin, hdr,
                        (int32_t*)(base + L.uniq), (int32_t*)(base + L.seg_start), (int32_t*)(base + L.long_seg),
                        (int32_t*)(base + L.long_cb), (int32_t*)(base + L.long_cnt), (int32_t*)(base + L.part_owner),
                        (int4*)(base + L.multi), (int32_t*)(base + L.msrc));
        DR_CHECK_LAUNCH("bpr_finish_kernel");
        return DR_OK;
    }
    DR_LAUNCH(bpr_meta_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, (const int32_t*)(base + L.vals_out),
              (const uint32_t*)(base + L.keys_out), u, B, nu, ni, fused ? 1 : 0, (int32_t*)(base + L.msrc),
              (unsigned char*)(base + L.solo));
    DR_CHECK_LAUNCH("bpr_meta_kernel");
    return DR_OK;
}
}  // namespace dr

extern "C" int64_t dr_segment_plan_bytes(int64_t n, int32_t k_max) {
    if (n < 0 || k_max < 1) return DR_ERR_ARG;
    PlanLayout L;
    if (plan_layout(n, k_max, &L, last_error_buf()) != 0) return DR_ERR_LAUNCH;
    return L.total;
}

extern "C" int32_t dr_segment_plan(const void* ids, int64_t n, int32_t id_bits, int64_t rows, void* plan,
                                   int64_t plan_bytes, void* stream) {
    DR_REQUIRE(n >= 0 && n < (1ll << 31) - 1, "dr_segment_plan: n=%lld out of range", (long long)n);
    DR_REQUIRE(rows >= 0 && rows < (1ll << 31) - 1, "dr_segment_plan: rows=%lld out of range", (long long)rows);
    DR_REQUIRE(id_bits == 32 || id_bits == 64, "dr_segment_plan: id_bits must be 32 or 64");
    DR_REQUIRE(plan != nullptr && ((uintptr_t)plan % 256 == 0), "dr_segment_plan: plan must be 256-byte aligned");
    DR_REQUIRE(ids != nullptr || n == 0, "dr_segment_plan: ids is NULL");
    PlanLayout L;
    if (plan_layout(n, 1, &L, last_error_buf()) != 0) return DR_ERR_LAUNCH;
    if (plan_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_segment_plan: workspace %lld < required %lld", (long long)plan_bytes,
                (long long)L.total);
    char* base = (char*)plan;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        DR_LAUNCH(plan_empty_kernel, dim3(1), dim3(1), 0, st, (int32_t*)(base + L.hdr), (int32_t*)(base + L.seg_start));
        DR_CHECK_LAUNCH("dr_segment_plan(empty)");
        return DR_OK;
    }
    unsigned end_bit = 1;
    while (end_bit < 32 && ((uint64_t)rows >> end_bit) != 0) ++end_bit;  // enough bits for the sentinel `rows`
    if (id_bits == 32)
        return plan_finish(LsdIds<int32_t>{(const int32_t*)ids, rows}, base, L, n, end_bit, 0xFFFFFFFFu, st, true);
    return plan_finish(LsdIds<int64_t>{(const int64_t*)ids, rows}, base, L, n, end_bit, 0xFFFFFFFFu, st, true);
}

extern "C" int32_t dr_segment_plan_view(void* plan, int64_t n, const int32_t** sorted_pos, const int32_t** uniq_ids,
                                        const int32_t** seg_start, const int32_t** num_segments) {
    DR_REQUIRE(plan != nullptr && n >= 0, "dr_segment_plan_view: bad arguments");
    PlanLayout L;
    if (plan_layout(n, 1, &L, last_error_buf()) != 0) return DR_ERR_LAUNCH;
    char* base = (char*)plan;
    if (sorted_pos) *sorted_pos = (const int32_t*)(base + L.vals_out);
    if (uniq_ids) *uniq_ids = (const int32_t*)(base + L.uniq);
    if (seg_start) *seg_start = (const int32_t*)(base + L.seg_start);
    if (num_segments) *num_segments = (const int32_t*)(base + L.hdr);
    return DR_OK;
}

namespace dr {
// shared by scatter-add, sparse update and the fused steps
// msrc[y] = source row of the y-th sorted contribution; depends on the ids and src_idx only, so fused steps
// run it with the plan, off the critical path
int seg_meta(void* plan, int64_t n, const int32_t* src_idx, int64_t src_rows, hipStream_t st) {
    PlanLayout L;
    if (plan_layout(n, 1, &L, last_error_buf()) != 0) return DR_ERR_LAUNCH;
    if (n == 0) return DR_OK;
    char* base = (char*)plan;
    DR_LAUNCH(seg_meta_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, (const int32_t*)(base + L.vals_out), src_idx, n,
              src_rows, (int32_t*)(base + L.msrc));
    DR_CHECK_LAUNCH("seg_meta_kernel");
    return DR_OK;
}

hipStream_t seg_side_stream(hipStream_t main, hipEvent_t* fork, hipEvent_t* join);

// solo flags of a BPR plan (one byte per original position; see bpr_meta_kernel)
const unsigned char* plan_solo_flags(void* plan, int64_t n, int k) {
    PlanLayout L;
    if (plan_layout(n, k, &L, last_error_buf()) != 0) return nullptr;
    return (const unsigned char*)((char*)plan + L.solo);
}

// n: contributions the plan was built for; n_table: those of them that belong to this table (grid sizing);
// sel: which part of a combined two-table plan this launch owns ({0,0,0} = an ordinary single-table plan)
static int seg_fill(SegArgs& a, const PlanLayout& L, char* base, float* table, float* slot0, float* slot1, int64_t rows,
                    int k, const float* src, int64_t src_rows, const int32_t* src_idx, const float* coef, int64_t n,
                    void* plan, int kind, const dr_opt_params* opt, hipStream_t st, bool meta_ready, SegSel sel,
                    const float* src2, bool skip_single) {
    a.table = table;
    a.slot0 = slot0;
    a.slot1 = slot1;
    a.rows = rows;
    a.k = k;
    a.src = src;
    a.src_rows = src_rows;
    a.src2 = src2;
    a.skip_single = skip_single ? 1 : 0;
    const int32_t* sorted_pos = (const int32_t*)(base + L.vals_out);
    a.spos = sorted_pos;
    a.coef = coef;
    a.id_lo = sel.id_lo;
    a.part = sel.part;
    a.pos_lo = sel.pos_lo;
    if (meta_ready) {
        a.msrc = (const int32_t*)(base + L.msrc);
    } else if (src_idx == nullptr && src_rows >= n) {
        a.msrc = sorted_pos;  // contribution e reads row e: the sorted positions are the source rows
    } else {
        int rc = seg_meta(plan, n, src_idx, src_rows, st);
        if (rc != DR_OK) return rc;
        a.msrc = (const int32_t*)(base + L.msrc);
    }
    a.uniq = (const int32_t*)(base + L.uniq);
    a.seg_start = (const int32_t*)(base + L.seg_start);
    a.hdr = (const int32_t*)(base + L.hdr);
    a.long_seg = (const int32_t*)(base + L.long_seg);
    a.long_cb = (const int32_t*)(base + L.long_cb);
    a.part_owner = (const int32_t*)(base + L.part_owner);
    a.partials = (float*)(base + L.partials);
    a.kind = kind;
    if (opt) a.opt = make_optp(opt);
    else memset(&a.opt, 0, sizeof(a.opt));
    return DR_OK;
}

int seg_launch(float* table, float* slot0, float* slot1, int64_t rows, int k, const float* src, int64_t src_rows,
               const int32_t* src_idx, const float* coef, int64_t n, void* plan, int64_t plan_bytes, int kind,
               const dr_opt_params* opt, hipStream_t st, bool meta_ready, SegSel sel, int64_t n_table, const float* src2,
               bool skip_single) {
    PlanLayout L;
    if (plan_layout(n, k, &L, last_error_buf()) != 0) return DR_ERR_LAUNCH;
    if (plan_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "segment kernel: plan workspace %lld < dr_segment_plan_bytes(n=%lld, k=%d) = %lld",
                (long long)plan_bytes, (long long)n, k, (long long)L.total);
    SegArgs a;
    int rc = seg_fill(a, L, (char*)plan, table, slot0, slot1, rows, k, src, src_rows, src_idx, coef, n, plan, kind, opt, st,
                      meta_ready, sel, src2, skip_single);
    if (rc != DR_OK) return rc;
    if (dispatch_seg(a, n_table > 0 ? n_table : n, st) != 0) DR_FAIL(DR_ERR_ARG, "segment kernel: unsupported k=%d", k);
    DR_CHECK_LAUNCH("seg_update_kernel");
    return DR_OK;
}

// The fused BPR step's tail (bpr_tail_kernel): every row with two or more contributions, of both tables, one launch.
// The combined plan of the batch (plan_build_bpr, fused) must be in `plan`; stash / coefQ as in bpr_step_impl.
int bpr_tail_launch(float* P, float* Ps0, float* Ps1, int64_t nu, float* Q, float* Qs0, float* Qs1, int64_t ni, int k,
                    int64_t B, const float* stash, const float* coefQ, void* plan, int64_t plan_bytes,
                    const dr_opt_params* opt, hipStream_t st, const float* loss_b, float* loss_out) {
    const int64_t n = 3 * B;
    PlanLayout L;
    if (plan_layout(n, k, &L, last_error_buf()) != 0) return DR_ERR_LAUNCH;
    if (plan_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "bpr tail: plan workspace %lld < required %lld", (long long)plan_bytes, (long long)L.total);
    char* base = (char*)plan;
    TailArgs t;
    int rc = seg_fill(t.q, L, base, Q, Qs0, Qs1, ni, k, P, nu, nullptr, coefQ, n, plan, opt->kind, opt, st, true,
                      SegSel{(uint32_t)nu, 2, (int)B}, stash, true);
    if (rc != DR_OK) return rc;
    rc = seg_fill(t.p, L, base, P, Ps0, Ps1, nu, k, stash, B, nullptr, nullptr, n, plan, opt->kind, opt, st, true,
                  SegSel{0, 1, 0}, nullptr, true);
    if (rc != DR_OK) return rc;
    t.multi = (const int4*)(base + L.multi);
    t.long_cnt = (int32_t*)(base + L.long_cnt);
    t.loss_b = loss_b;
    t.loss_out = loss_out;
    t.loss_n = B;
    RowShape s = row_shape(k);
    bool ok = false;
#define DR_T(V_, L_)                          \
    if (!ok && s.V == V_ && s.LPR == L_) {    \
        launch_tail<V_, L_>(t, n, st);        \
        ok = true;                            \
    }
    DR_T(4, 1) DR_T(4, 2) DR_T(4, 4) DR_T(4, 8) DR_T(4, 16) DR_T(4, 32) DR_T(4, 64)
    DR_T(2, 1) DR_T(2, 2) DR_T(2, 4) DR_T(2, 8) DR_T(2, 16) DR_T(2, 32) DR_T(2, 64)
    DR_T(1, 1) DR_T(1, 2) DR_T(1, 4) DR_T(1, 8) DR_T(1, 16) DR_T(1, 32) DR_T(1, 64)
#undef DR_T
    DR_REQUIRE(ok, "bpr tail: unsupported k=%d", k);
    DR_CHECK_LAUNCH("bpr_tail_kernel");
    return DR_OK;
}

// ---- row-sharded step (shard.hip) ---------------------------------------------------------------------------------
// Plan over a rank's combined key list ckey [u_loc ; nuL + local item row] whose length lives on the device (n_dev[1];
// n_dev[0] = number of user entries): LSD sort + bpr_finish_kernel<FinShard>, all launches sized for the capacity cap_k.
int64_t plan_shard_bytes(int64_t cap_k, int k) {
    PlanLayout L;
    if (plan_layout(cap_k, k, &L, last_error_buf()) != 0) return -1;
    return L.total;
}
int plan_build_shard(const uint32_t* ckey, const int32_t* n_dev, const int32_t* osrc, int64_t cap_s, int64_t cap_k,
                     int64_t nuL, int64_t niL, unsigned char* solo, void* plan, int64_t plan_bytes, int k, hipStream_t st) {
    DR_REQUIRE(cap_k >= 1 && cap_k < (1ll << 31) - 4096 && nuL + niL < (1ll << 32) - 2 && 2 * cap_s + cap_k < (1ll << 31) - 1,
               "shard plan: sizes exceed 32-bit keys / positions");
    PlanLayout L;
    if (plan_layout(cap_k, k, &L, last_error_buf()) != 0) return DR_ERR_LAUNCH;
    if (plan_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "shard plan: workspace %lld < required %lld", (long long)plan_bytes, (long long)L.total);
    char* base = (char*)plan;
    int32_t* hdr = (int32_t*)(base + L.hdr);
    unsigned end_bit = 1;
    while (end_bit < 32 && ((uint64_t)(nuL + niL) >> end_bit) != 0) ++end_bit;
    int rc = lsd_sort(LsdKeysOnly{ckey}, cap_k, end_bit, (uint32_t*)(base + L.keys_in), (int32_t*)(base + L.vals_in),
                      (uint32_t*)(base + L.keys_out), (int32_t*)(base + L.vals_out), base + L.lsd, hdr, st, n_dev + 1);
    if (rc != DR_OK) return rc;
    FinShard fin{n_dev, osrc, cap_s, nuL, niL, solo};
    DR_LAUNCH_NAMED("bpr_finish_kernel", (bpr_finish_kernel<FinShard>), dim3((unsigned)((cap_k + FIN_TILE - 1) / FIN_TILE)),
                    dim3(256), 0, st, (const uint32_t*)(base + L.keys_out), (const int32_t*)(base + L.vals_out), cap_k, fin, hdr,
                    (int32_t*)(base + L.uniq), (int32_t*)(base + L.seg_start), (int32_t*)(base + L.long_seg),
                    (int32_t*)(base + L.long_cb), (int32_t*)(base + L.long_cnt), (int32_t*)(base + L.part_owner),
                    (int4*)(base + L.multi), (int32_t*)(base + L.msrc));
    DR_CHECK_LAUNCH("bpr_finish_kernel<shard>");
    return DR_OK;
}

// The sharded step's tail: bpr_tail_kernel over the plan above.  Sources: stash rows [0, cap_s) = g (q_i - q_j) of the
// local samples (user side), [cap_s, 2 cap_s) = their pre-step user rows (item side, coefficient +-g from `coef`), and
// the rows received from the peers (source indices from 2 cap_s on, coefficient 1: the sender multiplied already).
int shard_tail_launch(float* P, float* Ps0, float* Ps1, int64_t nuL, float* Q, float* Qs0, float* Qs1, int64_t niL, int k,
                      int64_t cap_s, int64_t cap_k, const float* stash, const float* recv, const float* coef, void* plan,
                      int64_t plan_bytes, const dr_opt_params* opt, hipStream_t st, const float* loss_b, float* loss_out) {
    PlanLayout L;
    if (plan_layout(cap_k, k, &L, last_error_buf()) != 0) return DR_ERR_LAUNCH;
    if (plan_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "shard tail: plan workspace %lld < required %lld", (long long)plan_bytes, (long long)L.total);
    char* base = (char*)plan;
    TailArgs t;
    int rc = seg_fill(t.q, L, base, Q, Qs0, Qs1, niL, k, stash, 2 * cap_s, nullptr, coef, cap_k, plan, opt->kind, opt, st, true,
                      SegSel{(uint32_t)nuL, 2, 0}, recv, true);
    if (rc != DR_OK) return rc;
    rc = seg_fill(t.p, L, base, P, Ps0, Ps1, nuL, k, stash, 2 * cap_s, nullptr, nullptr, cap_k, plan, opt->kind, opt, st, true,
                  SegSel{0, 1, 0}, nullptr, true);
    if (rc != DR_OK) return rc;
    t.multi = (const int4*)(base + L.multi);
    t.long_cnt = (int32_t*)(base + L.long_cnt);
    t.loss_b = loss_b;
    t.loss_out = loss_out;
    t.loss_n = cap_s;
    RowShape s = row_shape(k);
    bool ok = false;
#define DR_T(V_, L_)                          \
    if (!ok && s.V == V_ && s.LPR == L_) {    \
        launch_tail<V_, L_>(t, cap_k, st);    \
        ok = true;                            \
    }
    DR_T(4, 1) DR_T(4, 2) DR_T(4, 4) DR_T(4, 8) DR_T(4, 16) DR_T(4, 32) DR_T(4, 64)
    DR_T(2, 1) DR_T(2, 2) DR_T(2, 4) DR_T(2, 8) DR_T(2, 16) DR_T(2, 32) DR_T(2, 64)
    DR_T(1, 1) DR_T(1, 2) DR_T(1, 4) DR_T(1, 8) DR_T(1, 16) DR_T(1, 32) DR_T(1, 64)
#undef DR_T
    DR_REQUIRE(ok, "shard tail: unsupported k=%d", k);
    DR_CHECK_LAUNCH("bpr_tail_kernel<shard>");
    return DR_OK;
}

int check_opt(const dr_opt_params* opt, const float* slot0, const float* slot1, const char* who) {
    DR_REQUIRE(opt != nullptr, "%s: opt is NULL", who);
    DR_REQUIRE(opt->kind >= DR_OPT_SGD && opt->kind <= DR_OPT_RMSPROP, "%s: bad optimizer kind %d", who, opt->kind);
    DR_REQUIRE(opt->reg_mode >= DR_REG_TOUCHED && opt->reg_mode <= DR_REG_NONE, "%s: bad reg_mode %d", who,
               opt->reg_mode);
    if (opt->kind != DR_OPT_SGD) DR_REQUIRE(slot0 != nullptr, "%s: optimizer kind %d needs slot0", who, opt->kind);
    if (opt->kind == DR_OPT_ADAM || opt->kind == DR_OPT_RMSPROP)
        DR_REQUIRE(slot1 != nullptr, "%s: optimizer kind %d needs slot1", who, opt->kind);
    return DR_OK;
}
}  // namespace dr

extern "C" int32_t dr_scatter_add_sorted_f32(float* gtable, int64_t rows, int32_t k, const float* src,
                                             int64_t src_rows, const int32_t* src_idx, const float* coef, int64_t n,
                                             void* plan, int64_t plan_bytes, void* stream) {
    DR_REQUIRE(rows >= 0 && k >= 1 && n >= 0 && src_rows >= 0, "dr_scatter_add_sorted_f32: bad sizes");
    if (n == 0) return DR_OK;
    DR_REQUIRE(gtable && src && plan, "dr_scatter_add_sorted_f32: NULL pointer");
    return seg_launch(gtable, nullptr, nullptr, rows, k, src, src_rows, src_idx, coef, n, plan, plan_bytes, KIND_ACC,
                      nullptr, (hipStream_t)stream, false, SegSel{0, 0, 0}, 0, nullptr, false);
}

extern "C" int32_t dr_sparse_row_update_f32(float* table, float* slot0, float* slot1, int64_t rows, int32_t k,
                                            const float* src, int64_t src_rows, const int32_t* src_idx,
                                            const float* coef, int64_t n, void* plan, int64_t plan_bytes,
                                            const dr_opt_params* opt, void* stream) {
    DR_REQUIRE(rows >= 0 && k >= 1 && n >= 0 && src_rows >= 0, "dr_sparse_row_update_f32: bad sizes");
    int rc = check_opt(opt, slot0, slot1, "dr_sparse_row_update_f32");
    if (rc != DR_OK) return rc;
    DR_REQUIRE(opt->reg_mode != DR_REG_DENSE,
               "dr_sparse_row_update_f32: DR_REG_DENSE is served by dr_scatter_add_sorted_f32 + dr_dense_update_f32");
    if (n == 0) return DR_OK;
    DR_REQUIRE(table && src && plan, "dr_sparse_row_update_f32: NULL pointer");
    return seg_launch(table, slot0, slot1, rows, k, src, src_rows, src_idx, coef, n, plan, plan_bytes, opt->kind, opt,
                      (hipStream_t)stream, false, SegSel{0, 0, 0}, 0, nullptr, false);
}

#ifndef DR_DENSE_NT_BYTES
#define DR_DENSE_NT_BYTES (128ll << 20)
#endif
static int32_t dense_update_launch(const char* who, float* w, float* slot0, float* slot1, const float* grad,
                                   int64_t count, const dr_opt_params* opt, const uint8_t* skip, int k, void* stream) {
    DR_REQUIRE(count >= 0, "%s: negative count", who);
    int rc = check_opt(opt, slot0, slot1, who);
    if (rc != DR_OK) return rc;
    if (count == 0) return DR_OK;
    DR_REQUIRE(w != nullptr, "%s: w is NULL", who);
    OptP o = make_optp(opt);
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = count % 4 == 0 && k % 4 == 0 &&
                    (((uintptr_t)w | (uintptr_t)slot0 | (uintptr_t)slot1 | (uintptr_t)grad) % 16 == 0);
    if (v4) {
        const int64_t c4 = count / 4;
        const int grid = grid_for(c4, 256, 2);
        const bool nt = DR_DENSE_NT_BYTES > 0 && count * 4 >= (int64_t)DR_DENSE_NT_BYTES;
#define DR_DU4(KIND_)                                                                                              \
    if (nt)                                                                                                        \
        DR_LAUNCH((dense_update4_kernel<KIND_, true>), dim3(grid), dim3(256), 0, st, (float4*)w, (float4*)slot0,   \
                  (float4*)slot1, (const float4*)grad, c4, o, skip, k / 4);                                        \
    else                                                                                                           \
        DR_LAUNCH((dense_update4_kernel<KIND_>), dim3(grid), dim3(256), 0, st, (float4*)w, (float4*)slot0,         \
                  (float4*)slot1, (const float4*)grad, c4, o, skip, k / 4)
        switch (opt->kind) {
            case DR_OPT_SGD: DR_DU4(DR_OPT_SGD); break;
            case DR_OPT_ADAM: DR_DU4(DR_OPT_ADAM); break;
            case DR_OPT_ADAGRAD: DR_DU4(DR_OPT_ADAGRAD); break;
            default: DR_DU4(DR_OPT_RMSPROP); break;
        }
#undef DR_DU4
        DR_CHECK_LAUNCH(who);
        return DR_OK;
    }
    const int grid = grid_for(count, 256, 4);
#define DR_DU1(KIND_) \
    DR_LAUNCH((dense_update_kernel<KIND_>), dim3(grid), dim3(256), 0, st, w, slot0, slot1, grad, count, o, skip, k)
    switch (opt->kind) {
        case DR_OPT_SGD: DR_DU1(DR_OPT_SGD); break;
        case DR_OPT_ADAM: DR_DU1(DR_OPT_ADAM); break;
        case DR_OPT_ADAGRAD: DR_DU1(DR_OPT_ADAGRAD); break;
        default: DR_DU1(DR_OPT_RMSPROP); break;
    }
#undef DR_DU1
    DR_CHECK_LAUNCH(who);
    return DR_OK;
}

extern "C" int32_t dr_dense_update_f32(float* w, float* slot0, float* slot1, const float* grad, int64_t count,
                                       const dr_opt_params* opt, void* stream) {
    return dense_update_launch("dr_dense_update_f32", w, slot0, slot1, grad, count, opt, nullptr, 4, stream);
}

extern "C" int32_t dr_dense_update_masked_f32(float* table, float* slot0, float* slot1, int64_t rows, int32_t k,
                                              const uint8_t* skip, const dr_opt_params* opt, void* stream) {
    DR_REQUIRE(rows >= 0 && k >= 1, "dr_dense_update_masked_f32: bad sizes");
    DR_REQUIRE(rows == 0 || skip != nullptr, "dr_dense_update_masked_f32: skip is NULL");
    return dense_update_launch("dr_dense_update_masked_f32", table, slot0, slot1, nullptr, rows * (int64_t)k, opt, skip,
                               (int)k, stream);
}

__global__ __launch_bounds__(256) void mark_rows_kernel(uint8_t* __restrict__ flags, int64_t rows,
                                                        const int32_t* __restrict__ ids, int64_t n, uint8_t value) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t r = ids[e];
        if (r >= 0 && r < rows) flags[r] = value;      // duplicates write the same byte
    }
}

extern "C" int32_t dr_mark_rows_u8(uint8_t* flags, int64_t rows, const int32_t* ids, int64_t n, int32_t value,
                                   void* stream) {
    DR_REQUIRE(rows >= 0 && n >= 0, "dr_mark_rows_u8: bad sizes");
    if (n == 0) return DR_OK;
    DR_REQUIRE(flags && ids, "dr_mark_rows_u8: NULL pointer");
    DR_LAUNCH(mark_rows_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, flags, rows, ids, n,
              (uint8_t)(value != 0));
    DR_CHECK_LAUNCH("dr_mark_rows_u8");
    return DR_OK;
}

extern "C" int32_t dr_rownorm_clip_f32(float* table, int64_t rows, int32_t k, float clip, void* stream) {
    DR_REQUIRE(rows >= 0 && k >= 1 && clip > 0.0f, "dr_rownorm_clip_f32: bad arguments");
    if (rows == 0) return DR_OK;
    DR_REQUIRE(table != nullptr, "dr_rownorm_clip_f32: table is NULL");
    RowShape s = row_shape(k);
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
#define DR_C(V_, L_)                                                                                            \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                      \
        DR_LAUNCH((rownorm_clip_kernel<V_, L_>), dim3(grid_for(rows, 1, 256 / L_)), dim3(256), 0, st,  \
                           table, rows, k, clip);                                                               \
        ok = true;                                                                                              \
    }
    DR_C(4, 1) DR_C(4, 2) DR_C(4, 4) DR_C(4, 8) DR_C(4, 16) DR_C(4, 32) DR_C(4, 64)
    DR_C(2, 1) DR_C(2, 2) DR_C(2, 4) DR_C(2, 8) DR_C(2, 16) DR_C(2, 32) DR_C(2, 64)
    DR_C(1, 1) DR_C(1, 2) DR_C(1, 4) DR_C(1, 8) DR_C(1, 16) DR_C(1, 32) DR_C(1, 64)
#undef DR_C
    DR_REQUIRE(ok, "dr_rownorm_clip_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("dr_rownorm_clip_f32");
    return DR_OK;
}

// ---------------------------------------------------------------------------------------
// stable bucketing of small keys (routing a batch to the ranks that own its rows)
// ---------------------------------------------------------------------------------------
namespace dr {
struct BucketLayout {
    int64_t keys_in, keys_out, vals_in, temp, temp_bytes, total;
};
static int bucket_layout(int64_t n, BucketLayout* L) {
    size_t t_sort = 0;
    const int64_t nn = n > 0 ? n : 1;
    hipError_t e = rocprim::radix_sort_pairs<rocprim::default_config, uint32_t*, uint32_t*, int32_t*, int32_t*>(
        nullptr, t_sort, nullptr, nullptr, nullptr, nullptr, (size_t)nn, 0, 32, 0, false);
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
    L->temp_bytes = (int64_t)t_sort + 256;
    L->temp = take(L->temp_bytes);
    L->total = off;
    return 0;
}

// keys out of range go to the extra bucket nkeys (never produced by the sharding code; kept out of the counts)
__global__ __launch_bounds__(256) void bucket_keys_kernel(const int32_t* __restrict__ keys, int64_t n, int nkeys,
                                                          uint32_t* __restrict__ kout, int32_t* __restrict__ vout,
                                                          long long* __restrict__ counts) {
    __shared__ int hist[257];
    for (int b = threadIdx.x; b <= nkeys; b += 256) hist[b] = 0;
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int kx = keys[e];
        const int b = (kx >= 0 && kx < nkeys) ? kx : nkeys;
        kout[e] = (uint32_t)b;
        vout[e] = (int32_t)e;
        atomicAdd(&hist[b], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nkeys; b += 256)
        if (hist[b]) atomicAdd((unsigned long long*)&counts[b], (unsigned long long)hist[b]);
}
}  // namespace dr

extern "C" int64_t dr_bucket_order_bytes(int64_t n) {
    if (n < 0) return DR_ERR_ARG;
    BucketLayout L;
    if (bucket_layout(n, &L) != 0) return DR_ERR_LAUNCH;
    return L.total;
}

extern "C" int32_t dr_bucket_order_i32(const int32_t* keys, int64_t n, int32_t nkeys, int32_t* order, int64_t* counts,
                                       void* ws, int64_t ws_bytes, void* stream) {
    DR_REQUIRE(n >= 0 && n < (1ll << 31) - 1 && nkeys >= 1 && nkeys <= 256, "dr_bucket_order_i32: bad sizes (nkeys in [1,256])");
    DR_REQUIRE(counts != nullptr, "dr_bucket_order_i32: counts is NULL");
    hipStream_t st = (hipStream_t)stream;
    DR_HIP(hipMemsetAsync(counts, 0, (size_t)nkeys * 8, st));
    if (n == 0) return DR_OK;
    DR_REQUIRE(keys && order && ws && ((uintptr_t)ws % 256 == 0), "dr_bucket_order_i32: NULL / misaligned pointer");
    BucketLayout L;
    if (bucket_layout(n, &L) != 0) DR_FAIL(DR_ERR_LAUNCH, "dr_bucket_order_i32: rocprim size query failed");
    if (ws_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_bucket_order_i32: workspace %lld < required %lld", (long long)ws_bytes,
                (long long)L.total);
    char* base = (char*)ws;
    uint32_t* kin = (uint32_t*)(base + L.keys_in);
    uint32_t* kout = (uint32_t*)(base + L.keys_out);
    int32_t* vin = (int32_t*)(base + L.vals_in);
    DR_LAUNCH(bucket_keys_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, keys, n, (int)nkeys, kin, vin,
              (long long*)counts);
    DR_CHECK_LAUNCH("bucket_keys_kernel");
    unsigned end_bit = 1;
    while ((1u << end_bit) <= (unsigned)nkeys) ++end_bit;  // keys 0 .. nkeys
    size_t tb = (size_t)L.temp_bytes;
    DR_HIP((rocprim::radix_sort_pairs<rocprim::default_config>(base + L.temp, tb, kin, kout, vin, order, (size_t)n, 0,
                                                               end_bit, st, false)));
    return DR_OK;
}
