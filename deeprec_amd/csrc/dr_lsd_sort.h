// Stable LSD radix sort of (key, list position) pairs: the sort under every segment plan (segment.hip).
// Integer-only, latency bound work on 10^4 .. 10^6 keys; no MFMA, nothing to tile for.
//
// Keys here are table rows (< 2^24 for the C4 tables): THREE 8-bit digits.  A pass is two launches:
//   lsd_hist_kernel     one workgroup per tile of 2048 keys: digit histogram of the tile (LDS atomics, order-free)
//   lsd_scatter_kernel  one workgroup per tile: adds up the tile histograms itself (digit totals + the share of the
//                       tiles before it: 512 B per tile, from L2), ranks its keys by digit with a stable in-LDS
//                       block sort of (digit, local index) -- one internal pass at 8 bits -- and writes (key,
//                       position) to their final places
// No workgroup waits for another one (no look-back, no device-scope fences).  Above LSD_INLINE_TILES tiles a third
// launch (lsd_colscan_kernel) does the column sums once instead of once per workgroup.
// The first pass reads its keys through a functor (ids -> keys on the fly), so no key list is materialised; the
// passes ping-pong between the caller's two buffer pairs and end in (keys_out, vals_out).
// Measured on MI355X (tools/probes/lsd_probe.hip, 196 608 random 24-bit keys, whole sort): 512 threads x 4 keys x 8 bits
// 43 us in 6 launches; 1024 x 4 x 12 bits 60 us in 4 launches (its scatter launch: 12 us of 384 KB histogram reads per
// workgroup + 10 us of two-pass block sort + 6 us of writes); 256 x 4 x 8: 50; 1024 x 4 x 8: 50; 256 x 8 x 8: 51.
// Beside the fused forward kernel (which fills every CU) the 512-thread workgroups find room where the 1024-thread
// ones wait for whole CUs: 5 + 12.5 us per pass inside the step against 28 + 41 us, forward kernel 100.7 vs 105.5 us.
// Replaces rocPRIM's device radix sort (4096-key block sort + 6 merge launches at 196 608 keys, 56 us).  The in-tile
// rank uses rocprim::block_radix_sort, a block-level primitive.
#pragma once
#include "dr_common.h"

#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/block/block_scan.hpp>

namespace dr {

#ifndef LSD_CFG_T
#define LSD_CFG_T 512
#define LSD_CFG_IPT 4
#define LSD_CFG_BITS 8
#endif
constexpr int LSD_T = LSD_CFG_T, LSD_IPT = LSD_CFG_IPT, LSD_TILE = LSD_T * LSD_IPT;  // keys per workgroup
constexpr int LSD_BITS = LSD_CFG_BITS, LSD_BINS = 1 << LSD_BITS;
constexpr int LSD_BPT = LSD_BINS >= LSD_T ? LSD_BINS / LSD_T : 1;   // bins a thread owns in the prologue (threads beyond the bins: none)
constexpr int LSD_INLINE_TILES = (LSD_BINS <= 256) ? 256 : 64;   // 128 KB / 512 KB of histogram rows per workgroup
#ifndef LSD_ABL
#define LSD_ABL 0   // tools/probes/lsd_probe.hip builds variants with parts removed
#endif

static inline int64_t lsd_tiles(int64_t n) { return (n + LSD_TILE - 1) / LSD_TILE; }
// scratch of one sort of n keys: tile histograms [tiles][4096] u16, then (above LSD_INLINE_TILES) their column
// prefixes [tiles][4096] u32 and the digit totals [4096] u32
static inline int64_t lsd_hist_bytes(int64_t n) { return align_up((lsd_tiles(n) + 1) * LSD_BINS * 2, 256); }
static inline int64_t lsd_scan_bytes(int64_t n) {
    return lsd_tiles(n) > LSD_INLINE_TILES ? align_up((lsd_tiles(n) + 1) * (int64_t)LSD_BINS * 4, 256) : 0;
}
static inline int64_t lsd_scratch_bytes(int64_t n) { return lsd_hist_bytes(n) + lsd_scan_bytes(n); }

// ---- key sources ----------------------------------------------------------------------------------------------
struct LsdPairs {  // a previous pass's output
    const uint32_t* keys;
    const int32_t* vals;
    __device__ __forceinline__ uint32_t key(int64_t e) const { return keys[e]; }
    __device__ __forceinline__ int32_t val(int64_t e) const { return vals[e]; }
};
struct LsdKeysOnly {  // a given key list; the payload is the list position
    const uint32_t* keys;
    __device__ __forceinline__ uint32_t key(int64_t e) const { return keys[e]; }
    __device__ __forceinline__ int32_t val(int64_t e) const { return (int32_t)e; }
};
template <typename IdT>
struct LsdIds {  // an id list against a table of `rows` rows: ids out of range -> the sentinel key `rows`
    const IdT* ids;
    int64_t rows;
    __device__ __forceinline__ uint32_t key(int64_t e) const {
        const int64_t id = (int64_t)ids[e];
        return (id < 0 || id >= rows) ? (uint32_t)rows : (uint32_t)id;
    }
    __device__ __forceinline__ int32_t val(int64_t e) const { return (int32_t)e; }
};
// combined key list of one BPR batch: [u ; num_user + i ; num_user + j], invalid ids -> num_user + num_item
struct LsdBpr {
    const int32_t *u, *i, *j;
    int64_t B, nu, ni;
    __device__ __forceinline__ uint32_t key(int64_t e) const {
        const uint32_t sentinel = (uint32_t)(nu + ni);
        if (e < B) {
            const int64_t id = u[e];
            return (id < 0 || id >= nu) ? sentinel : (uint32_t)id;
        }
        const int64_t id = (e < 2 * B) ? i[e - B] : j[e - 2 * B];
        return (id < 0 || id >= ni) ? sentinel : (uint32_t)(nu + id);
    }
    __device__ __forceinline__ int32_t val(int64_t e) const { return (int32_t)e; }
};

// ---- pass, launch 1: digit histogram of every tile --------------------------------------------------------------
// zero3: three plan-header counters that have to start at zero ride along (NULL: none)
// n_dev (may be NULL): the list length lives on the device (the launch is then sized for the capacity `n` and the
// workgroups beyond the real length leave at once)
template <class Src>
__global__ __launch_bounds__(LSD_T) void lsd_hist_kernel(Src src, int64_t n, const int32_t* __restrict__ n_dev,
                                                         unsigned shift, uint32_t mask, uint16_t* __restrict__ hist,
                                                         int32_t* __restrict__ zero3) {
    __shared__ int bins[LSD_BINS];
    const int tid = threadIdx.x;
    if (n_dev) n = *n_dev;
    if (zero3 && blockIdx.x == 0 && tid == 0) {
        zero3[1] = 0;
        zero3[2] = 0;
        zero3[4] = 0;
    }
    const int64_t base = (int64_t)blockIdx.x * LSD_TILE;
    if (base >= n) return;   // block-uniform
    for (int d = tid; d < LSD_BINS; d += LSD_T) bins[d] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < LSD_IPT; ++q) {
        const int64_t e = base + q * LSD_T + tid;
        if (e < n) atomicAdd(&bins[(src.key(e) >> shift) & mask], 1);
    }
    __syncthreads();
    for (int d = tid; d < LSD_BINS; d += LSD_T) hist[(int64_t)blockIdx.x * LSD_BINS + d] = (uint16_t)bins[d];
}

// ---- (large lists) column prefixes of the tile histograms, once -------------------------------------------------
static __global__ __launch_bounds__(256) void lsd_colscan_kernel(const uint16_t* __restrict__ hist, int ntiles,
                                                          const int32_t* __restrict__ n_dev,
                                                          uint32_t* __restrict__ colpre, uint32_t* __restrict__ total) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= LSD_BINS) return;
    if (n_dev) ntiles = (int)(((int64_t)*n_dev + LSD_TILE - 1) / LSD_TILE);
    uint32_t run = 0;
    int t = 0;
    for (; t + 8 <= ntiles; t += 8) {
        uint32_t h[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = hist[(int64_t)(t + q) * LSD_BINS + d];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            colpre[(int64_t)(t + q) * LSD_BINS + d] = run;
            run += h[q];
        }
    }
    for (; t < ntiles; ++t) {
        colpre[(int64_t)t * LSD_BINS + d] = run;
        run += hist[(int64_t)t * LSD_BINS + d];
    }
    total[d] = run;
}

// ---- pass, launch 2: ranked scatter -----------------------------------------------------------------------------
template <class Src>
__global__ __launch_bounds__(LSD_T) void lsd_scatter_kernel(Src src, int64_t n, const int32_t* __restrict__ n_dev,
                                                            unsigned shift, uint32_t mask,
                                                            unsigned bits, const uint16_t* __restrict__ hist, int ntiles,
                                                            const uint32_t* __restrict__ colpre,
                                                            const uint32_t* __restrict__ total,
                                                            uint32_t* __restrict__ keys_out,
                                                            int32_t* __restrict__ vals_out) {
    using Sort = rocprim::block_radix_sort<uint32_t, LSD_T, LSD_IPT, int32_t>;
    using Scan = rocprim::block_scan<uint32_t, LSD_T>;
    __shared__ union {
        typename Sort::storage_type sort;
        typename Scan::storage_type scan;
    } tmp;
    __shared__ uint32_t tkeys[LSD_TILE];
    __shared__ int32_t tvals[LSD_TILE];
    __shared__ uint32_t gbase[LSD_BINS];
    __shared__ uint16_t sdig[LSD_TILE];
    __shared__ uint16_t rstart[LSD_BINS];
    const int tid = threadIdx.x, tile = blockIdx.x;
    const int64_t base = (int64_t)tile * LSD_TILE;
    if (n_dev) {
        n = *n_dev;
        ntiles = (int)((n + LSD_TILE - 1) / LSD_TILE);
    }
    if (base >= n) return;   // block-uniform
    const int nt = (n - base < LSD_TILE) ? (int)(n - base) : LSD_TILE;
    // the tile's keys, blocked: thread t holds list positions 4t .. 4t+3 of the tile, so that rank order is list order.
    // Keys beyond the list get the largest digit and the largest local indices: a stable sort leaves them last.
    uint32_t dig[LSD_IPT];
    int32_t li[LSD_IPT];
#pragma unroll
    for (int q = 0; q < LSD_IPT; ++q) {
        const int l = tid * LSD_IPT + q;
        li[q] = l;
        dig[q] = mask;
        if (l < nt) {
            const uint32_t kx = src.key(base + l);
            tkeys[l] = kx;
            tvals[l] = src.val(base + l);
            dig[q] = (kx >> shift) & mask;
        }
    }
    // where every digit's keys of this tile start in the output: all keys with a smaller digit + the same digit's keys
    // in earlier tiles
    uint32_t tot[LSD_BPT], pre[LSD_BPT];
#pragma unroll
    for (int x = 0; x < LSD_BPT; ++x) tot[x] = pre[x] = 0u;
    const int d0 = tid * LSD_BPT;
    const bool own = d0 < LSD_BINS;
    if (colpre) {
        if (own) {
#pragma unroll
            for (int x = 0; x < LSD_BPT; ++x) {
                pre[x] = colpre[(int64_t)tile * LSD_BINS + d0 + x];
                tot[x] = total[d0 + x];
            }
        }
    } else if (LSD_ABL != 1 && own) {
        // 16 tile rows requested per round trip (written as one load-add per iteration this loop was a chain of `ntiles`
        // dependent L2 round trips: 18 us of a 32 us launch at 48 tiles)
        for (int t0 = 0; t0 < ntiles; t0 += 16) {
            uint16_t h[16][LSD_BPT];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int t = (t0 + q < ntiles) ? t0 + q : ntiles - 1;
#pragma unroll
                for (int x = 0; x < LSD_BPT; ++x) h[q][x] = hist[(int64_t)t * LSD_BINS + d0 + x];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t in = (t0 + q < ntiles) ? 1u : 0u, m = (t0 + q < tile) ? 1u : 0u;
#pragma unroll
                for (int x = 0; x < LSD_BPT; ++x) {
                    tot[x] += in * h[q][x];
                    pre[x] += m * h[q][x];
                }
            }
        }
    }
    uint32_t mine = 0, excl = 0;
#pragma unroll
    for (int x = 0; x < LSD_BPT; ++x) mine += tot[x];
    Scan().exclusive_scan(mine, excl, 0u, tmp.scan);
    if (own) {
#pragma unroll
        for (int x = 0; x < LSD_BPT; ++x) {
            gbase[d0 + x] = excl + pre[x];
            excl += tot[x];
        }
    }
    __syncthreads();
    if (LSD_ABL != 2) Sort().sort(dig, li, tmp.sort, 0u, bits);
#pragma unroll
    for (int q = 0; q < LSD_IPT; ++q) sdig[tid * LSD_IPT + q] = (uint16_t)dig[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < LSD_IPT; ++q) {
        const int r = tid * LSD_IPT + q;
        if (r == 0 || sdig[r - 1] != (uint16_t)dig[q]) rstart[dig[q]] = (uint16_t)r;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < LSD_IPT; ++q) {
        const int r = tid * LSD_IPT + q;
        if (li[q] < nt && LSD_ABL != 3) {
            uint32_t pos = gbase[dig[q]] + (uint32_t)(r - (int)rstart[dig[q]]);
            if (LSD_ABL != 0 && (int64_t)pos >= n) pos = (uint32_t)(n - 1);   // ablated builds compute wrong places: stay in bounds
            keys_out[pos] = tkeys[li[q]];
            vals_out[pos] = tvals[li[q]];
        }
    }
}

static inline unsigned lsd_passes(unsigned end_bit) { return end_bit <= LSD_BITS ? 1u : (end_bit + LSD_BITS - 1) / LSD_BITS; }

// Sort the n keys of `src` on bits [0, end_bit) (stable; payload = src.val).  The result lands in (keys_out, vals_out);
// (keys_tmp, vals_tmp) is the other buffer pair of the ping-pong; scratch: lsd_scratch_bytes(n), 256-byte aligned.
// zero3, n_dev: see lsd_hist_kernel (with n_dev, n is the capacity the launches are sized for).
// Launches: 2 per pass (3 above LSD_INLINE_TILES tiles).
template <class Src>
static int lsd_sort(Src src, int64_t n, unsigned end_bit, uint32_t* keys_tmp, int32_t* vals_tmp, uint32_t* keys_out,
                    int32_t* vals_out, void* scratch, int32_t* zero3, hipStream_t st, const int32_t* n_dev = nullptr) {
    if (n <= 0) return DR_OK;
    DR_REQUIRE(n < (1ll << 31) - LSD_TILE, "lsd_sort: n=%lld out of range", (long long)n);
    if (end_bit < 1) end_bit = 1;
    if (end_bit > 32) end_bit = 32;
    const unsigned passes = lsd_passes(end_bit);
    const unsigned bpp = (end_bit + passes - 1) / passes;
    const int ntiles = (int)lsd_tiles(n);
    uint16_t* hist = (uint16_t*)scratch;
    uint32_t* colpre = nullptr;
    uint32_t* total = nullptr;
    if (ntiles > LSD_INLINE_TILES) {
        colpre = (uint32_t*)((char*)scratch + lsd_hist_bytes(n));
        total = colpre + (int64_t)ntiles * LSD_BINS;
    }
    for (unsigned q = 0; q < passes; ++q) {
        const unsigned shift = q * bpp;
        const unsigned bits = (end_bit - shift < bpp) ? end_bit - shift : bpp;
        const uint32_t mask = (1u << bits) - 1u;
        const bool to_out = ((passes - 1 - q) & 1u) == 0;
        uint32_t* ko = to_out ? keys_out : keys_tmp;
        int32_t* vo = to_out ? vals_out : vals_tmp;
        if (q == 0) {
            DR_LAUNCH_NAMED("lsd_hist_kernel", (lsd_hist_kernel<Src>), dim3(ntiles), dim3(LSD_T), 0, st, src, n, n_dev, shift, mask,
                            hist, zero3);
        } else {
            LsdPairs prev{to_out ? keys_tmp : keys_out, to_out ? vals_tmp : vals_out};
            DR_LAUNCH_NAMED("lsd_hist_kernel", (lsd_hist_kernel<LsdPairs>), dim3(ntiles), dim3(LSD_T), 0, st, prev, n, n_dev, shift,
                            mask, hist, (int32_t*)nullptr);
        }
        DR_CHECK_LAUNCH("lsd_hist_kernel");
        if (colpre) {
            DR_LAUNCH_NAMED("lsd_colscan_kernel", lsd_colscan_kernel, dim3((LSD_BINS + 255) / 256), dim3(256), 0, st,
                            (const uint16_t*)hist, ntiles, n_dev, colpre, total);
            DR_CHECK_LAUNCH("lsd_colscan_kernel");
        }
        if (q == 0) {
            DR_LAUNCH_NAMED("lsd_scatter_kernel", (lsd_scatter_kernel<Src>), dim3(ntiles), dim3(LSD_T), 0, st, src, n, n_dev, shift,
                            mask, bits, (const uint16_t*)hist, ntiles, (const uint32_t*)colpre, (const uint32_t*)total, ko, vo);
        } else {
            LsdPairs prev{to_out ? keys_tmp : keys_out, to_out ? vals_tmp : vals_out};
            DR_LAUNCH_NAMED("lsd_scatter_kernel", (lsd_scatter_kernel<LsdPairs>), dim3(ntiles), dim3(LSD_T), 0, st, prev, n, n_dev,
                            shift, mask, bits, (const uint16_t*)hist, ntiles, (const uint32_t*)colpre, (const uint32_t*)total, ko, vo);
        }
        DR_CHECK_LAUNCH("lsd_scatter_kernel");
    }
    return DR_OK;
}

}  // namespace dr
