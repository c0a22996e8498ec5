// K7, round 3: the row-sharded BPR step as the FUSED step (deeprec_amd/sharding.py; the reference is single-device,
// bprmf.py:114-123 is the step).  One process per GPU, tables dealt round-robin (owner = row mod W), ids replicated by
// one all-gather.  A rank has two roles and handles both with the single-GPU step's machinery:
//   * it PROCESSES the samples whose user row it owns (n_s of the W B samples, in global batch order), and
//   * it OWNS the item rows item mod W == me: n_g (part, sample) entries of the global batch in canonical order
//     (i-part before j-part, then global batch position) -- whoever processes the sample.
// Its combined key list [user rows of its samples ; nuL + its item rows] is exactly the list the single-GPU step plans
// (and IS that list at W = 1), so the same plan (LSD sort + bpr_finish_kernel), the same tail kernels and a forward
// kernel of the same shape apply.  What W > 1 adds:
//   * an item row of a processed sample that lives elsewhere is read from the fetch buffer (distinct rows per owner,
//     ascending: one all-to-all-v), its gradient row (+-g) p_u is written by the forward kernel straight into the
//     send buffer at its wire position (second all-to-all-v);
//   * an owned entry whose sample a peer processes takes its contribution from the receive buffer with coefficient 1
//     (the sender multiplied: the same fp32 product the owner would have formed), in canonical order like any other.
// Single-contribution rows whose one contribution is local are applied by the forward kernel; everything else by the
// tail.  Same blocked position order as one process over the concatenated batch => tables bit-identical at every W.
//
// Route (ids only, a batch ahead): shard_count_kernel + shard_compact_kernel (stable compaction of the four lists a
// rank needs, one thread per global sample -- streaming 12 W B bytes, no sort of foreign keys), at W > 1 one LSD sort
// of the <= 4 B (W-1)/W remote entries of both roles + shard_vcount / shard_vpack (distinct rows, wire positions,
// split sizes), then the plan.  Every list length stays on the device; launches are sized for the capacities.
#include "dr_common.h"
#include "dr_lsd_sort.h"
#include "dr_segment.h"

namespace dr {

int64_t plan_shard_bytes(int64_t cap_k, int k);
int plan_build_shard(const uint32_t* ckey, const int32_t* n_dev, const int32_t* osrc, int64_t cap_s, int64_t cap_k,
                     int64_t nuL, int64_t niL, unsigned char* solo, void* plan, int64_t plan_bytes, int k, hipStream_t st);
int shard_tail_launch(float* P, float* Ps0, float* Ps1, int64_t nuL, float* Q, float* Qs0, float* Qs1, int64_t niL, int k,
                      int64_t cap_s, int64_t cap_k, const float* stash, const float* recv, const float* coef, void* plan,
                      int64_t plan_bytes, const dr_opt_params* opt, hipStream_t st, const float* loss_b, float* loss_out);
int check_opt(const dr_opt_params* opt, const float* slot0, const float* slot1, const char* who);

constexpr int SH_MAX_WORLD = 64;
constexpr int RT_PER = 8, RT_TILE = 256 * RT_PER;     // samples per workgroup of the route kernels
constexpr int VT_PER = 16, VT_TILE = 256 * VT_PER;    // sorted view entries per workgroup
// counts (int32, device; copied to the host for the split sizes)
enum { C_NS = 0, C_NK = 1, C_NV0 = 2, C_NV = 3, C_BAD = 4, C_OVER = 5, C_NG = 6, C_NV1 = 7, C_LISTS = 8 };
// then W entries each: send_f, send_g (view 0, per owner), recv_f, recv_g (view 1, per processing rank), hist_p, hist_o

struct ShardLayout {
    int64_t counts, tile_counts, ckey, osrc, coef, ref, ow, solo, vkey, vpay, vkey_t, vpay_t, vkey_o, vpay_o, vtile, vlsd,
        serve, loss_b, stash, plan, plan_bytes, total;
    int64_t cap_s, cap_g, cap_k, cap_v, tiles, vtiles;
    int cw;
};

static int shard_layout(int world, int64_t B, int k, int64_t cap_s, int64_t cap_g, ShardLayout* L) {
    const int64_t WB = (int64_t)world * B;
    L->cap_s = cap_s;
    L->cap_g = cap_g;
    L->cap_k = cap_s + cap_g;
    L->cap_v = world > 1 ? 2 * cap_s + cap_g : 0;
    L->tiles = (WB + RT_TILE - 1) / RT_TILE;
    L->vtiles = (L->cap_v + VT_TILE - 1) / VT_TILE;
    L->cw = 8 + 2 * world;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += align_up(bytes > 0 ? bytes : 1, 256);
        return o;
    };
    L->counts = take((C_LISTS + 6 * world) * 4);
    L->tile_counts = take((L->tiles + 1) * L->cw * 4);
    L->ckey = take(L->cap_k * 4);
    L->osrc = take(L->cap_k * 4);
    L->coef = take(L->cap_k * 4);
    L->ref = take(2 * cap_s * 4);
    L->ow = take(2 * cap_s * 4);
    L->solo = take(3 * cap_s);
    L->vkey = take(L->cap_v * 4);
    L->vpay = take(L->cap_v * 4);
    L->vkey_t = take(L->cap_v * 4);
    L->vpay_t = take(L->cap_v * 4);
    L->vkey_o = take(L->cap_v * 4);
    L->vpay_o = take(L->cap_v * 4);
    L->vtile = take((L->vtiles + 1) * 2 * 4);
    L->vlsd = take(L->cap_v > 0 ? lsd_scratch_bytes(L->cap_v) : 0);
    L->serve = take(cap_g * 4);
    L->loss_b = take(cap_s * 4);
    L->stash = take(2 * cap_s * (int64_t)k * 4);
    L->plan_bytes = plan_shard_bytes(L->cap_k, k);
    if (L->plan_bytes < 0) return -1;
    L->plan = take(L->plan_bytes);
    L->total = off;
    return 0;
}

struct RouteArgs {
    const int32_t* ids;   // [W][3][B]; NULL at world 1: the three lists below
    const int32_t *u1, *i1, *j1;
    int64_t B, WB, nu, ni;
    int W, me;
    uint32_t rpo;         // rows per owner (upper bound): view keys = major * rpo + local row
    int64_t nuL;
    int64_t cap_s, cap_g, cap_k, cap_v;
    int tiles, cw;
};

struct Sample {
    int64_t u, i, j;
    bool ok;
};
__device__ __forceinline__ Sample read_sample(const RouteArgs& a, int64_t g) {
    Sample x;
    if (a.ids) {
        const int64_t s = g / a.B, pos = g - s * a.B;
        const int32_t* base = a.ids + s * 3 * a.B;
        x.u = base[pos];
        x.i = base[a.B + pos];
        x.j = base[2 * a.B + pos];
    } else {
        x.u = a.u1[g];
        x.i = a.i1[g];
        x.j = a.j1[g];
    }
    x.ok = x.u >= 0 && x.u < a.nu && x.i >= 0 && x.i < a.ni && x.j >= 0 && x.j < a.ni;
    return x;
}
// the seven list counters of one sample: S, O_i, O_j, V0_i, V0_j, V1_i, V1_j
__device__ __forceinline__ void sample_flags(const RouteArgs& a, const Sample& x, int (&f)[7]) {
    const int p = (int)(x.u % a.W), oi = (int)(x.i % a.W), oj = (int)(x.j % a.W);
    const bool mine = p == a.me;
    f[0] = mine;
    f[1] = oi == a.me;
    f[2] = oj == a.me;
    f[3] = mine && oi != a.me;
    f[4] = mine && oj != a.me;
    f[5] = !mine && oi == a.me;
    f[6] = !mine && oj == a.me;
}

// ---- route, launch 1: per tile of RT_TILE global samples, how many entries each list gets -----------------------------
__global__ __launch_bounds__(256) void shard_count_kernel(RouteArgs a, int32_t* __restrict__ tile_counts) {
    __shared__ int hist[2 * SH_MAX_WORLD];
    __shared__ int red[4][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = tid; b < 2 * a.W; b += 256) hist[b] = 0;
    __syncthreads();
    int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t g0 = (int64_t)blockIdx.x * RT_TILE + (int64_t)tid * RT_PER;
#pragma unroll
    for (int q = 0; q < RT_PER; ++q) {
        const int64_t g = g0 + q;
        if (g >= a.WB) break;
        const Sample x = read_sample(a, g);
        if (!x.ok) {
            ++c[7];
            continue;
        }
        int f[7];
        sample_flags(a, x, f);
#pragma unroll
        for (int z = 0; z < 7; ++z) c[z] += f[z];
        atomicAdd(&hist[(int)(x.u % a.W)], 1);               // samples per processing rank
        atomicAdd(&hist[a.W + (int)(x.i % a.W)], 1);         // item entries per owner
        atomicAdd(&hist[a.W + (int)(x.j % a.W)], 1);
    }
#pragma unroll
    for (int z = 0; z < 8; ++z) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c[z] += __shfl_xor(c[z], o, 64);
        if (lane == 0) red[wave][z] = c[z];
    }
    __syncthreads();
    int32_t* out = tile_counts + (int64_t)blockIdx.x * a.cw;
    if (tid < 8) out[tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    for (int b = tid; b < 2 * a.W; b += 256) out[8 + b] = hist[b];
}

// ---- route, launch 2: stable compaction into the rank's lists --------------------------------------------------------
struct CompactOut {
    int32_t* counts;
    uint32_t* ckey;     // [cap_k]  combined key list: [u_loc ; nuL + local item row]
    int32_t* osrc;      // [cap_k]  item entries: 2 s + slot (local sample) or -1 - y (receive-buffer row)
    float* coef;        // [cap_k]  item entries: +-g (written by the forward kernel) or 1 (received rows)
    int32_t* ref;       // [2][cap_s] item row of (slot, s): local row, or niL + fetch-buffer row
    int32_t* ow;        // [2][cap_s] position of the entry in the combined list, or -1 - wire position in the send buffer
    unsigned char* solo;  // [3][cap_s]
    uint32_t* vkey;     // [cap_v]  remote entries of both roles: view 0 (I process, a peer owns) then view 1
    int32_t* vpay;
};
__global__ __launch_bounds__(256) void shard_compact_kernel(RouteArgs a, const int32_t* __restrict__ tile_counts,
                                                            CompactOut o) {
    __shared__ int tot[8 + 2 * SH_MAX_WORLD], pre[8];
    __shared__ int wsum[4][7];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // column sums of the tile counters: all tiles (totals) and the tiles before this one
    if (tid < a.cw) {
        int t_all = 0, t_pre = 0;
        for (int t0 = 0; t0 < a.tiles; t0 += 8) {
            int v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = tile_counts[(int64_t)((t0 + q < a.tiles) ? t0 + q : a.tiles - 1) * a.cw + tid];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (t0 + q < a.tiles) t_all += v[q];
                if (t0 + q < (int)blockIdx.x) t_pre += v[q];
            }
        }
        tot[tid] = t_all;
        if (tid < 8) pre[tid] = t_pre;
    }
    __syncthreads();
    // A rank over its capacity is seen by every rank (the histograms are global): all refuse the batch together, and no
    // list is written at all (their lengths are reported as zero, so every later launch of this route is a no-op).
    __shared__ int over_s;
    if (tid == 0) {
        int over = 0;
        for (int r = 0; r < a.W; ++r) over |= (tot[8 + r] > a.cap_s || tot[8 + a.W + r] > a.cap_g) ? 1 : 0;
        over_s = over;
    }
    __syncthreads();
    const bool over = over_s != 0;
    const int64_t n_s = tot[0], n_g = (int64_t)tot[1] + tot[2];
    const int64_t n_v0 = (int64_t)tot[3] + tot[4], n_v1 = (int64_t)tot[5] + tot[6];
    if (blockIdx.x == 0) {
        if (tid == 0) {
            o.counts[C_NS] = over ? 0 : (int32_t)n_s;
            o.counts[C_NK] = over ? 0 : (int32_t)(n_s + n_g);
            o.counts[C_NV0] = over ? 0 : (int32_t)n_v0;
            o.counts[C_NV] = over ? 0 : (int32_t)(n_v0 + n_v1);
            o.counts[C_BAD] = tot[7];
            o.counts[C_OVER] = over ? 1 : 0;
            o.counts[C_NG] = over ? 0 : (int32_t)n_g;
            o.counts[C_NV1] = over ? 0 : (int32_t)n_v1;
        }
        for (int b = tid; b < 4 * a.W; b += 256) o.counts[C_LISTS + b] = 0;           // shard_vpack_kernel adds into these
        for (int b = tid; b < 2 * a.W; b += 256) o.counts[C_LISTS + 4 * a.W + b] = tot[8 + b];
    }
    if (over) return;   // block-uniform
    // this thread's RT_PER consecutive samples
    const int64_t g0 = (int64_t)blockIdx.x * RT_TILE + (int64_t)tid * RT_PER;
    Sample x[RT_PER];
    int f[RT_PER][7];
    int c[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < RT_PER; ++q) {
#pragma unroll
        for (int z = 0; z < 7; ++z) f[q][z] = 0;
        x[q].ok = false;
        if (g0 + q < a.WB) {
            x[q] = read_sample(a, g0 + q);
            if (x[q].ok) sample_flags(a, x[q], f[q]);
        }
#pragma unroll
        for (int z = 0; z < 7; ++z) c[z] += f[q][z];
    }
    int base[7];
#pragma unroll
    for (int z = 0; z < 7; ++z) {
        int incl = c[z];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[wave][z] = incl;
        base[z] = incl - c[z];
    }
    __syncthreads();
#pragma unroll
    for (int z = 0; z < 7; ++z) {
        for (int w = 0; w < wave; ++w) base[z] += wsum[w][z];
        base[z] += pre[z];
    }
    // list offsets of the j-part entries: behind every i-part entry (canonical order is part-major)
    base[2] += tot[1];
    base[4] += tot[3];
    base[6] += tot[5];
    const uint32_t voff = (uint32_t)a.W * a.rpo;
#pragma unroll
    for (int q = 0; q < RT_PER; ++q) {
        if (!x[q].ok) continue;
        const int p = (int)(x[q].u % a.W);
        const bool mine = f[q][0] != 0;
        const int64_t s = base[0];
        if (mine) {
            if (s < a.cap_s) o.ckey[s] = (uint32_t)(x[q].u / a.W);
            base[0] += 1;
        }
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int64_t it = part ? x[q].j : x[q].i;
            const int own = (int)(it % a.W);
            const uint32_t row = (uint32_t)(it / a.W);
            if (own == a.me) {
                const int64_t pp = n_s + base[1 + part];
                base[1 + part] += 1;
                const bool fits = pp < a.cap_k && (pp - n_s) < a.cap_g;
                if (fits) o.ckey[pp] = (uint32_t)a.nuL + row;
                if (mine) {
                    if (fits && s < a.cap_s) {
                        o.osrc[pp] = (int32_t)(2 * s + part);
                        o.ref[part * a.cap_s + s] = (int32_t)row;
                        o.ow[part * a.cap_s + s] = (int32_t)pp;
                    }
                } else {
                    const int64_t z = n_v0 + base[5 + part];
                    base[5 + part] += 1;
                    if (fits) o.coef[pp] = 1.0f;
                    if (fits && z < a.cap_v) {
                        o.vkey[z] = voff + (uint32_t)p * a.rpo + row;
                        o.vpay[z] = (int32_t)pp;
                    }
                }
            } else if (mine) {
                const int64_t z = base[3 + part];
                base[3 + part] += 1;
                if (s < a.cap_s) {
                    o.solo[(1 + part) * a.cap_s + s] = 0;
                    if (z < a.cap_v) {
                        o.vkey[z] = (uint32_t)own * a.rpo + row;
                        o.vpay[z] = (int32_t)(2 * s + part);
                    }
                }
            }
        }
    }
}

// ---- route (W > 1), after the view sort: distinct rows, wire positions, split sizes -----------------------------------
__global__ __launch_bounds__(256) void shard_vcount_kernel(const uint32_t* __restrict__ sk, const int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ vtile) {
    __shared__ int red[4][2];
    const int n = counts[C_NV], n0 = counts[C_NV0];
    const int64_t base = (int64_t)blockIdx.x * VT_TILE;
    if (base >= n) return;
    int c0 = 0, c1 = 0;
#pragma unroll
    for (int q = 0; q < VT_PER; ++q) {
        const int64_t y = base + q * 256 + threadIdx.x;
        if (y < n && (y == 0 || sk[y] != sk[y - 1])) {
            if (y < n0) ++c0;
            else ++c1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c0 += __shfl_xor(c0, o, 64);
        c1 += __shfl_xor(c1, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6][0] = c0;
        red[threadIdx.x >> 6][1] = c1;
    }
    __syncthreads();
    if (threadIdx.x < 2) vtile[blockIdx.x * 2 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void shard_vpack_kernel(RouteArgs a, const uint32_t* __restrict__ sk,
                                                          const int32_t* __restrict__ sv, const int32_t* __restrict__ vtile,
                                                          int64_t niL, int32_t* __restrict__ counts, int32_t* __restrict__ osrc,
                                                          int32_t* __restrict__ ref, int32_t* __restrict__ ow,
                                                          int32_t* __restrict__ serve) {
    __shared__ int hist[4 * SH_MAX_WORLD];
    __shared__ int red[4][2], wtot[4][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = counts[C_NV], n0 = counts[C_NV0];
    const int64_t tile0 = (int64_t)blockIdx.x * VT_TILE;
    if (tile0 >= n) return;
    for (int b = tid; b < 4 * a.W; b += 256) hist[b] = 0;
    // distinct keys of each view that start in earlier tiles
    int b0 = 0, b1 = 0;
    for (int t = tid; t < (int)blockIdx.x; t += 256) {
        b0 += vtile[2 * t];
        b1 += vtile[2 * t + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        b0 += __shfl_xor(b0, o, 64);
        b1 += __shfl_xor(b1, o, 64);
    }
    if (lane == 0) {
        red[wave][0] = b0;
        red[wave][1] = b1;
    }
    const int64_t y0 = tile0 + (int64_t)tid * VT_PER;
    uint32_t key[VT_PER + 1];
    key[0] = (y0 > 0 && y0 - 1 < n) ? sk[y0 - 1] : 0u;
#pragma unroll
    for (int q = 0; q < VT_PER; ++q) key[q + 1] = (y0 + q < n) ? sk[y0 + q] : 0u;
    int m0 = 0, m1 = 0;
#pragma unroll
    for (int q = 0; q < VT_PER; ++q) {
        const int64_t y = y0 + q;
        if (y < n && (y == 0 || key[q + 1] != key[q])) {
            if (y < n0) ++m0;
            else ++m1;
        }
    }
    int i0 = m0, i1 = m1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v0 = __shfl_up(i0, off, 64), v1 = __shfl_up(i1, off, 64);
        if (lane >= off) {
            i0 += v0;
            i1 += v1;
        }
    }
    if (lane == 63) {
        wtot[wave][0] = i0;
        wtot[wave][1] = i1;
    }
    __syncthreads();
    int seg0 = red[0][0] + red[1][0] + red[2][0] + red[3][0] + i0 - m0;
    int seg1 = red[0][1] + red[1][1] + red[2][1] + red[3][1] + i1 - m1;
    for (int w = 0; w < wave; ++w) {
        seg0 += wtot[w][0];
        seg1 += wtot[w][1];
    }
    const uint32_t voff = (uint32_t)a.W * a.rpo;
#pragma unroll
    for (int q = 0; q < VT_PER; ++q) {
        const int64_t y = y0 + q;
        if (y >= n) break;
        const bool head = (y == 0 || key[q + 1] != key[q]);
        const int32_t pay = sv[y];
        if (y < n0) {   // view 0: I process the sample, rank `major` owns the item row
            const uint32_t kx = key[q + 1];
            const int major = (int)(kx / a.rpo);
            if (head) {
                ++seg0;
                atomicAdd(&hist[major], 1);
            }
            atomicAdd(&hist[a.W + major], 1);
            const int s = pay >> 1, part = pay & 1;
            ref[part * a.cap_s + s] = (int32_t)(niL + (seg0 - 1));      // row of the fetch buffer (owner-major, rows ascending)
            ow[part * a.cap_s + s] = (int32_t)(-1 - y);                 // wire position in the send buffer (same order)
        } else {        // view 1: I own the row, rank `major` processes the sample
            const uint32_t kx = key[q + 1] - voff;
            const int major = (int)(kx / a.rpo);
            if (head) {
                serve[seg1] = (int32_t)(kx - (uint32_t)major * a.rpo);  // rows the peers fetch from me, in their order
                ++seg1;
                atomicAdd(&hist[2 * a.W + major], 1);
            }
            atomicAdd(&hist[3 * a.W + major], 1);
            osrc[pay] = (int32_t)(-1 - (y - n0));                       // row of the receive buffer
        }
    }
    __syncthreads();
    for (int b = tid; b < 4 * a.W; b += 256)
        if (hist[b]) atomicAdd(&counts[C_LISTS + b], hist[b]);
}

// ---- the step's forward kernel: bpr_forward_update_1col_kernel with indirect item rows and a send buffer -------------
struct ShardFwd {
    SegArgs p, q;              // table / slots / rows / k / kind / opt (apply_row_loaded)
    const float* qf;           // fetched item rows (refs >= q.rows)
    const int32_t* counts;
    const int32_t* u_loc;      // = ckey[0 .. n_s)
    const int32_t* ref;
    const int32_t* ow;
    const unsigned char* solo;
    float* coef;
    float* loss_b;
    float* stash;
    float* send;
    int64_t cap_s;
};
template <int V, int LPR, int KIND>
__global__ __launch_bounds__(256) void shard_forward_kernel(ShardFwd t) {
    t.p.opt = resolve(t.p.opt);
    t.q.opt = t.p.opt;
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int k = t.p.k;
    const int64_t niL = t.q.rows, cap = t.cap_s;
    const int64_t n_s = t.counts[C_NS];
    const int nsl = (KIND == DR_OPT_ADAM) ? 2 : kind_slots(t.p.kind);
    const int64_t iters = (cap + SPB - 1) / SPB;
    for (int64_t it = blockIdx.x; it < iters; it += gridDim.x) {
        const int64_t s = it * SPB + sub;
        if (s >= n_s) {
            if (s < cap && lane == 0) t.loss_b[s] = 0.0f;   // the loss sum runs over the capacity
            continue;
        }
        const int64_t iu = t.u_loc[s];
        const int64_t ri = t.ref[s], rj = t.ref[cap + s];
        const int owi = t.ow[s], owj = t.ow[cap + s];
        const bool su = t.solo[s] != 0, si = t.solo[cap + s] != 0, sj = t.solo[2 * cap + s] != 0;
        const float* pi = (ri < niL ? t.q.table + ri * k : t.qf + (ri - niL) * k) + lane * V;
        const float* pj = (rj < niL ? t.q.table + rj * k : t.qf + (rj - niL) * k) + lane * V;
        const int64_t ou = iu * k + lane * V;
        float pu[V], qi[V], qj[V], mu[V], vu2[V], mi[V], vi2[V], mj[V], vj2[V];
        vload_nt<V>(pu, t.p.table + ou);      // rows touched once per step carry the non-temporal hint (pairwise.hip)
        vload<V>(qi, pi);
        vload<V>(qj, pj);
        if (su) {
            if (nsl >= 1) vload_nt<V>(mu, t.p.slot0 + ou);
            if (nsl >= 2) vload_nt<V>(vu2, t.p.slot1 + ou);
        }
        if (si) {   // a solo item row is a local row
            if (nsl >= 1) vload_nt<V>(mi, t.q.slot0 + ri * k + lane * V);
            if (nsl >= 2) vload_nt<V>(vi2, t.q.slot1 + ri * k + lane * V);
        }
        if (sj) {
            if (nsl >= 1) vload_nt<V>(mj, t.q.slot0 + rj * k + lane * V);
            if (nsl >= 2) vload_nt<V>(vj2, t.q.slot1 + rj * k + lane * V);
        }
        float xui = 0.0f, xuj = 0.0f;
#pragma unroll
        for (int x = 0; x < V; ++x) {
            xui = xui + pu[x] * qi[x];
            xuj = xuj + pu[x] * qj[x];
        }
        xui = subwave_sum<LPR>(xui);
        xuj = subwave_sum<LPR>(xuj);
        const float sg = 1.0f / (1.0f + expf(-(xui - xuj)));
        const float l = -logf(sg);
        const float g = -(1.0f - sg);
        const float ng = -g;
        if (lane == 0) {
            t.loss_b[s] = l;
            if (owi >= 0) t.coef[owi] = g;
            if (owj >= 0) t.coef[owj] = ng;
        }
        float d[V], tot[V];
#pragma unroll
        for (int x = 0; x < V; ++x) d[x] = g * (qi[x] - qj[x]);
        // item rows that live elsewhere: the gradient row goes to its wire position (the owner adds it with coefficient 1)
        if (owi < 0) {
#pragma unroll
            for (int x = 0; x < V; ++x) tot[x] = g * pu[x];
            vstore<V>(t.send + (int64_t)(-1 - owi) * k + lane * V, tot);
        }
        if (owj < 0) {
#pragma unroll
            for (int x = 0; x < V; ++x) tot[x] = ng * pu[x];
            vstore<V>(t.send + (int64_t)(-1 - owj) * k + lane * V, tot);
        }
        if (si) {
#pragma unroll
            for (int x = 0; x < V; ++x) tot[x] = 0.0f + (0.0f + g * pu[x]);
            apply_row_loaded<V, KIND, true>(t.q, (uint32_t)ri, lane, tot, qi, mi, vi2);
        }
        if (sj) {
#pragma unroll
            for (int x = 0; x < V; ++x) tot[x] = 0.0f + (0.0f + ng * pu[x]);
            apply_row_loaded<V, KIND, true>(t.q, (uint32_t)rj, lane, tot, qj, mj, vj2);
        }
        // pre-step copy of the user row for the tail pass (local item rows that other contributions share)
        if ((owi >= 0 && !si) || (owj >= 0 && !sj)) vstore<V>(t.stash + (cap + s) * k + lane * V, pu);
        if (su) {
#pragma unroll
            for (int x = 0; x < V; ++x) tot[x] = 0.0f + (0.0f + d[x]);
            apply_row_loaded<V, KIND, true>(t.p, (uint32_t)iu, lane, tot, pu, mu, vu2);
        } else {
            vstore<V>(t.stash + s * k + lane * V, d);
        }
    }
}

static int shard_args_ok(int32_t world, int32_t me, int64_t B, int64_t num_user, int64_t num_item, int32_t k,
                         int64_t cap_s, int64_t cap_g) {
    DR_REQUIRE(world >= 1 && world <= SH_MAX_WORLD && me >= 0 && me < world, "dr_shard_bpr: world in [1,%d], 0 <= me < world",
               SH_MAX_WORLD);
    DR_REQUIRE(B >= 1 && num_user >= 1 && num_item >= 1 && k >= 1, "dr_shard_bpr: bad sizes");
    DR_REQUIRE(cap_s >= 1 && cap_g >= 1 && cap_s <= (int64_t)world * B && cap_g <= 2 * (int64_t)world * B,
               "dr_shard_bpr: capacities out of range");
    DR_REQUIRE(4 * (int64_t)world * B < (1ll << 31) - 8192 && 2 * (num_item + world) < (1ll << 31) - 2,
               "dr_shard_bpr: batch / item table too large for 32-bit list positions");
    RowShape s = row_shape(k);
    DR_REQUIRE(k == s.V * s.LPR, "dr_shard_bpr: k=%d is not V * 2^n lanes (k in {1,2,4,8,...,256} x vector width)", k);
    return DR_OK;
}
}  // namespace dr

using namespace dr;

// Workspace of one routed batch (route lists, plan, stash, per-sample losses) for `world` ranks with B samples each;
// cap_s / cap_g: the most samples a rank may process / item entries it may own (a batch beyond them is refused).
extern "C" int64_t dr_shard_bpr_bytes(int32_t world, int64_t B, int32_t k, int64_t cap_s, int64_t cap_g) {
    if (world < 1 || world > SH_MAX_WORLD || B < 1 || k < 1 || cap_s < 1 || cap_g < 1) return DR_ERR_ARG;
    ShardLayout L;
    if (shard_layout(world, B, k, cap_s, cap_g, &L) != 0) return DR_ERR_LAUNCH;
    return L.total;
}

// Device address of one list inside the workspace (the host side needs the counts and the serve list; the others are
// there for tests): which = 0 counts (int32 [8 + 6 world]), 1 serve rows (int32 [cap_g]: local rows the peers fetch from
// this rank, peer-major, ascending), 2 combined key list (uint32 [cap_s + cap_g]), 3 entry sources (int32 [cap_s + cap_g]),
// 4 item row references (int32 [2][cap_s]), 5 list / wire positions (int32 [2][cap_s]), 6 solo flags (uint8 [3][cap_s]),
// 7 coefficients (float [cap_s + cap_g]).
extern "C" int32_t dr_shard_bpr_list(int32_t world, int64_t B, int32_t k, int64_t cap_s, int64_t cap_g, void* ws,
                                     int32_t which, void** ptr) {
    DR_REQUIRE(ws != nullptr && ptr != nullptr && which >= 0 && which <= 7, "dr_shard_bpr_list: bad arguments");
    ShardLayout L;
    if (shard_layout(world, B, k, cap_s, cap_g, &L) != 0) return DR_ERR_LAUNCH;
    const int64_t off[8] = {L.counts, L.serve, L.ckey, L.osrc, L.ref, L.ow, L.solo, L.coef};
    *ptr = (char*)ws + off[which];
    return DR_OK;
}

// Everything about a step that depends on the ids only.  ids = [world][3][B] int32 (every rank's u, i, j); at world 1
// ids may be NULL and the batch given as its three lists u, i, j [B] (no copy into a stacked buffer).  counts_host
// (may be NULL): pinned host memory that receives the counts by an asynchronous copy at the end of the route.
// counts (8 + 6 world int32, dr_shard_bpr_list which = 0): [0] samples this rank processes, [1] entries of its combined key list,
// [2] remote item entries it sends gradient rows for, [3] those plus the ones it receives, [4] samples with an id out
// of range, [5] some rank exceeds a capacity, [6] item entries it owns, [7] gradient rows it receives; then per rank:
// rows to fetch from it / gradient rows to send to it / rows it fetches from me / gradient rows it sends me /
// samples it processes / item entries it owns.  No host synchronisation.
extern "C" int32_t dr_shard_bpr_route(const int32_t* ids, const int32_t* u, const int32_t* i, const int32_t* j,
                                      int32_t world, int32_t me, int64_t B, int64_t num_user, int64_t num_item, int32_t k,
                                      int64_t cap_s, int64_t cap_g, void* ws, int64_t ws_bytes, int32_t* counts_host,
                                      void* stream) {
    int rc = shard_args_ok(world, me, B, num_user, num_item, k, cap_s, cap_g);
    if (rc != DR_OK) return rc;
    DR_REQUIRE(ws != nullptr && ((uintptr_t)ws % 256 == 0), "dr_shard_bpr_route: NULL / misaligned workspace");
    DR_REQUIRE(ids != nullptr || (world == 1 && u && i && j), "dr_shard_bpr_route: ids is NULL (u, i, j serve at world 1 only)");
    ShardLayout L;
    if (shard_layout(world, B, k, cap_s, cap_g, &L) != 0) return DR_ERR_LAUNCH;
    if (ws_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_shard_bpr_route: workspace %lld < required %lld", (long long)ws_bytes, (long long)L.total);
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    RouteArgs a;
    a.ids = ids;
    a.u1 = u;
    a.i1 = i;
    a.j1 = j;
    a.B = B;
    a.WB = (int64_t)world * B;
    a.nu = num_user;
    a.ni = num_item;
    a.W = world;
    a.me = me;
    a.rpo = (uint32_t)((num_item + world - 1) / world);
    a.nuL = (num_user - me + world - 1) / world;
    const int64_t niL = (num_item - me + world - 1) / world;
    a.cap_s = L.cap_s;
    a.cap_g = L.cap_g;
    a.cap_k = L.cap_k;
    a.cap_v = L.cap_v;
    a.tiles = (int)L.tiles;
    a.cw = L.cw;
    int32_t* counts = (int32_t*)(base + L.counts);
    int32_t* tile_counts = (int32_t*)(base + L.tile_counts);
    DR_LAUNCH_NAMED("shard_count_kernel", shard_count_kernel, dim3((unsigned)L.tiles), dim3(256), 0, st, a, tile_counts);
    DR_CHECK_LAUNCH("shard_count_kernel");
    CompactOut o;
    o.counts = counts;
    o.ckey = (uint32_t*)(base + L.ckey);
    o.osrc = (int32_t*)(base + L.osrc);
    o.coef = (float*)(base + L.coef);
    o.ref = (int32_t*)(base + L.ref);
    o.ow = (int32_t*)(base + L.ow);
    o.solo = (unsigned char*)(base + L.solo);
    o.vkey = (uint32_t*)(base + L.vkey);
    o.vpay = (int32_t*)(base + L.vpay);
    DR_LAUNCH_NAMED("shard_compact_kernel", shard_compact_kernel, dim3((unsigned)L.tiles), dim3(256), 0, st, a,
                    (const int32_t*)tile_counts, o);
    DR_CHECK_LAUNCH("shard_compact_kernel");
    if (world > 1) {
        unsigned bits = 1;
        while (bits < 32 && ((2ull * world * a.rpo) >> bits) != 0) ++bits;
        rc = lsd_sort(LsdPairs{o.vkey, o.vpay}, L.cap_v, bits, (uint32_t*)(base + L.vkey_t), (int32_t*)(base + L.vpay_t),
                      (uint32_t*)(base + L.vkey_o), (int32_t*)(base + L.vpay_o), base + L.vlsd, nullptr, st, counts + C_NV);
        if (rc != DR_OK) return rc;
        DR_LAUNCH_NAMED("shard_vcount_kernel", shard_vcount_kernel, dim3((unsigned)L.vtiles), dim3(256), 0, st,
                        (const uint32_t*)(base + L.vkey_o), (const int32_t*)counts, (int32_t*)(base + L.vtile));
        DR_CHECK_LAUNCH("shard_vcount_kernel");
        DR_LAUNCH_NAMED("shard_vpack_kernel", shard_vpack_kernel, dim3((unsigned)L.vtiles), dim3(256), 0, st, a,
                        (const uint32_t*)(base + L.vkey_o), (const int32_t*)(base + L.vpay_o), (const int32_t*)(base + L.vtile),
                        niL, counts, o.osrc, o.ref, o.ow, (int32_t*)(base + L.serve));
        DR_CHECK_LAUNCH("shard_vpack_kernel");
    }
    rc = plan_build_shard(o.ckey, counts, o.osrc, L.cap_s, L.cap_k, a.nuL, niL, o.solo, base + L.plan, L.plan_bytes, k, st);
    if (rc != DR_OK) return rc;
    if (counts_host)
        DR_HIP(hipMemcpyAsync(counts_host, counts, (size_t)(C_LISTS + 6 * world) * 4, hipMemcpyDeviceToHost, st));
    return DR_OK;
}

// Lookups, dots, loss; every row with one local contribution updated in place; gradient rows of remote items into
// `send` (counts[2] rows, wire order).  P / Q: this rank's shards (num_user / num_item are the GLOBAL sizes); fetched:
// the item rows received from their owners (sum of counts' send_f rows), NULL at world 1.
extern "C" int32_t dr_shard_bpr_forward(float* P, float* Pslot0, float* Pslot1, float* Q, float* Qslot0, float* Qslot1,
                                        int32_t world, int32_t me, int64_t B, int64_t num_user, int64_t num_item, int32_t k,
                                        int64_t cap_s, int64_t cap_g, const float* fetched, float* send,
                                        const dr_opt_params* opt, void* ws, int64_t ws_bytes, void* stream) {
    int rc = shard_args_ok(world, me, B, num_user, num_item, k, cap_s, cap_g);
    if (rc != DR_OK) return rc;
    rc = check_opt(opt, Pslot0, Pslot1, "dr_shard_bpr_forward(P)");
    if (rc != DR_OK) return rc;
    rc = check_opt(opt, Qslot0, Qslot1, "dr_shard_bpr_forward(Q)");
    if (rc != DR_OK) return rc;
    DR_REQUIRE(opt->reg_mode != DR_REG_DENSE, "dr_shard_bpr_forward: DR_REG_DENSE would stream the whole shard every step");
    DR_REQUIRE(P && Q && ws && ((uintptr_t)ws % 256 == 0), "dr_shard_bpr_forward: NULL / misaligned pointer");
    DR_REQUIRE(world == 1 || (fetched && send), "dr_shard_bpr_forward: fetched / send buffers are NULL");
    ShardLayout L;
    if (shard_layout(world, B, k, cap_s, cap_g, &L) != 0) return DR_ERR_LAUNCH;
    if (ws_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_shard_bpr_forward: workspace %lld < required %lld", (long long)ws_bytes, (long long)L.total);
    char* base = (char*)ws;
    ShardFwd t;
    memset(&t, 0, sizeof(t));
    t.p.table = P; t.p.slot0 = Pslot0; t.p.slot1 = Pslot1; t.p.k = k; t.p.kind = opt->kind;
    t.p.rows = (num_user - me + world - 1) / world;
    t.p.opt = make_optp(opt);
    t.q = t.p;
    t.q.table = Q; t.q.slot0 = Qslot0; t.q.slot1 = Qslot1;
    t.q.rows = (num_item - me + world - 1) / world;
    t.qf = fetched;
    t.counts = (const int32_t*)(base + L.counts);
    t.u_loc = (const int32_t*)(base + L.ckey);
    t.ref = (const int32_t*)(base + L.ref);
    t.ow = (const int32_t*)(base + L.ow);
    t.solo = (const unsigned char*)(base + L.solo);
    t.coef = (float*)(base + L.coef);
    t.loss_b = (float*)(base + L.loss_b);
    t.stash = (float*)(base + L.stash);
    t.send = send;
    t.cap_s = L.cap_s;
    RowShape s = row_shape(k);
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
#define DR_F(V_, L_)                                                                                                  \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                            \
        const int64_t wb_ = (L.cap_s + 256 / L_ - 1) / (256 / L_);                                                    \
        if (opt->kind == DR_OPT_ADAM)                                                                                 \
            DR_LAUNCH_NAMED("shard_forward_kernel", (shard_forward_kernel<V_, L_, DR_OPT_ADAM>),                      \
                            dim3(grid_resident(shard_forward_kernel<V_, L_, DR_OPT_ADAM>, wb_)), dim3(256), 0, st, t); \
        else                                                                                                          \
            DR_LAUNCH_NAMED("shard_forward_kernel", (shard_forward_kernel<V_, L_, KIND_ANY>),                         \
                            dim3(grid_resident(shard_forward_kernel<V_, L_, KIND_ANY>, wb_)), dim3(256), 0, st, t);    \
        ok = true;                                                                                                    \
    }
    DR_F(4, 1) DR_F(4, 2) DR_F(4, 4) DR_F(4, 8) DR_F(4, 16) DR_F(4, 32) DR_F(4, 64)
    DR_F(2, 1) DR_F(2, 2) DR_F(2, 4) DR_F(2, 8) DR_F(2, 16) DR_F(2, 32) DR_F(2, 64)
    DR_F(1, 1) DR_F(1, 2) DR_F(1, 4) DR_F(1, 8) DR_F(1, 16) DR_F(1, 32) DR_F(1, 64)
#undef DR_F
    DR_REQUIRE(ok, "dr_shard_bpr_forward: unsupported k=%d", k);
    DR_CHECK_LAUNCH("shard_forward_kernel");
    return DR_OK;
}

// Every row with two or more contributions (or one that a peer sent), both tables; received: the gradient rows from
// the peers (counts[7] rows, peer-major in wire order), NULL at world 1.  loss_out: sum of the losses of the samples
// this rank processed.
extern "C" int32_t dr_shard_bpr_finish(float* P, float* Pslot0, float* Pslot1, float* Q, float* Qslot0, float* Qslot1,
                                       int32_t world, int32_t me, int64_t B, int64_t num_user, int64_t num_item, int32_t k,
                                       int64_t cap_s, int64_t cap_g, const float* received, const dr_opt_params* opt,
                                       float* loss_out, void* ws, int64_t ws_bytes, void* stream) {
    int rc = shard_args_ok(world, me, B, num_user, num_item, k, cap_s, cap_g);
    if (rc != DR_OK) return rc;
    rc = check_opt(opt, Pslot0, Pslot1, "dr_shard_bpr_finish(P)");
    if (rc != DR_OK) return rc;
    rc = check_opt(opt, Qslot0, Qslot1, "dr_shard_bpr_finish(Q)");
    if (rc != DR_OK) return rc;
    DR_REQUIRE(P && Q && ws && ((uintptr_t)ws % 256 == 0), "dr_shard_bpr_finish: NULL / misaligned pointer");
    DR_REQUIRE(world == 1 || received, "dr_shard_bpr_finish: receive buffer is NULL");
    ShardLayout L;
    if (shard_layout(world, B, k, cap_s, cap_g, &L) != 0) return DR_ERR_LAUNCH;
    if (ws_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_shard_bpr_finish: workspace %lld < required %lld", (long long)ws_bytes, (long long)L.total);
    char* base = (char*)ws;
    const int64_t nuL = (num_user - me + world - 1) / world, niL = (num_item - me + world - 1) / world;
    return shard_tail_launch(P, Pslot0, Pslot1, nuL, Q, Qslot0, Qslot1, niL, k, L.cap_s, L.cap_k, (const float*)(base + L.stash),
                             received, (const float*)(base + L.coef), base + L.plan, L.plan_bytes, opt, (hipStream_t)stream,
                             (const float*)(base + L.loss_b), loss_out);
}
