// Caser's horizontal convolution block (reference: models/seq_rec/Caser.py:132-154) -- SURVEY 8f rank 3, row a16.
//
// For every filter height h = 1 .. L the reference runs conv2d(kernel (h, d), C filters) over e = Q[item_seq] [B, L, d],
// ReLU, and a max over the L - h + 1 positions.  All heights share ONE filter matrix here, W [C L (L + 1) / 2, d] with
// row C h (h - 1) / 2 + c h + r = row r of filter c of height h, so that
//   * the products are one library GEMM per column chunk, G = e W^T (G[b, p, (h, c, r)] = <e[b, p], W[h, c, r]>);
//   * hconv_pool_fwd_kernel turns G into the pooled output: s[t] = sum_r G[b, t + r, (h, c, r)] + bias, out = max(0,
//     max_t s[t]), arg = the winning position (-1: the ReLU is inactive);
//   * the backward never materialises dG: only the h rows of the winning window of each (b, h, c) carry gradient, so
//     d e (hconv_bwd_input_kernel) and d W (hconv_bwd_filter_kernel) are sums of C L (L + 1) / 2 scaled rows per
//     sequence -- 1 / (2 L) of the arithmetic of the two dense backward GEMMs.
// Under autograd the strided window sums went through index_add (atomics) in the backward: 63 of 197 ms per step at
// B = 4096, L = 50, d = 128.  All sums here have a fixed order.
#include "dr_common.h"

namespace dr {

constexpr int HC_MAX_L = 64;
constexpr int HC_MAX_D = 256;
constexpr int HC_MAX_C = 64;

__host__ __device__ static inline int64_t hc_off(int C, int h) { return (int64_t)C * h * (h - 1) / 2; }

// Pooling over a column chunk of G that holds heights h0 .. h1.  One workgroup per (sequence, height, band of filters):
// the band's [L, cb * h] slab of G (cb = as many filters as fit HC_BAND floats per row) is read coalesced into LDS (odd
// row stride), one thread per (filter, position) sums its window's diagonal, one thread per filter takes the maximum.
constexpr int HC_BAND = 120;   // 24 KB of LDS per workgroup at L = 50: six workgroups per CU hide the slab loads' latency

__host__ __device__ static inline int hc_cb(int C, int h) {
    int cb = HC_BAND / h;
    return cb < 1 ? 1 : (cb > C ? C : cb);
}
__host__ __device__ static inline int hc_bands(int C, int h) {
    const int cb = hc_cb(C, h);
    return (C + cb - 1) / cb;
}

__global__ __launch_bounds__(256) void hconv_pool_fwd_kernel(const float* __restrict__ G, int64_t ldg,
                                                             const float* __restrict__ bias, int64_t B, int L, int C,
                                                             int h0, int h1, float* __restrict__ out,
                                                             int32_t* __restrict__ arg) {
    extern __shared__ float lds[];
    int h = h0, band = blockIdx.x;
    while (h < h1 && band >= hc_bands(C, h)) {
        band -= hc_bands(C, h);
        ++h;
    }
    const int cbmax = hc_cb(C, h);
    const int c0 = band * cbmax;
    const int cb = (C - c0 < cbmax) ? C - c0 : cbmax;
    const int Wd = cb * h, Wp = Wd | 1, T = L - h + 1;
    float* X = lds;
    float* S2 = lds + L * Wp;      // [cb][T]
    const int64_t col = hc_off(C, h) - hc_off(C, h0) + (int64_t)c0 * h;
    const int tid = threadIdx.x;
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        __syncthreads();
        const float* src = G + b * L * ldg + col;
        // eight loads in flight per thread before the first LDS store (written as X[..] = src[..] in one loop the
        // compiler waits for each load before its store: one memory round trip per element)
        for (int base = 0; base < L * Wd; base += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid;
                const int p = idx / Wd, x = idx - p * Wd;
                v[u] = idx < L * Wd ? src[(int64_t)p * ldg + x] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid;
                const int p = idx / Wd, x = idx - p * Wd;
                if (idx < L * Wd) X[p * Wp + x] = v[u];
            }
        }
        __syncthreads();
        for (int it = tid; it < cb * T; it += 256) {
            const int c = it / T, t = it - c * T;
            const float* x = X + t * Wp + c * h;
            float sum = 0.0f;
            for (int r = 0; r < h; ++r) sum += x[r * Wp + r];
            S2[it] = sum + bias[(h - 1) * C + c0 + c];
        }
        __syncthreads();
        if (tid < cb) {
            float best = -INFINITY;
            int bt = 0;
            for (int t = 0; t < T; ++t) {
                const float v = S2[tid * T + t];
                if (v > best) {
                    best = v;
                    bt = t;
                }
            }
            const int64_t o = b * ((int64_t)C * L) + (h - 1) * C + c0 + tid;
            out[o] = best > 0.0f ? best : 0.0f;
            arg[o] = best > 0.0f ? bt : -1;
        }
    }
}

// ---- direct forward: no G --------------------------------------------------------------------------------------
// out[b, h, c] = max(0, max_t (bias + sum_{r < h} <e[b, t + r], W[h, c, r]>)) computed where it is needed only:
// L (L + 1) (L + 2) / 6 row products per (sequence, filter) instead of the L^2 (L + 1) / 2 of G = e W^T (2.9x fewer at
// L = 50), and nothing round-trips through HBM.  v_mfma_f32_16x16x4_f32 with M = 16 SEQUENCES (so every MFMA row is
// useful whatever L - h + 1 is), N = 16 filters of one height, K = the window's h d elements.
//
// One workgroup owns 16 sequences and the PAIR of heights (hA, hB = L + 1 - hA): every pair costs the same 2 hA hB row
// products per column, and its L + 1 positions split evenly over the four waves (hA's extra positions go to the first
// waves, hB's to the last: 94 % balance over all pairs at L = 50).  The wave keeps the accumulators of its positions
// (<= 16 of hA, <= 8 of hB, <= 18 together) in registers across the whole K loop: for each 16-column chunk cb the workgroup stages
// e[16 sequences, L rows, 16 columns] (L KB) into LDS with coalesced loads, then each wave runs over the filter rows r:
// one 16-byte B operand per lane from global memory (the four waves read the same rows: vector-L1 hits, requested four
// rows ahead) and one ds_read_b128 per (r, position) for A, four MFMAs each.  Measured on the way here (C5 shape): the
// same MFMA stream with operands taken from L2 by per-lane loads ran at 51 % of the fp32 MFMA peak, 95 % with the loads
// removed -- 18.7 GB of half-used cache lines and an exposed reload at every (position block, column chunk) start.
// Lane l: i = l % 16 (sequence as A row / filter as B column), q = l / 16; its float4 holds columns 16 cb + 4 q + j,
// and MFMA j contracts over q.  D: lane holds (sequence 4 q + v, filter i), v = 0..3.
typedef float hc_f32x4 __attribute__((ext_vector_type(4)));

#define HC_LD4(p) (*reinterpret_cast<const hc_f32x4*>(p))
#ifndef HC_ABL
#define HC_ABL 0   // timing ablations (dev; results wrong by construction): 1 no staging, 2 no filter-row loads, 3 no staging +
                   // no barriers, 4 no LDS reads, 5 staging reads 16 hot sequences, 6 one workgroup per CU, 7 staging
                   // without the LDS stores, 8 without the global loads
#endif
// Accumulators of a wave: its positions of one height are taken in groups of four ("quads", MFMAs of a quad interleaved
// so that no accumulator is touched twice in a row) plus a tail of up to three.  One file of four quads serves both
// heights -- the lower height's quads fill it from the front, the upper height's from the back: a wave holds at most
// (T_A + 3) / 4 + (T_B + 3) / 4 <= (L + 1) / 4 + 2 <= 18 positions, i.e. at most 4 whole quads -- and each height has its
// own three tail accumulators.  The position COUNTS are run-time values tested by uniform branches; every accumulator
// index is a compile-time constant.  (The first version of this kernel dispatched on the count with a switch over
// templates: the compiler then kept two copies of the accumulator file, ~60 v_mov_b64 around every case, 219 VGPRs.)
constexpr int HC_NQ = 4;

// positions t_start .. t_start + 4 nq + nt - 1 of height hh, one column chunk: acc += sum_r <slab row t + r, filter row r>.
// slab: the staged chunk + 4 * lane (row s at slab + 256 s); wr: the lane's filter row 0 of this chunk (row r at wr + r d).
// Quad x accumulates in accq[x] (DIR = 1) or accq[HC_NQ - 1 - x] (DIR = -1).
template <int DIR>
__device__ __forceinline__ void hc_rows(hc_f32x4 (&accq)[HC_NQ][4], hc_f32x4 (&tail)[3], const float* __restrict__ slab,
                                        const float* __restrict__ wr, int d, int hh, int t_start, int nq, int nt) {
    constexpr int RD = 4;                   // filter rows in flight (8 for waves with few positions: no faster)
    hc_f32x4 b[RD];
#pragma unroll
    for (int u = 0; u < RD; ++u) b[u] = HC_LD4(wr + (int64_t)(u < hh ? u : hh - 1) * d);
    const float* ap = slab + t_start * 256;
    const float* at = ap + 4 * nq * 256;    // the tail's first position
    for (int r0 = 0; r0 < hh; r0 += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            if (r0 + u < hh) {
                const float* ar = ap + (r0 + u) * 256;
#pragma unroll
                for (int x = 0; x < HC_NQ; ++x)
                    if (x < nq) {
                        hc_f32x4 a[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#if HC_ABL == 4
                            a[k] = b[(u + k) % RD];
#else
                            a[k] = HC_LD4(ar + (4 * x + k) * 256);
#endif
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                hc_f32x4& c = accq[DIR > 0 ? x : HC_NQ - 1 - x][k];
                                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][j], b[u][j], c, 0, 0, 0);
                            }
                    }
                if (nt > 0) {
                    const float* atr = at + (r0 + u) * 256;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (k < nt) {
                            const hc_f32x4 a = HC_LD4(atr + k * 256);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                tail[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[u][j], tail[k], 0, 0, 0);
                        }
                }
            }
            const int rn = r0 + u + RD;     // rows past the end: clamped, unused
#if HC_ABL != 2
            b[u] = HC_LD4(wr + (int64_t)(rn < hh ? rn : hh - 1) * d);
#endif
        }
    }
}

// the wave's best position of one height -> LDS (float best, int position) per (lane, v); positions in ascending order
template <int DIR>
__device__ __forceinline__ void hc_wave_best(const hc_f32x4 (&accq)[HC_NQ][4], const hc_f32x4 (&tail)[3], int nq, int nt,
                                             int t_start, float bv, float* bests, int* bts, int wave, int lane) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float best = -INFINITY;
        int bt = 0;
#pragma unroll
        for (int x = 0; x < HC_NQ; ++x)
            if (x < nq) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float val = accq[DIR > 0 ? x : HC_NQ - 1 - x][k][v] + bv;
                    if (val > best) {
                        best = val;
                        bt = t_start + 4 * x + k;
                    }
                }
            }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < nt) {
                const float val = tail[k][v] + bv;
                if (val > best) {
                    best = val;
                    bt = t_start + 4 * nq + k;
                }
            }
        bests[(wave * 4 + v) * 64 + lane] = best;
        bts[(wave * 4 + v) * 64 + lane] = bt;
    }
}

#ifndef HC_LB
#define HC_LB 3   // workgroups per CU the register budget is cut for (LDS: L KB each)
#endif
__global__ __launch_bounds__(256, HC_LB) void hconv_direct_fwd_kernel(const float* __restrict__ e, const float* __restrict__ W,
                                                                  const float* __restrict__ bias, int64_t B, int L, int C,
                                                                  int d, int nsg, float* __restrict__ out,
                                                                  int32_t* __restrict__ arg) {
    extern __shared__ float lds[];          // [L rows][16 sequences][16 columns]; the epilogue's 16 KB reuse it
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int npairs = (L + 1) / 2;
    const int pr = npairs - 1 - (int)(blockIdx.x / nsg);        // costliest pairs (hA ~ hB) first
    const int hA = pr + 1, hB = L - pr;                         // hA <= hB; equal: the middle height of an odd L, once
    const int64_t s0 = (int64_t)(blockIdx.x % nsg) * 16;
    const int TA = L + 1 - hA, TB = (hB != hA) ? L + 1 - hB : 0;
    const int nA = TA / 4 + (wave < TA % 4 ? 1 : 0);
    const int tA = wave * (TA / 4) + (wave < TA % 4 ? wave : TA % 4);
    const int xB = 4 - TB % 4;                                  // waves xB .. 3 take hB's extra positions
    const int nB = TB / 4 + ((TB % 4) && wave >= xB ? 1 : 0);
    const int tB = wave * (TB / 4) + ((TB % 4) && wave > xB ? wave - xB : 0);
    const int64_t seq = (s0 + i < B) ? s0 + i : B - 1;
    const int filt = i < C ? i : C - 1;
    const float* eg = e + (HC_ABL == 5 ? (int64_t)i : seq) * L * d + 4 * q;                                   // + s d + 16 cb
    const float* wA = W + (hc_off(C, hA) + (int64_t)filt * hA) * d + 4 * q;      // + r d + 16 cb
    const float* wB = W + (hc_off(C, hB) + (int64_t)filt * hB) * d + 4 * q;
    hc_f32x4 accq[HC_NQ][4], tailA[3], tailB[3];
#pragma unroll
    for (int x = 0; x < HC_NQ; ++x)
#pragma unroll
        for (int k = 0; k < 4; ++k) accq[x][k] = (hc_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; ++k) tailA[k] = tailB[k] = (hc_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const float* slab = lds + 4 * lane;
    const int ncb = d / 16;
    for (int cb = 0; cb < ncb; ++cb) {
#ifndef HC_SB
#define HC_SB 13
#endif
        // wave w stages rows w, w + 4, ...: the loads of a batch of HC_SB rows in flight before the barrier / the stores
        for (int sb = 0; sb < ((HC_ABL == 1 || HC_ABL == 3) ? 0 : L); sb += 4 * HC_SB) {
            hc_f32x4 v[HC_SB];
#pragma unroll
            for (int k = 0; k < HC_SB; ++k) {
                const int s = sb + 4 * k + wave;
#if HC_ABL == 8
                v[k] = (hc_f32x4){0.01f * s, 0.02f * lane, 0.03f * cb, 0.5f};
#else
                v[k] = HC_LD4(eg + (int64_t)(s < L ? s : L - 1) * d + cb * 16);
#endif
            }
#if HC_ABL != 3
            if (sb == 0) __syncthreads();   // the previous chunk's readers are done
#endif
#pragma unroll
            for (int k = 0; k < HC_SB; ++k) {
                const int s = sb + 4 * k + wave;
#if HC_ABL == 7
                if (s < L && v[k][0] == 123.456f) *reinterpret_cast<hc_f32x4*>(lds + s * 256 + 4 * lane) = v[k];
#else
                if (s < L) *reinterpret_cast<hc_f32x4*>(lds + s * 256 + 4 * lane) = v[k];
#endif
            }
        }
#if HC_ABL != 3
        __syncthreads();
#endif
        hc_rows<1>(accq, tailA, slab, wA + cb * 16, d, hA, tA, nA >> 2, nA & 3);
        if (nB) hc_rows<-1>(accq, tailB, slab, wB + cb * 16, d, hB, tB, nB >> 2, nB & 3);
    }
    __syncthreads();
    float* bests = lds;                                         // [height A / B][wave][v][lane]
    int* bts = reinterpret_cast<int*>(lds + 2 * 1024);
    hc_wave_best<1>(accq, tailA, nA >> 2, nA & 3, tA, bias[(hA - 1) * C + filt], bests, bts, wave, lane);
    hc_wave_best<-1>(accq, tailB, nB >> 2, nB & 3, tB, bias[(hB - 1) * C + filt], bests + 1024, bts + 1024, wave, lane);
    __syncthreads();
    // thread (wave, lane) finishes v = wave of both heights: waves in ascending position order, the first maximum wins
    const int64_t sq = s0 + 4 * q + wave;
    if (i < C && sq < B) {
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
            if (hx == 1 && TB == 0) break;
            const int hh = hx ? hB : hA;
            float best = -INFINITY;
            int bt = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float v = bests[hx * 1024 + (w * 4 + wave) * 64 + lane];
                if (v > best) {
                    best = v;
                    bt = bts[hx * 1024 + (w * 4 + wave) * 64 + lane];
                }
            }
            const int64_t o = sq * ((int64_t)C * L) + (hh - 1) * C + i;
            out[o] = best > 0.0f ? best : 0.0f;
            arg[o] = best > 0.0f ? bt : -1;
        }
    }
}

// ---- the same kernel over 32-column chunks, two height pairs per workgroup (d % 32 == 0) -----------------------------
// What the 16-column version above still pays for is its staging: every wave-load there covers 16 sequences x 64 bytes,
// half cache lines.  Here a chunk is 32 columns = one full 128-byte line per (sequence, row): a staging wave-load covers
// 8 sequences x 128 bytes, the chunk (2 L KB) is staged once for TWO height pairs of neighbouring cost (waves 0-3 and
// 4-7 of a 512-thread workgroup; one workgroup per CU), and there are half as many barriers per sequence group.
// LDS row s: [half 0 / 1][q][sequence slot][4 floats], slot = sequence ^ (2 q + half) -- lane (i, q) reads columns 8 q .. 8 q + 3 (half 0) and 8 q + 4 .. 8 q + 7
// (half 1) as two conflict-free ds_read_b128; MFMA j < 8 contracts column 8 q + j over q.
template <int DIR>
__device__ __forceinline__ void hc2_rows(hc_f32x4 (&accq)[HC_NQ][4], hc_f32x4 (&tail)[3], const float* __restrict__ slab,
                                         const float* __restrict__ slab1, const float* __restrict__ wr, int d, int hh, int t_start, int nq, int nt) {
    constexpr int RD = 4;
    hc_f32x4 b[RD][2];
#pragma unroll
    for (int u = 0; u < RD; ++u) {
        const float* p = wr + (int64_t)(u < hh ? u : hh - 1) * d;
        b[u][0] = HC_LD4(p);
        b[u][1] = HC_LD4(p + 4);
    }
    const float* ap = slab + t_start * 512;
    const float* ap1 = slab1 + t_start * 512;
    const float* at = ap + 4 * nq * 512;
    const float* at1 = ap1 + 4 * nq * 512;
    for (int r0 = 0; r0 < hh; r0 += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            if (r0 + u < hh) {
                const float* ar = ap + (r0 + u) * 512;
                const float* ar1 = ap1 + (r0 + u) * 512;
#pragma unroll
                for (int x = 0; x < HC_NQ; ++x)
                    if (x < nq) {
                        hc_f32x4 a[4][2];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            a[k][0] = HC_LD4(ar + (4 * x + k) * 512);
                            a[k][1] = HC_LD4(ar1 + (4 * x + k) * 512);
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                hc_f32x4& c = accq[DIR > 0 ? x : HC_NQ - 1 - x][k];
                                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][j >> 2][j & 3], b[u][j >> 2][j & 3], c, 0, 0, 0);
                            }
                    }
                if (nt > 0) {
                    const float* atr = at + (r0 + u) * 512;
                    const float* atr1 = at1 + (r0 + u) * 512;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (k < nt) {
                            const hc_f32x4 a0 = HC_LD4(atr + k * 512), a1 = HC_LD4(atr1 + k * 512);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                tail[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b[u][0][j], tail[k], 0, 0, 0);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                tail[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b[u][1][j], tail[k], 0, 0, 0);
                        }
                }
            }
            const int rn = r0 + u + RD;     // rows past the end: clamped, unused
            const float* p = wr + (int64_t)(rn < hh ? rn : hh - 1) * d;
            b[u][0] = HC_LD4(p);
            b[u][1] = HC_LD4(p + 4);
        }
    }
}

__global__ __launch_bounds__(512, 1) void hconv_direct2_fwd_kernel(const float* __restrict__ e, const float* __restrict__ W,
                                                                   const float* __restrict__ bias, int64_t B, int L, int C,
                                                                   int d, int nsg, float* __restrict__ out,
                                                                   int32_t* __restrict__ arg) {
    extern __shared__ float lds[];          // [L rows][2 halves][4 q][16 sequences][4]; the epilogue's 32 KB reuse it
    const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, side = wave8 >> 2;
    const int i = lane & 15, q = lane >> 4;
    const int npairs = (L + 1) / 2;
    // duo dd = the pairs of cost rank 2 dd and 2 dd + 1 (neighbours: nearly the same cost); costliest duos first
    const int dd = (int)(blockIdx.x / nsg);
    const int pr = npairs - 1 - (2 * dd + side);
    const bool live = pr >= 0;              // an odd number of pairs: the last duo's second half only keeps the barriers
    const int hA = live ? pr + 1 : 1, hB = live ? L - pr : 1;
    const int64_t s0 = (int64_t)(blockIdx.x % nsg) * 16;
    const int TA = live ? L + 1 - hA : 0, TB = (live && hB != hA) ? L + 1 - hB : 0;
    const int nA = TA / 4 + (wave < TA % 4 ? 1 : 0);
    const int tA = wave * (TA / 4) + (wave < TA % 4 ? wave : TA % 4);
    const int xB = 4 - TB % 4;
    const int nB = TB / 4 + ((TB % 4) && wave >= xB ? 1 : 0);
    const int tB = wave * (TB / 4) + ((TB % 4) && wave > xB ? wave - xB : 0);
    const int filt = i < C ? i : C - 1;
    const float* wA = W + (hc_off(C, hA) + (int64_t)filt * hA) * d + 8 * q;      // + r d + 32 cb
    const float* wB = W + (hc_off(C, hB) + (int64_t)filt * hB) * d + 8 * q;
    // staging: lane = (sequence % 8, 16-byte column group g = 2 q' + half); a wave-load = 8 sequences x one 128-byte line
    const int sg = lane >> 3, cg = lane & 7;
    const int64_t seq0 = (s0 + sg < B) ? s0 + sg : B - 1, seq1 = (s0 + 8 + sg < B) ? s0 + 8 + sg : B - 1;
    const float* eg0 = e + seq0 * L * d + 4 * cg;
    const float* eg1 = e + seq1 * L * d + 4 * cg;
    // float offset inside an LDS row; the sequence slot is XORed with the column group so that the eight lanes one LDS
    // cycle serves (one sequence, eight column groups) hit eight different bank quads -- readers undo it (slab / slab1)
    const int so0 = (cg & 1) * 256 + (cg >> 1) * 64 + (sg ^ cg) * 4, so1 = so0 + 32;
    hc_f32x4 accq[HC_NQ][4], tailA[3], tailB[3];
#pragma unroll
    for (int x = 0; x < HC_NQ; ++x)
#pragma unroll
        for (int k = 0; k < 4; ++k) accq[x][k] = (hc_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; ++k) tailA[k] = tailB[k] = (hc_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const float* slab = lds + q * 64 + (i ^ (2 * q)) * 4;
    const float* slab1 = lds + 256 + q * 64 + (i ^ (2 * q + 1)) * 4;
    const int ncb = d / 32;
    for (int cb = 0; cb < ncb; ++cb) {
        // wave w stages rows w, w + 8, ...: both halves of the 16 sequences, all loads in flight before the barrier
        constexpr int SB = (HC_MAX_L + 7) / 8;
        hc_f32x4 v[SB][2];
#pragma unroll
        for (int k = 0; k < SB; ++k) {
            const int s = 8 * k + wave8;
            const int64_t o = (int64_t)(s < L ? s : L - 1) * d + cb * 32;
            v[k][0] = HC_LD4(eg0 + o);
            v[k][1] = HC_LD4(eg1 + o);
        }
        __syncthreads();                    // the previous chunk's readers are done
#pragma unroll
        for (int k = 0; k < SB; ++k) {
            const int s = 8 * k + wave8;
            if (s < L) {
                *reinterpret_cast<hc_f32x4*>(lds + s * 512 + so0) = v[k][0];
                *reinterpret_cast<hc_f32x4*>(lds + s * 512 + so1) = v[k][1];
            }
        }
        __syncthreads();
        if (nA) hc2_rows<1>(accq, tailA, slab, slab1, wA + cb * 32, d, hA, tA, nA >> 2, nA & 3);
        if (nB) hc2_rows<-1>(accq, tailB, slab, slab1, wB + cb * 32, d, hB, tB, nB >> 2, nB & 3);
    }
    __syncthreads();
    float* bests = lds + side * 4096;                           // [side][height A / B][wave][v][lane]
    int* bts = reinterpret_cast<int*>(lds + side * 4096 + 2048);
    hc_wave_best<1>(accq, tailA, nA >> 2, nA & 3, tA, bias[(hA - 1) * C + filt], bests, bts, wave, lane);
    hc_wave_best<-1>(accq, tailB, nB >> 2, nB & 3, tB, bias[(hB - 1) * C + filt], bests + 1024, bts + 1024, wave, lane);
    __syncthreads();
    const int64_t sq = s0 + 4 * q + wave;
    if (live && i < C && sq < B) {
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
            if (hx == 1 && TB == 0) break;
            const int hh = hx ? hB : hA;
            float best = -INFINITY;
            int bt = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float v = bests[hx * 1024 + (w * 4 + wave) * 64 + lane];
                if (v > best) {
                    best = v;
                    bt = bts[hx * 1024 + (w * 4 + wave) * 64 + lane];
                }
            }
            const int64_t o = sq * ((int64_t)C * L) + (hh - 1) * C + i;
            out[o] = best > 0.0f ? best : 0.0f;
            arg[o] = best > 0.0f ? bt : -1;
        }
    }
}

// d e: one workgroup per sequence accumulates its [L, d] tile in LDS; columns V (tid % LPR) .. +V-1 of row p belong to
// one thread (p % (256 / LPR) = tid / LPR), contributions added in ascending (h, c, r)
template <int V, int LPR>
__global__ __launch_bounds__(256) void hconv_bwd_input_kernel(const float* __restrict__ gout,
                                                              const int32_t* __restrict__ arg,
                                                              const float* __restrict__ W, int64_t B, int L, int C,
                                                              int d, float* __restrict__ ge) {
    extern __shared__ float lds[];
    constexpr int NG = 256 / LPR;
    float* acc = lds;
    float* gs = acc + L * d;
    int32_t* as = reinterpret_cast<int32_t*>(gs + C * L);
    const int tid = threadIdx.x, k = (tid % LPR) * V, grp = tid / LPR;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int idx = tid; idx < L * d; idx += 256) acc[idx] = 0.0f;
        for (int idx = tid; idx < C * L; idx += 256) {
            gs[idx] = gout[b * C * L + idx];
            as[idx] = arg[b * C * L + idx];
        }
        __syncthreads();
        if (k < d) {
            for (int h = 1; h <= L; ++h) {
                const float* wh = W + hc_off(C, h) * d + k;
                for (int c = 0; c < C; ++c) {
                    const int t0 = as[(h - 1) * C + c];
                    const float g = gs[(h - 1) * C + c];
                    if (t0 < 0 || g == 0.0f) continue;
                    const float* w = wh + (int64_t)c * h * d;
                    int r = (grp - t0) % NG;
                    if (r < 0) r += NG;
                    float* a = acc + (t0 + r) * d + k;
#pragma unroll 2
                    for (; r < h; r += NG, a += NG * d) {
                        float wv[V], av[V];
                        vload<V>(wv, w + (int64_t)r * d);
                        vload<V>(av, a);
#pragma unroll
                        for (int y = 0; y < V; ++y) av[y] = fmaf(g, wv[y], av[y]);
                        vstore<V>(a, av);
                    }
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < L * d; idx += 256) ge[b * L * d + idx] = acc[idx];
    }
}

// (Round 2, tried and removed: filter rows shared by eight sequences through LDS chunks, every thread keeping positions
// p = pg + 4 i of its sequence in 16 register slots and testing 0 <= p - t0 < h per slot.  Bit-identical, but 2.5 ms
// against this kernel's 2.3 ms at C5, 5.3 ms with 8192-float chunks: the per-slot range tests are vector instructions for
// ALL 16 slots of every filter -- twelve times the useful FMAs -- because a register file cannot be indexed by t0.)
// d W: one workgroup per (height, filter, batch chunk); the h x d partial lives in registers (LPR lanes of V floats
// per row; a thread keeps columns V (tid % LPR) .. +V-1 of the rows r with r % (256 / LPR) = tid / LPR), sequences in
// ascending order; the chunks' partials are summed afterwards
template <int V, int LPR>
__global__ __launch_bounds__(256) void hconv_bwd_filter_kernel(const float* __restrict__ gout,
                                                               const int32_t* __restrict__ arg,
                                                               const float* __restrict__ e, int64_t B, int L, int C,
                                                               int d, int S, float* __restrict__ partial) {
    constexpr int NG = 256 / LPR;
    constexpr int RMAX = (HC_MAX_L + NG - 1) / NG;
    __shared__ float gs[256];
    __shared__ int32_t as[256];
    const int tid = threadIdx.x, k = (tid % LPR) * V, grp = tid / LPR;
    const int64_t rows = hc_off(C, L + 1);
    // chunk-major: the workgroups resident together read the same few sequences' rows (L2), tall filters first
    const int npairs = L * C;
    const int s = blockIdx.x / npairs, pair = blockIdx.x % npairs;
    const int h = L - pair / C, c = pair % C;
    const int64_t o = (int64_t)(h - 1) * C + c;
    const int64_t per = (B + S - 1) / S;
    const int64_t b0 = s * per, b1 = (b0 + per < B) ? b0 + per : B;
    float acc[RMAX][V];
#pragma unroll
    for (int i = 0; i < RMAX; ++i)
#pragma unroll
        for (int x = 0; x < V; ++x) acc[i][x] = 0.0f;
    for (int64_t bb = b0; bb < b1; bb += 256) {
        __syncthreads();
        if (bb + tid < b1) {
            gs[tid] = gout[(bb + tid) * C * L + o];
            as[tid] = arg[(bb + tid) * C * L + o];
        }
        __syncthreads();
        const int n = (int)((b1 - bb < 256) ? b1 - bb : 256);
        for (int x = 0; x < n; ++x) {
            const int t0 = as[x];
            const float g = gs[x];
            if (t0 < 0 || g == 0.0f) continue;
            const float* eb = e + ((bb + x) * L + t0) * d + k;
#pragma unroll
            for (int i = 0; i < RMAX; ++i) {
                const int r = grp + i * NG;
                if (r < h && k < d) {
                    float v[V];
                    vload<V>(v, eb + (int64_t)r * d);
#pragma unroll
                    for (int y = 0; y < V; ++y) acc[i][y] = fmaf(g, v[y], acc[i][y]);
                }
            }
        }
    }
    float* dst = partial + ((int64_t)s * rows + hc_off(C, h) + (int64_t)c * h) * d + k;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        const int r = grp + i * NG;
        if (r < h && k < d) vstore<V>(dst + (int64_t)r * d, acc[i]);
    }
}

// (Round 2, tried and removed: every sequence's [L, d] slab staged in LDS once per 256 rows of dW, 16 rows per thread in
// registers, up to three sequences requested ahead.  Bit-identical, but 3.0 ms with a branch per row (32 dependent LDS
// round trips per sequence), 2.4 ms branch-free with one sequence ahead, 2.2 ms with three -- against this kernel's
// 1.9 ms at C5: per useful FMA the (g, t0) look-ups, address arithmetic and selects cost more vector instructions than
// the per-(height, filter) workgroups above spend, and 8 waves per CU hide less latency than their 32.)

// (Round 3, tried and removed -- tools/hconv_bwd_probe.py at C5, rocprofv3: d e 2.31 ms, d W 1.87 ms for the two kernels above.
//  (a) "Second form": a thread owns ONE output row over a 32-column slice with its 32 sums in registers -- position p of
//  sequence s for d e (lanes = positions, 96 filter rows per double-buffered LDS stage shared by 10 sequences), row r of
//  filter (h, c) for d W (lanes = filter rows, 2 sequences per stage) -- and per contribution reads the other operand's
//  row slice with eight ds_read_b128 and does 16 v_pk_fma_f32.  Bit-identical, 3.5x fewer vector instructions per FMA, but
//  d e 2.69 ms and d W 2.11-2.21 ms (3.8 ms when cut to 96 VGPRs: spills): every contribution's 128-byte slice crosses the
//  LDS once per LANE (43 GB per kernel, 0.55 ms at 128 B/clk/CU if every lane were useful); with lanes = positions only
//  h / L of them are covered (SQ_ACTIVE_INST_LDS 73 % of SQ_BUSY_CYCLES), with lanes = filter rows the 16-lane groups of
//  ds_read_b128 straddle filters with unrelated winners (SQ_LDS_BANK_CONFLICT = SQ_ACTIVE_INST_LDS: 2-way on average).
//  (b) The kernels above with the next filter's rows (d e) / two or four sequences' rows (d W) requested before the current
//  FMAs: 3.87 / 3.04 ms (4.37 with four) -- the extra registers cost more occupancy than the overlap returns.
//  A dense MFMA form does L times the useful products (one winner among L - h + 1 positions), i.e. the forward's 3.2 ms.)
// ---- vertical convolution (Caser.py:118-130): out[b, col * n_v + v] = bias[v] + sum_l w[v, l] e[b, l, col] -------------
// One thread per (sequence, column): the L slab elements of its column are read once (coalesced over the columns), the
// n_v filters sit in LDS.  HBM / L2 bound on the slab (B L d floats); no MFMA: n_v = 4 outputs per L = 50 inputs.
constexpr int VC_MAX_NV = 16;

__global__ __launch_bounds__(256) void vconv_fwd_kernel(const float* __restrict__ e, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int64_t B, int L, int d, int nv,
                                                        float* __restrict__ out) {
    __shared__ float ws[VC_MAX_NV * HC_MAX_L];
    for (int x = threadIdx.x; x < nv * L; x += 256) ws[x] = w[x];
    __syncthreads();
    const int64_t n = B * d;
    for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < n; x += (int64_t)gridDim.x * 256) {
        const int64_t b = x / d;
        const int col = (int)(x - b * d);
        float acc[VC_MAX_NV];
#pragma unroll
        for (int v = 0; v < VC_MAX_NV; ++v) acc[v] = 0.0f;
        const float* ep = e + b * L * d + col;
        for (int l = 0; l < L; ++l) {
            const float ev = ep[(int64_t)l * d];
#pragma unroll
            for (int v = 0; v < VC_MAX_NV; ++v)
                if (v < nv) acc[v] = fmaf(ws[v * L + l], ev, acc[v]);
        }
#pragma unroll
        for (int v = 0; v < VC_MAX_NV; ++v)
            if (v < nv) out[x * nv + v] = acc[v] + bias[v];
    }
}

// backward: ge[b, l, col] = sum_v w[v, l] g[b, col, v] (overwritten); partial[block, v, l] = this block's share of
// dLoss/dw[v, l] = sum_{b, col} g[b, col, v] e[b, l, col], column l = L holds the bias gradient sum g[b, col, v].
// Wave-level shuffles, lane 0 adds into the wave's LDS row in a fixed order; waves and blocks are summed in index order.
__global__ __launch_bounds__(256) void vconv_bwd_kernel(const float* __restrict__ e, const float* __restrict__ w,
                                                        const float* __restrict__ g, int64_t B, int L, int d, int nv,
                                                        float* __restrict__ ge, float* __restrict__ partial) {
    __shared__ float ws[VC_MAX_NV * HC_MAX_L];
    __shared__ float wacc[4][VC_MAX_NV * (HC_MAX_L + 1)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int L1 = L + 1;
    for (int x = threadIdx.x; x < nv * L; x += 256) ws[x] = w[x];
    for (int x = threadIdx.x; x < 4 * VC_MAX_NV * (HC_MAX_L + 1); x += 256) (&wacc[0][0])[x] = 0.0f;
    __syncthreads();
    const int64_t n = B * d;
    const int64_t n_up = (n + 255) / 256 * 256;          // whole waves take part in the shuffles
    for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < n_up; x += (int64_t)gridDim.x * 256) {
        const bool live = x < n;
        const int64_t b = live ? x / d : 0;
        const int col = live ? (int)(x - b * d) : 0;
        float gv[VC_MAX_NV];
#pragma unroll
        for (int v = 0; v < VC_MAX_NV; ++v) gv[v] = (live && v < nv) ? g[x * nv + v] : 0.0f;
        const float* ep = e + b * L * d + col;
        float* gp = ge + b * L * d + col;
        for (int l = 0; l <= L; ++l) {
            const float ev = l < L ? (live ? ep[(int64_t)l * d] : 0.0f) : 1.0f;
            if (l < L && live) {
                float s = 0.0f;
#pragma unroll
                for (int v = 0; v < VC_MAX_NV; ++v)
                    if (v < nv) s = fmaf(ws[v * L + l], gv[v], s);
                gp[(int64_t)l * d] = s;
            }
#pragma unroll
            for (int v = 0; v < VC_MAX_NV; ++v)
                if (v < nv) {
                    float p = gv[v] * ev;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) p += __shfl_down(p, o, 64);
                    if (lane == 0) wacc[wave][v * L1 + l] += p;
                }
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < nv * L1; x += 256)
        partial[(int64_t)blockIdx.x * nv * L1 + x] = ((wacc[0][x] + wacc[1][x]) + wacc[2][x]) + wacc[3][x];
}

static int hc_check(const char* who, int64_t B, int32_t L, int32_t C, int32_t d) {
    DR_REQUIRE(B >= 0 && L >= 1 && L <= HC_MAX_L && C >= 1 && C <= HC_MAX_C && d >= 1 && d <= HC_MAX_D,
               "%s: need B >= 0, 1 <= L <= %d, 1 <= C <= %d, 1 <= d <= %d (got B=%lld L=%d C=%d d=%d)", who, HC_MAX_L,
               HC_MAX_C, HC_MAX_D, (long long)B, L, C, d);
    return DR_OK;
}

}  // namespace dr

using namespace dr;

extern "C" int32_t dr_caser_hconv_pool_f32(const float* G, int64_t ldg, const float* bias, int64_t B, int32_t L,
                                           int32_t C, int32_t h0, int32_t h1, float* out, int32_t* arg, void* stream) {
    if (int rc = hc_check("dr_caser_hconv_pool_f32", B, L, C, 1)) return rc;
    DR_REQUIRE(h0 >= 1 && h0 <= h1 && h1 <= L, "dr_caser_hconv_pool_f32: heights %d..%d outside 1..%d", h0, h1, L);
    DR_REQUIRE(ldg >= hc_off(C, h1 + 1) - hc_off(C, h0), "dr_caser_hconv_pool_f32: ldg %lld < the chunk's %lld columns",
               (long long)ldg, (long long)(hc_off(C, h1 + 1) - hc_off(C, h0)));
    if (B == 0) return DR_OK;
    DR_REQUIRE(G && bias && out && arg, "dr_caser_hconv_pool_f32: NULL pointer");
    int nbands = 0;
    size_t lds = 0;
    for (int h = h0; h <= h1; ++h) {
        nbands += hc_bands(C, h);
        const int cb = hc_cb(C, h);
        const size_t need = ((size_t)L * ((cb * h) | 1) + (size_t)cb * (L - h + 1)) * 4;
        if (need > lds) lds = need;
    }
    if (lds > 48 * 1024)
        DR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hconv_pool_fwd_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int gy = (int)(B < 32768 ? B : 32768);
    DR_LAUNCH_NAMED("hconv_pool_fwd_kernel", hconv_pool_fwd_kernel, dim3(nbands, gy), dim3(256), lds,
                    (hipStream_t)stream, G, ldg, bias, B, (int)L, (int)C, (int)h0, (int)h1, out, arg);
    DR_CHECK_LAUNCH("hconv_pool_fwd_kernel");
    return DR_OK;
}

extern "C" int32_t dr_caser_hconv_direct_f32(const float* e, const float* W, const float* bias, int64_t B, int32_t L,
                                             int32_t C, int32_t d, float* out, int32_t* arg, void* stream) {
    if (int rc = hc_check("dr_caser_hconv_direct_f32", B, L, C, d)) return rc;
    DR_REQUIRE(C <= 16 && d % 16 == 0, "dr_caser_hconv_direct_f32: needs C <= 16 filters per height and d %% 16 == 0 (got C=%d d=%d)",
               C, d);
    if (B == 0) return DR_OK;
    DR_REQUIRE(e && W && bias && out && arg && (((uintptr_t)e | (uintptr_t)W) % 16 == 0),
               "dr_caser_hconv_direct_f32: NULL / misaligned pointer");
    const int64_t nsg = (B + 15) / 16;
    const int npairs = (L + 1) / 2;
    if (d % 32 == 0 && DR_EXP_ENV("DR_HCONV_16COL") == nullptr) {    // 32-column chunks, two height pairs per workgroup
        const int nduo = (npairs + 1) / 2;
        DR_REQUIRE(nsg * nduo < (1ll << 31) - 1, "dr_caser_hconv_direct_f32: grid too large");
        size_t lds2 = (size_t)L * 2048;
        if (lds2 < 32 * 1024) lds2 = 32 * 1024;
        DR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hconv_direct2_fwd_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        DR_LAUNCH_NAMED("hconv_direct2_fwd_kernel", hconv_direct2_fwd_kernel, dim3((unsigned)(nsg * nduo)), dim3(512), lds2,
                        (hipStream_t)stream, e, W, bias, B, (int)L, (int)C, (int)d, (int)nsg, out, arg);
        DR_CHECK_LAUNCH("hconv_direct2_fwd_kernel");
        return DR_OK;
    }
    DR_REQUIRE(nsg * npairs < (1ll << 31) - 1, "dr_caser_hconv_direct_f32: grid too large");
    size_t lds = (size_t)L * 1024;          // the staged chunk; the epilogue needs 16 KB
    if (lds < 16 * 1024) lds = 16 * 1024;
#if HC_ABL == 6
    lds = 100 * 1024;                       // one workgroup per CU
#endif
    if (lds > 48 * 1024)
        DR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hconv_direct_fwd_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DR_LAUNCH_NAMED("hconv_direct_fwd_kernel", hconv_direct_fwd_kernel, dim3((unsigned)(nsg * npairs)), dim3(256), lds,
                    (hipStream_t)stream, e, W, bias, B, (int)L, (int)C, (int)d, (int)nsg, out, arg);
    DR_CHECK_LAUNCH("hconv_direct_fwd_kernel");
    return DR_OK;
}

extern "C" int32_t dr_caser_vconv_forward_f32(const float* e, const float* w, const float* bias, int64_t B, int32_t L,
                                              int32_t d, int32_t nv, float* out, void* stream) {
    DR_REQUIRE(B >= 0 && L >= 1 && L <= HC_MAX_L && d >= 1 && nv >= 1 && nv <= VC_MAX_NV,
               "dr_caser_vconv_forward_f32: need 1 <= L <= %d, 1 <= n_v <= %d", HC_MAX_L, VC_MAX_NV);
    if (B == 0) return DR_OK;
    DR_REQUIRE(e && w && bias && out, "dr_caser_vconv_forward_f32: NULL pointer");
    DR_LAUNCH(vconv_fwd_kernel, dim3(grid_for(B * d, 256)), dim3(256), 0, (hipStream_t)stream, e, w, bias, B, (int)L,
              (int)d, (int)nv, out);
    DR_CHECK_LAUNCH("vconv_fwd_kernel");
    return DR_OK;
}

extern "C" int32_t dr_caser_vconv_backward_f32(const float* e, const float* w, const float* g, int64_t B, int32_t L,
                                               int32_t d, int32_t nv, int32_t S, float* ge, float* partial,
                                               void* stream) {
    DR_REQUIRE(B >= 0 && L >= 1 && L <= HC_MAX_L && d >= 1 && nv >= 1 && nv <= VC_MAX_NV && S >= 1 && S <= 65535,
               "dr_caser_vconv_backward_f32: need 1 <= L <= %d, 1 <= n_v <= %d, 1 <= S <= 65535", HC_MAX_L, VC_MAX_NV);
    DR_REQUIRE(partial != nullptr, "dr_caser_vconv_backward_f32: partial is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        DR_HIP(hipMemsetAsync(partial, 0, (size_t)S * nv * (L + 1) * 4, st));
        return DR_OK;
    }
    DR_REQUIRE(e && w && g && ge, "dr_caser_vconv_backward_f32: NULL pointer");
    DR_LAUNCH(vconv_bwd_kernel, dim3(S), dim3(256), 0, st, e, w, g, B, (int)L, (int)d, (int)nv, ge, partial);
    DR_CHECK_LAUNCH("vconv_bwd_kernel");
    return DR_OK;
}

extern "C" int32_t dr_caser_hconv_backward_input_f32(const float* gout, const int32_t* arg, const float* W, int64_t B,
                                                     int32_t L, int32_t C, int32_t d, float* ge, void* stream) {
    if (int rc = hc_check("dr_caser_hconv_backward_input_f32", B, L, C, d)) return rc;
    if (B == 0) return DR_OK;
    DR_REQUIRE(gout && arg && W && ge, "dr_caser_hconv_backward_input_f32: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)L * d + 2 * (size_t)C * L) * 4;
    const int grid = (int)(B < 2048 ? B : 2048);
#define DR_HC(V_, LPR_)                                                                                                 \
    do {                                                                                                                 \
        if (lds > 48 * 1024)                                                                                             \
            DR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hconv_bwd_input_kernel<V_, LPR_>),                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
        DR_LAUNCH_NAMED("hconv_bwd_input_kernel", (hconv_bwd_input_kernel<V_, LPR_>), dim3(grid), dim3(256), lds, st,   \
                        gout, arg, W, B, (int)L, (int)C, (int)d, ge);                                                    \
    } while (0)
    if (d % 4 == 0 && ((uintptr_t)W % 16 == 0)) {
        if (d <= 64)
            DR_HC(4, 16);
        else if (d <= 128)
            DR_HC(4, 32);
        else
            DR_HC(4, 64);
    } else if (d % 2 == 0 && ((uintptr_t)W % 8 == 0)) {
        if (d <= 64)
            DR_HC(2, 32);
        else if (d <= 128)
            DR_HC(2, 64);
        else
            DR_HC(2, 128);
    } else {
        if (d <= 64)
            DR_HC(1, 64);
        else if (d <= 128)
            DR_HC(1, 128);
        else
            DR_HC(1, 256);
    }
#undef DR_HC
    DR_CHECK_LAUNCH("hconv_bwd_input_kernel");
    return DR_OK;
}

extern "C" int32_t dr_caser_hconv_backward_filter_f32(const float* gout, const int32_t* arg, const float* e, int64_t B,
                                                      int32_t L, int32_t C, int32_t d, int32_t S, float* partial,
                                                      void* stream) {
    if (int rc = hc_check("dr_caser_hconv_backward_filter_f32", B, L, C, d)) return rc;
    DR_REQUIRE(S >= 1 && S <= 64, "dr_caser_hconv_backward_filter_f32: 1 <= S <= 64 batch chunks");
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {       // no sequences: the partial sums are zero
        DR_REQUIRE(partial != nullptr, "dr_caser_hconv_backward_filter_f32: partial is NULL");
        DR_HIP(hipMemsetAsync(partial, 0, (size_t)S * (size_t)hc_off(C, L + 1) * (size_t)d * 4, st));
        return DR_OK;
    }
    DR_REQUIRE(gout && arg && e && partial, "dr_caser_hconv_backward_filter_f32: NULL pointer");
    const bool v4 = d % 4 == 0 && (((uintptr_t)e | (uintptr_t)partial) % 16 == 0);
    const int grid = L * C * S;
#define DR_HF(V_, LPR_)                                                                                                  \
    DR_LAUNCH_NAMED("hconv_bwd_filter_kernel", (hconv_bwd_filter_kernel<V_, LPR_>), dim3(grid), dim3(256), 0, st, gout, \
                    arg, e, B, (int)L, (int)C, (int)d, (int)S, partial)
    const bool v2 = d % 2 == 0 && (((uintptr_t)e | (uintptr_t)partial) % 8 == 0);
    if (v4) {
        if (d <= 64)
            DR_HF(4, 16);
        else if (d <= 128)
            DR_HF(4, 32);
        else
            DR_HF(4, 64);
    } else if (v2) {          // the reference's default num_factor = 150
        if (d <= 64)
            DR_HF(2, 32);
        else if (d <= 128)
            DR_HF(2, 64);
        else
            DR_HF(2, 128);
    } else {
        if (d <= 64)
            DR_HF(1, 64);
        else if (d <= 128)
            DR_HF(1, 128);
        else
            DR_HF(1, 256);
    }
#undef DR_HF
    DR_CHECK_LAUNCH("hconv_bwd_filter_kernel");
    return DR_OK;
}
