// Small dense towers behind the embeddings: NeuMF's MLP 128->128->80->40->20->10 (neumf.py:80-113),
// NNMF's 210->210->50->50->50->1 (nnmf.py:60-80), NFM's hidden layer (nfm.py:64-69).  At the reference's batch sizes
// (B = 128 .. 1000) a layer is a few hundred kFLOP: as library GEMMs + bias + activation kernels under autograd the
// tower is ~45 launches of 10-20 us each, all of it latency.  Here: one forward kernel, one delta kernel, one
// weight-gradient kernel for the WHOLE tower; fp32 fmaf chains in a fixed order (k ascending / batch row ascending),
// so results are repeatable.  Plain VALU on purpose: the work is far too small for MFMA tiles to matter.
#include "dr_common.h"

namespace dr {

constexpr int MLP_MAX_LAYERS = 8;
constexpr int MLP_MAX_DIM = 256;
constexpr int MLP_TB = 16;                    // batch rows per workgroup
constexpr int MLP_LD = MLP_MAX_DIM + 4;       // LDS row stride: 16-byte aligned rows (float4 reads along k)

struct MlpArgs {
    int L;
    int dims[MLP_MAX_LAYERS + 1];
    int act[MLP_MAX_LAYERS];                  // DR_ACT_*
    int off[MLP_MAX_LAYERS + 1];              // column offset of layer l's output inside acts / deltas rows
    const float* W[MLP_MAX_LAYERS];           // [dims[l], dims[l+1]] row-major
    const float* b[MLP_MAX_LAYERS];           // [dims[l+1]] or NULL
    float* gW[MLP_MAX_LAYERS];
    float* gb[MLP_MAX_LAYERS];
    int width;                                // sum of dims[1..L] = row length of acts / deltas
};

__device__ __forceinline__ float act_fwd(int kind, float x) {
    if (kind == DR_ACT_RELU) return x > 0.0f ? x : 0.0f;
    if (kind == DR_ACT_SIGMOID) return 1.0f / (1.0f + expf(-x));
    return x;
}
// derivative expressed through the activation's OUTPUT y
__device__ __forceinline__ float act_bwd(int kind, float y) {
    if (kind == DR_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    if (kind == DR_ACT_SIGMOID) return y * (1.0f - y);
    return 1.0f;
}

// LDS plan shared by the forward and delta kernels: two activation buffers [MLP_TB][MLP_LD] and one weight chunk.
// The weight matrix of a layer is staged through LDS in chunks of MLP_KC rows (forward) or MLP_KC columns (delta) with
// one cooperative, coalesced round of loads per chunk: a loop that fetches its weight from global memory every
// iteration is a chain of K dependent round trips on a nearly empty CU (measured: 101 us for NeuMF's tower at B=256).
constexpr int MLP_KC = 64;
constexpr int MLP_WS = MLP_MAX_DIM + 1;                      // weight chunk row stride (odd)
constexpr int MLP_LDS_FLOATS = 2 * MLP_TB * MLP_LD + MLP_KC * MLP_WS;

// Cooperative global -> LDS staging of `total` elements by 256 threads: MLP_SU loads are issued back to back into
// registers before the first LDS store.  Written as a plain `lds[f(i)] = g[h(i)]` loop the compiler waits for every
// load before its store, i.e. one memory round trip per element and thread: measured 7 us per 32-row chunk of the
// weight-gradient kernel, 23 us for one 128 x 128 layer of the forward kernel.
constexpr int MLP_SU = 16;
template <typename Load, typename Store>
__device__ __forceinline__ void stage(int total, int tid, Load load, Store store) {
    for (int base = 0; base < total; base += 256 * MLP_SU) {
        float t[MLP_SU];
#pragma unroll
        for (int u = 0; u < MLP_SU; ++u) {
            const int idx = base + u * 256 + tid;
            t[u] = idx < total ? load(idx) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < MLP_SU; ++u) {
            const int idx = base + u * 256 + tid;
            if (idx < total) store(idx, t[u]);
        }
    }
}

// thread layout of a layer whose per-thread dimension (output column, or input column in the delta pass) has `n`
// entries: np = power of two >= n (>= 16), 256 / np groups, each owning rpg = 16 / groups rows of the tile
struct Split {
    int np, rpg;
};
__device__ __forceinline__ Split split_for(int n) {
    int np = 16;
    while (np < n) np <<= 1;
    Split s;
    s.np = np;
    s.rpg = MLP_TB / (256 / np);
    return s;
}

// acc[q] += sum_k in[q][k] * w[k * ws], k = 0 .. kc4 (a multiple of 4): the inputs are read four k at a time
// (ds_read_b128, the same address for every lane of a group), so a step of 4 k is RPG + 4 LDS reads for 4 * RPG FMAs
template <int RPG>
__device__ __forceinline__ void fma_rows(float (&acc)[MLP_TB], const float* in, const float* w, int ws, int kc4) {
    for (int k = 0; k < kc4; k += 4) {
        const float w0 = w[(k + 0) * ws], w1 = w[(k + 1) * ws], w2 = w[(k + 2) * ws], w3 = w[(k + 3) * ws];
#pragma unroll
        for (int q = 0; q < RPG; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(in + q * MLP_LD + k);
            acc[q] = fmaf(v.x, w0, acc[q]);
            acc[q] = fmaf(v.y, w1, acc[q]);
            acc[q] = fmaf(v.z, w2, acc[q]);
            acc[q] = fmaf(v.w, w3, acc[q]);
        }
    }
}

// acts[r, off[l] .. off[l] + dims[l+1]) = h_{l+1}[r, :];  thread n owns output column n for the tile's MLP_TB rows
__global__ __launch_bounds__(256) void mlp_forward_kernel(MlpArgs a, const float* __restrict__ x, int64_t B,
                                                          float* __restrict__ acts) {
    extern __shared__ float lds[];
    float* buf0 = lds;
    float* buf1 = lds + MLP_TB * MLP_LD;
    float* wch = lds + 2 * MLP_TB * MLP_LD;                  // [MLP_KC][N | 1]
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * MLP_TB;
    const int rows = (int)((B - r0) < MLP_TB ? (B - r0) : MLP_TB);
    const int d04 = (a.dims[0] + 3) & ~3;             // columns up to the next multiple of 4 are zero (float4 reads)
    {
        const int d0 = a.dims[0];
        stage(MLP_TB * d04, tid,
              [&](int idx) {
                  const int r = idx / d04, c = idx - r * d04;
                  return (r < rows && c < d0) ? x[(r0 + r) * d0 + c] : 0.0f;
              },
              [&](int idx, float v) {
                  const int r = idx / d04, c = idx - r * d04;
                  buf0[r * MLP_LD + c] = v;
              });
    }
    float* in = buf0;
    float* out = buf1;
    for (int l = 0; l < a.L; ++l) {
        const int Kd = a.dims[l], N = a.dims[l + 1], S = N | 1;
        const float* __restrict__ W = a.W[l];
        // narrow layers split the tile's rows over thread groups so that all four waves work: group g owns rows
        // g * rpg .. + rpg of output column n
        const Split sp = split_for(N);
        const int n = tid & (sp.np - 1), rbase = (tid / sp.np) * sp.rpg;
        float acc[MLP_TB];
        const float bias = (n < N && a.b[l]) ? a.b[l][n] : 0.0f;
#pragma unroll
        for (int r = 0; r < MLP_TB; ++r) acc[r] = bias;
        for (int k0 = 0; k0 < Kd; k0 += MLP_KC) {
            const int kc = (Kd - k0) < MLP_KC ? (Kd - k0) : MLP_KC;
            const int kc4 = (kc + 3) & ~3;
            __syncthreads();                                 // previous chunk consumed / input tile written
            {   // rows k0 .. k0 + kc of W are one contiguous run; rows kc .. kc4 of the chunk are zero (stale input
                // columns times 0)
                const float* __restrict__ Wc = W + (int64_t)k0 * N;
                const int live = kc * N;
                stage(kc4 * N, tid, [&](int idx) { return idx < live ? Wc[idx] : 0.0f; },
                      [&](int idx, float v) {
                          const int k = idx / N, c = idx - k * N;
                          wch[k * S + c] = v;
                      });
            }
            __syncthreads();
            if (n < N) {
                switch (sp.rpg) {
                    case 16: fma_rows<16>(acc, in + rbase * MLP_LD + k0, wch + n, S, kc4); break;
                    case 8: fma_rows<8>(acc, in + rbase * MLP_LD + k0, wch + n, S, kc4); break;
                    case 4: fma_rows<4>(acc, in + rbase * MLP_LD + k0, wch + n, S, kc4); break;
                    case 2: fma_rows<2>(acc, in + rbase * MLP_LD + k0, wch + n, S, kc4); break;
                    default: fma_rows<1>(acc, in + rbase * MLP_LD + k0, wch + n, S, kc4); break;
                }
            }
        }
        if (n < N) {
            for (int q = 0; q < sp.rpg; ++q) {
                const int r = rbase + q;
                const float y = act_fwd(a.act[l], acc[q]);
                out[r * MLP_LD + n] = y;
                if (r < rows) acts[(r0 + r) * a.width + a.off[l] + n] = y;
            }
        } else if (n < ((N + 3) & ~3)) {
            for (int q = 0; q < sp.rpg; ++q) out[(rbase + q) * MLP_LD + n] = 0.0f;
        }
        float* t = in;
        in = out;
        out = t;
    }
}

// deltas[r, off[l] ..) = dLoss/d(pre-activation of layer l's output);  gx = dLoss/dx (may be NULL)
// thread k owns input column k of the layer being back-propagated through, for the tile's MLP_TB rows
__global__ __launch_bounds__(256) void mlp_delta_kernel(MlpArgs a, int64_t B, const float* __restrict__ acts,
                                                        const float* __restrict__ gout, float* __restrict__ deltas,
                                                        float* __restrict__ gx) {
    extern __shared__ float lds[];
    float* buf0 = lds;
    float* buf1 = lds + MLP_TB * MLP_LD;
    float* wch = lds + 2 * MLP_TB * MLP_LD;                  // [MLP_KC][MLP_WS]: columns n0 .. of W, transposed
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * MLP_TB;
    const int rows = (int)((B - r0) < MLP_TB ? (B - r0) : MLP_TB);
    const int NL = a.dims[a.L];
    const int NL4 = (NL + 3) & ~3;
    {
        const int oL = a.off[a.L - 1], kL = a.act[a.L - 1];
        stage(MLP_TB * NL4, tid,
              [&](int idx) {
                  const int r = idx / NL4, c = idx - r * NL4;
                  if (!(r < rows && c < NL)) return 0.0f;
                  const float y = acts[(r0 + r) * a.width + oL + c];
                  return gout[(r0 + r) * NL + c] * act_bwd(kL, y);
              },
              [&](int idx, float d) {
                  const int r = idx / NL4, c = idx - r * NL4;
                  if (r < rows && c < NL) deltas[(r0 + r) * a.width + oL + c] = d;
                  buf0[r * MLP_LD + c] = d;
              });
    }
    float* din = buf0;
    float* dout = buf1;
    for (int l = a.L - 1; l >= 0; --l) {  // through W_l: delta of layer l's output -> delta of layer l-1's output (or gx)
        if (l == 0 && gx == nullptr) break;
        const int Kd = a.dims[l], N = a.dims[l + 1];
        const float* __restrict__ W = a.W[l];
        const Split sp = split_for(Kd);
        const int k = tid & (sp.np - 1), rbase = (tid / sp.np) * sp.rpg;
        float acc[MLP_TB];
#pragma unroll
        for (int r = 0; r < MLP_TB; ++r) acc[r] = 0.0f;
        for (int n0 = 0; n0 < N; n0 += MLP_KC) {
            const int nc = (N - n0) < MLP_KC ? (N - n0) : MLP_KC;
            const int nc4 = (nc + 3) & ~3;
            __syncthreads();
            stage(Kd * nc4, tid,                                  // columns nc .. nc4 are zero
                  [&](int idx) {
                      const int kk = idx / nc4, c = idx - kk * nc4;
                      return c < nc ? W[(int64_t)kk * N + n0 + c] : 0.0f;
                  },
                  [&](int idx, float v) {
                      const int kk = idx / nc4, c = idx - kk * nc4;
                      wch[c * MLP_WS + kk] = v;                          // transposed: [column][k]
                  });
            __syncthreads();
            if (k < Kd) {
                switch (sp.rpg) {
                    case 16: fma_rows<16>(acc, din + rbase * MLP_LD + n0, wch + k, MLP_WS, nc4); break;
                    case 8: fma_rows<8>(acc, din + rbase * MLP_LD + n0, wch + k, MLP_WS, nc4); break;
                    case 4: fma_rows<4>(acc, din + rbase * MLP_LD + n0, wch + k, MLP_WS, nc4); break;
                    case 2: fma_rows<2>(acc, din + rbase * MLP_LD + n0, wch + k, MLP_WS, nc4); break;
                    default: fma_rows<1>(acc, din + rbase * MLP_LD + n0, wch + k, MLP_WS, nc4); break;
                }
            }
        }
        if (k < Kd) {
            for (int q = 0; q < sp.rpg; ++q) {
                const int r = rbase + q;
                if (l == 0) {
                    if (r < rows) gx[(r0 + r) * Kd + k] = acc[q];
                } else {
                    float d = 0.0f;
                    if (r < rows) {
                        const float y = acts[(r0 + r) * a.width + a.off[l - 1] + k];
                        d = acc[q] * act_bwd(a.act[l - 1], y);
                        deltas[(r0 + r) * a.width + a.off[l - 1] + k] = d;
                    }
                    dout[r * MLP_LD + k] = d;
                }
            }
        } else if (l > 0 && k < ((Kd + 3) & ~3)) {
            for (int q = 0; q < sp.rpg; ++q) dout[(rbase + q) * MLP_LD + k] = 0.0f;
        }
        float* t = din;
        din = dout;
        dout = t;
    }
}

// gW_l[k, n] = sum_r h_l[r, k] * delta_{l+1}[r, n], gb_l[n] = sum_r delta_{l+1}[r, n] (h_0 = x), batch rows in ascending
// order.  One workgroup per (layer, tile of MLP_WK input columns): batch rows go through LDS MLP_WR at a time (two
// coalesced rounds of loads per chunk), thread n accumulates its MLP_WK weights and, in the layer's first tile, the bias.
constexpr int MLP_WK = 16, MLP_WR = 32;
__global__ __launch_bounds__(256) void mlp_wgrad_kernel(MlpArgs a, const float* __restrict__ x, int64_t B,
                                                        const float* __restrict__ acts,
                                                        const float* __restrict__ deltas) {
    __shared__ __attribute__((aligned(16))) float hs[MLP_WR][MLP_WK + 4];
    __shared__ float ds[MLP_WR][MLP_MAX_DIM + 1];
    int tile = blockIdx.x, l = 0;
    for (; l < a.L; ++l) {
        const int nt = (a.dims[l] + MLP_WK - 1) / MLP_WK;
        if (tile < nt) break;
        tile -= nt;
    }
    if (l >= a.L) return;
    const int Kd = a.dims[l], N = a.dims[l + 1];
    const int k0 = tile * MLP_WK;
    const int kc = (Kd - k0) < MLP_WK ? (Kd - k0) : MLP_WK;
    const float* __restrict__ h = (l == 0) ? x : (acts + a.off[l - 1]);
    const int64_t hstride = (l == 0) ? a.dims[0] : a.width;
    const float* __restrict__ dl = deltas + a.off[l];
    const int tid = threadIdx.x, n = tid;
    float acc[MLP_WK], accb = 0.0f;
#pragma unroll
    for (int q = 0; q < MLP_WK; ++q) acc[q] = 0.0f;
    for (int64_t rb = 0; rb < B; rb += MLP_WR) {
        const int rc = (int)((B - rb) < MLP_WR ? (B - rb) : MLP_WR);
        __syncthreads();
        stage(MLP_WR * MLP_WK, tid,
              [&](int idx) {
                  const int r = idx / MLP_WK, q = idx - r * MLP_WK;
                  return (r < rc && q < kc) ? h[(rb + r) * hstride + k0 + q] : 0.0f;
              },
              [&](int idx, float v) { hs[idx / MLP_WK][idx % MLP_WK] = v; });
        stage(MLP_WR * N, tid,
              [&](int idx) {
                  const int r = idx / N, c = idx - r * N;
                  return r < rc ? dl[(rb + r) * a.width + c] : 0.0f;
              },
              [&](int idx, float v) { ds[idx / N][idx % N] = v; });
        __syncthreads();
        if (n < N) {
#pragma unroll 4
            for (int r = 0; r < MLP_WR; ++r) {
                const float d = ds[r][n];
                accb = accb + d;
#pragma unroll
                for (int q = 0; q < MLP_WK; q += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(&hs[r][q]);
                    acc[q] = fmaf(v.x, d, acc[q]);
                    acc[q + 1] = fmaf(v.y, d, acc[q + 1]);
                    acc[q + 2] = fmaf(v.z, d, acc[q + 2]);
                    acc[q + 3] = fmaf(v.w, d, acc[q + 3]);
                }
            }
        }
    }
    if (n < N) {
        for (int q = 0; q < kc; ++q) a.gW[l][(int64_t)(k0 + q) * N + n] = acc[q];
        if (tile == 0 && a.gb[l] != nullptr) a.gb[l][n] = accb;
    }
}

static int mlp_args(MlpArgs* a, int L, const int32_t* dims, const int32_t* act, const float* const* W,
                    const float* const* b, const char* who) {
    DR_REQUIRE(L >= 1 && L <= MLP_MAX_LAYERS && dims && act && W, "%s: 1..%d layers, dims / act / W required", who,
               MLP_MAX_LAYERS);
    memset(a, 0, sizeof(*a));
    a->L = L;
    int off = 0;
    for (int l = 0; l <= L; ++l) {
        DR_REQUIRE(dims[l] >= 1 && dims[l] <= MLP_MAX_DIM, "%s: layer width %d outside [1,%d]", who, dims[l], MLP_MAX_DIM);
        a->dims[l] = dims[l];
    }
    for (int l = 0; l < L; ++l) {
        DR_REQUIRE(act[l] >= DR_ACT_NONE && act[l] <= DR_ACT_SIGMOID, "%s: bad activation %d", who, act[l]);
        DR_REQUIRE(W[l] != nullptr, "%s: W[%d] is NULL", who, l);
        a->act[l] = act[l];
        a->W[l] = W[l];
        a->b[l] = b ? b[l] : nullptr;
        a->off[l] = off;
        off += dims[l + 1];
    }
    a->width = off;
    return DR_OK;
}

}  // namespace dr

using namespace dr;

extern "C" int32_t dr_mlp_forward_f32(const float* x, int64_t B, int32_t L, const int32_t* dims, const int32_t* act,
                                      const float* const* W, const float* const* b, float* acts, void* stream) {
    MlpArgs a;
    int rc = mlp_args(&a, L, dims, act, W, b, "dr_mlp_forward_f32");
    if (rc != DR_OK) return rc;
    DR_REQUIRE(B >= 0, "dr_mlp_forward_f32: negative batch");
    if (B == 0) return DR_OK;
    DR_REQUIRE(x && acts, "dr_mlp_forward_f32: NULL pointer");
    const size_t smem = (size_t)MLP_LDS_FLOATS * 4;
    DR_HIP(hipFuncSetAttribute((const void*)mlp_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DR_LAUNCH(mlp_forward_kernel, dim3((unsigned)((B + MLP_TB - 1) / MLP_TB)), dim3(256), smem, (hipStream_t)stream, a, x,
              B, acts);
    DR_CHECK_LAUNCH("mlp_forward_kernel");
    return DR_OK;
}

extern "C" int32_t dr_mlp_backward_f32(const float* x, int64_t B, int32_t L, const int32_t* dims, const int32_t* act,
                                       const float* const* W, const float* acts, const float* gout, float* deltas,
                                       float* gx, float* const* gW, float* const* gb, void* stream) {
    MlpArgs a;
    int rc = mlp_args(&a, L, dims, act, W, nullptr, "dr_mlp_backward_f32");
    if (rc != DR_OK) return rc;
    DR_REQUIRE(B >= 0 && gW != nullptr, "dr_mlp_backward_f32: bad arguments");
    for (int l = 0; l < L; ++l) {
        DR_REQUIRE(gW[l] != nullptr, "dr_mlp_backward_f32: gW[%d] is NULL", l);
        a.gW[l] = gW[l];
        a.gb[l] = gb ? gb[l] : nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        for (int l = 0; l < L; ++l) {
            DR_HIP(hipMemsetAsync(gW[l], 0, (size_t)dims[l] * dims[l + 1] * 4, st));
            if (a.gb[l]) DR_HIP(hipMemsetAsync(a.gb[l], 0, (size_t)dims[l + 1] * 4, st));
        }
        return DR_OK;
    }
    DR_REQUIRE(x && acts && gout && deltas, "dr_mlp_backward_f32: NULL pointer");
    const size_t smem = (size_t)MLP_LDS_FLOATS * 4;
    DR_HIP(hipFuncSetAttribute((const void*)mlp_delta_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    DR_LAUNCH(mlp_delta_kernel, dim3((unsigned)((B + MLP_TB - 1) / MLP_TB)), dim3(256), smem, st, a, B, acts, gout, deltas,
              gx);
    DR_CHECK_LAUNCH("mlp_delta_kernel");
    int tiles = 0;
    for (int l = 0; l < L; ++l) tiles += (dims[l] + MLP_WK - 1) / MLP_WK;
    DR_LAUNCH(mlp_wgrad_kernel, dim3((unsigned)tiles), dim3(256), 0, st, a, x, B, acts, (const float*)deltas);
    DR_CHECK_LAUNCH("mlp_wgrad_kernel");
    return DR_OK;
}
