// Full-catalog scores of a pairwise tower (NeuMF: neumf.py:80-115 evaluated for every (user, item) pair by
// RankingMetrics.py:82-101 through predict, neumf.py:213-214) without materialising the per-pair activations.
//
//   h1[u, i]  = relu(uh[u] + ih[i])                      uh = mlp_P[u] W0[user half] + b0, ih = mlp_Q W0[item half]
//   h_{l+1}   = relu(h_l W_l + b_l), l = 1 .. 4          (128 -> 80 -> 40 -> 20 -> 10 at the reference's defaults)
//   out[u, i] += sum_n h5[n]                              (out arrives holding the GMF dot <P[u], Q[i]>)
//
// 14.4 k multiply-adds per pair; as library GEMMs the chain writes and re-reads 512 + 320 + 160 + 80 + 40 bytes per
// pair and layer, here nothing but the item halves (L2) and one float per pair move.  v_mfma_f32_16x16x4_f32 with the
// OUTPUT FEATURES along M and 16 PAIRS (one user x 16 consecutive items) along N, i.e. every layer is computed
// transposed, H_{l+1}^T = W_l^T H_l^T: the D layout of one layer (lane (pair, q) holds features 16 t + 4 q + v of
// tile t in register v) is then exactly the B operand of the next layer's step (t, v), so the whole tower chains in
// registers -- no LDS transposes between layers.  A wave owns 4 pair tiles (64 items of one user), which amortises
// every weight operand (one ds_read_b128 per (output tile, input tile): the A values of 4 steps) over 16 MFMAs.
// Weights of all layers sit in LDS in operand order (63 KB at the default shape, zero padded to 16 x 16 tiles: padded
// features come out as relu(0) = 0 and add nothing).  Workgroups are persistent; consecutive waves take the same item
// block for consecutive users, so an item half is fetched from HBM once and re-read from L2.
#include "dr_common.h"

namespace dr {

typedef float tw_f4 __attribute__((ext_vector_type(4)));

struct TowerArgs {
    const float* uh;     // [nu, h1]
    const float* ih;     // [ni, h1]
    const float* W[4];   // W_l [dims[l], dims[l + 1]] row-major
    const float* b[4];   // b_l [dims[l + 1]]
    int dims[5];         // h1, n2, n3, n4, n5
    int64_t nu, ni, ldo;
    float* out;          // [nu, ldo]: += tower sum
};

// W [K, N] row-major -> LDS operand order: entry ((mt KT + t) 64 + lane) 4 + v = W[16 t + 4 q + v][16 mt + lane % 16]
__device__ __forceinline__ void tower_stage(float* dst, const float* __restrict__ W, int K, int N, int KT, int MT) {
    const int total = MT * KT * 256;
    for (int x = threadIdx.x; x < total; x += blockDim.x) {
        const int v = x & 3, lane = (x >> 2) & 63, tile = x >> 8;
        const int mt = tile / KT, t = tile - mt * KT;
        const int k = 16 * t + 4 * (lane >> 4) + v, n = 16 * mt + (lane & 15);
        dst[x] = (k < K && n < N) ? W[(int64_t)k * N + n] : 0.0f;
    }
}
__device__ __forceinline__ void tower_stage_bias(float* dst, const float* __restrict__ b, int N, int MT) {
    for (int x = threadIdx.x; x < MT * 16; x += blockDim.x) dst[x] = x < N ? b[x] : 0.0f;
}

// out^T = W^T in^T for 4 pair tiles: in[t][pt] = features 16 t + 4 q + v of pair tile pt (bias + ReLU applied)
template <int KT, int MT>
__device__ __forceinline__ void tower_layer(const tw_f4 (&in)[KT][4], const tw_f4* __restrict__ wl, tw_f4 (&out)[MT][4],
                                            int lane) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) out[mt][pt] = (tw_f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const tw_f4 a = wl[(mt * KT + t) * 64 + lane];
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
                    out[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v], in[t][pt][v], out[mt][pt], 0, 0, 0);
        }
}

template <int MT>
__device__ __forceinline__ void tower_act(tw_f4 (&h)[MT][4], const tw_f4* __restrict__ bl, int q) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const tw_f4 bv = bl[mt * 4 + q];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int v = 0; v < 4; ++v) h[mt][pt][v] = fmaxf(h[mt][pt][v] + bv[v], 0.0f);
    }
}

template <int T1, int M2, int M3, int M4, int M5>
__global__ __launch_bounds__(256, 2) void tower_catalog_kernel(TowerArgs a) {
    extern __shared__ float lds[];
    constexpr int O2 = 0, O3 = O2 + M2 * T1 * 256, O4 = O3 + M3 * M2 * 256, O5 = O4 + M4 * M3 * 256,
                  OB = O5 + M5 * M4 * 256;
    tower_stage(lds + O2, a.W[0], a.dims[0], a.dims[1], T1, M2);
    tower_stage(lds + O3, a.W[1], a.dims[1], a.dims[2], M2, M3);
    tower_stage(lds + O4, a.W[2], a.dims[2], a.dims[3], M3, M4);
    tower_stage(lds + O5, a.W[3], a.dims[3], a.dims[4], M4, M5);
    tower_stage_bias(lds + OB, a.b[0], a.dims[1], M2);
    tower_stage_bias(lds + OB + 16 * M2, a.b[1], a.dims[2], M3);
    tower_stage_bias(lds + OB + 16 * (M2 + M3), a.b[2], a.dims[3], M4);
    tower_stage_bias(lds + OB + 16 * (M2 + M3 + M4), a.b[3], a.dims[4], M5);
    __syncthreads();
    const tw_f4* w2 = reinterpret_cast<const tw_f4*>(lds + O2);
    const tw_f4* w3 = reinterpret_cast<const tw_f4*>(lds + O3);
    const tw_f4* w4 = reinterpret_cast<const tw_f4*>(lds + O4);
    const tw_f4* w5 = reinterpret_cast<const tw_f4*>(lds + O5);
    const tw_f4* bl = reinterpret_cast<const tw_f4*>(lds + OB);
    const int lane = threadIdx.x & 63, q = lane >> 4, pr = lane & 15;
    const int h1 = 16 * T1;
    const int64_t nib = (a.ni + 63) / 64;
    const int64_t total = nib * a.nu;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < total; w += nwaves) {
        const int64_t ib = w / a.nu, u = w - ib * a.nu;         // consecutive waves: the same items, consecutive users
        asm volatile("" ::: "memory");  // the weights never change: without this the compiler hoists every LDS read
                                        // of the tower out of the work loop and spills hundreds of registers
        const float* ip[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            int64_t item = ib * 64 + pt * 16 + pr;
            if (item >= a.ni) item = a.ni - 1;                   // clamped: computed, not stored
            ip[pt] = a.ih + item * h1 + 4 * q;
        }
        const float* up = a.uh + u * h1 + 4 * q;
        tw_f4 acc2[M2][4];
#pragma unroll
        for (int mt = 0; mt < M2; ++mt)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc2[mt][pt] = (tw_f4){0.0f, 0.0f, 0.0f, 0.0f};
        tw_f4 cur[4], nxt[4], uc, un;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) cur[pt] = *reinterpret_cast<const tw_f4*>(ip[pt]);
        uc = *reinterpret_cast<const tw_f4*>(up);
#pragma unroll
        for (int t = 0; t < T1; ++t) {
            if (t + 1 < T1) {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) nxt[pt] = *reinterpret_cast<const tw_f4*>(ip[pt] + 16 * (t + 1));
                un = *reinterpret_cast<const tw_f4*>(up + 16 * (t + 1));
            }
            tw_f4 x[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int v = 0; v < 4; ++v) x[pt][v] = fmaxf(cur[pt][v] + uc[v], 0.0f);
#pragma unroll
            for (int mt = 0; mt < M2; ++mt) {
                const tw_f4 wv = w2[(mt * T1 + t) * 64 + lane];
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt)
                        acc2[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[v], x[pt][v], acc2[mt][pt], 0, 0, 0);
            }
            if (t + 1 < T1) {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) cur[pt] = nxt[pt];
                uc = un;
            }
        }
        tower_act<M2>(acc2, bl, q);
        tw_f4 acc3[M3][4];
        tower_layer<M2, M3>(acc2, w3, acc3, lane);
        tower_act<M3>(acc3, bl + 4 * M2, q);
        tw_f4 acc4[M4][4];
        tower_layer<M3, M4>(acc3, w4, acc4, lane);
        tower_act<M4>(acc4, bl + 4 * (M2 + M3), q);
        tw_f4 acc5[M5][4];
        tower_layer<M4, M5>(acc4, w5, acc5, lane);
        tower_act<M5>(acc5, bl + 4 * (M2 + M3 + M4), q);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            float s = 0.0f;
#pragma unroll
            for (int mt = 0; mt < M5; ++mt)
#pragma unroll
                for (int v = 0; v < 4; ++v) s += acc5[mt][pt][v];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const int64_t item = ib * 64 + pt * 16 + pr;
            if (q == 0 && item < a.ni) a.out[u * a.ldo + item] += s;
        }
    }
}

}  // namespace dr

using namespace dr;

extern "C" int32_t dr_tower_catalog_scores_f32(const float* uh, int64_t nu, const float* ih, int64_t ni,
                                               const int32_t* dims, const float* const* W, const float* const* b,
                                               float* out, int64_t ldo, void* stream) {
    DR_REQUIRE(dims != nullptr && W != nullptr && b != nullptr, "dr_tower_catalog_scores_f32: NULL dims / W / b array");
    DR_REQUIRE(nu >= 0 && ni >= 0 && ldo >= ni, "dr_tower_catalog_scores_f32: need nu, ni >= 0 and ldo >= ni");
    int tiles[5];
    for (int l = 0; l < 5; ++l) {
        DR_REQUIRE(dims[l] >= 1 && dims[l] <= 128, "dr_tower_catalog_scores_f32: layer width %d outside 1..128", dims[l]);
        tiles[l] = (dims[l] + 15) / 16;
    }
    DR_REQUIRE(dims[0] % 16 == 0, "dr_tower_catalog_scores_f32: the first width (%d) must be a multiple of 16", dims[0]);
    if (nu == 0 || ni == 0) return DR_OK;
    DR_REQUIRE(uh && ih && out && (((uintptr_t)uh | (uintptr_t)ih) % 16 == 0),
               "dr_tower_catalog_scores_f32: NULL / misaligned pointer");
    TowerArgs a;
    a.uh = uh; a.ih = ih; a.nu = nu; a.ni = ni; a.ldo = ldo; a.out = out;
    for (int l = 0; l < 4; ++l) {
        DR_REQUIRE(W[l] && b[l], "dr_tower_catalog_scores_f32: NULL weight / bias of layer %d", l + 1);
        a.W[l] = W[l];
        a.b[l] = b[l];
    }
    for (int l = 0; l < 5; ++l) a.dims[l] = dims[l];
    const int64_t total_waves = ((ni + 63) / 64) * nu;
    int64_t grid = (total_waves + 3) / 4;
    if (grid > 512) grid = 512;             // 256 CUs x 2 persistent workgroups
#define DR_TW(T1_, M2_, M3_, M4_, M5_)                                                                                  \
    if (tiles[0] == T1_ && tiles[1] == M2_ && tiles[2] == M3_ && tiles[3] == M4_ && tiles[4] == M5_) {                  \
        const size_t lds = ((size_t)(M2_ * T1_ + M3_ * M2_ + M4_ * M3_ + M5_ * M4_) * 256 +                            \
                            16 * (M2_ + M3_ + M4_ + M5_)) * 4;                                                          \
        DR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tower_catalog_kernel<T1_, M2_, M3_, M4_, M5_>),        \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                              \
        DR_LAUNCH_NAMED("tower_catalog_kernel", (tower_catalog_kernel<T1_, M2_, M3_, M4_, M5_>), dim3((unsigned)grid), \
                        dim3(256), lds, (hipStream_t)stream, a);                                                        \
        DR_CHECK_LAUNCH("tower_catalog_kernel");                                                                        \
        return DR_OK;                                                                                                   \
    }
    DR_TW(8, 5, 3, 2, 1)    // neumf.py:61 defaults: num_factor_mlp = 64, hidden_dimension = 10
    DR_TW(4, 5, 3, 2, 1)    // num_factor_mlp = 32
    DR_TW(2, 3, 2, 1, 1)    // small shapes (tests: widths 32 -> 40 -> 20 -> 10 -> 5)
#undef DR_TW
    DR_FAIL(DR_ERR_ARG, "dr_tower_catalog_scores_f32: tower %d-%d-%d-%d-%d has no instantiation (16-wide tiles %d,%d,%d,%d,%d)",
            dims[0], dims[1], dims[2], dims[3], dims[4], tiles[0], tiles[1], tiles[2], tiles[3], tiles[4]);
}
