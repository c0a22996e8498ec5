// Shared host/device helpers for the deeprec_amd HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/deeprec_amd.h"

namespace dr {

// ---- host-side error plumbing ---------------------------------------------
char* last_error_buf();  // thread-local, defined in abi.hip

#define DR_FAIL(code, ...)                                         \
    do {                                                           \
        snprintf(dr::last_error_buf(), 512, __VA_ARGS__);          \
        return (code);                                             \
    } while (0)

#define DR_REQUIRE(cond, ...)                                      \
    do {                                                           \
        if (!(cond)) DR_FAIL(DR_ERR_ARG, __VA_ARGS__);             \
    } while (0)

#define DR_CHECK_LAUNCH(what)                                                          \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) DR_FAIL(DR_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

// hipGetLastError() is sticky per thread: an unrelated earlier failure (e.g. a probe made while another
// library initialised HIP) must not be blamed on our launch, so the slate is cleared first.
#define DR_LAUNCH(...)                   \
    do {                                 \
        (void)hipGetLastError();         \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

#define DR_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) DR_FAIL(DR_ERR_LAUNCH, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// ---- optional per-kernel timing (dr_profile_enable / dr_profile_collect, defined in abi.hip) ----
// DR_LAUNCH_NAMED launches like DR_LAUNCH; while profiling is enabled it goes through hipExtLaunchKernelGGL with a
// start / stop event pair, which the runtime stamps with the dispatch's own begin / end times (what rocprofv3's
// kernel trace reports).  Disabled (the default) it costs one load and a branch.  Never enable under capture.
extern bool g_prof_on;
bool prof_slot(const char* name, hipEvent_t* e0, hipEvent_t* e1);
#define DR_LAUNCH_NAMED(name, kernel, grid, block, shmem, stream, ...)                              \
    do {                                                                                            \
        hipEvent_t pe0_ = nullptr, pe1_ = nullptr;                                                  \
        (void)hipGetLastError();                                                                    \
        if (dr::g_prof_on && dr::prof_slot(name, &pe0_, &pe1_))                                     \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, pe0_, pe1_, 0, __VA_ARGS__);  \
        else                                                                                        \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                    \
    } while (0)

// Set (per host thread) while an entry point that must not use helper streams is running (dr_bpr_step_static_f32):
// the helper-stream providers then hand back the caller's stream.  Defined in abi.hip.
extern thread_local bool g_no_side_streams;

// Helper streams / events of the calling host thread on the current device: created by dr_init() ONLY (abi.hip) -- no
// entry point creates a HIP object.  helpers() returns NULL until dr_init() has run on this thread for this device;
// the steps then run on the caller's stream alone (same results, no overlap of plan and forward kernel).
struct Helpers {
    hipStream_t plan = nullptr, seg = nullptr;                    // plan beside the forward kernel; long-segment passes
    hipEvent_t fork = nullptr, q_ready = nullptr, p_ready = nullptr, seg_fork = nullptr, seg_join = nullptr;
    bool ready = false;
};
Helpers* helpers();

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Experiment switches (grid caps, ablations) exist only in -DDR_EXPERIMENTS builds (`make experiments`, used by tools/):
// the release library reads no environment variable, so a stray one cannot change which kernels run.
#ifdef DR_EXPERIMENTS
#define DR_EXP_ENV(name) getenv(name)
#else
#define DR_EXP_ENV(name) ((const char*)nullptr)
#endif

// Memory-bound launches: cap the grid at CUs * blocks-per-CU and grid-stride the rest
// (256 CUs on MI355X; 8 blocks of 256 threads per CU = full 32 waves/CU).
static inline int grid_for(int64_t work_items, int block, int items_per_block_iter = 1) {
    int64_t per = (int64_t)block * items_per_block_iter;
    int64_t g = (work_items + per - 1) / per;
    if (g < 1) g = 1;
    static const int64_t cap = DR_EXP_ENV("DR_GRID_CAP") ? atoll(DR_EXP_ENV("DR_GRID_CAP")) : 256 * 8;
    return (int)(g > cap ? cap : g);
}

// Grid of a grid-stride memory-bound kernel = the number of its blocks the device holds at once (occupancy x CUs):
// with the fixed 8-blocks-per-CU cap above, a kernel that fits only 5 or 6 blocks per CU runs 1.6 or 1.33 "waves" of
// blocks and the partial last wave idles part of the chip (measured on the fused BPR step: 2048 blocks 184 us,
// 1536 blocks 196 us, 1280 blocks = 5 per CU 177 us).  Cached per kernel; falls back to grid_for's cap.
int occupancy_blocks(const void* kernel, int block);  // abi.hip
template <typename KernelT>
static inline int grid_resident(KernelT kernel, int64_t work_blocks, int block = 256) {
    static const bool off = DR_EXP_ENV("DR_GRID_CAP") != nullptr;
    int cap = off ? 0 : occupancy_blocks(reinterpret_cast<const void*>(kernel), block);
    if (cap <= 0) return grid_for(work_blocks, 1, 1);
    if (work_blocks < 1) work_blocks = 1;
    return (int)(work_blocks > cap ? cap : work_blocks);
}

// ---- vector-width selection -------------------------------------------------
// A row of k floats is moved as k/V vectors of V floats (V = 4, 2 or 1) by LPR lanes
// per row (power of two <= 64), so 64/LPR rows are in flight per wavefront.
struct RowShape {
    int V;    // floats per lane-vector
    int LPR;  // lanes per row
};
static inline RowShape row_shape(int k) {
    RowShape s;
    s.V = (k % 4 == 0) ? 4 : ((k % 2 == 0) ? 2 : 1);
    int cpr = k / s.V;
    int l = 1;
    while (l < cpr && l < 64) l <<= 1;
    s.LPR = l;
    return s;
}

template <int V>
struct VecT;
template <>
struct VecT<4> { using type = float4; };
template <>
struct VecT<2> { using type = float2; };
template <>
struct VecT<1> { using type = float; };

template <int V>
__device__ __forceinline__ void vload(float (&r)[V], const float* p) {
    if constexpr (V == 4) {
        float4 t = *reinterpret_cast<const float4*>(p);
        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else if constexpr (V == 2) {
        float2 t = *reinterpret_cast<const float2*>(p);
        r[0] = t.x; r[1] = t.y;
    } else {
        r[0] = *p;
    }
}
template <int V>
__device__ __forceinline__ void vstore(float* p, const float (&r)[V]) {
    if constexpr (V == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
    } else if constexpr (V == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(r[0], r[1]);
    } else {
        *p = r[0];
    }
}

// the same with the non-temporal hint (rows that are touched once per step: no reuse to keep in L2 / Infinity Cache)
template <int V>
__device__ __forceinline__ void vload_nt(float (&r)[V], const float* p) {
    if constexpr (V == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else if constexpr (V == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 t = __builtin_nontemporal_load(reinterpret_cast<const f2*>(p));
        r[0] = t.x; r[1] = t.y;
    } else {
        r[0] = __builtin_nontemporal_load(p);
    }
}
template <int V>
__device__ __forceinline__ void vstore_nt(float* p, const float (&r)[V]) {
    if constexpr (V == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 t = {r[0], r[1], r[2], r[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<f4*>(p));
    } else if constexpr (V == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 t = {r[0], r[1]};
        __builtin_nontemporal_store(t, reinterpret_cast<f2*>(p));
    } else {
        __builtin_nontemporal_store(r[0], p);
    }
}

// sum over the LPR lanes that share a row (xor butterfly; every lane gets the result)
template <int LPR>
__device__ __forceinline__ float subwave_sum(float x) {
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// ---- the library's fixed-order sum of n floats by ONE block (dr_sum_f32; the fused BPR step's loss) ----
// 1024 "virtual threads": virtual thread t owns the float4 groups t, t + 1024, ... added in index order (lanes of a
// group in order), then a fixed tree over the 1024 partial sums in LDS; the < 4 trailing elements are added last.  A
// block of NT threads (NT divides 1024) runs 1024 / NT virtual threads per thread, so every NT gives the same bits.
template <int NT>
__device__ __forceinline__ void fixed_sum(const float* __restrict__ x, int64_t n, float* __restrict__ out, float* red) {
    constexpr int VT = 1024 / NT;
    const bool vec = ((uintptr_t)x % 16) == 0;
    const int64_t ngrp = vec ? n / 4 : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float acc[VT];
#pragma unroll
    for (int j = 0; j < VT; ++j) acc[j] = 0.0f;
    constexpr int DEPTH = VT >= 4 ? 32 / VT : 8;   // loads in flight per thread: VT * DEPTH
    int64_t i = threadIdx.x;
    for (; i + (VT - 1) * NT + (DEPTH - 1) * 1024 < ngrp; i += DEPTH * 1024) {
        float4 v[VT][DEPTH];
#pragma unroll
        for (int j = 0; j < VT; ++j)
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) v[j][q] = x4[i + j * NT + q * 1024];
#pragma unroll
        for (int j = 0; j < VT; ++j)
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) acc[j] = (((acc[j] + v[j][q].x) + v[j][q].y) + v[j][q].z) + v[j][q].w;
    }
#pragma unroll
    for (int j = 0; j < VT; ++j)
        for (int64_t g = i + j * NT; g < ngrp; g += 1024) {
            const float4 v = x4[g];
            acc[j] = (((acc[j] + v.x) + v.y) + v.z) + v.w;
        }
#pragma unroll
    for (int j = 0; j < VT; ++j) red[threadIdx.x + j * NT] = acc[j];
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        for (int t = threadIdx.x; t < s; t += NT) red[t] += red[t + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float t = red[0];
        for (int64_t e = ngrp * 4; e < n; ++e) t += x[e];
        out[0] = t;
    }
}

// ---- optimizer row math ([TF-ext] ApplyAdam / ApplyAdagrad / ApplyRMSProp / SGD) ----
// Separate roundings (the library is built with -ffp-contract=off) so the numpy oracle
// reproduces the same sequence.
struct OptP {
    int kind, reg_mode;
    float lr, alpha, omb1, omb2, eps, omrho, momentum, reg;
    const float* alpha_dev;
    const float* reg_dev;
};
// kernels call this once at entry: a device-resident alpha (graph replay) overrides the host value
__device__ __forceinline__ OptP resolve(OptP o) {
    if (o.alpha_dev) o.alpha = *o.alpha_dev;
    if (o.reg_dev) o.reg = *o.reg_dev;
    return o;
}
static inline OptP make_optp(const dr_opt_params* o) {
    OptP p;
    p.kind = o->kind;
    p.reg_mode = o->reg_mode;
    p.lr = o->lr;
    p.alpha = o->alpha;
    p.omb1 = 1.0f - o->beta1;
    p.omb2 = 1.0f - o->beta2;
    p.eps = o->eps;
    p.omrho = 1.0f - o->rho;
    p.momentum = o->momentum;
    p.reg = o->reg;
    p.alpha_dev = o->alpha_dev;
    p.reg_dev = o->reg_dev;
    return p;
}

template <int KIND>
__device__ __forceinline__ void opt_apply(float& w, float& s0, float& s1, float g, const OptP& o) {
    if constexpr (KIND == DR_OPT_SGD) {
        w = w - o.lr * g;
    } else if constexpr (KIND == DR_OPT_ADAM) {
        s0 = s0 + (g - s0) * o.omb1;
        s1 = s1 + (g * g - s1) * o.omb2;
        w = w - (s0 * o.alpha) / (sqrtf(s1) + o.eps);
    } else if constexpr (KIND == DR_OPT_ADAGRAD) {
        s0 = s0 + g * g;
        w = w - (g * o.lr) * (1.0f / sqrtf(s0));
    } else {  // RMSPROP
        s0 = s0 + (g * g - s0) * o.omrho;
        s1 = s1 * o.momentum + (g * o.lr) / sqrtf(s0 + o.eps);
        w = w - s1;
    }
}

}  // namespace dr
