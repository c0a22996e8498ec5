// K3/K4 (+K5 orchestration): per-sample interaction + loss kernels and the fused training steps.
//
// One sub-wave of LPR lanes owns one sample: it reads the sample's embedding rows as V-float
// vectors, reduces the dot products with an xor butterfly inside the sub-wave, evaluates the
// loss derivative once and writes the per-sample gradient seed g[b] plus the "stash" row that
// the P-side update sums (so table rows are only ever re-read in their pre-step state:
// Q is updated first from (g, P_old), P afterwards from the stash -- see DESIGN.md).
#include "dr_common.h"
#include "dr_segment.h"

#include <stdlib.h>

namespace dr {

int seg_launch(float* table, float* slot0, float* slot1, int64_t rows, int k, const float* src, int64_t src_rows,
               const int32_t* src_idx, const float* coef, int64_t n, void* plan, int64_t plan_bytes, int kind,
               const dr_opt_params* opt, hipStream_t st, bool meta_ready, SegSel sel, int64_t n_table,
               const float* src2 = nullptr, bool skip_single = false);
int plan_build_bpr(const int32_t* u, const int32_t* i, const int32_t* j, int64_t B, int64_t nu, int64_t ni, void* plan,
                   int64_t plan_bytes, int k, hipStream_t st, bool fused);
const unsigned char* plan_solo_flags(void* plan, int64_t n, int k);
int bpr_tail_launch(float* P, float* Ps0, float* Ps1, int64_t nu, float* Q, float* Qs0, float* Qs1, int64_t ni, int k,
                    int64_t B, const float* stash, const float* coefQ, void* plan, int64_t plan_bytes,
                    const dr_opt_params* opt, hipStream_t st, const float* loss_b, float* loss_out);
int check_opt(const dr_opt_params* opt, const float* slot0, const float* slot1, const char* who);

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- BPR forward (bprmf.py:84-92,118-119) ------------------------------------------------
template <int V, int LPR>
__global__ __launch_bounds__(256) void bpr_forward_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                          int64_t nu, int64_t ni, int k,
                                                          const int32_t* __restrict__ u, const int32_t* __restrict__ i,
                                                          const int32_t* __restrict__ j, int64_t B,
                                                          float* __restrict__ g_out, float* __restrict__ loss_b,
                                                          float* __restrict__ stash, float* __restrict__ coef2) {
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int cpr = k / V;
    const int64_t iters = (B + SPB - 1) / SPB;
    for (int64_t it = blockIdx.x; it < iters; it += gridDim.x) {
        const int64_t b = it * SPB + sub;
        const bool live = b < B;
        int64_t iu = 0, ii = 0, ij = 0;
        bool ok = false;
        if (live) {
            iu = u[b]; ii = i[b]; ij = j[b];
            ok = iu >= 0 && iu < nu && ii >= 0 && ii < ni && ij >= 0 && ij < ni;
        }
        float xui = 0.0f, xuj = 0.0f;
        float d0[V];
#pragma unroll
        for (int x = 0; x < V; ++x) d0[x] = 0.0f;
        if (ok) {
            for (int c = lane; c < cpr; c += LPR) {
                float pu[V], qi[V], qj[V];
                vload<V>(pu, P + iu * k + c * V);
                vload<V>(qi, Q + ii * k + c * V);
                vload<V>(qj, Q + ij * k + c * V);
#pragma unroll
                for (int x = 0; x < V; ++x) {
                    xui = xui + pu[x] * qi[x];
                    xuj = xuj + pu[x] * qj[x];
                }
                if (c == lane) {
#pragma unroll
                    for (int x = 0; x < V; ++x) d0[x] = qi[x] - qj[x];
                }
            }
        }
        xui = subwave_sum<LPR>(xui);
        xuj = subwave_sum<LPR>(xuj);
        float g = 0.0f, l = 0.0f;
        if (ok) {
            const float s = sigmoidf_(xui - xuj);
            l = -logf(s);
            g = -(1.0f - s);
        }
        if (live) {
            if (lane == 0) {
                g_out[b] = g;
                loss_b[b] = l;
                if (coef2) {  // coefficient list of the Q-side contributions [i ; j]
                    coef2[b] = g;
                    coef2[B + b] = -g;
                }
            }
            for (int c = lane; c < cpr; c += LPR) {
                float d[V];
                if (c == lane) {
#pragma unroll
                    for (int x = 0; x < V; ++x) d[x] = d0[x];
                } else if (ok) {
                    float qi[V], qj[V];
                    vload<V>(qi, Q + ii * k + c * V);
                    vload<V>(qj, Q + ij * k + c * V);
#pragma unroll
                    for (int x = 0; x < V; ++x) d[x] = qi[x] - qj[x];
                } else {
#pragma unroll
                    for (int x = 0; x < V; ++x) d[x] = 0.0f;
                }
#pragma unroll
                for (int x = 0; x < V; ++x) d[x] = g * d[x];
                vstore<V>(stash + b * k + c * V, d);
            }
        }
    }
}

// ---- BPR forward that also applies every single-contribution row (the planned step in a sparse regulariser mode) ----
// A row that exactly one (sample, slot) of the batch touches -- at C4 every user row but a few hundred and about
// half of the item rows -- needs no segment machinery: the sub-wave that has just read it for the dot products adds
// its one gradient contribution, L2 and the optimizer and writes it back (solo[] flags from the plan).  Saves the
// stash round trip, the second read of the row and of its source row, and most of the two segment launches.
// Arithmetic = the segment kernel's for a one-contribution segment (tot = 0 + c; tot = 0 + tot; apply_row), so the
// tables stay bit-identical to the unfused step.  stash row b = g * (q_i - q_j) for shared user rows (the tail sums them),
// stash row B + b = pre-step P[u_b] whenever one of the sample's item rows is shared (the tail reads it, SegArgs.src2).
struct FusedTables {
    SegArgs p, q;  // only table / slots / rows / k / kind / opt are used (apply_row)
};
template <int V, int LPR, int KIND>
__global__ __launch_bounds__(256) void bpr_forward_update_kernel(FusedTables t, const int32_t* __restrict__ u,
                                                                 const int32_t* __restrict__ i,
                                                                 const int32_t* __restrict__ j, int64_t B,
                                                                 const unsigned char* __restrict__ solo,
                                                                 float* __restrict__ g_out, float* __restrict__ loss_b,
                                                                 float* __restrict__ stash, float* __restrict__ coef2) {
    t.p.opt = resolve(t.p.opt);
    t.q.opt = t.p.opt;
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int k = t.p.k;
    const int cpr = k / V;
    const int64_t nu = t.p.rows, ni = t.q.rows;
    const float* __restrict__ P = t.p.table;
    const float* __restrict__ Q = t.q.table;
    const int64_t iters = (B + SPB - 1) / SPB;
    for (int64_t it = blockIdx.x; it < iters; it += gridDim.x) {
        const int64_t b = it * SPB + sub;
        const bool live = b < B;
        int64_t iu = 0, ii = 0, ij = 0;
        bool ok = false, su = false, si = false, sj = false;
        if (live) {
            iu = u[b]; ii = i[b]; ij = j[b];
            su = solo[b] != 0;
            si = solo[B + b] != 0;
            sj = solo[2 * B + b] != 0;
            ok = iu >= 0 && iu < nu && ii >= 0 && ii < ni && ij >= 0 && ij < ni;
        }
        const bool vu = live && iu >= 0 && iu < nu;  // a solo flag implies a valid id
        float xui = 0.0f, xuj = 0.0f;
        float p0[V], qi0[V], qj0[V];
        // optimizer slots of the solo rows (first column of this lane): requested together with the rows themselves,
        // so that the sample's whole traffic is one round of loads instead of two dependent ones
        float mu[V], vu2[V], mi[V], vi2[V], mj[V], vj2[V];
        const int nsl = (KIND == DR_OPT_ADAM) ? 2 : kind_slots(t.p.kind);
#pragma unroll
        for (int x = 0; x < V; ++x) p0[x] = qi0[x] = qj0[x] = mu[x] = vu2[x] = mi[x] = vi2[x] = mj[x] = vj2[x] = 0.0f;
        if (ok) {
            if (lane < cpr) {
                const int64_t ou = iu * k + lane * V, oi = ii * k + lane * V, oj = ij * k + lane * V;
                if (su && nsl >= 1) vload<V>(mu, t.p.slot0 + ou);
                if (su && nsl >= 2) vload<V>(vu2, t.p.slot1 + ou);
                if (si && nsl >= 1) vload<V>(mi, t.q.slot0 + oi);
                if (si && nsl >= 2) vload<V>(vi2, t.q.slot1 + oi);
                if (sj && nsl >= 1) vload<V>(mj, t.q.slot0 + oj);
                if (sj && nsl >= 2) vload<V>(vj2, t.q.slot1 + oj);
            }
            for (int c = lane; c < cpr; c += LPR) {
                float pu[V], qi[V], qj[V];
                vload<V>(pu, P + iu * k + c * V);
                vload<V>(qi, Q + ii * k + c * V);
                vload<V>(qj, Q + ij * k + c * V);
#pragma unroll
                for (int x = 0; x < V; ++x) {
                    xui = xui + pu[x] * qi[x];
                    xuj = xuj + pu[x] * qj[x];
                }
                if (c == lane) {
#pragma unroll
                    for (int x = 0; x < V; ++x) {
                        p0[x] = pu[x];
                        qi0[x] = qi[x];
                        qj0[x] = qj[x];
                    }
                }
            }
        }
        xui = subwave_sum<LPR>(xui);
        xuj = subwave_sum<LPR>(xuj);
        float g = 0.0f, l = 0.0f;
        if (ok) {
            const float s = sigmoidf_(xui - xuj);
            l = -logf(s);
            g = -(1.0f - s);
        }
        if (!live) continue;
        if (lane == 0) {
            g_out[b] = g;
            loss_b[b] = l;
            coef2[b] = g;
            coef2[B + b] = -g;
        }
        const bool need_p_old = vu && !(si && sj);  // some item-side contribution of this sample goes through the tail
        for (int c = lane; c < cpr; c += LPR) {
            float pu[V], qi[V], qj[V];
            if (c == lane && ok) {
#pragma unroll
                for (int x = 0; x < V; ++x) {
                    pu[x] = p0[x];
                    qi[x] = qi0[x];
                    qj[x] = qj0[x];
                }
            } else {
#pragma unroll
                for (int x = 0; x < V; ++x) pu[x] = qi[x] = qj[x] = 0.0f;
                if (vu) vload<V>(pu, P + iu * k + c * V);
                if (ok) {
                    vload<V>(qi, Q + ii * k + c * V);
                    vload<V>(qj, Q + ij * k + c * V);
                }
            }
            // the sample's own three contributions (zero rows when the sample is invalid: g = 0, d = 0)
            float d[V];
#pragma unroll
            for (int x = 0; x < V; ++x) d[x] = g * (ok ? (qi[x] - qj[x]) : 0.0f);
            const bool pre = ok && c == lane;  // rows and slots of this column are in registers
            // pre-step copy of the user row for the tail pass (its item-side contributions never read P in place)
            if (need_p_old) vstore<V>(stash + (B + b) * k + c * V, pu);
            if (su) {
                float tot[V];
#pragma unroll
                for (int x = 0; x < V; ++x) tot[x] = 0.0f + (0.0f + d[x]);
                if (pre) {
                    float w[V];
#pragma unroll
                    for (int x = 0; x < V; ++x) w[x] = pu[x];
                    apply_row_loaded<V, KIND>(t.p, (uint32_t)iu, c, tot, w, mu, vu2);
                } else {
                    apply_row<V, KIND>(t.p, (uint32_t)iu, c, tot);
                }
            } else {
                vstore<V>(stash + b * k + c * V, d);
            }
            if (si) {
                float tot[V];
#pragma unroll
                for (int x = 0; x < V; ++x) tot[x] = vu ? (0.0f + (0.0f + g * pu[x])) : 0.0f;
                if (pre) apply_row_loaded<V, KIND>(t.q, (uint32_t)ii, c, tot, qi, mi, vi2);
                else apply_row<V, KIND>(t.q, (uint32_t)ii, c, tot);
            }
            if (sj) {
                float tot[V];
                const float ng = -g;
#pragma unroll
                for (int x = 0; x < V; ++x) tot[x] = vu ? (0.0f + (0.0f + ng * pu[x])) : 0.0f;
                if (pre) apply_row_loaded<V, KIND>(t.q, (uint32_t)ij, c, tot, qj, mj, vj2);
                else apply_row<V, KIND>(t.q, (uint32_t)ij, c, tot);
            }
        }
    }
}

// The same for k == V * LPR (every lane owns exactly one vector column; k = 128, 64, 32, ...): no column loops and no
// reload paths, every row and optimizer slot of the sample is requested in one round, and the per-id validity makes
// the update path uniform (an invalid sample contributes zero rows to its valid solo rows, like the segment path).
// stash rows [B, 2B): pre-step P[u_b] of samples with a shared item row (read by their item-side contributions).
// DR_FWD_NT: non-temporal hint on the rows the fused forward kernel touches ONCE per step -- 1: the user row and its slots
// (loads and stores), 2: + the stores of solo item rows, 3: + the slot loads of solo item rows.  Those rows have no
// reuse to keep in L2 / Infinity Cache, and without the hint they evict the item rows that do (Zipf head).  Measured on the
// C4 step (bench.py, median of 25 windows): 0.1555 ms -> 0.1488 (1) -> 0.1482 (2) -> 0.1465 (3); the same hint in the tail
// kernel: no further change; on the row LOADS of solo item rows as well (a branch per row): 0.1484, slightly slower.
#ifndef DR_FWD_NT
#define DR_FWD_NT 3
#endif
template <int V, int LPR, int KIND>
__global__ __launch_bounds__(256) void bpr_forward_update_1col_kernel(FusedTables t, const int32_t* __restrict__ u,
                                                                      const int32_t* __restrict__ i,
                                                                      const int32_t* __restrict__ j, int64_t B,
                                                                      const unsigned char* __restrict__ solo,
                                                                      float* __restrict__ g_out,
                                                                      float* __restrict__ loss_b,
                                                                      float* __restrict__ stash,
                                                                      float* __restrict__ coef2) {
    t.p.opt = resolve(t.p.opt);
    t.q.opt = t.p.opt;
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int k = t.p.k;
    const int64_t nu = t.p.rows, ni = t.q.rows;
    const int nsl = (KIND == DR_OPT_ADAM) ? 2 : kind_slots(t.p.kind);
    const int64_t iters = (B + SPB - 1) / SPB;
    for (int64_t it = blockIdx.x; it < iters; it += gridDim.x) {
        const int64_t b = it * SPB + sub;
        const bool live = b < B;
        int64_t iu = 0, ii = 0, ij = 0;
        bool su = false, si = false, sj = false;
        if (live) {
            iu = u[b]; ii = i[b]; ij = j[b];
            su = solo[b] != 0;
            si = solo[B + b] != 0;
            sj = solo[2 * B + b] != 0;
        }
        const bool vu = live && iu >= 0 && iu < nu, vi = live && ii >= 0 && ii < ni, vj = live && ij >= 0 && ij < ni;
        const bool ok = vu && vi && vj;
        // Row loads are UNCONDITIONAL (an id out of range reads row 0 and is zeroed by a select afterwards) and the slot
        // registers carry no initial value: written as `x = 0; if (valid) load(x)` every load was followed by
        // s_waitcnt vmcnt(0) -- the zero written on the other path may not overtake a load in flight into the same
        // register -- so the three row fetches of a sample were three random-HBM round trips in a row, not one.
        const int64_t ou = (vu ? iu : 0) * k + lane * V, oi = (vi ? ii : 0) * k + lane * V, oj = (vj ? ij : 0) * k + lane * V;
        float pu[V], qi[V], qj[V], mu[V], vu2[V], mi[V], vi2[V], mj[V], vj2[V];
#if DR_FWD_NT >= 1
        vload_nt<V>(pu, t.p.table + ou);
#else
        vload<V>(pu, t.p.table + ou);
#endif
        vload<V>(qi, t.q.table + oi);
        vload<V>(qj, t.q.table + oj);
        if (su) {   // solo rows are valid rows; slots beyond the optimizer's are never read
#if DR_FWD_NT >= 1
            if (nsl >= 1) vload_nt<V>(mu, t.p.slot0 + ou);
            if (nsl >= 2) vload_nt<V>(vu2, t.p.slot1 + ou);
#else
            if (nsl >= 1) vload<V>(mu, t.p.slot0 + ou);
            if (nsl >= 2) vload<V>(vu2, t.p.slot1 + ou);
#endif
        }
        if (si) {
#if DR_FWD_NT >= 3
            if (nsl >= 1) vload_nt<V>(mi, t.q.slot0 + oi);
            if (nsl >= 2) vload_nt<V>(vi2, t.q.slot1 + oi);
#else
            if (nsl >= 1) vload<V>(mi, t.q.slot0 + oi);
            if (nsl >= 2) vload<V>(vi2, t.q.slot1 + oi);
#endif
        }
        if (sj) {
#if DR_FWD_NT >= 3
            if (nsl >= 1) vload_nt<V>(mj, t.q.slot0 + oj);
            if (nsl >= 2) vload_nt<V>(vj2, t.q.slot1 + oj);
#else
            if (nsl >= 1) vload<V>(mj, t.q.slot0 + oj);
            if (nsl >= 2) vload<V>(vj2, t.q.slot1 + oj);
#endif
        }
#pragma unroll
        for (int x = 0; x < V; ++x) {
            pu[x] = vu ? pu[x] : 0.0f;
            qi[x] = vi ? qi[x] : 0.0f;
            qj[x] = vj ? qj[x] : 0.0f;
        }
        float xui = 0.0f, xuj = 0.0f;
        if (ok) {
#pragma unroll
            for (int x = 0; x < V; ++x) {
                xui = xui + pu[x] * qi[x];
                xuj = xuj + pu[x] * qj[x];
            }
        }
        xui = subwave_sum<LPR>(xui);
        xuj = subwave_sum<LPR>(xuj);
        float g = 0.0f, l = 0.0f;
        if (ok) {
            const float s = sigmoidf_(xui - xuj);
            l = -logf(s);
            g = -(1.0f - s);
        }
        if (!live) continue;
        if (lane == 0) {
            g_out[b] = g;
            loss_b[b] = l;
            coef2[b] = g;
            coef2[B + b] = -g;
        }
        float d[V];
#pragma unroll
        for (int x = 0; x < V; ++x) d[x] = g * (ok ? (qi[x] - qj[x]) : 0.0f);
        float tot[V];
        if (si) {
#pragma unroll
            for (int x = 0; x < V; ++x) tot[x] = vu ? (0.0f + (0.0f + g * pu[x])) : 0.0f;
            apply_row_loaded<V, KIND, (DR_FWD_NT >= 2)>(t.q, (uint32_t)ii, lane, tot, qi, mi, vi2);
        }
        if (sj) {
            const float ng = -g;
#pragma unroll
            for (int x = 0; x < V; ++x) tot[x] = vu ? (0.0f + (0.0f + ng * pu[x])) : 0.0f;
            apply_row_loaded<V, KIND, (DR_FWD_NT >= 2)>(t.q, (uint32_t)ij, lane, tot, qj, mj, vj2);
        }
        // pre-step copy of the user row (stash row B + b) for the tail pass: item-side contributions never read P in place
        if (vu && !(si && sj)) vstore<V>(stash + (B + b) * k + lane * V, pu);
        if (su) {
#pragma unroll
            for (int x = 0; x < V; ++x) tot[x] = 0.0f + (0.0f + d[x]);
            apply_row_loaded<V, KIND, (DR_FWD_NT >= 1)>(t.p, (uint32_t)iu, lane, tot, pu, mu, vu2);
        } else {
            vstore<V>(stash + b * k + lane * V, d);   // a shared user row: summed and applied by the tail pass
        }
    }
}

// idsQ = [i ; j], srcQ = [u ; u]   (needs only the batch ids: runs beside the forward kernel)
__global__ __launch_bounds__(256) void bpr_qids_kernel(const int32_t* __restrict__ u, const int32_t* __restrict__ i,
                                                       const int32_t* __restrict__ j, int64_t B,
                                                       int32_t* __restrict__ idsQ, int32_t* __restrict__ srcQ) {
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (int64_t)gridDim.x * 256) {
        idsQ[b] = i[b];
        idsQ[B + b] = j[b];
        srcQ[b] = u[b];
        srcQ[B + b] = u[b];
    }
}

// coefQ = [g ; -g]
__global__ __launch_bounds__(256) void bpr_qcoef_kernel(const float* __restrict__ g, int64_t B,
                                                        float* __restrict__ coefQ) {
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (int64_t)gridDim.x * 256) {
        const float gb = g[b];
        coefQ[b] = gb;
        coefQ[B + b] = -gb;
    }
}

// Helper stream on which the segment plan of a step is built while the forward kernel runs on the caller's stream
// (it depends on the batch ids only): dr_init()'s objects (dr_common.h: helpers()); fork / join with events, so the
// caller still sees a single in-order stream (also under graph capture).  Without dr_init(): the caller's stream alone.
struct SideStream {
    hipStream_t s;
    hipEvent_t fork, q_ready, p_ready;
};
static bool side_stream(SideStream* out) {
    static const bool off = DR_EXP_ENV("DR_NO_OVERLAP") != nullptr;
    Helpers* h = off ? nullptr : helpers();
    if (!h) return false;
    out->s = h->plan;
    out->fork = h->fork;
    out->q_ready = h->q_ready;
    out->p_ready = h->p_ready;
    return true;
}

// ---- pointwise forward: GMF BCE (gmf.py:66-75) / MF squared error (mf.py:50-55,96-97) ----
template <int V, int LPR, bool SQ>
__global__ __launch_bounds__(256) void pointwise_forward_kernel(
    const float* __restrict__ P, const float* __restrict__ Q, const float* __restrict__ BU,
    const float* __restrict__ BI, float mu, int64_t nu, int64_t ni, int k, const int32_t* __restrict__ u,
    const int32_t* __restrict__ i, const float* __restrict__ y, int64_t B, float* __restrict__ g_out,
    float* __restrict__ loss_b, float* __restrict__ pred, float* __restrict__ stash) {
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int cpr = k / V;
    const int64_t iters = (B + SPB - 1) / SPB;
    for (int64_t it = blockIdx.x; it < iters; it += gridDim.x) {
        const int64_t b = it * SPB + sub;
        const bool live = b < B;
        int64_t iu = 0, ii = 0;
        bool ok = false;
        if (live) {
            iu = u[b]; ii = i[b];
            ok = iu >= 0 && iu < nu && ii >= 0 && ii < ni;
        }
        float z = 0.0f;
        float q0[V];
#pragma unroll
        for (int x = 0; x < V; ++x) q0[x] = 0.0f;
        if (ok) {
            for (int c = lane; c < cpr; c += LPR) {
                float pu[V], qi[V];
                vload<V>(pu, P + iu * k + c * V);
                vload<V>(qi, Q + ii * k + c * V);
#pragma unroll
                for (int x = 0; x < V; ++x) z = z + pu[x] * qi[x];
                if (c == lane) {
#pragma unroll
                    for (int x = 0; x < V; ++x) q0[x] = qi[x];
                }
            }
        }
        z = subwave_sum<LPR>(z);
        float g = 0.0f, l = 0.0f, out = 0.0f;
        if (ok) {
            if constexpr (SQ) {
                z = z + BU[iu] + BI[ii] + mu;
                out = z;
                if (y) {
                    const float d = y[b] - z;
                    l = d * d;
                    g = -2.0f * d;
                }
            } else {
                const float p = sigmoidf_(z);
                out = p;
                if (y) {
                    const float yy = y[b];
                    const float e = 1e-10f;
                    l = -(yy * logf(p + e) + (1.0f - yy) * logf(1.0f - p + e));
                    const float dp = -(yy / (p + e)) + (1.0f - yy) / (1.0f - p + e);
                    g = dp * (p * (1.0f - p));
                }
            }
        }
        if (live) {
            if (lane == 0) {
                if (g_out) g_out[b] = g;
                if (loss_b) loss_b[b] = l;
                if (pred) pred[b] = out;
            }
            if (stash)
                for (int c = lane; c < cpr; c += LPR) {
                    float d[V];
                    if (c == lane) {
#pragma unroll
                        for (int x = 0; x < V; ++x) d[x] = q0[x];
                    } else if (ok) {
                        vload<V>(d, Q + ii * k + c * V);
                    } else {
#pragma unroll
                        for (int x = 0; x < V; ++x) d[x] = 0.0f;
                    }
#pragma unroll
                    for (int x = 0; x < V; ++x) d[x] = g * d[x];
                    vstore<V>(stash + b * k + c * V, d);
                }
        }
    }
}

// ---- CML hinge forward (cml.py:68-77): stash[b] = dL/dq_i, stash[B+b] = dL/dq_j ----------
template <int V, int LPR>
__global__ __launch_bounds__(256) void hinge_forward_kernel(
    const float* __restrict__ P, const float* __restrict__ Q, int64_t nu, int64_t ni, int k,
    const int32_t* __restrict__ u, const int32_t* __restrict__ i, const int32_t* __restrict__ j,
    const uint8_t* __restrict__ mask_pos, const uint8_t* __restrict__ mask_neg, float keep, float margin, int64_t B,
    float* __restrict__ loss_b, float* __restrict__ stash) {
    constexpr int SPB = 256 / LPR;
    const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
    const int cpr = k / V;
    const float sc = 1.0f / keep;
    const int64_t iters = (B + SPB - 1) / SPB;
    for (int64_t it = blockIdx.x; it < iters; it += gridDim.x) {
        const int64_t b = it * SPB + sub;
        const bool live = b < B;
        int64_t iu = 0, ii = 0, ij = 0;
        bool ok = false;
        if (live) {
            iu = u[b]; ii = i[b]; ij = j[b];
            ok = iu >= 0 && iu < nu && ii >= 0 && ii < ni && ij >= 0 && ij < ni;
        }
        float ap = 0.0f, an = 0.0f;
        if (ok) {
            for (int c = lane; c < cpr; c += LPR) {
                float pu[V], qi[V], qj[V];
                vload<V>(pu, P + iu * k + c * V);
                vload<V>(qi, Q + ii * k + c * V);
                vload<V>(qj, Q + ij * k + c * V);
#pragma unroll
                for (int x = 0; x < V; ++x) {
                    const float mp = (mask_pos ? (float)mask_pos[b * k + c * V + x] : 1.0f) * sc;
                    const float mn = (mask_neg ? (float)mask_neg[b * k + c * V + x] : 1.0f) * sc;
                    const float dp = pu[x] - qi[x], dn = pu[x] - qj[x];
                    ap = ap + (dp * dp) * mp;
                    an = an + (dn * dn) * mn;
                }
            }
        }
        ap = subwave_sum<LPR>(ap);
        an = subwave_sum<LPR>(an);
        const float a = ap - an + margin;
        const float act = (ok && a >= 0.0f) ? 1.0f : 0.0f;
        if (live) {
            if (lane == 0) loss_b[b] = ok ? fmaxf(a, 0.0f) : 0.0f;
            for (int c = lane; c < cpr; c += LPR) {
                float gi[V], gj[V];
                if (ok) {
                    float pu[V], qi[V], qj[V];
                    vload<V>(pu, P + iu * k + c * V);
                    vload<V>(qi, Q + ii * k + c * V);
                    vload<V>(qj, Q + ij * k + c * V);
#pragma unroll
                    for (int x = 0; x < V; ++x) {
                        const float mp = (mask_pos ? (float)mask_pos[b * k + c * V + x] : 1.0f) * sc;
                        const float mn = (mask_neg ? (float)mask_neg[b * k + c * V + x] : 1.0f) * sc;
                        gi[x] = -(act * ((2.0f * mp) * (pu[x] - qi[x])));
                        gj[x] = act * ((2.0f * mn) * (pu[x] - qj[x]));
                    }
                } else {
#pragma unroll
                    for (int x = 0; x < V; ++x) gi[x] = gj[x] = 0.0f;
                }
                vstore<V>(stash + b * k + c * V, gi);
                vstore<V>(stash + (B + b) * k + c * V, gj);
            }
        }
    }
}

__global__ __launch_bounds__(256) void hinge_lists_kernel(const int32_t* __restrict__ u, const int32_t* __restrict__ i,
                                                          const int32_t* __restrict__ j, int64_t B,
                                                          int32_t* __restrict__ idsP, int32_t* __restrict__ idsQ,
                                                          float* __restrict__ coefP) {
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (int64_t)gridDim.x * 256) {
        idsP[b] = u[b];
        idsP[B + b] = u[b];
        idsQ[b] = i[b];
        idsQ[B + b] = j[b];
        coefP[b] = -1.0f;
        coefP[B + b] = -1.0f;
    }
}

// ---- per-pair k-ordered fmaf chain (bprmf.py:168-173) --------------------------------------
__global__ __launch_bounds__(256) void pair_dot_kernel(const float* __restrict__ A, int64_t rowsA,
                                                       const float* __restrict__ Bm, int64_t rowsB, int k,
                                                       const int32_t* __restrict__ ai, const int32_t* __restrict__ bi,
                                                       int64_t n, float* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t a = ai[e], b = bi[e];
        float acc = 0.0f;
        if (a >= 0 && a < rowsA && b >= 0 && b < rowsB) {
            const float* pa = A + a * k;
            const float* pb = Bm + b * k;
            for (int c = 0; c < k; ++c) acc = __builtin_fmaf(pa[c], pb[c], acc);
        }
        out[e] = acc;
    }
}

#define DR_SHAPE_DISPATCH(MACRO)                                                                     \
    MACRO(4, 1) MACRO(4, 2) MACRO(4, 4) MACRO(4, 8) MACRO(4, 16) MACRO(4, 32) MACRO(4, 64)           \
    MACRO(2, 1) MACRO(2, 2) MACRO(2, 4) MACRO(2, 8) MACRO(2, 16) MACRO(2, 32) MACRO(2, 64)           \
    MACRO(1, 1) MACRO(1, 2) MACRO(1, 4) MACRO(1, 8) MACRO(1, 16) MACRO(1, 32) MACRO(1, 64)

// workspace layout of the fused steps
struct StepLayout {
    int64_t g, loss_b, stash, idsA, idsB, coef, src, planA, planB, gdenseP, gdenseQ, total, plan_bytesA, plan_bytesB;
};
}  // namespace dr

using namespace dr;

extern "C" int64_t dr_segment_plan_bytes(int64_t n, int32_t k_max);
extern "C" int32_t dr_segment_plan(const void* ids, int64_t n, int32_t id_bits, int64_t rows, void* plan,
                                   int64_t plan_bytes, void* stream);

namespace dr {
// stash_rows: B for BPR / pointwise, 2B for hinge.  dense tables only in DR_REG_DENSE.
static int step_layout(int64_t B, int k, int64_t stash_rows, int64_t nu, int64_t ni, bool dense, StepLayout* L) {
    int64_t BB = B > 0 ? B : 1;
    int64_t pa = dr_segment_plan_bytes(3 * BB, k), pb = dr_segment_plan_bytes(2 * BB, k);
    if (pa < 0 || pb < 0) return -1;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    L->g = take(BB * 4);
    L->loss_b = take(BB * 4);
    L->stash = take((stash_rows > 0 ? stash_rows : 1) * (int64_t)k * 4);
    L->idsA = take(2 * BB * 4);
    L->idsB = take(2 * BB * 4);
    L->coef = take(2 * BB * 4);
    L->src = take(2 * BB * 4);
    L->planA = take(pa);
    L->planB = take(pb);
    L->plan_bytesA = pa;
    L->plan_bytesB = pb;
    L->gdenseP = dense ? take(nu * (int64_t)k * 4) : 0;
    L->gdenseQ = dense ? take(ni * (int64_t)k * 4) : 0;
    L->total = off;
    return 0;
}

// One table's update from a contribution list whose plan is already built, in either sparse (touched / none)
// or dense mode.
static int table_apply(float* T, float* s0, float* s1, int64_t rows, int k, int64_t n, const float* src,
                       int64_t src_rows, const int32_t* src_idx, const float* coef, void* plan, int64_t plan_bytes,
                       float* gdense, const dr_opt_params* opt, hipStream_t st, bool meta_ready = false,
                       SegSel sel = SegSel{0, 0, 0}, int64_t n_table = 0, const float* src2 = nullptr,
                       bool skip_single = false) {
    int rc;
    if (opt->reg_mode == DR_REG_DENSE) {
        DR_HIP(hipMemsetAsync(gdense, 0, (size_t)rows * k * 4, st));
        rc = seg_launch(gdense, nullptr, nullptr, rows, k, src, src_rows, src_idx, coef, n, plan, plan_bytes, KIND_ACC, nullptr, st,
                        meta_ready, sel, n_table);
        if (rc != DR_OK) return rc;
        return dr_dense_update_f32(T, s0, s1, gdense, rows * (int64_t)k, opt, st);
    }
    return seg_launch(T, s0, s1, rows, k, src, src_rows, src_idx, coef, n, plan, plan_bytes, opt->kind, opt, st, meta_ready, sel,
                      n_table, src2, skip_single);
}

static int table_update(float* T, float* s0, float* s1, int64_t rows, int k, const int32_t* ids, int64_t n,
                        const float* src, int64_t src_rows, const int32_t* src_idx, const float* coef, void* plan,
                        int64_t plan_bytes, float* gdense, const dr_opt_params* opt, hipStream_t st) {
    int rc = dr_segment_plan(ids, n, 32, rows, plan, plan_bytes, st);
    if (rc != DR_OK) return rc;
    return table_apply(T, s0, s1, rows, k, n, src, src_rows, src_idx, coef, plan, plan_bytes, gdense, opt, st);
}
}  // namespace dr

namespace dr {
static int bpr_forward(const float* P, const float* Q, int64_t num_user, int64_t num_item, int32_t k, const int32_t* u,
                       const int32_t* i, const int32_t* j, int64_t B, float* g_out, float* loss_b, float* stash,
                       float* coef2, void* stream) {
    DR_REQUIRE(num_user >= 0 && num_item >= 0 && k >= 1 && B >= 0, "dr_bpr_forward_f32: bad sizes");
    if (B == 0) return DR_OK;
    DR_REQUIRE(P && Q && u && i && j && g_out && loss_b && stash, "dr_bpr_forward_f32: NULL pointer");
    RowShape s = row_shape(k);
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
#define DR_F(V_, L_)                                                                                                \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                          \
        DR_LAUNCH_NAMED("bpr_forward_kernel", (bpr_forward_kernel<V_, L_>),                                         \
                        dim3(grid_resident(bpr_forward_kernel<V_, L_>, (B + 256 / L_ - 1) / (256 / L_))),           \
                        dim3(256), 0, st, P, Q, num_user, num_item, k, u, i, j, B, g_out, loss_b, stash, coef2);    \
        ok = true;                                                                                                  \
    }
    DR_SHAPE_DISPATCH(DR_F)
#undef DR_F
    DR_REQUIRE(ok, "dr_bpr_forward_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("dr_bpr_forward_f32");
    return DR_OK;
}
}  // namespace dr

namespace dr {
static int bpr_forward_update(float* P, float* Ps0, float* Ps1, int64_t num_user, float* Q, float* Qs0, float* Qs1,
                              int64_t num_item, int32_t k, const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                              const dr_opt_params* opt, const unsigned char* solo, float* g_out, float* loss_b,
                              float* stash, float* coef2, hipStream_t st) {
    FusedTables t;
    memset(&t, 0, sizeof(t));
    t.p.table = P; t.p.slot0 = Ps0; t.p.slot1 = Ps1; t.p.rows = num_user; t.p.k = k; t.p.kind = opt->kind;
    t.p.opt = make_optp(opt);
    t.q = t.p;
    t.q.table = Q; t.q.slot0 = Qs0; t.q.slot1 = Qs1; t.q.rows = num_item;
    RowShape s = row_shape(k);
    bool ok = false;
    static const int64_t fwd_grid = DR_EXP_ENV("DR_FWD_GRID") ? atoll(DR_EXP_ENV("DR_FWD_GRID")) : 0;
#define DR_F(V_, L_)                                                                                                  \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                            \
        int64_t wb_ = (B + 256 / L_ - 1) / (256 / L_);                                                                 \
        if (fwd_grid > 0 && wb_ > fwd_grid) wb_ = fwd_grid;                                                           \
        if (k == V_ * L_ && opt->kind == DR_OPT_ADAM)                                                                 \
            DR_LAUNCH_NAMED("bpr_forward_update_kernel", (bpr_forward_update_1col_kernel<V_, L_, DR_OPT_ADAM>),       \
                            dim3(grid_resident(bpr_forward_update_1col_kernel<V_, L_, DR_OPT_ADAM>, wb_)), dim3(256),  \
                            0, st, t, u, i, j, B, solo, g_out, loss_b, stash, coef2);                                 \
        else if (k == V_ * L_)                                                                                        \
            DR_LAUNCH_NAMED("bpr_forward_update_kernel", (bpr_forward_update_1col_kernel<V_, L_, KIND_ANY>),          \
                            dim3(grid_resident(bpr_forward_update_1col_kernel<V_, L_, KIND_ANY>, wb_)), dim3(256), 0,  \
                            st, t, u, i, j, B, solo, g_out, loss_b, stash, coef2);                                    \
        else                                                                                                          \
            DR_LAUNCH_NAMED("bpr_forward_update_kernel", (bpr_forward_update_kernel<V_, L_, KIND_ANY>),               \
                            dim3(grid_resident(bpr_forward_update_kernel<V_, L_, KIND_ANY>, wb_)), dim3(256), 0, st,   \
                            t, u, i, j, B, solo, g_out, loss_b, stash, coef2);                                        \
        ok = true;                                                                                                    \
    }
    DR_SHAPE_DISPATCH(DR_F)
#undef DR_F
    DR_REQUIRE(ok, "dr_bpr_step_planned_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("bpr_forward_update_kernel");
    return DR_OK;
}
}  // namespace dr

extern "C" int32_t dr_bpr_forward_f32(const float* P, const float* Q, int64_t num_user, int64_t num_item, int32_t k,
                                      const int32_t* u, const int32_t* i, const int32_t* j, int64_t B, float* g_out,
                                      float* loss_b, float* stash, void* stream) {
    return dr::bpr_forward(P, Q, num_user, num_item, k, u, i, j, B, g_out, loss_b, stash, nullptr, stream);
}

extern "C" int64_t dr_pairwise_step_bytes(int64_t B, int32_t k, int64_t num_user, int64_t num_item,
                                          int32_t reg_mode) {
    if (B < 0 || k < 1) return DR_ERR_ARG;
    StepLayout L;
    if (step_layout(B, k, 2 * B, num_user, num_item, reg_mode == DR_REG_DENSE, &L) != 0) return DR_ERR_LAUNCH;
    return L.total;
}

namespace dr {
// mode 0: whole step; 1: only the id lists + both segment plans (they depend on the batch ids alone);
// 2: the step, with the plans of THIS batch already in ws (built by a mode-1 call)
static int bpr_step_impl(float* P, float* Pslot0, float* Pslot1, int64_t num_user, float* Q, float* Qslot0,
                         float* Qslot1, int64_t num_item, int32_t k, const int32_t* u, const int32_t* i,
                         const int32_t* j, int64_t B, const dr_opt_params* opt, int reg_mode, float* loss_out, void* ws,
                         int64_t ws_bytes, void* stream, int mode) {
    DR_REQUIRE(num_user >= 0 && num_item >= 0 && k >= 1 && B >= 0, "dr_bpr_step_f32: bad sizes");
    int rc;
    if (mode != 1) {
        rc = check_opt(opt, Pslot0, Pslot1, "dr_bpr_step_f32(P)");
        if (rc != DR_OK) return rc;
        rc = check_opt(opt, Qslot0, Qslot1, "dr_bpr_step_f32(Q)");
        if (rc != DR_OK) return rc;
        DR_REQUIRE(P && Q, "dr_bpr_step_f32: table pointer is NULL");
        reg_mode = opt->reg_mode;
    }
    DR_REQUIRE(ws && ((uintptr_t)ws % 256 == 0), "dr_bpr_step_f32: NULL / misaligned workspace");
    DR_REQUIRE(B == 0 || (u && i && j), "dr_bpr_step_f32: id pointer is NULL");
    StepLayout L;
    if (step_layout(B, k, 2 * B, num_user, num_item, reg_mode == DR_REG_DENSE, &L) != 0) return DR_ERR_LAUNCH;
    if (ws_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_bpr_step_f32: workspace %lld < required %lld", (long long)ws_bytes,
                (long long)L.total);
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    float* g = (float*)(base + L.g);
    float* loss_b = (float*)(base + L.loss_b);
    float* stash = (float*)(base + L.stash);
    int32_t* idsQ = (int32_t*)(base + L.idsA);
    float* coefQ = (float*)(base + L.coef);
    int32_t* srcQ = (int32_t*)(base + L.src);
    const bool any = B > 0 || reg_mode == DR_REG_DENSE;
    // Fork (mode 0): the id lists and both segment plans depend on the batch ids only, so they are built on the
    // helper stream while the forward kernel runs here; the updates join on events.
    SideStream side_obj;
    SideStream* side = (mode == 0 && B > 0 && side_stream(&side_obj)) ? &side_obj : nullptr;
    hipStream_t ps = side ? side->s : st;
    if (side) {
        DR_HIP(hipEventRecord(side->fork, st));
        DR_HIP(hipStreamWaitEvent(side->s, side->fork, 0));
    }
    // The planned step in a sparse regulariser mode is FUSED: the forward kernel applies every single-contribution row
    // itself (flags in the plan), the segment kernels only see rows with two or more contributions.
    const bool fused = mode != 0 && reg_mode != DR_REG_DENSE;
    if (mode != 1 && B > 0) {  // enqueued first: the host then feeds the helper stream while this runs
        if (fused && mode == 2) {
            const unsigned char* solo = plan_solo_flags(base + L.planA, 3 * B, k);
            DR_REQUIRE(solo != nullptr, "dr_bpr_step_planned_f32: plan layout failed");
            rc = bpr_forward_update(P, Pslot0, Pslot1, num_user, Q, Qslot0, Qslot1, num_item, k, u, i, j, B, opt, solo, g,
                                    loss_b, stash, coefQ, st);
        } else {
            rc = bpr_forward(P, Q, num_user, num_item, k, u, i, j, B, g, loss_b, stash, coefQ, st);
        }
        if (rc != DR_OK) return rc;
    }
    if (mode != 2) {
        if (any) {
            rc = plan_build_bpr(u, i, j, B, num_user, num_item, base + L.planA, L.plan_bytesA, k, ps, fused);
            if (rc != DR_OK) return rc;
            if (side) DR_HIP(hipEventRecord(side->q_ready, side->s));
        }
        if (mode == 1) return DR_OK;
    }
    if (any && fused && mode == 2) {
        // every row with two or more contributions, both tables, one launch (bpr_tail_kernel): rows of the two tables
        // are disjoint, Q-side sources are pre-step P rows (in place for non-solo users, the stash copy for solo ones)
        // and P-side sources are stash rows, none of which this launch writes
        rc = bpr_tail_launch(P, Pslot0, Pslot1, num_user, Q, Qslot0, Qslot1, num_item, k, B, stash, coefQ,
                             base + L.planA, L.plan_bytesA, opt, st, loss_b, loss_out);
        if (rc != DR_OK) return rc;
        return DR_OK;   // the loss was summed by block 0 of the tail's last launch (same order and bits as dr_sum_f32)
    } else if (any) {
        // one combined plan: segments below the split are P rows (contributions = stash rows b), the others Q
        // rows (key - num_user; contributions (+-g_b) * P[u_b]).  Q first: its gradient needs the pre-step P rows.
        if (side) DR_HIP(hipStreamWaitEvent(st, side->q_ready, 0));
        rc = table_apply(Q, Qslot0, Qslot1, num_item, k, 3 * B, P, num_user, nullptr, coefQ, base + L.planA,
                         L.plan_bytesA, (float*)(base + L.gdenseQ), opt, st, true,
                         SegSel{(uint32_t)num_user, 2, (int)B}, 2 * B);
        if (rc != DR_OK) return rc;
        rc = table_apply(P, Pslot0, Pslot1, num_user, k, 3 * B, stash, B, nullptr, nullptr, base + L.planA,
                         L.plan_bytesA, (float*)(base + L.gdenseP), opt, st, true, SegSel{0, 1, 0}, B);
        if (rc != DR_OK) return rc;
    }
    if (loss_out) return dr_sum_f32(loss_b, B, loss_out, st);
    return DR_OK;
}
}  // namespace dr

extern "C" int32_t dr_bpr_step_f32(float* P, float* Pslot0, float* Pslot1, int64_t num_user, float* Q, float* Qslot0,
                                   float* Qslot1, int64_t num_item, int32_t k, const int32_t* u, const int32_t* i,
                                   const int32_t* j, int64_t B, const dr_opt_params* opt, float* loss_out, void* ws,
                                   int64_t ws_bytes, void* stream) {
    return bpr_step_impl(P, Pslot0, Pslot1, num_user, Q, Qslot0, Qslot1, num_item, k, u, i, j, B, opt, 0, loss_out, ws,
                         ws_bytes, stream, 0);
}

extern "C" int32_t dr_bpr_plan_f32(int64_t num_user, int64_t num_item, int32_t k, const int32_t* u, const int32_t* i,
                                   const int32_t* j, int64_t B, int32_t reg_mode, void* ws, int64_t ws_bytes,
                                   void* stream) {
    return bpr_step_impl(nullptr, nullptr, nullptr, num_user, nullptr, nullptr, nullptr, num_item, k, u, i, j, B, nullptr,
                         reg_mode, nullptr, ws, ws_bytes, stream, 1);
}

extern "C" int32_t dr_bpr_step_planned_f32(float* P, float* Pslot0, float* Pslot1, int64_t num_user, float* Q,
                                           float* Qslot0, float* Qslot1, int64_t num_item, int32_t k, const int32_t* u,
                                           const int32_t* i, const int32_t* j, int64_t B, const dr_opt_params* opt,
                                           float* loss_out, void* ws, int64_t ws_bytes, void* stream) {
    return bpr_step_impl(P, Pslot0, Pslot1, num_user, Q, Qslot0, Qslot1, num_item, k, u, i, j, B, opt, 0, loss_out, ws,
                         ws_bytes, stream, 2);
}

// ---- dr_bpr_step_static_f32: dr_bpr_step_f32 for callers that cannot allocate ------------------------------------
// The step's workspace (g, per-sample loss, stash rows, id / coefficient lists, the combined segment plan with its
// block partials) lives in a file-scope device array instead of a caller-provided buffer.  `scratch` below is the
// HOST-side stand-in of that array: its device address is looked up once per device with hipGetSymbolAddress and
// only that pointer is ever handed to kernels.  Every GPU holds its own copy until the process ends.
#define DR_STATIC_WS_BYTES (4ll << 20)
static __device__ __align__(256) unsigned char scratch[DR_STATIC_WS_BYTES];

namespace dr {
static void* g_scratch_ptr[32] = {nullptr};
// device address of this device's `scratch`: looked up by dr_init() (abi.hip), or -- for callers that never called it --
// on the first static step on each device
int static_scratch_init() {
    int dev = -1;
    DR_HIP(hipGetDevice(&dev));
    DR_REQUIRE(dev >= 0 && dev < 32, "dr_bpr_step_static_f32: device index %d out of range", dev);
    if (__atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE)) return DR_OK;
    void* p = nullptr;
    DR_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    DR_REQUIRE(p != nullptr && ((uintptr_t)p % 256 == 0), "dr_bpr_step_static_f32: bad scratch address");
    __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);  // racing threads store the same address
    return DR_OK;
}
static int static_scratch(void** out) {
    int rc = static_scratch_init();
    if (rc != DR_OK) return rc;
    int dev = -1;
    DR_HIP(hipGetDevice(&dev));
    *out = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    return DR_OK;
}
}  // namespace dr

extern "C" int32_t dr_bpr_step_static_f32(float* P, float* Pslot0, float* Pslot1, int64_t num_user, float* Q,
                                          float* Qslot0, float* Qslot1, int64_t num_item, int32_t k, const int32_t* u,
                                          const int32_t* i, const int32_t* j, int64_t B, const dr_opt_params* opt,
                                          float* loss_out, void* stream) {
    DR_REQUIRE(B >= 0 && B <= DR_STATIC_MAX_B, "dr_bpr_step_static_f32: B=%lld exceeds the static maximum %d",
               (long long)B, DR_STATIC_MAX_B);
    DR_REQUIRE(k >= 1 && k <= DR_STATIC_MAX_K, "dr_bpr_step_static_f32: k=%d exceeds the static maximum %d", k,
               DR_STATIC_MAX_K);
    DR_REQUIRE(opt != nullptr && opt->reg_mode != DR_REG_DENSE,
               "dr_bpr_step_static_f32: DR_REG_DENSE needs table-sized dense gradients; use dr_bpr_step_f32");
    StepLayout L;
    if (step_layout(B, k, 2 * B, num_user, num_item, false, &L) != 0) return DR_ERR_LAUNCH;
    if (L.total > DR_STATIC_WS_BYTES)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_bpr_step_static_f32: step needs %lld bytes, static scratch holds %lld",
                (long long)L.total, (long long)DR_STATIC_WS_BYTES);
    void* ws = nullptr;
    int rc = static_scratch(&ws);
    if (rc != DR_OK) return rc;
    // 3B <= 4096 keys: the plan is ONE workgroup (plan_small_kernel), no device-wide sort; no helper streams or events
    // are used (g_no_side_streams), nothing is created, allocated or freed.  The plan kernels write every header field
    // and every partial row before it is read, so the previous call's contents need no clearing.
    g_no_side_streams = true;
    rc = bpr_step_impl(P, Pslot0, Pslot1, num_user, Q, Qslot0, Qslot1, num_item, k, u, i, j, B, opt, 0, loss_out, ws,
                       DR_STATIC_WS_BYTES, stream, 0);
    g_no_side_streams = false;
    return rc;
}

extern "C" int32_t dr_pointwise_forward_f32(const float* P, const float* Q, const float* BU, const float* BI, float mu,
                                            int64_t num_user, int64_t num_item, int32_t k, int32_t loss_kind,
                                            const int32_t* u, const int32_t* i, const float* y, int64_t B,
                                            float* g_out, float* loss_b, float* pred, float* stash, void* stream) {
    DR_REQUIRE(num_user >= 0 && num_item >= 0 && k >= 1 && B >= 0, "dr_pointwise_forward_f32: bad sizes");
    DR_REQUIRE(loss_kind == 0 || loss_kind == 1, "dr_pointwise_forward_f32: loss_kind must be 0 (bce) or 1 (sq)");
    DR_REQUIRE(loss_kind == 0 || (BU && BI), "dr_pointwise_forward_f32: squared-error kind needs bias vectors");
    if (B == 0) return DR_OK;
    DR_REQUIRE(P && Q && u && i, "dr_pointwise_forward_f32: NULL pointer");
    DR_REQUIRE(y || (!g_out && !loss_b && !stash), "dr_pointwise_forward_f32: labels needed for gradients");
    RowShape s = row_shape(k);
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
#define DR_F(V_, L_)                                                                                                 \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                           \
        if (loss_kind == 1)                                                                                          \
            DR_LAUNCH((pointwise_forward_kernel<V_, L_, true>), dim3(grid_for(B, 1, 256 / L_)), dim3(256), 0, \
                               st, P, Q, BU, BI, mu, num_user, num_item, k, u, i, y, B, g_out, loss_b, pred, stash); \
        else                                                                                                         \
            DR_LAUNCH((pointwise_forward_kernel<V_, L_, false>), dim3(grid_for(B, 1, 256 / L_)), dim3(256),  \
                               0, st, P, Q, BU, BI, mu, num_user, num_item, k, u, i, y, B, g_out, loss_b, pred,      \
                               stash);                                                                               \
        ok = true;                                                                                                   \
    }
    DR_SHAPE_DISPATCH(DR_F)
#undef DR_F
    DR_REQUIRE(ok, "dr_pointwise_forward_f32: unsupported k=%d", k);
    DR_CHECK_LAUNCH("dr_pointwise_forward_f32");
    return DR_OK;
}

// GMF (loss_kind 0) / biased MF (loss_kind 1) training step.  Bias vectors are [rows] tables (k=1).
extern "C" int32_t dr_pointwise_step_f32(float* P, float* Pslot0, float* Pslot1, int64_t num_user, float* Q,
                                         float* Qslot0, float* Qslot1, int64_t num_item, int32_t k, float* BU,
                                         float* BUslot0, float* BUslot1, float* BI, float* BIslot0, float* BIslot1,
                                         float mu, int32_t loss_kind, const int32_t* u, const int32_t* i,
                                         const float* y, int64_t B, const dr_opt_params* opt, float* loss_out,
                                         void* ws, int64_t ws_bytes, void* stream) {
    DR_REQUIRE(num_user >= 0 && num_item >= 0 && k >= 1 && B >= 0, "dr_pointwise_step_f32: bad sizes");
    int rc = check_opt(opt, Pslot0, Pslot1, "dr_pointwise_step_f32(P)");
    if (rc != DR_OK) return rc;
    rc = check_opt(opt, Qslot0, Qslot1, "dr_pointwise_step_f32(Q)");
    if (rc != DR_OK) return rc;
    if (loss_kind == 1) {
        rc = check_opt(opt, BUslot0, BUslot1, "dr_pointwise_step_f32(BU)");
        if (rc != DR_OK) return rc;
        rc = check_opt(opt, BIslot0, BIslot1, "dr_pointwise_step_f32(BI)");
        if (rc != DR_OK) return rc;
    }
    DR_REQUIRE(P && Q && ws && ((uintptr_t)ws % 256 == 0), "dr_pointwise_step_f32: NULL / misaligned pointer");
    DR_REQUIRE(B == 0 || (u && i && y), "dr_pointwise_step_f32: id / label pointer is NULL");
    StepLayout L;
    if (step_layout(B, k, 2 * B, num_user, num_item, opt->reg_mode == DR_REG_DENSE, &L) != 0) return DR_ERR_LAUNCH;
    if (ws_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_pointwise_step_f32: workspace %lld < required %lld", (long long)ws_bytes,
                (long long)L.total);
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    float* g = (float*)(base + L.g);
    float* loss_b = (float*)(base + L.loss_b);
    float* stash = (float*)(base + L.stash);
    if (B > 0) {
        rc = dr_pointwise_forward_f32(P, Q, BU, BI, mu, num_user, num_item, k, loss_kind, u, i, y, B, g, loss_b,
                                      nullptr, stash, st);
        if (rc != DR_OK) return rc;
    }
    if (B > 0 || opt->reg_mode == DR_REG_DENSE) {
        // bias vectors first (they only need g), then Q from (g, P_old), then P from the stash
        if (loss_kind == 1) {
            rc = table_update(BU, BUslot0, BUslot1, num_user, 1, u, B, g, B, nullptr, nullptr, base + L.planA,
                              L.plan_bytesA, (float*)(base + L.gdenseP), opt, st);
            if (rc != DR_OK) return rc;
            rc = table_update(BI, BIslot0, BIslot1, num_item, 1, i, B, g, B, nullptr, nullptr, base + L.planA,
                              L.plan_bytesA, (float*)(base + L.gdenseQ), opt, st);
            if (rc != DR_OK) return rc;
        }
        rc = table_update(Q, Qslot0, Qslot1, num_item, k, i, B, P, num_user, u, g, base + L.planA, L.plan_bytesA,
                          (float*)(base + L.gdenseQ), opt, st);
        if (rc != DR_OK) return rc;
        rc = table_update(P, Pslot0, Pslot1, num_user, k, u, B, stash, B, nullptr, nullptr, base + L.planB,
                          L.plan_bytesB, (float*)(base + L.gdenseP), opt, st);
        if (rc != DR_OK) return rc;
    }
    if (loss_out) return dr_sum_f32(loss_b, B, loss_out, st);
    return DR_OK;
}

// CML step (cml.py:72-81,120): hinge on squared distances, sparse Adagrad, whole-table row-norm clip.
extern "C" int32_t dr_hinge_step_f32(float* P, float* Pacc, int64_t num_user, float* Q, float* Qacc, int64_t num_item,
                                     int32_t k, const int32_t* u, const int32_t* i, const int32_t* j,
                                     const uint8_t* mask_pos, const uint8_t* mask_neg, float keep, float margin,
                                     float clip, int64_t B, const dr_opt_params* opt, float* loss_out, void* ws,
                                     int64_t ws_bytes, void* stream) {
    DR_REQUIRE(num_user >= 0 && num_item >= 0 && k >= 1 && B >= 0, "dr_hinge_step_f32: bad sizes");
    DR_REQUIRE(keep > 0.0f && keep <= 1.0f, "dr_hinge_step_f32: keep must be in (0,1]");
    int rc = check_opt(opt, Pacc, Pacc, "dr_hinge_step_f32(P)");
    if (rc != DR_OK) return rc;
    rc = check_opt(opt, Qacc, Qacc, "dr_hinge_step_f32(Q)");
    if (rc != DR_OK) return rc;
    DR_REQUIRE(opt->reg_mode == DR_REG_NONE, "dr_hinge_step_f32: CML has no L2 term (cml.py:77), use DR_REG_NONE");
    DR_REQUIRE(P && Q && ws && ((uintptr_t)ws % 256 == 0), "dr_hinge_step_f32: NULL / misaligned pointer");
    DR_REQUIRE(B == 0 || (u && i && j), "dr_hinge_step_f32: id pointer is NULL");
    StepLayout L;
    if (step_layout(B, k, 2 * B, num_user, num_item, false, &L) != 0) return DR_ERR_LAUNCH;
    if (ws_bytes < L.total)
        DR_FAIL(DR_ERR_WORKSPACE, "dr_hinge_step_f32: workspace %lld < required %lld", (long long)ws_bytes,
                (long long)L.total);
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    float* loss_b = (float*)(base + L.loss_b);
    float* stash = (float*)(base + L.stash);
    int32_t* idsP = (int32_t*)(base + L.idsA);
    int32_t* idsQ = (int32_t*)(base + L.idsB);
    float* coefP = (float*)(base + L.coef);
    if (B > 0) {
        RowShape s = row_shape(k);
        bool ok = false;
#define DR_F(V_, L_)                                                                                              \
    if (!ok && s.V == V_ && s.LPR == L_) {                                                                        \
        DR_LAUNCH((hinge_forward_kernel<V_, L_>), dim3(grid_for(B, 1, 256 / L_)), dim3(256), 0, st, P, Q, \
                           num_user, num_item, k, u, i, j, mask_pos, mask_neg, keep, margin, B, loss_b, stash);   \
        ok = true;                                                                                                \
    }
        DR_SHAPE_DISPATCH(DR_F)
#undef DR_F
        DR_REQUIRE(ok, "dr_hinge_step_f32: unsupported k=%d", k);
        DR_CHECK_LAUNCH("hinge_forward_kernel");
        DR_LAUNCH(hinge_lists_kernel, dim3(grid_for(B, 256)), dim3(256), 0, st, u, i, j, B, idsP, idsQ, coefP);
        DR_CHECK_LAUNCH("hinge_lists_kernel");
        // both updates read only the stash, so their order is free
        rc = table_update(Q, Qacc, nullptr, num_item, k, idsQ, 2 * B, stash, 2 * B, nullptr, nullptr, base + L.planA,
                          L.plan_bytesA, nullptr, opt, st);
        if (rc != DR_OK) return rc;
        rc = table_update(P, Pacc, nullptr, num_user, k, idsP, 2 * B, stash, 2 * B, nullptr, coefP, base + L.planB,
                          L.plan_bytesB, nullptr, opt, st);
        if (rc != DR_OK) return rc;
    }
    if (clip > 0.0f) {
        rc = dr_rownorm_clip_f32(P, num_user, k, clip, st);
        if (rc != DR_OK) return rc;
        rc = dr_rownorm_clip_f32(Q, num_item, k, clip, st);
        if (rc != DR_OK) return rc;
    }
    if (loss_out) return dr_sum_f32(loss_b, B, loss_out, st);
    return DR_OK;
}

extern "C" int32_t dr_pair_dot_f32(const float* A, int64_t rowsA, const float* B, int64_t rowsB, int32_t k,
                                   const int32_t* a_idx, const int32_t* b_idx, int64_t n, float* out, void* stream) {
    DR_REQUIRE(rowsA >= 0 && rowsB >= 0 && k >= 1 && n >= 0, "dr_pair_dot_f32: bad sizes");
    if (n == 0) return DR_OK;
    DR_REQUIRE(A && B && a_idx && b_idx && out, "dr_pair_dot_f32: NULL pointer");
    DR_LAUNCH(pair_dot_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, A, rowsA, B, rowsB, k,
                       a_idx, b_idx, n, out);
    DR_CHECK_LAUNCH("dr_pair_dot_f32");
    return DR_OK;
}
