// Segment-update argument block and the row-application code shared by the segment kernels (segment.hip) and the
// fused BPR forward kernel (pairwise.hip), which applies single-contribution rows itself.
#pragma once
#include "dr_common.h"

namespace dr {

constexpr int KIND_ACC = -1;  // plain accumulate (dr_scatter_add_sorted_f32)

struct SegSel {
    uint32_t id_lo;
    int part;
    int pos_lo;
};

struct SegArgs {
    float* table;
    float* slot0;
    float* slot1;
    int64_t rows;
    int k;
    const float* src;
    int64_t src_rows;
    const float* src2;     // source rows src_rows, src_rows + 1, ... (NULL: none): a second source table
    int skip_single;       // 1: segments with exactly one contribution are somebody else's (fused BPR forward)
    const int32_t* msrc;   // per SORTED contribution: source row (or -1 = skip)
    const float* coef;     // per contribution in ORIGINAL order (read through spos), NULL = 1
    const int32_t* spos;   // sorted position list
    uint32_t id_lo;        // destination row = segment key - id_lo (combined two-table plans)
    int part;              // 0: all segments, 1: those below the plan's split, 2: those at / above it
    int pos_lo;            // coef is indexed by (position - pos_lo)
    const int32_t* uniq;
    const int32_t* seg_start;
    const int32_t* hdr;       // [0] nseg, [1] n long segments
    const int32_t* long_seg;
    const int32_t* long_cb;
    const int32_t* part_owner;
    float* partials;
    int kind;
    OptP opt;
};

// source row `si` of a contribution: rows at / beyond src_rows live in the second source table
__device__ __forceinline__ const float* seg_src_row(const SegArgs& a, int si) {
    return (a.src2 != nullptr && (int64_t)si >= a.src_rows) ? a.src2 + ((int64_t)si - a.src_rows) * a.k
                                                             : a.src + (int64_t)si * a.k;
}

// gradient total -> destination row (accumulate, or L2 + optimizer application)
constexpr int KIND_ANY = -2;  // optimizer kind read from the arguments at run time
template <int V, int KIND = KIND_ANY>
__device__ __forceinline__ void apply_row(const SegArgs& a, uint32_t r, int c, float (&tot)[V]) {
    const int64_t off = (int64_t)r * a.k + c * V;
    float w[V], s0[V], s1[V];
    if (KIND == KIND_ACC) {  // plain lookup backward (dr_scatter_add_sorted_f32)
        vload<V>(w, a.table + off);
#pragma unroll
        for (int x = 0; x < V; ++x) w[x] = w[x] + tot[x];
        vstore<V>(a.table + off, w);
        return;
    }
    if (KIND == DR_OPT_ADAM) {  // the hot instantiation: no switch, all three row loads issued together
        vload<V>(w, a.table + off);
        vload<V>(s0, a.slot0 + off);
        vload<V>(s1, a.slot1 + off);
        if (a.opt.reg_mode == DR_REG_TOUCHED) {
#pragma unroll
            for (int x = 0; x < V; ++x) tot[x] = tot[x] + a.opt.reg * w[x];
        }
#pragma unroll
        for (int x = 0; x < V; ++x) opt_apply<DR_OPT_ADAM>(w[x], s0[x], s1[x], tot[x], a.opt);
        vstore<V>(a.slot0 + off, s0);
        vstore<V>(a.slot1 + off, s1);
        vstore<V>(a.table + off, w);
        return;
    }
    vload<V>(w, a.table + off);
    if (a.kind == KIND_ACC) {
#pragma unroll
        for (int x = 0; x < V; ++x) w[x] = w[x] + tot[x];
        vstore<V>(a.table + off, w);
        return;
    }
    if (a.opt.reg_mode == DR_REG_TOUCHED) {
#pragma unroll
        for (int x = 0; x < V; ++x) tot[x] = tot[x] + a.opt.reg * w[x];
    }
    switch (a.kind) {
        case DR_OPT_SGD:
#pragma unroll
            for (int x = 0; x < V; ++x) opt_apply<DR_OPT_SGD>(w[x], s0[x], s1[x], tot[x], a.opt);
            break;
        case DR_OPT_ADAM:
            vload<V>(s0, a.slot0 + off);
            vload<V>(s1, a.slot1 + off);
#pragma unroll
            for (int x = 0; x < V; ++x) opt_apply<DR_OPT_ADAM>(w[x], s0[x], s1[x], tot[x], a.opt);
            vstore<V>(a.slot0 + off, s0);
            vstore<V>(a.slot1 + off, s1);
            break;
        case DR_OPT_ADAGRAD:
            vload<V>(s0, a.slot0 + off);
#pragma unroll
            for (int x = 0; x < V; ++x) opt_apply<DR_OPT_ADAGRAD>(w[x], s0[x], s1[x], tot[x], a.opt);
            vstore<V>(a.slot0 + off, s0);
            break;
        default:
            vload<V>(s0, a.slot0 + off);
            vload<V>(s1, a.slot1 + off);
#pragma unroll
            for (int x = 0; x < V; ++x) opt_apply<DR_OPT_RMSPROP>(w[x], s0[x], s1[x], tot[x], a.opt);
            vstore<V>(a.slot0 + off, s0);
            vstore<V>(a.slot1 + off, s1);
            break;
    }
    vstore<V>(a.table + off, w);
}

// apply_row with the destination row's column already in registers: w always, s0 / s1 iff the optimizer kind uses
// them (kind_slots).  Same operations in the same order as apply_row.
__device__ __forceinline__ int kind_slots(int kind) {
    return kind == DR_OPT_SGD ? 0 : ((kind == DR_OPT_ADAM || kind == DR_OPT_RMSPROP) ? 2 : 1);
}
template <int V, int KIND = KIND_ANY, bool NT = false>
__device__ __forceinline__ void apply_row_loaded(const SegArgs& a, uint32_t r, int c, float (&tot)[V], float (&w)[V],
                                                 float (&s0)[V], float (&s1)[V]) {
    const int64_t off = (int64_t)r * a.k + c * V;
    if (a.opt.reg_mode == DR_REG_TOUCHED) {
#pragma unroll
        for (int x = 0; x < V; ++x) tot[x] = tot[x] + a.opt.reg * w[x];
    }
    if (KIND == DR_OPT_ADAM) {  // the hot instantiation: no switch
#pragma unroll
        for (int x = 0; x < V; ++x) opt_apply<DR_OPT_ADAM>(w[x], s0[x], s1[x], tot[x], a.opt);
        if (NT) {
            vstore_nt<V>(a.slot0 + off, s0);
            vstore_nt<V>(a.slot1 + off, s1);
            vstore_nt<V>(a.table + off, w);
        } else {
            vstore<V>(a.slot0 + off, s0);
            vstore<V>(a.slot1 + off, s1);
            vstore<V>(a.table + off, w);
        }
        return;
    }
    switch (a.kind) {
        case DR_OPT_SGD:
#pragma unroll
            for (int x = 0; x < V; ++x) opt_apply<DR_OPT_SGD>(w[x], s0[x], s1[x], tot[x], a.opt);
            break;
        case DR_OPT_ADAM:
#pragma unroll
            for (int x = 0; x < V; ++x) opt_apply<DR_OPT_ADAM>(w[x], s0[x], s1[x], tot[x], a.opt);
            vstore<V>(a.slot0 + off, s0);
            vstore<V>(a.slot1 + off, s1);
            break;
        case DR_OPT_ADAGRAD:
#pragma unroll
            for (int x = 0; x < V; ++x) opt_apply<DR_OPT_ADAGRAD>(w[x], s0[x], s1[x], tot[x], a.opt);
            vstore<V>(a.slot0 + off, s0);
            break;
        default:
#pragma unroll
            for (int x = 0; x < V; ++x) opt_apply<DR_OPT_RMSPROP>(w[x], s0[x], s1[x], tot[x], a.opt);
            vstore<V>(a.slot0 + off, s0);
            vstore<V>(a.slot1 + off, s1);
            break;
    }
    vstore<V>(a.table + off, w);
}

}  // namespace dr
