/* oracle.c -- plain-C restatement of the pieces of the DeepRec hot path whose exact fp32
 * operation order matters (k-ordered fmaf chains) or that serve as the timed CPU baseline.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg; never by the deeprec_amd package.
 *
 * Reference lines restated (relative to /root/reference):
 *   or_pair_dot / or_score_matrix kind 0 : bprmf.py:91-92,168-173 ; SeqRecMetrics.py:213
 *   or_score_matrix kind 1               : cml.py:72,151-153 (-sum (p-q)^2, expanded form)
 *   or_score_matrix kind 2               : SeqRecMetrics.py:122-123
 *   or_bpr_step_*                        : bprmf.py:114-123 (forward, loss, gradient, update)
 * Parity unpinned for the TensorFlow arithmetic (TensorFlow is not available here).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

float or_counter_normal(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    int s = (int)(z & 0xFFFF) + (int)((z >> 16) & 0xFFFF) + (int)((z >> 32) & 0xFFFF) + (int)((z >> 48) & 0xFFFF);
    return (float)(s - 131070) * (1.0f / 37837.227f);
}

void or_init_normal(float* w, int64_t count, uint64_t seed, float stddev) {
    for (int64_t i = 0; i < count; ++i) w[i] = stddev * or_counter_normal(seed, (uint64_t)i);
}

/* rows of a counter-initialised table without building the table: w[r, c] = stddev * normal(seed, rows[r] * k + c)
 * (what dr_init_normal_f32 writes at element rows[r] * k + c) -- full-size C4 tests check sampled rows this way */
void or_init_rows(float* w, const int64_t* rows, int64_t n, int k, uint64_t seed, float stddev) {
    for (int64_t r = 0; r < n; ++r)
        for (int c = 0; c < k; ++c) w[r * k + c] = stddev * or_counter_normal(seed, (uint64_t)rows[r] * (uint64_t)k + (uint64_t)c);
}

static float chain_dot(const float* a, const float* b, int k) {
    float acc = 0.0f;
    for (int c = 0; c < k; ++c) acc = fmaf(a[c], b[c], acc);
    return acc;
}

void or_pair_dot(const float* A, const float* B, int k, const int32_t* ai, const int32_t* bi, int64_t n, float* out) {
    for (int64_t e = 0; e < n; ++e) out[e] = chain_dot(A + (int64_t)ai[e] * k, B + (int64_t)bi[e] * k, k);
}

/* weighted negative squared norm: -sum_c (w*x_c)*x_c as one fmaf chain over [X ; X2] */
static float neg_sqnorm(const float* x, const float* x2, int k, float w1, float w2) {
    float acc = 0.0f;
    for (int c = 0; c < k; ++c) acc = fmaf(w1 * x[c], x[c], acc);
    if (x2)
        for (int c = 0; c < k; ++c) acc = fmaf(w2 * x2[c], x2[c], acc);
    return -acc;
}

/* kind 0: dot (+bias); 1: -||u-i||^2 as 2<u,i> - |u|^2 - |i|^2 ; 2: -a|u-i|^2 - (1-a)|u2-i2|^2 likewise */
void or_score_matrix(const float* U, const float* U2, int64_t Nu, const float* I, const float* I2, const float* bias,
                     int64_t Ni, int k, int kind, float alpha, float* out) {
    float* uadd = (float*)malloc(sizeof(float) * (size_t)(Nu > 0 ? Nu : 1));
    float* iadd = (float*)malloc(sizeof(float) * (size_t)(Ni > 0 ? Ni : 1));
    const float w1 = (kind == 2) ? alpha : 1.0f, w2 = 1.0f - alpha;
    if (kind != 0) {
        for (int64_t u = 0; u < Nu; ++u) uadd[u] = neg_sqnorm(U + u * k, kind == 2 ? U2 + u * k : NULL, k, w1, w2);
        for (int64_t i = 0; i < Ni; ++i) {
            iadd[i] = neg_sqnorm(I + i * k, kind == 2 ? I2 + i * k : NULL, k, w1, w2);
            if (bias) iadd[i] = iadd[i] + bias[i];
        }
    }
    for (int64_t u = 0; u < Nu; ++u)
        for (int64_t i = 0; i < Ni; ++i) {
            float acc = 0.0f;
            if (kind == 2) {
                for (int c = 0; c < k; ++c) acc = fmaf(w1 * U[u * k + c], I[i * k + c], acc);
                for (int c = 0; c < k; ++c) acc = fmaf(w2 * U2[u * k + c], I2[i * k + c], acc);
            } else {
                acc = chain_dot(U + u * k, I + i * k, k);
            }
            float s = acc;
            if (kind == 0) {
                if (bias) s = s + bias[i];
            } else {
                s = 2.0f * s;
                s = s + uadd[u];
                s = s + iadd[i];
            }
            out[u * Ni + i] = s;
        }
    free(uadd);
    free(iadd);
}

/* ---- CPU baseline: one BPR-MF step, touched-rows L2, SGD or Adam, scalar C -------------------
 * Same arithmetic as oracle/ref_ops.py:pairwise_step (blocked position-order duplicate sums are
 * replaced by plain position-order sums per row: this function is the timed baseline, the numpy
 * oracle is the checker). */
typedef struct { int32_t id; int32_t pos; } idpos;
static int cmp_idpos(const void* a, const void* b) {
    const idpos* x = (const idpos*)a; const idpos* y = (const idpos*)b;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

static void apply_row(float* w, float* m, float* v, const float* g, int k, int adam, float lr, float alpha, float omb1,
                      float omb2, float eps, float reg) {
    for (int c = 0; c < k; ++c) {
        float gg = g[c] + reg * w[c];
        if (adam) {
            m[c] = m[c] + (gg - m[c]) * omb1;
            v[c] = v[c] + (gg * gg - v[c]) * omb2;
            w[c] = w[c] - (m[c] * alpha) / (sqrtf(v[c]) + eps);
        } else {
            w[c] = w[c] - lr * gg;
        }
    }
}

double or_bpr_step(float* P, float* Pm, float* Pv, float* Q, float* Qm, float* Qv, int k, const int32_t* u,
                   const int32_t* i, const int32_t* j, int64_t B, int adam, float lr, float alpha, float beta1,
                   float beta2, float eps, float reg) {
    const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
    float* g = (float*)malloc(sizeof(float) * (size_t)B);
    float* stash = (float*)malloc(sizeof(float) * (size_t)B * k);
    float* acc = (float*)malloc(sizeof(float) * (size_t)k);
    idpos* lst = (idpos*)malloc(sizeof(idpos) * (size_t)(2 * B));
    double loss = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        const float* pu = P + (int64_t)u[b] * k; const float* qi = Q + (int64_t)i[b] * k; const float* qj = Q + (int64_t)j[b] * k;
        float xi = 0.0f, xj = 0.0f;
        for (int c = 0; c < k; ++c) { xi = xi + pu[c] * qi[c]; xj = xj + pu[c] * qj[c]; }
        const float s = 1.0f / (1.0f + expf(-(xi - xj)));
        loss += -logf(s);
        g[b] = -(1.0f - s);
        for (int c = 0; c < k; ++c) stash[b * k + c] = g[b] * (qi[c] - qj[c]);
    }
    /* Q from (g, P_old) */
    for (int64_t b = 0; b < B; ++b) { lst[b].id = i[b]; lst[b].pos = (int32_t)b; lst[B + b].id = j[b]; lst[B + b].pos = (int32_t)(B + b); }
    qsort(lst, (size_t)(2 * B), sizeof(idpos), cmp_idpos);
    for (int64_t s0 = 0; s0 < 2 * B;) {
        int64_t s1 = s0;
        memset(acc, 0, sizeof(float) * (size_t)k);
        while (s1 < 2 * B && lst[s1].id == lst[s0].id) {
            const int32_t p = lst[s1].pos; const int64_t b = p < B ? p : p - B; const float cf = p < B ? g[b] : -g[b];
            const float* pu = P + (int64_t)u[b] * k;
            for (int c = 0; c < k; ++c) acc[c] = acc[c] + cf * pu[c];
            ++s1;
        }
        const int64_t r = lst[s0].id;
        apply_row(Q + r * k, Qm ? Qm + r * k : NULL, Qv ? Qv + r * k : NULL, acc, k, adam, lr, alpha, omb1, omb2, eps, reg);
        s0 = s1;
    }
    /* P from the stash */
    for (int64_t b = 0; b < B; ++b) { lst[b].id = u[b]; lst[b].pos = (int32_t)b; }
    qsort(lst, (size_t)B, sizeof(idpos), cmp_idpos);
    for (int64_t s0 = 0; s0 < B;) {
        int64_t s1 = s0;
        memset(acc, 0, sizeof(float) * (size_t)k);
        while (s1 < B && lst[s1].id == lst[s0].id) {
            const float* d = stash + (int64_t)lst[s1].pos * k;
            for (int c = 0; c < k; ++c) acc[c] = acc[c] + d[c];
            ++s1;
        }
        const int64_t r = lst[s0].id;
        apply_row(P + r * k, Pm ? Pm + r * k : NULL, Pv ? Pv + r * k : NULL, acc, k, adam, lr, alpha, omb1, omb2, eps, reg);
        s0 = s1;
    }
    free(g); free(stash); free(acc); free(lst);
    return loss;
}

void or_gather(const float* table, int64_t rows, int k, const int32_t* ids, int64_t n, float* out) {
    for (int64_t e = 0; e < n; ++e) {
        if (ids[e] < 0 || ids[e] >= rows) memset(out + e * k, 0, sizeof(float) * (size_t)k);
        else memcpy(out + e * k, table + (int64_t)ids[e] * k, sizeof(float) * (size_t)k);
    }
}

/* ---- negative sampler: restates bprmf.py:126-133,175-181 (uniform draw from all items minus the user's
 * train items) with the counter-based generator in place of numpy's stream; checker for
 * dr_sample_negatives_i32.  Deliberately the plain formulation: build the candidate list, index it. ---- */
static uint32_t or_counter_u32(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

/* AttRec.py:448-466 (_get_neg_items_sbpr): pool = all items first_item .. num_item-1 minus the sample's own targets,
 * built as a list like the reference does; draw = pool[floor(r * |pool| / 2^32)] */
void or_sample_negatives_excl(const int32_t* targets, int64_t n, int32_t T, int32_t n_neg, int64_t num_item,
                              int64_t first_item, uint64_t seed, uint64_t offset, int32_t* out) {
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (size_t)(num_item > 0 ? num_item : 1));
    for (int64_t e = 0; e < n; ++e) {
        int64_t cnt = 0;
        for (int64_t it = first_item; it < num_item; ++it) {
            int is_t = 0;
            for (int q = 0; q < T; ++q) is_t |= (targets[e * T + q] == it);
            if (!is_t) cand[cnt++] = (int32_t)it;
        }
        for (int d = 0; d < n_neg; ++d) {
            const int64_t x = e * n_neg + d;
            if (cnt <= 0) {
                out[x] = -1;
                continue;
            }
            const uint64_t r = or_counter_u32(seed, offset + (uint64_t)x);
            out[x] = cand[(r * (uint64_t)cnt) >> 32];
        }
    }
    free(cand);
}

void or_sample_negatives(const int32_t* train_ptr, const int32_t* train_idx, const int32_t* users, int64_t n,
                         int32_t n_neg, int64_t num_user, int64_t num_item, int64_t first_item, uint64_t seed,
                         uint64_t offset, int32_t* out) {
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (size_t)(num_item > 0 ? num_item : 1));
    for (int64_t e = 0; e < n; ++e) {
        const int64_t u = users[e];
        int64_t cnt = 0;
        if (u >= 0 && u < num_user) {
            int p = train_ptr[u];
            const int pe = train_ptr[u + 1];
            for (int64_t it = first_item; it < num_item; ++it) {      /* neg_items[u], ascending */
                while (p < pe && train_idx[p] < it) ++p;
                if (p < pe && train_idx[p] == it) continue;
                cand[cnt++] = (int32_t)it;
            }
        }
        for (int d = 0; d < n_neg; ++d) {
            const int64_t x = e * n_neg + d;
            if (cnt <= 0) {
                out[x] = -1;
                continue;
            }
            const uint64_t r = or_counter_u32(seed, offset + (uint64_t)x);
            out[x] = cand[(r * (uint64_t)cnt) >> 32];
        }
    }
    free(cand);
}
