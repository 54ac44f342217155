/* oracle/dqn_oracle_omp.c -- all-core (OpenMP) form of the oracle's whole-update driver and actor step.
 * TEST INFRASTRUCTURE ONLY (see dqn_oracle.h). PARITY UNPINNED (see dqn_oracle.h).
 *
 * Used by bench.py's cpu_baseline leg ("C restatement with OpenMP, thread count =
 * os.cpu_count()") and pinned bit-for-bit to the scalar driver by tests/test_oracle.py: every per-row
 * quantity is computed by exactly the scalar code on a slice of the batch, and every weight-gradient element is
 * the same i-ascending fmaf chain as in orc_grads -- only WHICH thread computes a row / an element differs.
 *
 * Restates (like the scalar driver): General/QLearning/q_agent.py:146-169 (_step), :137-141 (_policy), :176-183
 * (env loop); q_learning_functions.py:23-25, :35-36, :52-61; LunarLander/dddqn.py:24-31.
 */
#include "dqn_oracle.h"
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

int32_t orc_omp_threads(void) { return (int32_t)omp_get_max_threads(); }
void orc_omp_set_threads(int32_t n) { if (n > 0) omp_set_num_threads(n); }

typedef struct { int64_t w1, b1, w2, b2, wv, bv, wa, ba; } offs_t;
static offs_t offsets(orc_dims m) {
    offs_t o; int64_t p = 0;
    o.w1 = p; p += (int64_t)m.D * m.H1;  o.b1 = p; p += m.H1;
    o.w2 = p; p += (int64_t)m.H1 * m.H2; o.b2 = p; p += m.H2;
    o.wv = p; p += m.H2;                 o.bv = p; p += 1;
    o.wa = p; p += (int64_t)m.H2 * m.A;  o.ba = p; p += m.A;
    return o;
}

static inline float huber(float e) {
    const float ae = fabsf(e);
    const float qd = ae < 1.0f ? ae : 1.0f;
    return 0.5f * (qd * qd) + (ae - qd);
}

/* slice [lo, hi) of B rows for thread t of nt */
static inline void slice(int32_t B, int t, int nt, int32_t *lo, int32_t *hi) {
    const int32_t per = (B + nt - 1) / nt;
    *lo = t * per; if (*lo > B) *lo = B;
    *hi = *lo + per; if (*hi > B) *hi = B;
}

/* orc_per_sample with the descents spread over threads (each sample is independent; max is order-free) */
static void per_sample_omp(const orc_per *t, int64_t size, int32_t B, float beta, uint64_t seed, uint64_t ctr,
                           int32_t *idx, float *isw) {
    const float total = t->tree[1];
    const float seg = total / (float)B;
    float wmax = 0.0f;
#pragma omp parallel for schedule(static) reduction(max : wmax)
    for (int32_t k = 0; k < B; ++k) {
        uint32_t c[4] = { (uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)k, ORC_STREAM_PER };
        uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) }, o[4];
        orc_philox4x32_10(c, key, o);
        float u = ((float)k + orc_u01(o[0])) * seg;
        int64_t node = 1;
        while (node < t->N) {
            const float l = t->tree[2 * node];
            if (u < l) node = 2 * node;
            else { u = u - l; node = 2 * node + 1; }
        }
        int64_t leaf = node - t->N;
        if (leaf >= size) leaf = size - 1;
        idx[k] = (int32_t)leaf;
        const float w = orc_pow_det(((float)size * t->tree[t->N + leaf]) / total, -beta);
        isw[k] = w;
        if (w > wmax) wmax = w;
    }
#pragma omp parallel for schedule(static)
    for (int32_t k = 0; k < B; ++k) isw[k] = isw[k] / wmax;
}

/* gradient of orc_loss, same arithmetic as orc_grads: phase 1 per row (threads own row slices), phase 2 per
 * weight element (threads own rows of each weight matrix), every sum in ascending i */
static void grads_omp(orc_dims m, const float *P, const float *s, const float *targets, const float *isw,
                      int32_t B, float *grad, float *loss_out) {
    const int D = m.D, H1 = m.H1, H2 = m.H2, A = m.A, A1 = 1 + m.A;
    const offs_t o = offsets(m);
    float *pred = (float *)malloc(sizeof(float) * (size_t)B * A);
    float *h1 = (float *)malloc(sizeof(float) * (size_t)B * H1);
    float *h2 = (float *)malloc(sizeof(float) * (size_t)B * H2);
    float *dz3 = (float *)malloc(sizeof(float) * (size_t)B * A1);     /* [dv, dadv_0..] per row */
    float *dz2 = (float *)malloc(sizeof(float) * (size_t)B * H2);
    float *dz1 = (float *)malloc(sizeof(float) * (size_t)B * H1);
    float *rowl = (float *)malloc(sizeof(float) * (size_t)B);
    const float invB = 1.0f / (float)B;
#pragma omp parallel
    {
        int32_t lo, hi;
        slice(B, omp_get_thread_num(), omp_get_num_threads(), &lo, &hi);
        if (hi > lo)
            orc_forward(m, P, s + (int64_t)lo * D, hi - lo, pred + (int64_t)lo * A, h1 + (int64_t)lo * H1, h2 + (int64_t)lo * H2);
        for (int32_t i = lo; i < hi; ++i) {
            float g[64];
            float row = 0.0f, gsum = 0.0f;
            const float w = isw ? isw[i] : 1.0f;
            for (int k = 0; k < A; ++k) {
                const float e = pred[(int64_t)i * A + k] - targets[(int64_t)i * A + k];
                row = row + huber(e);
                const float c = e > 1.0f ? 1.0f : (e < -1.0f ? -1.0f : e);
                g[k] = (w * c) * invB;
                gsum = gsum + g[k];
            }
            rowl[i] = isw ? w * row : row;
            float *z3 = dz3 + (int64_t)i * A1;
            const float dv = gsum, gmean = gsum / (float)A;
            z3[0] = dv;
            for (int k = 0; k < A; ++k) z3[1 + k] = g[k] - gmean;
            const float *h2i = h2 + (int64_t)i * H2, *h1i = h1 + (int64_t)i * H1;
            float *z2 = dz2 + (int64_t)i * H2, *z1 = dz1 + (int64_t)i * H1;
            for (int j = 0; j < H2; ++j) {
                float t = P[o.wv + j] * dv;
                for (int k = 0; k < A; ++k) t = fmaf(P[o.wa + (int64_t)j * A + k], z3[1 + k], t);
                z2[j] = h2i[j] > 0.0f ? t : 0.0f;
            }
            for (int k = 0; k < H1; ++k) {
                const float *wr = P + o.w2 + (int64_t)k * H2;
                float t = 0.0f;
                for (int j = 0; j < H2; ++j) t = fmaf(wr[j], z2[j], t);
                z1[k] = h1i[k] > 0.0f ? t : 0.0f;
            }
        }
#pragma omp barrier
        /* phase 2: grad[w + k*N + j] = fmaf chain over i of in[i][k] * dz[i][j], from 0 (as orc_grads from a zeroed buffer) */
#pragma omp for schedule(static) nowait
        for (int k = 0; k < H2; ++k) {                                  /* heads: wv[k], wa[k][:] */
            float acc[65];
            for (int c = 0; c < A1; ++c) acc[c] = 0.0f;
            for (int32_t i = 0; i < B; ++i) {
                const float hk = h2[(int64_t)i * H2 + k];
                const float *z3 = dz3 + (int64_t)i * A1;
                for (int c = 0; c < A1; ++c) acc[c] = fmaf(hk, z3[c], acc[c]);
            }
            grad[o.wv + k] = acc[0];
            for (int c = 0; c < A; ++c) grad[o.wa + (int64_t)k * A + c] = acc[1 + c];
        }
#pragma omp for schedule(static) nowait
        for (int k = 0; k < H1; ++k) {                                  /* layer 2: w2[k][:] */
            float *gw = grad + o.w2 + (int64_t)k * H2;
            for (int j = 0; j < H2; ++j) gw[j] = 0.0f;
            for (int32_t i = 0; i < B; ++i) {
                const float hk = h1[(int64_t)i * H1 + k];
                const float *z2 = dz2 + (int64_t)i * H2;
                for (int j = 0; j < H2; ++j) gw[j] = fmaf(hk, z2[j], gw[j]);
            }
        }
#pragma omp for schedule(static) nowait
        for (int k = 0; k < D; ++k) {                                   /* layer 1: w1[k][:] */
            float *gw = grad + o.w1 + (int64_t)k * H1;
            for (int j = 0; j < H1; ++j) gw[j] = 0.0f;
            for (int32_t i = 0; i < B; ++i) {
                const float xk = s[(int64_t)i * D + k];
                const float *z1 = dz1 + (int64_t)i * H1;
                for (int j = 0; j < H1; ++j) gw[j] = fmaf(xk, z1[j], gw[j]);
            }
        }
        /* biases: plain i-ascending sums (grad[b] = grad[b] + dz[i], from 0) */
#pragma omp for schedule(static) nowait
        for (int j = 0; j < H2; ++j) {
            float t = 0.0f;
            for (int32_t i = 0; i < B; ++i) t = t + dz2[(int64_t)i * H2 + j];
            grad[o.b2 + j] = t;
        }
#pragma omp for schedule(static) nowait
        for (int j = 0; j < H1; ++j) {
            float t = 0.0f;
            for (int32_t i = 0; i < B; ++i) t = t + dz1[(int64_t)i * H1 + j];
            grad[o.b1 + j] = t;
        }
#pragma omp for schedule(static)
        for (int c = 0; c < A1; ++c) {
            float t = 0.0f;
            for (int32_t i = 0; i < B; ++i) t = t + dz3[(int64_t)i * A1 + c];
            if (c == 0) grad[o.bv] = t; else grad[o.ba + c - 1] = t;
        }
    }
    float acc = 0.0f;
    for (int32_t i = 0; i < B; ++i) acc = acc + rowl[i];
    if (loss_out) *loss_out = acc / (float)B;
    free(pred); free(h1); free(h2); free(dz3); free(dz2); free(dz1); free(rowl);
}

/* orc_adam_step, elements spread over threads (each element is independent) */
static void adam_omp(orc_opt o, float *P, const float *g, float *mu, float *nu, int32_t *count, double *b1pow,
                     double *b2pow, int64_t n) {
    *count += 1;
    *b1pow *= (double)o.b1;
    *b2pow *= (double)o.b2;
    const float c1 = (float)(1.0 - *b1pow), c2 = (float)(1.0 - *b2pow);
    const float omb1 = 1.0f - o.b1, omb2 = 1.0f - o.b2, neglr = -o.lr;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float gi = g[i] * 1.0f;
        const float m = (o.b1 * mu[i]) + (omb1 * gi);
        const float v = (o.b2 * nu[i]) + (omb2 * (gi * gi));
        mu[i] = m; nu[i] = v;
        const float mhat = m / c1, vhat = v / c2;
        float u = mhat / (sqrtf(vhat) + o.eps);
        if (o.adamw) u = u + (o.wd * P[i]);
        P[i] = P[i] + (neglr * u);
    }
}

/* orc_learner_update on all cores; same result, bit for bit (one-step returns only) */
float orc_learner_update_omp(orc_learner *l, int32_t B) {
    float loss = 0.0f;
    const int D = l->m.D, A = l->m.A;
    if (l->per) per_sample_omp(l->per, l->rb->size, B, l->beta, l->seed, l->ctr, l->idx, l->isw);
    else        orc_uniform_indices(l->rb->size, B, l->seed, l->ctr, l->idx);
    l->ctr += 1;
#pragma omp parallel
    {
        int32_t lo, hi;
        slice(B, omp_get_thread_num(), omp_get_num_threads(), &lo, &hi);
        if (hi > lo) {
            orc_replay_gather(l->rb, l->idx + lo, hi - lo, l->s + (int64_t)lo * D, l->a + lo, l->r + lo,
                              l->s2 + (int64_t)lo * D, l->d + lo);
            for (int32_t i = lo; i < hi; ++i) l->df[i] = l->d[i] ? 1.0f : 0.0f;
            orc_q_targets(l->m, l->P, l->Pt, l->s + (int64_t)lo * D, l->a + lo, l->r + lo, l->s2 + (int64_t)lo * D,
                          l->df + lo, l->n_step > 1 ? l->gamma_n : l->gamma, hi - lo, l->targets + (int64_t)lo * A,
                          NULL, NULL, NULL, NULL, l->delta + lo);
        }
    }
    grads_omp(l->m, l->P, l->s, l->targets, l->per ? l->isw : NULL, B, l->grad, &loss);
    adam_omp(l->opt, l->P, l->grad, l->mu, l->nu, &l->count, &l->b1pow, &l->b2pow, orc_param_count(l->m));
    if (l->per) {
        for (int32_t i = 0; i < B; ++i) l->delta[i] = fabsf(l->delta[i]);
        orc_per_update(l->per, l->idx, l->delta, B);
    }
    return loss;
}

/* orc_learner_actor_step on all cores (one-step returns): forward / policy / synthetic transition per env in parallel,
 * ring + tree insert in order */
void orc_learner_actor_step_omp(orc_learner *l, float *obs, int32_t n, float epsilon, float p_done, uint64_t *env_ctr) {
    const int32_t D = l->m.D, A = l->m.A;
    int32_t *a = (int32_t *)malloc(sizeof(int32_t) * n), *slots = (int32_t *)malloc(sizeof(int32_t) * n);
    float *next = (float *)malloc(sizeof(float) * (size_t)n * D), *r = (float *)malloc(sizeof(float) * n);
    float *q = (float *)malloc(sizeof(float) * (size_t)n * A);
    uint8_t *d = (uint8_t *)malloc(n);
    const uint64_t ctr = *env_ctr, seed = l->seed;
#pragma omp parallel
    {
        int32_t lo, hi;
        slice(n, omp_get_thread_num(), omp_get_num_threads(), &lo, &hi);
        if (hi > lo) orc_forward(l->m, l->P, obs + (int64_t)lo * D, hi - lo, q + (int64_t)lo * A, NULL, NULL);
        const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
        for (int32_t i = lo; i < hi; ++i) {
            /* q_agent.py:137-141 (as orc_act) */
            uint32_t c[4] = { (uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)i, ORC_STREAM_POLICY }, out[4];
            orc_philox4x32_10(c, key, out);
            if (epsilon < orc_u01(out[0])) {
                int best = 0;
                for (int k = 1; k < A; ++k) if (q[(int64_t)i * A + k] > q[(int64_t)i * A + best]) best = k;
                a[i] = best;
            } else {
                a[i] = (int32_t)(((uint64_t)out[1] * (uint64_t)A) >> 32);
            }
            /* synthetic transition (as orc_synth_env) */
            const uint32_t base = (uint32_t)i * (uint32_t)(D + 1);
            uint32_t ce[4] = { (uint32_t)ctr, (uint32_t)(ctr >> 32), 0, ORC_STREAM_ENV }, o[4];
            for (int32_t e = 0; e < D; ++e) {
                ce[2] = base + (uint32_t)e;
                orc_philox4x32_10(ce, key, o);
                next[(int64_t)i * D + e] = (((orc_u01(o[0]) + orc_u01(o[1])) + (orc_u01(o[2]) + orc_u01(o[3]))) - 2.0f) * 1.73205078f;
            }
            ce[2] = base + (uint32_t)D;
            orc_philox4x32_10(ce, key, o);
            const int done = orc_u01(o[0]) < p_done;
            float rew = (((orc_u01(o[1]) + orc_u01(o[2])) + (orc_u01(o[3]) + orc_u01(o[0]))) - 2.0f) * 1.73205078f;
            if (done) rew = (o[1] & 1u) ? 100.0f : -100.0f;
            r[i] = rew;
            d[i] = done ? 1 : 0;
        }
    }
    orc_replay_add(l->rb, obs, a, r, next, d, n, slots);
    if (l->per) orc_per_add(l->per, slots, n);
    memcpy(obs, next, sizeof(float) * (size_t)n * D);
    *env_ctr += 1;
    free(a); free(slots); free(next); free(r); free(q); free(d);
}
