/* oracle/dqn_oracle_net.c -- CPU restatement: dueling Q-net, double-Q targets, Huber
 * loss, backward, Adam/AdamW, epsilon-greedy, and a whole-update driver.
 * TEST INFRASTRUCTURE ONLY (see dqn_oracle.h). PARITY UNPINNED (see dqn_oracle.h).
 *
 * f32 throughout. Dot products are k-ordered fmaf chains starting from 0 (the same
 * arithmetic as gfx950's v_mfma_f32_*_f32), bias added afterwards (hk.Linear: x@w + b);
 * the heads' K = hidden2 sums are four such chains over interleaved k groups (heads_row).
 * Everything else is one IEEE rounding per operation (-ffp-contract=off).
 */
#include "dqn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int64_t orc_param_count(orc_dims m) {
    return (int64_t)m.D * m.H1 + m.H1 + (int64_t)m.H1 * m.H2 + m.H2 + m.H2 + 1 + (int64_t)m.H2 * m.A + m.A;
}

typedef struct { int64_t w1, b1, w2, b2, wv, bv, wa, ba; } offs_t;
static offs_t offsets(orc_dims m) {
    offs_t o; int64_t p = 0;
    o.w1 = p; p += (int64_t)m.D * m.H1;  o.b1 = p; p += m.H1;
    o.w2 = p; p += (int64_t)m.H1 * m.H2; o.b2 = p; p += m.H2;
    o.wv = p; p += m.H2;                 o.bv = p; p += 1;
    o.wa = p; p += (int64_t)m.H2 * m.A;  o.ba = p; p += m.A;
    return o;
}

/* y[n] = (sum_k x[k]*w[k,n], k ascending, fmaf chain from 0) + b[n] */
static void linear_row(const float *x, const float *w, const float *b, int K, int N, float *y) {
    for (int n = 0; n < N; ++n) y[n] = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float xk = x[k];
        const float *wr = w + (int64_t)k * N;
        for (int n = 0; n < N; ++n) y[n] = fmaf(xk, wr[n], y[n]);
    }
    for (int n = 0; n < N; ++n) y[n] = y[n] + b[n];
}

/* The two heads (dddqn.py:29-30; K = hidden2, a multiple of 16): four partial fmaf chains from 0, chain j over the k with
 * (k / 4) % 4 == j in ascending order, combined as ((c0 + c1) + (c2 + c3)) + b[n]. On the GPU chain j is the j-th
 * v_mfma_f32_16x16x4_f32 of every 16-deep k-block (dqn_net.hip) / wave j of a k_actor workgroup: a quarter of the dependent
 * chain length of the plain k-ordered sum. */
static void heads_row(const float *x, const float *w, const float *b, int K, int N, float *y) {
    for (int n = 0; n < N; ++n) {
        float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = 0; k < K; ++k) c[(k >> 2) & 3] = fmaf(x[k], w[(int64_t)k * N + n], c[(k >> 2) & 3]);
        y[n] = ((c[0] + c[1]) + (c[2] + c[3])) + b[n];
    }
}

void orc_forward(orc_dims m, const float *P, const float *x, int32_t B,
                 float *q, float *h1_out, float *h2_out) {
    /* LunarLander/dddqn.py:24-31 */
    offs_t o = offsets(m);
    float *h1 = (float *)malloc(sizeof(float) * m.H1);
    float *h2 = (float *)malloc(sizeof(float) * m.H2);
    float adv[64];
    for (int32_t i = 0; i < B; ++i) {
        linear_row(x + (int64_t)i * m.D, P + o.w1, P + o.b1, m.D, m.H1, h1);   /* :25 */
        for (int j = 0; j < m.H1; ++j) h1[j] = h1[j] > 0.0f ? h1[j] : 0.0f;   /* :26 */
        linear_row(h1, P + o.w2, P + o.b2, m.H1, m.H2, h2);                    /* :27 */
        for (int j = 0; j < m.H2; ++j) h2[j] = h2[j] > 0.0f ? h2[j] : 0.0f;   /* :28 */
        float v;
        heads_row(h2, P + o.wv, P + o.bv, m.H2, 1, &v);                        /* :29 */
        heads_row(h2, P + o.wa, P + o.ba, m.H2, m.A, adv);                     /* :30 */
        float sum = 0.0f;
        for (int a = 0; a < m.A; ++a) sum = sum + adv[a];
        const float mean = sum / (float)m.A;
        for (int a = 0; a < m.A; ++a) q[(int64_t)i * m.A + a] = (v + adv[a]) - mean; /* :31 */
        if (h1_out) memcpy(h1_out + (int64_t)i * m.H1, h1, sizeof(float) * m.H1);
        if (h2_out) memcpy(h2_out + (int64_t)i * m.H2, h2, sizeof(float) * m.H2);
    }
    free(h1); free(h2);
}

void orc_q_targets(orc_dims m, const float *P, const float *Pt, const float *s,
                   const int32_t *a, const float *r, const float *s2, const float *d,
                   float gamma, int32_t B, float *targets,
                   float *q_out, float *nq_out, float *nt_out, int32_t *astar_out, float *delta_out) {
    /* General/QLearning/q_learning_functions.py:52-61 */
    const int A = m.A;
    float *q  = (float *)malloc(sizeof(float) * (size_t)B * A);
    float *nq = (float *)malloc(sizeof(float) * (size_t)B * A);
    float *nt = (float *)malloc(sizeof(float) * (size_t)B * A);
    orc_forward(m, P,  s,  B, q,  NULL, NULL);   /* :52 */
    orc_forward(m, P,  s2, B, nq, NULL, NULL);   /* :53 */
    orc_forward(m, Pt, s2, B, nt, NULL, NULL);   /* :54 */
    for (int32_t i = 0; i < B; ++i) {
        const float *nqi = nq + (int64_t)i * A;
        int astar = 0;                            /* :55 argmax, first max wins */
        for (int k = 1; k < A; ++k) if (nqi[k] > nqi[astar]) astar = k;
        const float qa = q[(int64_t)i * A + a[i]];
        /* :58  r + (1 - done) * (gamma * next_q_tm[max_action] - q[action])   (quirk Q3) */
        const float t1 = gamma * nt[(int64_t)i * A + astar];
        const float t2 = t1 - qa;
        const float t3 = (1.0f - d[i]) * t2;
        const float delta = r[i] + t3;
        /* :59  q + target_val * one_hot(action)                              (quirk Q4) */
        for (int k = 0; k < A; ++k)
            targets[(int64_t)i * A + k] = q[(int64_t)i * A + k] + delta * (k == a[i] ? 1.0f : 0.0f);
        if (astar_out) astar_out[i] = astar;
        if (delta_out) delta_out[i] = delta;
    }
    if (q_out)  memcpy(q_out,  q,  sizeof(float) * (size_t)B * A);
    if (nq_out) memcpy(nq_out, nq, sizeof(float) * (size_t)B * A);
    if (nt_out) memcpy(nt_out, nt, sizeof(float) * (size_t)B * A);
    free(q); free(nq); free(nt);
}

static inline float huber(float e) {
    /* optax.huber_loss(delta=1): 0.5*min(|e|,1)^2 + (|e| - min(|e|,1)) */
    const float ae = fabsf(e);
    const float qd = ae < 1.0f ? ae : 1.0f;
    return 0.5f * (qd * qd) + (ae - qd);
}

float orc_loss(orc_dims m, const float *P, const float *s, const float *targets,
               const float *isw, int32_t B) {
    /* q_learning_functions.py:35-36: mean_i sum_a huber(pred - target) */
    const int A = m.A;
    float *pred = (float *)malloc(sizeof(float) * (size_t)B * A);
    orc_forward(m, P, s, B, pred, NULL, NULL);
    float acc = 0.0f;
    for (int32_t i = 0; i < B; ++i) {
        float row = 0.0f;
        for (int k = 0; k < A; ++k) row = row + huber(pred[(int64_t)i * A + k] - targets[(int64_t)i * A + k]);
        if (isw) row = isw[i] * row;
        acc = acc + row;
    }
    free(pred);
    return acc / (float)B;
}

void orc_grads(orc_dims m, const float *P, const float *s, const float *targets,
               const float *isw, int32_t B, float *grad, float *loss_out, float *dq_out) {
    /* jax.grad(compute_loss) (q_learning_functions.py:23), hand-derived:
     *   dL/dpred = w_i * clip(pred - target, -1, 1) / B
     *   dueling: dv = sum_a g_a ; dadv_j = g_j - (1/A) sum_a g_a
     *   relu' = 1 where the activation is > 0, else 0 */
    const int D = m.D, H1 = m.H1, H2 = m.H2, A = m.A;
    offs_t o = offsets(m);
    const int64_t n = orc_param_count(m);
    float *pred = (float *)malloc(sizeof(float) * (size_t)B * A);
    float *h1 = (float *)malloc(sizeof(float) * (size_t)B * H1);
    float *h2 = (float *)malloc(sizeof(float) * (size_t)B * H2);
    float *dz2 = (float *)malloc(sizeof(float) * H2);
    float *dz1 = (float *)malloc(sizeof(float) * H1);
    orc_forward(m, P, s, B, pred, h1, h2);
    memset(grad, 0, sizeof(float) * (size_t)n);
    float acc = 0.0f;
    const float invB = 1.0f / (float)B;
    for (int32_t i = 0; i < B; ++i) {
        float g[64], dadv[64];
        float row = 0.0f, gsum = 0.0f;
        const float w = isw ? isw[i] : 1.0f;
        for (int k = 0; k < A; ++k) {
            const float e = pred[(int64_t)i * A + k] - targets[(int64_t)i * A + k];
            row = row + huber(e);
            const float c = e > 1.0f ? 1.0f : (e < -1.0f ? -1.0f : e);
            g[k] = (w * c) * invB;
            gsum = gsum + g[k];
            if (dq_out) dq_out[(int64_t)i * A + k] = g[k];
        }
        acc = acc + (isw ? w * row : row);
        const float dv = gsum;
        const float gmean = gsum / (float)A;
        for (int k = 0; k < A; ++k) dadv[k] = g[k] - gmean;
        const float *h2i = h2 + (int64_t)i * H2, *h1i = h1 + (int64_t)i * H1, *xi = s + (int64_t)i * D;
        /* heads */
        for (int j = 0; j < H2; ++j) {
            grad[o.wv + j] = fmaf(h2i[j], dv, grad[o.wv + j]);
            for (int k = 0; k < A; ++k)
                grad[o.wa + (int64_t)j * A + k] = fmaf(h2i[j], dadv[k], grad[o.wa + (int64_t)j * A + k]);
        }
        grad[o.bv] = grad[o.bv] + dv;
        for (int k = 0; k < A; ++k) grad[o.ba + k] = grad[o.ba + k] + dadv[k];
        /* dh2 = dv*wv^T + dadv*wa^T, through ReLU */
        for (int j = 0; j < H2; ++j) {
            float t = P[o.wv + j] * dv;
            for (int k = 0; k < A; ++k) t = fmaf(P[o.wa + (int64_t)j * A + k], dadv[k], t);
            dz2[j] = h2i[j] > 0.0f ? t : 0.0f;
        }
        /* layer 2 */
        for (int k = 0; k < H1; ++k) {
            const float hk = h1i[k];
            float *gw = grad + o.w2 + (int64_t)k * H2;
            for (int j = 0; j < H2; ++j) gw[j] = fmaf(hk, dz2[j], gw[j]);
        }
        for (int j = 0; j < H2; ++j) grad[o.b2 + j] = grad[o.b2 + j] + dz2[j];
        for (int k = 0; k < H1; ++k) {
            const float *wr = P + o.w2 + (int64_t)k * H2;
            float t = 0.0f;
            for (int j = 0; j < H2; ++j) t = fmaf(wr[j], dz2[j], t);
            dz1[k] = h1i[k] > 0.0f ? t : 0.0f;
        }
        /* layer 1 */
        for (int k = 0; k < D; ++k) {
            const float xk = xi[k];
            float *gw = grad + o.w1 + (int64_t)k * H1;
            for (int j = 0; j < H1; ++j) gw[j] = fmaf(xk, dz1[j], gw[j]);
        }
        for (int j = 0; j < H1; ++j) grad[o.b1 + j] = grad[o.b1 + j] + dz1[j];
    }
    if (loss_out) *loss_out = acc / (float)B;
    free(pred); free(h1); free(h2); free(dz2); free(dz1);
}

void orc_adam_step(orc_opt o, float *P, const float *g, float *mu, float *nu,
                   int32_t *count, double *b1pow, double *b2pow, int64_t n, float grad_scale) {
    /* optax scale_by_adam -> (adamw: add_decayed_weights) -> scale(-lr) -> apply_updates
     * (call sites: Test/lunar_lander.py:48, q_learning_functions.py:24-25) */
    *count += 1;
    *b1pow *= (double)o.b1;
    *b2pow *= (double)o.b2;
    const float c1 = (float)(1.0 - *b1pow), c2 = (float)(1.0 - *b2pow);
    const float omb1 = 1.0f - o.b1, omb2 = 1.0f - o.b2, neglr = -o.lr;
    for (int64_t i = 0; i < n; ++i) {
        const float gi = g[i] * grad_scale;
        const float m = (o.b1 * mu[i]) + (omb1 * gi);
        const float v = (o.b2 * nu[i]) + (omb2 * (gi * gi));
        mu[i] = m; nu[i] = v;
        const float mhat = m / c1, vhat = v / c2;
        float u = mhat / (sqrtf(vhat) + o.eps);
        if (o.adamw) u = u + (o.wd * P[i]);
        P[i] = P[i] + (neglr * u);
    }
}

void orc_act(orc_dims m, const float *P, const float *s, int32_t n, float epsilon,
             uint64_t seed, uint64_t ctr, int32_t *actions) {
    /* q_agent.py:137-141: greedy iff epsilon < U(0,1) else randint(0, A);
     * q_learning_functions.py:70: argmax over the (1,A) output */
    float *q = (float *)malloc(sizeof(float) * (size_t)n * m.A);
    orc_forward(m, P, s, n, q, NULL, NULL);
    for (int32_t i = 0; i < n; ++i) {
        uint32_t c[4] = { (uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)i, ORC_STREAM_POLICY };
        uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) }, out[4];
        orc_philox4x32_10(c, key, out);
        if (epsilon < orc_u01(out[0])) {
            int best = 0;
            for (int k = 1; k < m.A; ++k) if (q[(int64_t)i * m.A + k] > q[(int64_t)i * m.A + best]) best = k;
            actions[i] = best;
        } else {
            actions[i] = (int32_t)(((uint64_t)out[1] * (uint64_t)m.A) >> 32);
        }
    }
    free(q);
}

void orc_obs_augment(const float *obs, const int32_t *step, int32_t max_steps,
                     int32_t n, int32_t D, float *out) {
    /* LunarLander/env.py:19-21: append(observation, step/max_steps).astype(float32)
     * (python float division is f64, then cast) */
    for (int32_t i = 0; i < n; ++i) {
        memcpy(out + (int64_t)i * (D + 1), obs + (int64_t)i * D, sizeof(float) * D);
        out[(int64_t)i * (D + 1) + D] = (float)((double)step[i] / (double)max_steps);
    }
}

/* ------------------------------------------------------------ whole update */
int orc_learner_init(orc_learner *l, orc_dims m, orc_opt opt, float gamma, int32_t maxB,
                     orc_replay *rb, orc_per *per, const float *P0, uint64_t seed) {
    memset(l, 0, sizeof(*l));
    l->m = m; l->opt = opt; l->gamma = gamma; l->beta = 0.4f; l->rb = rb; l->per = per;
    l->maxB = maxB; l->seed = seed; l->ctr = 0; l->count = 0; l->b1pow = 1.0; l->b2pow = 1.0;
    const int64_t n = orc_param_count(m);
    l->P = (float *)malloc(sizeof(float) * n);  l->Pt = (float *)malloc(sizeof(float) * n);
    l->mu = (float *)calloc(n, sizeof(float));  l->nu = (float *)calloc(n, sizeof(float));
    l->grad = (float *)malloc(sizeof(float) * n);
    memcpy(l->P, P0, sizeof(float) * n); memcpy(l->Pt, P0, sizeof(float) * n);
    l->idx = (int32_t *)malloc(sizeof(int32_t) * maxB); l->a = (int32_t *)malloc(sizeof(int32_t) * maxB);
    l->isw = (float *)malloc(sizeof(float) * maxB);
    l->s = (float *)malloc(sizeof(float) * (size_t)maxB * m.D); l->s2 = (float *)malloc(sizeof(float) * (size_t)maxB * m.D);
    l->r = (float *)malloc(sizeof(float) * maxB); l->df = (float *)malloc(sizeof(float) * maxB);
    l->d = (uint8_t *)malloc(maxB);
    l->targets = (float *)malloc(sizeof(float) * (size_t)maxB * m.A);
    l->delta = (float *)malloc(sizeof(float) * maxB);
    return 0;
}

void orc_learner_set_nstep(orc_learner *l, int32_t n_step, int32_t n_envs) {
    free(l->hist_s); free(l->hist_a); free(l->hist_r); free(l->hist_d);
    l->hist_s = NULL; l->hist_a = NULL; l->hist_r = NULL; l->hist_d = NULL;
    l->n_step = n_step; l->hist_n = n_envs; l->hist_steps = 0; l->gamma_n = l->gamma;
    if (n_step <= 1) return;
    for (int32_t i = 1; i < n_step; ++i) l->gamma_n = l->gamma_n * l->gamma;
    l->hist_s = (float *)calloc((size_t)n_step * n_envs * l->m.D, sizeof(float));
    l->hist_a = (int32_t *)calloc((size_t)n_step * n_envs, sizeof(int32_t));
    l->hist_r = (float *)calloc((size_t)n_step * n_envs, sizeof(float));
    l->hist_d = (uint8_t *)calloc((size_t)n_step * n_envs, 1);
}

void orc_learner_free(orc_learner *l) {
    free(l->hist_s); free(l->hist_a); free(l->hist_r); free(l->hist_d);
    free(l->P); free(l->Pt); free(l->mu); free(l->nu); free(l->grad); free(l->idx); free(l->a);
    free(l->isw); free(l->s); free(l->s2); free(l->r); free(l->df); free(l->d); free(l->targets); free(l->delta);
    memset(l, 0, sizeof(*l));
}

float orc_learner_update(orc_learner *l, int32_t B) {
    /* q_agent.py:146-169 (_step): sample -> preprocessing -> q_targets -> train_step */
    float loss = 0.0f;
    if (l->per) orc_per_sample(l->per, l->rb->size, B, l->beta, l->seed, l->ctr, l->idx, l->isw);
    else        orc_uniform_indices(l->rb->size, B, l->seed, l->ctr, l->idx);
    l->ctr += 1;
    orc_replay_gather(l->rb, l->idx, B, l->s, l->a, l->r, l->s2, l->d);
    for (int32_t i = 0; i < B; ++i) l->df[i] = l->d[i] ? 1.0f : 0.0f;   /* preprocessing :84 */
    orc_q_targets(l->m, l->P, l->Pt, l->s, l->a, l->r, l->s2, l->df, l->n_step > 1 ? l->gamma_n : l->gamma, B,
                  l->targets, NULL, NULL, NULL, NULL, l->delta);
    orc_grads(l->m, l->P, l->s, l->targets, l->per ? l->isw : NULL, B, l->grad, &loss, NULL);
    orc_adam_step(l->opt, l->P, l->grad, l->mu, l->nu, &l->count, &l->b1pow, &l->b2pow,
                  orc_param_count(l->m), 1.0f);
    if (l->per) {
        for (int32_t i = 0; i < B; ++i) l->delta[i] = fabsf(l->delta[i]);
        orc_per_update(l->per, l->idx, l->delta, B);
    }
    return loss;
}

/* ------------------------------------------------------------ synthetic actor */
static inline float ih_normal(const uint32_t o[4]) {
    return (((orc_u01(o[0]) + orc_u01(o[1])) + (orc_u01(o[2]) + orc_u01(o[3]))) - 2.0f) * 1.73205078f;
}

void orc_synth_env(int32_t n, int32_t D, uint64_t seed, uint64_t env_ctr, float p_done,
                   float *obs_next, float *r, uint8_t *d) {
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    for (int32_t i = 0; i < n; ++i) {
        const uint32_t base = (uint32_t)i * (uint32_t)(D + 1);
        uint32_t c[4] = { (uint32_t)env_ctr, (uint32_t)(env_ctr >> 32), 0, ORC_STREAM_ENV }, o[4];
        for (int32_t e = 0; e < D; ++e) {
            c[2] = base + (uint32_t)e;
            orc_philox4x32_10(c, key, o);
            obs_next[(int64_t)i * D + e] = ih_normal(o);
        }
        c[2] = base + (uint32_t)D;
        orc_philox4x32_10(c, key, o);
        const int done = orc_u01(o[0]) < p_done;
        float rew = (((orc_u01(o[1]) + orc_u01(o[2])) + (orc_u01(o[3]) + orc_u01(o[0]))) - 2.0f) * 1.73205078f;
        if (done) rew = (o[1] & 1u) ? 100.0f : -100.0f;
        r[i] = rew;
        d[i] = done ? 1 : 0;
    }
}

void orc_learner_actor_step(orc_learner *l, float *obs, int32_t n, float epsilon, float p_done, uint64_t *env_ctr) {
    const int32_t D = l->m.D;
    int32_t *a = (int32_t *)malloc(sizeof(int32_t) * n), *slots = (int32_t *)malloc(sizeof(int32_t) * n);
    float *next = (float *)malloc(sizeof(float) * (size_t)n * D), *r = (float *)malloc(sizeof(float) * n);
    uint8_t *d = (uint8_t *)malloc(n);
    orc_act(l->m, l->P, obs, n, epsilon, l->seed, *env_ctr, a);                 /* q_agent.py:176 */
    orc_synth_env(n, D, l->seed, *env_ctr, p_done, next, r, d);                  /* :177 (synthetic) */
    if (l->n_step > 1) {
        /* n-step: file this step in the history; once n_step entries exist, add the row of the oldest window */
        const int32_t ns = l->n_step;
        const int32_t pos = (int32_t)(l->hist_steps % (uint64_t)ns);
        memcpy(l->hist_s + (size_t)pos * n * D, obs, sizeof(float) * (size_t)n * D);
        memcpy(l->hist_a + (size_t)pos * n, a, sizeof(int32_t) * n);
        memcpy(l->hist_r + (size_t)pos * n, r, sizeof(float) * n);
        memcpy(l->hist_d + (size_t)pos * n, d, n);
        if (l->hist_steps + 1 >= (uint64_t)ns) {
            const int32_t pu = (pos + 1) % ns;                                   /* oldest entry of the window */
            float *R = (float *)malloc(sizeof(float) * n); uint8_t *dn = (uint8_t *)malloc(n);
            for (int32_t i = 0; i < n; ++i) {
                int32_t last = ns - 1;                                           /* window cut after the first done */
                for (int32_t k = 0; k < ns; ++k) if (l->hist_d[(size_t)((pu + k) % ns) * n + i]) { last = k; break; }
                float acc = l->hist_r[(size_t)((pu + last) % ns) * n + i];
                for (int32_t k = last - 1; k >= 0; --k) acc = l->hist_r[(size_t)((pu + k) % ns) * n + i] + l->gamma * acc;
                R[i] = acc;
                dn[i] = l->hist_d[(size_t)((pu + last) % ns) * n + i];
            }
            orc_replay_add(l->rb, l->hist_s + (size_t)pu * n * D, l->hist_a + (size_t)pu * n, R, next, dn, n, slots);
            if (l->per) orc_per_add(l->per, slots, n);
            free(R); free(dn);
        }
        l->hist_steps += 1;
        memcpy(obs, next, sizeof(float) * (size_t)n * D);
        *env_ctr += 1;
        free(a); free(slots); free(next); free(r); free(d);
        return;
    }
    orc_replay_add(l->rb, obs, a, r, next, d, n, slots);                         /* :182 */
    if (l->per) orc_per_add(l->per, slots, n);
    memcpy(obs, next, sizeof(float) * (size_t)n * D);                            /* :183 */
    *env_ctr += 1;
    free(a); free(slots); free(next); free(r); free(d);
}

/* The vector actor step with ObsWrapper's feature (LunarLander/env.py:19-31) as the last observation column: obs[n,D] holds the
 * current observations incl. the feature, t[n] the envs' step counters (both advanced in place). Per env: the synthetic
 * transition as orc_learner_actor_step (the draw of the last column is unused), next[D-1] = float32(float64(t + 1) / max_steps)
 * (env.py:20, :24), done |= t + 1 >= max_steps (q_agent.py:179-180); the row goes to the ring; the env continues from `next`,
 * after an episode end with feature 0 and t = 0 (env.py:28-30). One-step returns. */
void orc_learner_actor_step_tf(orc_learner *l, float *obs, int32_t *t, int32_t n, float epsilon, float p_done, int32_t max_steps,
                               uint64_t *env_ctr) {
    const int32_t D = l->m.D;
    int32_t *a = (int32_t *)malloc(sizeof(int32_t) * n), *slots = (int32_t *)malloc(sizeof(int32_t) * n);
    float *next = (float *)malloc(sizeof(float) * (size_t)n * D), *r = (float *)malloc(sizeof(float) * n);
    uint8_t *d = (uint8_t *)malloc(n);
    orc_act(l->m, l->P, obs, n, epsilon, l->seed, *env_ctr, a);                 /* q_agent.py:176 */
    orc_synth_env(n, D, l->seed, *env_ctr, p_done, next, r, d);                  /* :177 (synthetic) */
    for (int32_t i = 0; i < n; ++i) {
        const int32_t tt = t[i] + 1;                                            /* env.py:24 */
        next[(int64_t)i * D + D - 1] = (float)((double)tt / (double)max_steps); /* env.py:20 */
        if (tt >= max_steps) d[i] = 1;                                          /* q_agent.py:179-180 */
    }
    orc_replay_add(l->rb, obs, a, r, next, d, n, slots);                         /* :182 */
    if (l->per) orc_per_add(l->per, slots, n);
    memcpy(obs, next, sizeof(float) * (size_t)n * D);                            /* :183 */
    for (int32_t i = 0; i < n; ++i) {
        if (d[i]) { obs[(int64_t)i * D + D - 1] = 0.0f; t[i] = 0; }             /* env.py:28-30 */
        else t[i] = t[i] + 1;
    }
    *env_ctr += 1;
    free(a); free(slots); free(next); free(r); free(d);
}
