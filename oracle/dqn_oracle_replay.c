/* oracle/dqn_oracle_replay.c -- CPU restatement: RNG, replay ring, PER sum-tree.
 * TEST INFRASTRUCTURE ONLY (see dqn_oracle.h). PARITY UNPINNED (see dqn_oracle.h).
 * Integer / indexing work; every f32 op is a single IEEE rounding (-ffp-contract=off).
 */
#include "dqn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Philox */
/* Philox4x32-10 (Salmon et al., SC'11), published algorithm restated. */
static inline void mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float orc_u01(uint32_t x) { return (float)(x >> 8) * 0x1.0p-24f; }

static inline void philox_draw(uint64_t seed, uint64_t ctr, uint32_t k, uint32_t stream, uint32_t out[4]) {
    uint32_t c[4] = { (uint32_t)ctr, (uint32_t)(ctr >> 32), k, stream };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    orc_philox4x32_10(c, key, out);
}

/* ---------------------------------------------------------- deterministic pow */
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

float orc_log2_det(float x) {
    uint32_t u = f2bits(x);
    int32_t e = (int32_t)((u >> 23) & 0xFF) - 127;
    float m = bits2f((u & 0x007FFFFFu) | 0x3F800000u);       /* [1,2) */
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }           /* [0.7071,1.4142] */
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float p = 0.111111112f;                                   /* 1/9 */
    p = p * z; p = p + 0.142857149f;                          /* 1/7 */
    p = p * z; p = p + 0.2f;
    p = p * z; p = p + 0.333333343f;                          /* 1/3 */
    p = p * z; p = p + 1.0f;
    float ln_m = (2.0f * s) * p;
    float r = ln_m * 1.44269502f;                             /* 1/ln2 */
    return (float)e + r;
}

float orc_exp2_det(float y) {
    float fi = floorf(y + 0.5f);
    int32_t i = (int32_t)fi;
    float f = y - fi;                                         /* [-0.5,0.5] */
    float t = f * 0.693147182f;
    float p = 1.98412701e-4f;                                 /* 1/5040 */
    p = p * t; p = p + 1.38888892e-3f;                        /* 1/720 */
    p = p * t; p = p + 8.33333377e-3f;                        /* 1/120 */
    p = p * t; p = p + 4.16666679e-2f;                        /* 1/24 */
    p = p * t; p = p + 0.166666672f;                          /* 1/6 */
    p = p * t; p = p + 0.5f;
    p = p * t; p = p + 1.0f;
    p = p * t; p = p + 1.0f;
    if (i < -126) i = -126;
    if (i > 127) i = 127;
    return p * bits2f((uint32_t)(i + 127) << 23);
}

float orc_pow_det(float x, float a) { return orc_exp2_det(a * orc_log2_det(x)); }

/* ------------------------------------------------------------- replay ring */
int orc_replay_init(orc_replay *rb, int64_t capacity, int32_t obs_dim) {
    /* General/Base/replay_buffer.py:20-34 -- zero-filled SoA arrays */
    memset(rb, 0, sizeof(*rb));
    rb->capacity = capacity; rb->obs_dim = obs_dim;
    rb->states       = (float *)calloc((size_t)capacity * obs_dim, sizeof(float));
    rb->actions      = (int32_t *)calloc((size_t)capacity, sizeof(int32_t));
    rb->rewards      = (float *)calloc((size_t)capacity, sizeof(float));
    rb->observations = (float *)calloc((size_t)capacity * obs_dim, sizeof(float));
    rb->dones        = (uint8_t *)calloc((size_t)capacity, 1);
    return (rb->states && rb->actions && rb->rewards && rb->observations && rb->dones) ? 0 : -1;
}

void orc_replay_free(orc_replay *rb) {
    free(rb->states); free(rb->actions); free(rb->rewards); free(rb->observations); free(rb->dones);
    memset(rb, 0, sizeof(*rb));
}

void orc_replay_add(orc_replay *rb, const float *s, const int32_t *a, const float *r,
                    const float *s2, const uint8_t *d, int64_t n, int32_t *slots_out) {
    /* General/Base/replay_buffer.py:58-65, applied to n rows in order */
    const int32_t D = rb->obs_dim;
    for (int64_t j = 0; j < n; ++j) {
        int64_t k = rb->counter % rb->capacity;                          /* :59 */
        memcpy(rb->states + k * D, s + j * D, sizeof(float) * D);        /* :59 */
        rb->actions[k] = a[j];                                           /* :60 */
        rb->rewards[k] = r[j];                                           /* :61 */
        memcpy(rb->observations + k * D, s2 + j * D, sizeof(float) * D); /* :62 */
        rb->dones[k] = d[j] ? 1 : 0;                                     /* :63 */
        rb->counter += 1;                                                /* :64 */
        rb->size = rb->counter < rb->capacity ? rb->counter : rb->capacity; /* :65 */
        if (slots_out) slots_out[j] = (int32_t)k;
    }
}

void orc_uniform_indices(int64_t size, int32_t B, uint64_t seed, uint64_t ctr, int32_t *idx) {
    /* replay_buffer.py:77 randint(0, num_samples, batch_size): uniform, with replacement.
     * numba's RNG state is not reproducible from outside, so the build's spec is Philox. */
    for (int32_t k = 0; k < B; ++k) {
        uint32_t o[4];
        philox_draw(seed, ctr, (uint32_t)k, ORC_STREAM_UNIFORM, o);
        idx[k] = (int32_t)(((uint64_t)o[0] * (uint64_t)size) >> 32);
    }
}

void orc_replay_gather(const orc_replay *rb, const int32_t *idx, int32_t B,
                       float *s, int32_t *a, float *r, float *s2, uint8_t *d) {
    /* replay_buffer.py:78-84 */
    const int32_t D = rb->obs_dim;
    for (int32_t k = 0; k < B; ++k) {
        int64_t i = idx[k];
        memcpy(s + (int64_t)k * D, rb->states + i * D, sizeof(float) * D);
        a[k] = rb->actions[i];
        r[k] = rb->rewards[i];
        memcpy(s2 + (int64_t)k * D, rb->observations + i * D, sizeof(float) * D);
        d[k] = rb->dones[i];
    }
}

/* ---------------------------------------------------------------- sum-tree */
int orc_per_init(orc_per *t, int32_t L, float alpha, float eps) {
    memset(t, 0, sizeof(*t));
    t->L = L; t->N = (int64_t)1 << L;
    t->tree = (float *)calloc((size_t)(2 * t->N), sizeof(float));
    t->pmax = 1.0f; t->alpha = alpha; t->eps = eps;
    return t->tree ? 0 : -1;
}

void orc_per_free(orc_per *t) { free(t->tree); memset(t, 0, sizeof(*t)); }

static inline void walk_up(orc_per *t, int64_t node) {
    /* parents recomputed as left + right: never delta-added, so CPU == GPU bitwise */
    for (node >>= 1; node >= 1; node >>= 1)
        t->tree[node] = t->tree[2 * node] + t->tree[2 * node + 1];
}

void orc_per_add(orc_per *t, const int32_t *slots, int64_t n) {
    for (int64_t j = 0; j < n; ++j) {
        int64_t node = t->N + slots[j];
        t->tree[node] = t->pmax;
        walk_up(t, node);
    }
}

void orc_per_set(orc_per *t, const int32_t *idx, const float *prio, int32_t B) {
    /* sequential order => for duplicate idx the highest batch position wins */
    for (int32_t i = 0; i < B; ++i) {
        int64_t node = t->N + idx[i];
        t->tree[node] = prio[i];
        if (prio[i] > t->pmax) t->pmax = prio[i];
        walk_up(t, node);
    }
}

void orc_per_update(orc_per *t, const int32_t *idx, const float *td_abs, int32_t B) {
    float *p = (float *)malloc(sizeof(float) * (size_t)B);
    for (int32_t i = 0; i < B; ++i) p[i] = orc_pow_det(td_abs[i] + t->eps, t->alpha);
    orc_per_set(t, idx, p, B);
    free(p);
}

void orc_per_sample(const orc_per *t, int64_t size, int32_t B, float beta,
                    uint64_t seed, uint64_t ctr, int32_t *idx, float *isw) {
    const float total = t->tree[1];
    const float seg = total / (float)B;
    float wmax = 0.0f;
    for (int32_t k = 0; k < B; ++k) {
        uint32_t o[4];
        philox_draw(seed, ctr, (uint32_t)k, ORC_STREAM_PER, o);
        float u = ((float)k + orc_u01(o[0])) * seg;
        int64_t node = 1;
        while (node < t->N) {
            float l = t->tree[2 * node];
            if (u < l) node = 2 * node;
            else { u = u - l; node = 2 * node + 1; }
        }
        int64_t leaf = node - t->N;
        if (leaf >= size) leaf = size - 1;
        idx[k] = (int32_t)leaf;
        float p = t->tree[t->N + leaf];
        float w = orc_pow_det(((float)size * p) / total, -beta);
        isw[k] = w;
        if (w > wmax) wmax = w;
    }
    for (int32_t k = 0; k < B; ++k) isw[k] = isw[k] / wmax;
}
