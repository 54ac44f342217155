// csrc/dqn_net_common.h -- device helpers shared by the f32 (dqn_net.hip) and bf16 (dqn_net_bf16.hip)
// Q-network kernels: the per-row TD arithmetic, Huber loss, epsilon-greedy policy and the Adam element.
#pragma once
#include "dqn_device.h"
#include "dqn_launch.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Agent._policy (q_agent.py:137-141): greedy iff eps < U(0,1) else randint(0, A);
// compute_action (q_learning_functions.py:70): argmax, first max wins.
__device__ __forceinline__ int policy_row(const float *q, int A, float epsilon, unsigned long long seed,
                                          unsigned long long ctr, int i) {
    const u32x4 o = philox_draw(seed, ctr, (uint32_t)i, DQN_STREAM_POLICY);
    if (epsilon < u01(o.x)) {
        int act = 0;
        for (int k = 1; k < A; ++k) if (q[k] > q[act]) act = k;
        return act;
    }
    return (int)(((unsigned long long)o.y * (unsigned long long)A) >> 32);
}

// q_learning_functions.py:55-60 for one row. Returns delta; writes the target row.
__device__ __forceinline__ float td_row(const float *q, const float *nq, const float *nt, int a, float r,
                                        float d, float gamma, int A, float *target_row) {
    int astar = 0;                                             // :55 argmax, first max wins
    for (int k = 1; k < A; ++k) if (nq[k] > nq[astar]) astar = k;
    const float t1 = gamma * nt[astar];                        // :58, quirk Q3: (1-d) covers -q too
    const float t2 = t1 - q[a];
    const float t3 = (1.0f - d) * t2;
    const float delta = r + t3;
    for (int k = 0; k < A; ++k) target_row[k] = q[k] + delta * (k == a ? 1.0f : 0.0f);   // :59, quirk Q4
    return delta;
}

__device__ __forceinline__ float huber(float e) {              // optax.huber_loss(delta=1)
    const float ae = fabsf(e);
    const float qd = ae < 1.0f ? ae : 1.0f;
    return 0.5f * (qd * qd) + (ae - qd);
}

// optax scale_by_adam -> add_decayed_weights (adamw) -> scale(-lr) -> apply_updates, one element.
// One IEEE rounding per written operation: bit-exact against the CPU restatement. (sqrtf, not __fsqrt_rn: in this
// toolchain the intrinsic is the 1-ulp native square root; sqrtf and `/` are the correctly rounded forms.)
struct AdamCoef { float c1, c2, omb1, omb2, neglr; };

__device__ __forceinline__ AdamCoef adam_coef(const DqnState *st, float b1, float b2, double *b1pow, double *b2pow) {
    *b1pow = st->b1pow * (double)b1; *b2pow = st->b2pow * (double)b2;
    return AdamCoef{(float)(1.0 - *b1pow), (float)(1.0 - *b2pow), 1.0f - b1, 1.0f - b2, -st->lr};
}

__device__ __forceinline__ float adam_elem(const AdamCoef &c, float g, float *P, float *mu, float *nu, int i,
                                           int adamw, float b1, float b2, float eps, float wd, float grad_scale) {
    const float gi = g * grad_scale;
    const float mm = (b1 * mu[i]) + (c.omb1 * gi);
    const float vv = (b2 * nu[i]) + (c.omb2 * (gi * gi));
    mu[i] = mm; nu[i] = vv;
    const float mhat = __fdiv_rn(mm, c.c1), vhat = __fdiv_rn(vv, c.c2);
    float u = __fdiv_rn(mhat, sqrtf(vhat) + eps);
    float p = P[i];
    if (adamw) u = u + (wd * p);
    p = p + (c.neglr * u);
    P[i] = p;
    return p;
}

// index into a fragment-packed f32 matrix (dqn_net.hip) of element [k][c]; KQ = number of 16-row k-blocks
__device__ __host__ __forceinline__ long long pidx(int KQ, int k, int c) {
    const int kq = k >> 4, kk = k & 15;
    return ((long long)((c >> 4) * KQ + kq)) * 256 + ((((kk & 3) << 4) | (c & 15)) << 2) + (kk >> 2);
}

// the two f32 shadows the actor kernel (dqn_actor.hip) reads -- k-packed W2 and the transposed heads -- for parameter
// element i; in bf16 mode they live in their own f32 buffer beside the bf16 packs
__device__ __forceinline__ void scatter_actor_packs(const NetDims &m, int i, float v, float *packf) {
    const int o_w2 = (int)m.o_w2, o_b2 = (int)m.o_b2, o_wv = (int)m.o_wv, o_bv = (int)m.o_bv, o_wa = (int)m.o_wa, o_ba = (int)m.o_ba;
    if (i >= o_w2 && i < o_b2) {
        const int u = i - o_w2, k = u / m.H2, n = u - k * m.H2;
        packf[m.p_w2k + ((long long)(k >> 2) * m.H2 + n) * 4 + (k & 3)] = v;
        // behind the f32 shadows: W2 once more as bf16, eight consecutive k of a column per 16 bytes -- the bf16 actor's
        // register slab arrives ready-made (half the bytes of the f32 shadow, no conversion in its prologue)
        reinterpret_cast<__bf16 *>(packf + m.pack_floats)[((long long)(k >> 3) * m.H2 + n) * 8 + (k & 7)] = (__bf16)v;
    } else if (i >= o_wv && i < o_bv) {
        packf[m.p_wht + pidx(1, 0, i - o_wv)] = v;
    } else if (i >= o_wa && i < o_ba) {
        const int u = i - o_wa, k = u / m.A, a = u - k * m.A;
        packf[m.p_wht + pidx(1, 1 + a, k)] = v;
    }
}

