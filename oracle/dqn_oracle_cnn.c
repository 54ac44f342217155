/* oracle/dqn_oracle_cnn.c -- CPU restatement of the Nature-CNN dueling Q-network forward (BASELINE configs[4]).
 * TEST INFRASTRUCTURE ONLY (see dqn_oracle.h). PARITY UNPINNED: the reference has no CNN at all;
 * what it contributes is the dueling head (LunarLander/dddqn.py:29-31) and the TD rule the Q values feed
 * (General/QLearning/q_learning_functions.py:55-60), both restated in dqn_oracle_net.c.
 *
 * f32; every dot product is ONE fmaf chain from 0 over k = (kh, kw, c) ascending (the order of an HWIO weight tensor), bias
 * added afterwards, ReLU; frames are scaled as (float)u8 / 255.0f. Layout NHWC throughout.
 */
#include "dqn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int IH, IW, IC, OH, OW, OC, KH, KW, S; } cgeom;
static const cgeom G[4] = { {84, 84, 4, 20, 20, 32, 8, 8, 4}, {20, 20, 32, 9, 9, 64, 4, 4, 2}, {9, 9, 64, 7, 7, 64, 3, 3, 1},
                            {1, 1, 3136, 1, 1, 512, 1, 1, 1} };

int64_t orc_cnn_param_count(int32_t A) {
    int64_t p = 0;
    for (int l = 0; l < 4; ++l) p += (int64_t)G[l].KH * G[l].KW * G[l].IC * G[l].OC + G[l].OC;
    return p + 512 + 1 + 512 * (int64_t)A + A;
}

static void layer(const cgeom g, const float *in, const float *w, const float *b, float *out) {
    float acc[512];
    for (int oh = 0; oh < g.OH; ++oh)
        for (int ow = 0; ow < g.OW; ++ow) {
            for (int n = 0; n < g.OC; ++n) acc[n] = 0.0f;
            for (int kh = 0; kh < g.KH; ++kh)
                for (int kw = 0; kw < g.KW; ++kw) {
                    const float *px = in + ((int64_t)(oh * g.S + kh) * g.IW + (ow * g.S + kw)) * g.IC;
                    const float *wk = w + (int64_t)((kh * g.KW + kw) * g.IC) * g.OC;
                    for (int c = 0; c < g.IC; ++c) {
                        const float xv = px[c];
                        const float *wr = wk + (int64_t)c * g.OC;
                        for (int n = 0; n < g.OC; ++n) acc[n] = fmaf(xv, wr[n], acc[n]);
                    }
                }
            float *o = out + ((int64_t)oh * g.OW + ow) * g.OC;
            for (int n = 0; n < g.OC; ++n) { const float v = acc[n] + b[n]; o[n] = v > 0.0f ? v : 0.0f; }
        }
}

/* q[B][A]; feat (optional) = the 512 fc features */
void orc_cnn_forward(const float *P, const uint8_t *frames, int32_t B, int32_t A, float *q, float *feat) {
    float *x = (float *)malloc(sizeof(float) * 84 * 84 * 4), *a0 = (float *)malloc(sizeof(float) * 20 * 20 * 32);
    float *a1 = (float *)malloc(sizeof(float) * 9 * 9 * 64), *a2 = (float *)malloc(sizeof(float) * 3136), *a3 = (float *)malloc(sizeof(float) * 512);
    int64_t o[4], ob[4], p = 0;
    for (int l = 0; l < 4; ++l) { o[l] = p; p += (int64_t)G[l].KH * G[l].KW * G[l].IC * G[l].OC; ob[l] = p; p += G[l].OC; }
    const int64_t o_wv = p, o_bv = p + 512, o_wa = p + 513, o_ba = p + 513 + 512 * (int64_t)A;
    for (int32_t i = 0; i < B; ++i) {
        const uint8_t *f = frames + (int64_t)i * 84 * 84 * 4;
        for (int t = 0; t < 84 * 84 * 4; ++t) x[t] = (float)f[t] / 255.0f;
        layer(G[0], x, P + o[0], P + ob[0], a0);
        layer(G[1], a0, P + o[1], P + ob[1], a1);
        layer(G[2], a1, P + o[2], P + ob[2], a2);
        layer(G[3], a2, P + o[3], P + ob[3], a3);
        if (feat) memcpy(feat + (int64_t)i * 512, a3, sizeof(float) * 512);
        /* dueling head: LunarLander/dddqn.py:29-31 (plain k-ascending chains here) */
        float v = 0.0f, adv[16], sum = 0.0f;
        for (int j = 0; j < A; ++j) adv[j] = 0.0f;
        for (int k = 0; k < 512; ++k) {
            v = fmaf(a3[k], P[o_wv + k], v);
            for (int j = 0; j < A; ++j) adv[j] = fmaf(a3[k], P[o_wa + (int64_t)k * A + j], adv[j]);
        }
        for (int j = 0; j < A; ++j) { adv[j] = adv[j] + P[o_ba + j]; sum = sum + adv[j]; }
        v = v + P[o_bv];
        const float mean = sum / (float)A;
        for (int j = 0; j < A; ++j) q[(int64_t)i * A + j] = (v + adv[j]) - mean;
    }
    free(x); free(a0); free(a1); free(a2); free(a3);
}

/* ---- loss gradient (f32 restatement and f64 tolerance reference): see dqn_oracle_cnn_grad.inc ---- */
#define REAL float
#define FMA(a, b, c) fmaf((a), (b), (c))
#define NAME(x) x##_f32
#include "dqn_oracle_cnn_grad.inc"
#undef REAL
#undef FMA
#undef NAME
#define REAL double
#define FMA(a, b, c) ((a) * (b) + (c))
#define NAME(x) x##_f64
#include "dqn_oracle_cnn_grad.inc"
#undef REAL
#undef FMA
#undef NAME

void orc_cnn_grads(const float *P, const uint8_t *frames, const float *targets, const float *isw, int32_t B, int32_t A,
                   float *grad, float *loss_out) {
    cnn_grads_f32(P, frames, targets, isw, B, A, grad, loss_out, NULL);
}
void orc_cnn_grads_f64(const float *P, const uint8_t *frames, const float *targets, const float *isw, int32_t B, int32_t A,
                       double *grad, double *loss_out) {
    cnn_grads_f64(P, frames, targets, isw, B, A, grad, loss_out, NULL);
}
