// csrc/dqn_bf16_pack.h -- the bf16 fragment-packed weight shadows (dqn_net_bf16.hip: k_pack16), shared by the 16-row bf16
// kernels and the 64-row bf16 kernels (dqn_net_big16.hip):
//   packed16(M)[((ct*KQ + kq)*64 + lane)*8 + j] = bf16( M[32*kq + 8*(lane>>4) + j][16*ct + (lane&15)] )
#pragma once
#include "dqn_net_common.h"

__device__ __host__ __forceinline__ long long pidx16(int KQ, int k, int c) {
    const int kq = k >> 5, kk = k & 31;
    return ((((long long)((c >> 4) * KQ + kq)) * 64 + (((kk >> 3) << 4) | (c & 15))) << 3) + (kk & 7);
}

struct Dims16 {
    int KQ1, KQ2, KQH;                  // 32-row k-blocks of layer 1 (K = D), layer 2 (K = H1), heads (K = H2)
    long long p_w1, p_w2, p_wh, p_w2t, p_wht, pack_elems;
};

__host__ __device__ inline Dims16 make_dims16(const NetDims &m) {
    Dims16 d{};
    d.KQ1 = (m.D + 31) / 32; d.KQ2 = (m.H1 + 31) / 32; d.KQH = (m.H2 + 31) / 32;
    long long q = 0;
    d.p_w1 = q;  q += (long long)d.KQ1 * 32 * m.H1;          // K = D,  C = H1
    d.p_w2 = q;  q += (long long)d.KQ2 * 32 * m.H2;          // K = H1, C = H2
    d.p_wh = q;  q += (long long)d.KQH * 32 * 16;            // K = H2, C = 1+A -> 16
    d.p_w2t = q; q += (long long)d.KQH * 32 * m.H1;          // K = H2, C = H1  (W2 transposed)
    d.p_wht = q; q += (long long)32 * m.H2;                  // K = 1+A -> 32, C = H2
    d.pack_elems = q;
    return d;
}

__device__ __forceinline__ void scatter_packs16(const NetDims &m, const Dims16 &d, int i, float v, __bf16 *pack) {
    const __bf16 b = (__bf16)v;
    const int o_b1 = (int)m.o_b1, o_w2 = (int)m.o_w2, o_b2 = (int)m.o_b2, o_wv = (int)m.o_wv, o_bv = (int)m.o_bv,
              o_wa = (int)m.o_wa, o_ba = (int)m.o_ba;
    if (i < o_b1) {
        const int k = i / m.H1, n = i - k * m.H1;
        pack[d.p_w1 + pidx16(d.KQ1, k, n)] = b;
    } else if (i >= o_w2 && i < o_b2) {
        const int u = i - o_w2, k = u / m.H2, n = u - k * m.H2;
        pack[d.p_w2 + pidx16(d.KQ2, k, n)] = b;
        pack[d.p_w2t + pidx16(d.KQH, n, k)] = b;
    } else if (i >= o_wv && i < o_bv) {
        const int k = i - o_wv;
        pack[d.p_wh + pidx16(d.KQH, k, 0)] = b;
        pack[d.p_wht + pidx16(1, 0, k)] = b;
    } else if (i >= o_wa && i < o_ba) {
        const int u = i - o_wa, k = u / m.A, a = u - k * m.A;
        pack[d.p_wh + pidx16(d.KQH, k, 1 + a)] = b;
        pack[d.p_wht + pidx16(1, 1 + a, k)] = b;
    }
}

