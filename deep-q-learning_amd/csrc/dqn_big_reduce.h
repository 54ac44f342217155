// csrc/dqn_big_reduce.h -- the reduction / optimizer kernel of the large-batch paths (f32: dqn_net_big.hip, bf16:
// dqn_net_big16.hip): sums the split-K slab partials and the per-row-tile column sums in a fixed order, applies Adam / AdamW
// and refreshes the weight shadows of the precision mode (BF = false: f32 fragment packs; true: bf16 packs + actor shadows).
#pragma once
#include "dqn_device.h"
#include "dqn_launch.h"
#include "dqn_net_common.h"
#include "dqn_bf16_pack.h"

constexpr int BIG_H = 256;              // hidden width of the large-batch paths
constexpr int BIG_DW_TILES = 8;         // slab slots per batch slice

// Gradient element i of the flat parameter vector and its optimizer step (as k_dw's epilogue / k_adam):
//   weight blocks (blockIdx.x < wblocks): one thread per element, the sum of its slab partials over the batch slices, in
//     slice order;
//   bias blocks: one WAVE per bias element, the sum of the per-row-tile column sums -- lane l adds tiles l, l + 64, ... in
//     order, then a fixed shuffle tree (there can be thousands of row tiles: a serial loop per element would be the launch).
template <bool BF>
__device__ __forceinline__ void big_apply(const NetDims &m, int i, float gsum, float *grad, const AdamArgs &ad, const AdamCoef &co) {
    grad[i] = gsum;
    if (!ad.P) return;
    const float pnew = adam_elem(co, gsum, ad.P, ad.mu, ad.nu, i, ad.adamw, ad.b1, ad.b2, ad.eps, ad.wd, ad.grad_scale);
    if constexpr (BF) {                                                // bf16 mode: the bf16 packs + the actor's f32 shadows
        scatter_packs16(m, make_dims16(m), i, pnew, reinterpret_cast<__bf16 *>(ad.pack));
        if (ad.pack_act) scatter_actor_packs(m, i, pnew, ad.pack_act);
        return;
    }
    // the fragment-packed shadows of this element (dqn_net.hip: scatter_packs)
    const int o_b1 = (int)m.o_b1, o_w2 = (int)m.o_w2, o_b2 = (int)m.o_b2, o_wv = (int)m.o_wv, o_bv = (int)m.o_bv, o_wa = (int)m.o_wa, o_ba = (int)m.o_ba;
    if (i < o_b1) { const int k = i / m.H1, n = i - k * m.H1; ad.pack[m.p_w1 + pidx(m.KQ1, k, n)] = pnew; }
    else if (i >= o_w2 && i < o_b2) {
        const int u = i - o_w2, k = u / m.H2, n = u - k * m.H2;
        ad.pack[m.p_w2 + pidx(m.H1 / 16, k, n)] = pnew;
        ad.pack[m.p_w2t + pidx(m.H2 / 16, n, k)] = pnew;
        ad.pack[m.p_w2k + ((long long)(k >> 2) * m.H2 + n) * 4 + (k & 3)] = pnew;
    } else if (i >= o_wv && i < o_bv) {
        const int k = i - o_wv;
        ad.pack[m.p_wh + pidx(m.H2 / 16, k, 0)] = pnew;
        ad.pack[m.p_wht + pidx(1, 0, k)] = pnew;
    } else if (i >= o_wa && i < o_ba) {
        const int u = i - o_wa, k = u / m.A, a = u - k * m.A;
        ad.pack[m.p_wh + pidx(m.H2 / 16, k, 1 + a)] = pnew;
        ad.pack[m.p_wht + pidx(1, 1 + a, k)] = pnew;
    }
}

template <bool BF>
__global__ void __launch_bounds__(256)
k_big_reduce(NetDims m, const float *__restrict__ slab, int KS, const float *__restrict__ colsum, int row_tiles, int B,
             float *grad, const float *__restrict__ loss_part, float *loss_out, DqnState *st, int bump_ctr, AdamArgs ad, int wblocks) {
    const int nP = (int)m.P;
    double b1pow = 0.0, b2pow = 0.0;
    AdamCoef co{};
    if (ad.P) co = adam_coef(st, ad.b1, ad.b2, &b1pow, &b2pow);
    if ((int)blockIdx.x < wblocks) {
        // FOUR lanes per weight element (r03): lane `part` sums a contiguous quarter of the element's partials -- (slice, half) order,
        // all of its loads in flight together -- and the quarters are combined (p0 + p1) + (p2 + p3) by shuffles: a fixed order, so
        // the result is deterministic. One thread per element walked its 32-64 partials in four dependent batches with ~270
        // workgroups on the machine: 44 us for a 33 MB slab (0.7 TB/s, latency-bound) at B = 2^17.
        const int t = blockIdx.x * 256 + threadIdx.x, i = t >> 2, part = t & 3;
        // slab address of the element inside a slot, slot of the first partial, partials per slice (light tiles: two halves)
        long long off = -1; int nh = 1;
        if (i < (int)m.o_b1)      { const int k = i / m.H1, n = i - k * m.H1; off = 4ll * 128 * 128 + k * 256 + n; nh = 2; }
        else if (i >= (int)m.o_w2 && i < (int)m.o_b2) { const int u = i - (int)m.o_w2, k = u / m.H2, n = u - k * m.H2; off = (long long)(2 * (k >> 7) + (n >> 7)) * 128 * 128 + (k & 127) * 128 + (n & 127); }
        else if (i >= (int)m.o_wv && i < (int)m.o_bv) { const int k = i - (int)m.o_wv; off = 6ll * 128 * 128 + k * 32; nh = 2; }
        else if (i >= (int)m.o_wa && i < (int)m.o_ba) { const int u = i - (int)m.o_wa, k = u / m.A, a = u - k * m.A; off = 6ll * 128 * 128 + k * 32 + 1 + a; nh = 2; }
        const bool live = i < nP && off >= 0;
        float psum = 0.0f;
        if (live) {
            const float *p = slab + off;
            const long long ss = (long long)BIG_DW_TILES * 128 * 128, hs = 128 * 128;
            const int NPART = nh * KS, per = (NPART + 3) >> 2;
            int q = part * per;
            const int q1 = q + per < NPART ? q + per : NPART;
            auto addr = [&](int qq) { return nh == 1 ? p + qq * ss : p + (qq >> 1) * ss + (qq & 1) * hs; };
            for (; q + 8 <= q1; q += 8) {                          // eight partials in flight, added in order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *addr(q + u);
#pragma unroll
                for (int u = 0; u < 8; ++u) psum = psum + v[u];
            }
            for (; q < q1; ++q) psum = psum + *addr(q);
        }
        // (every lane takes part in the shuffles: the four lanes of an element are live or dead together)
        const float s01 = psum + __shfl_down(psum, 1, 64);
        const float gsum = s01 + __shfl_down(s01, 2, 64);
        if (live && part == 0) big_apply<BF>(m, i, gsum, grad, ad, co);
    } else {
        const int lane = threadIdx.x & 63;
        const int e = ((int)blockIdx.x - wblocks) * 4 + (threadIdx.x >> 6);          // bias element: b1 | b2 | bv, ba
        if (e < 2 * BIG_H + 1 + m.A) {
            const int i = e < BIG_H ? (int)m.o_b1 + e : (e < 2 * BIG_H ? (int)m.o_b2 + (e - BIG_H) : (e == 2 * BIG_H ? (int)m.o_bv : (int)m.o_ba + (e - 2 * BIG_H - 1)));
            float sacc = 0.0f;
            const float *pc = colsum + e;
            const long long cs = 2 * BIG_H + 16;
            int q = lane;
            for (; q + 7 * 64 < row_tiles; q += 8 * 64) {                // eight column sums in flight, added in tile order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = pc[(q + 64 * u) * cs];
#pragma unroll
                for (int u = 0; u < 8; ++u) sacc = sacc + v[u];
            }
            for (; q < row_tiles; q += 64) sacc = sacc + pc[q * cs];
            for (int o = 32; o > 0; o >>= 1) sacc = sacc + __shfl_xor(sacc, o, 64);
            if (lane == 0) big_apply<BF>(m, i, sacc, grad, ad, co);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        // loss = (sum of the per-16-row partial sums) / B: lane l adds entries l, l + 64, ... (eight loads in flight), then a
        // fixed shuffle tree
        const int lane = threadIdx.x, KQb = (B + 15) / 16;
        float s = 0.0f;
        int q = lane;
        for (; q + 7 * 64 < KQb; q += 8 * 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = loss_part[q + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = s + v[u];
        }
        for (; q < KQb; q += 64) s = s + loss_part[q];
        for (int o = 32; o > 0; o >>= 1) s = s + __shfl_xor(s, o, 64);
        if (lane == 0) {
            float Lv = __fdiv_rn(s, (float)B);
            if (st->err_count != 0u) Lv = __int_as_float(0x7fc00000);
            st->loss = Lv;
            if (loss_out) *loss_out = Lv;
            if (bump_ctr) { st->sample_ctr += 1ull; st->wmax = 0.0f; }
        }
    }
    if (ad.P) {
        // commit the optimizer counters once every block has read them (its stores above depend on the coefficients)
        LDS_BARRIER();
        if (threadIdx.x == 0) {
            const unsigned int ticket = atomicAdd(&st->arrive, 1u);
            if (ticket == gridDim.x - 1) { st->b1pow = b1pow; st->b2pow = b2pow; st->adam_count += 1; st->arrive = 0; }
        }
    }
}

