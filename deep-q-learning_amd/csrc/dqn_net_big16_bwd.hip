// csrc/dqn_net_big16_bwd.hip -- the UPDATE / GRADIENT form of the 64-row bf16 path (see dqn_net_big16.hip for the layouts and
// dqn_net_big16.h for the shared pieces): one workgroup takes a 64-row tile through online(s'), target(s'), online(s)
// (LunarLander/dddqn.py:24-31; q_learning_functions.py:52-54), the TD target / Huber gradient / dueling backward (:55-60,
// :35-36) and the row gradients dz2, dz1, and stashes x, h1, h2, dz1, dz2, dz3 k-packed for k_big_dw16. Every pass in the N
// orientation (lane = output column), ReLU gates as bit masks in two registers.
//
// r03 measurements that shaped what is NOT here (B = 2^17, rows kernel of the update): this kernel 250 us; with the forward
// kernel's packed epilogues + transposed non-stash passes + gates re-read from the stash 275 us; with ALL of W2^T and both gate
// sets requested ahead of the stash stores (stores count in vmcnt: the dz2 . W2^T product waits 11.7 K cycles for 2 K of MFMA
// behind the dz2 stash) 409 us -- the extra 200 live registers spill (184 scratch registers at the 512-register limit), and
// scratch traffic queues behind the same stores. The row backward wants the weight gradients fused into it (no stash at all);
// not built.
#include "dqn_net_big16.h"

template <bool X16>                      // obs_dim <= 16: layer 1 is one 16-deep k-step
__global__ void __launch_bounds__(256)
k_big_rows16_bwd(NetDims m, Dims16 d16, Big16Args g, int B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int A = m.A;
    const int ntiles = (B + 63) >> 6;
    __bf16 *lx = reinterpret_cast<__bf16 *>(smem);     // [64][SX]
    __bf16 *la = lx + 64 * SX;                         // [64][SA]   h1, then h2, then dz2
    __bf16 *l3 = la + 64 * SA;                         // [64][S3]   dz3
    float *lh = reinterpret_cast<float *>(l3 + 64 * S3);   // [64][16]   heads
    float *lq = lh + 64 * 16;                          // [3][64][16] Q rows of the passes
    float *lrow = lq + 3 * 64 * 16;                    // [64] per-row loss
    const int ct0 = 2 * wave;                          // this wave's two 32-column tiles

    // input rows of a pass: thread (g8 = tid >> 5, cc = tid & 31) takes column cc of the eight rows 8 g8 .. 8 g8 + 7 -- its LDS
    // image elements and ONE 16-byte piece of the k-packed x stash; requested one pass ahead
    float xv[8];
    auto x_request = [&](const Big16Pass &P, int rbase) {
        const int g8 = tid0 >> 5, cc = tid0 & 31;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = rbase + 8 * g8 + j;
            xv[j] = 0.0f;
            if (row < B && cc < m.D) xv[j] = P.x[(long long)row * m.D + cc];
        }
    };
    x_request(g.p[0], (int)blockIdx.x * 64);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * 64;
    unsigned long long m1 = 0ull, m2 = 0ull;           // ReLU gates of this lane's accumulator elements: bit (rt*2 + ct)*16 + r
    // the TD rule's per-row inputs (thread = row, tid < 64): asked for at the top of the tile, a whole forward pass before they are
    // used (the stamps had the TD block at 8.4 K cycles: four dependent global round trips per row)
    float td_wmax = 1.0f;
    int td_a = 0; float td_r = 0.0f, td_d = 0.0f, td_w = 1.0f, td_nq[4] = {0.f, 0.f, 0.f, 0.f}, td_nt[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.do_bwd && tid0 < 64 && row0 + tid0 < B) {
        const int i = row0 + tid0;
        td_a = g.g.a ? g.g.a[i] : 0;
        if (!g.g.targets) {
            td_r = g.g.r[i]; td_d = g.g.d_f32 ? g.g.d_f32[i] : (g.g.d_u8[i] ? 1.0f : 0.0f);      // preprocessing :84
            if (g.npass != 3 && A <= 4) for (int k2 = 0; k2 < A; ++k2) { td_nq[k2] = g.g.nq[(long long)i * A + k2]; td_nt[k2] = g.g.nt[(long long)i * A + k2]; }
        }
        td_w = g.g.w_raw ? g.g.w_raw[i] : (g.g.isw ? g.g.isw[i] : 1.0f);
        if (g.g.w_raw) td_wmax = g.st->wmax;
    }
    for (int ps = 0; ps < g.npass; ++ps) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));                  // opaque per pass (keeps the epilogue addresses out of the outer loops)
        const int lane = tid & 63, h = lane >> 5, c = lane & 31;
        const Big16Pass &P = g.p[ps];
        const bool last = ps == g.npass - 1;
        const bool stash = last && g.do_bwd;
        {
            const int g8 = tid >> 5, cc = tid & 31;
            bf16x8 xb;
#pragma unroll
            for (int j = 0; j < 8; ++j) { xb[j] = (__bf16)xv[j]; lx[(8 * g8 + j) * SX + cc] = xb[j]; }
            if (stash) *reinterpret_cast<bf16x8 *>(g.px + ((long long)((row0 >> 3) + g8) * 32 + cc) * 8) = xb;
        }
        float bias1[2], bias2[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) { bias1[ct] = P.params[m.o_b1 + 32 * (ct0 + ct) + c]; bias2[ct] = P.params[m.o_b2 + 32 * (ct0 + ct) + c]; }
        BigLayer16<1, 1, X16> L1; BigLayer16<HB / 32, 4, false> L2;
        L1.init(P.pack + d16.p_w1, ct0, lane); L1.prefetch();
        L2.init(P.pack + d16.p_w2, ct0, lane);
        L2.prefetch();                                 // layer 2's first k-blocks travel behind layer 1
        LDS_BARRIER();
        // ---- layer 1: h1 = relu(x @ w1 + b1)                                  dddqn.py:25-26
        {
            f32x16 acc[2][2];
            zero_acc16(acc);
            L1.run(lx, SX, lane, acc);
            if (!last) x_request(g.p[ps + 1], row0);
            else if (tile + (int)gridDim.x < ntiles) x_request(g.p[0], (tile + (int)gridDim.x) * 64);
            int rb8 = row0 >> 3;
            asm volatile("" : "+v"(rb8));
            auto epi1 = [&](auto stash_tag) {
                constexpr bool ST = decltype(stash_tag)::value;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const int col = 32 * (ct0 + ct) + c;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            bf16x4 pk;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int r = 4 * gq + u;
                                float v = acc[rt][ct][r] + bias1[ct];
                                v = v > 0.0f ? v : 0.0f;
                                pk[u] = (__bf16)v;
                                la[(32 * rt + row32b(r, h)) * SA + col] = pk[u];
                                if constexpr (ST) { if (v > 0.0f) m1 |= 1ull << ((rt * 2 + ct) * 16 + r); }
                            }
                            if constexpr (ST) *reinterpret_cast<bf16x4 *>(g.ph1 + ((long long)(rb8 + 4 * rt + gq) * HB + col) * 8 + 4 * h) = pk;
                        }
                    }
            };
            if (stash) epi1(std::true_type{}); else epi1(std::false_type{});
        }
        LDS_BARRIER();
        // ---- layer 2: h2 = relu(h1 @ w2 + b2)                                 dddqn.py:27-28
        bf16x8 wv[HB / 32];                            // the heads' weights (requested behind layer 2's MFMAs)
        {
            f32x16 acc[2][2];
            zero_acc16(acc);
            L2.run(la, SA, lane, acc);
            {
                const bf16x8 *wh = reinterpret_cast<const bf16x8 *>(P.pack + d16.p_wh) + lane;
#pragma unroll
                for (int kq = 0; kq < HB / 32; ++kq) wv[kq] = wh[kq * 64];
            }
            LDS_BARRIER();                             // every wave has read all of h1: h2 may replace it
            int rb8 = row0 >> 3, rb = row0;
            asm volatile("" : "+v"(rb8), "+v"(rb));
            auto epi2 = [&](auto stash_tag, auto feat_tag) {
                constexpr bool ST = decltype(stash_tag)::value, FT = decltype(feat_tag)::value;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const int col = 32 * (ct0 + ct) + c;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            bf16x4 pk;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int r = 4 * gq + u, rl = 32 * rt + row32b(r, h);
                                float v = acc[rt][ct][r] + bias2[ct];
                                v = v > 0.0f ? v : 0.0f;
                                pk[u] = (__bf16)v;
                                la[rl * SA + col] = pk[u];
                                if constexpr (FT) { if (rb + rl < B) g.feat[(long long)(rb + rl) * HB + col] = v; }   // :32-33
                                if constexpr (ST) { if (v > 0.0f) m2 |= 1ull << ((rt * 2 + ct) * 16 + r); }
                            }
                            if constexpr (ST) *reinterpret_cast<bf16x4 *>(g.ph2 + ((long long)(rb8 + 4 * rt + gq) * HB + col) * 8 + 4 * h) = pk;
                        }
                    }
            };
            const bool feat_on = last && g.feat != nullptr;
            if (stash) { if (feat_on) epi2(std::true_type{}, std::true_type{}); else epi2(std::true_type{}, std::false_type{}); }
            else { if (feat_on) epi2(std::false_type{}, std::true_type{}); else epi2(std::false_type{}, std::false_type{}); }
        }
        LDS_BARRIER();
        // ---- heads (dddqn.py:29-30): wave w takes rows 16w .. 16w+15 on 16x16x32 (A: lane (row l&15, k = 8(l>>4) + j))
        {
            const int kg = lane >> 4, r16 = lane & 15;
            f32x4 hc = {0.f, 0.f, 0.f, 0.f};
            const __bf16 *arow = la + (16 * wave + r16) * SA + 8 * kg;
            float biash = 0.0f;
            if (r16 == 0) biash = P.params[m.o_bv]; else if (r16 <= A) biash = P.params[m.o_ba + r16 - 1];
#pragma unroll
            for (int kq = 0; kq < HB / 32; ++kq) hc = MFMA16B(*reinterpret_cast<const bf16x8 *>(arow + 32 * kq), wv[kq], hc);
#pragma unroll
            for (int r = 0; r < 4; ++r) lh[(16 * wave + 4 * kg + r) * 16 + r16] = hc[r] + biash;      // C/D: col l&15, row 4(l>>4) + r
        }
        LDS_BARRIER();
        // ---- Q = val + adv - mean(adv)                                        dddqn.py:31
        if (tid < 64) {
            const float *hr = lh + tid * 16;
            float sum = 0.0f;
            for (int a = 0; a < A; ++a) sum = sum + hr[1 + a];
            const float mean = __fdiv_rn(sum, (float)A);
            for (int a = 0; a < A; ++a) {
                const float qv = (hr[0] + hr[1 + a]) - mean;
                lq[(ps * 64 + tid) * 16 + a] = qv;
                if (P.q && row0 + tid < B) P.q[(long long)(row0 + tid) * A + a] = qv;
            }
        }
        LDS_BARRIER();
    }
    if (!g.do_bwd) continue;
    STAMP(1, 12);
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, h = lane >> 5, c = lane & 31;

    // ---- TD target / Huber gradient / dueling backward of this tile's rows (the row arithmetic of k_bwd_rows, f32)
    const BwdArgs &bw = g.g;
    const __bf16 *bpack = reinterpret_cast<const __bf16 *>(bw.pack);
    const int pl = g.npass - 1;                        // pass that produced q = pred
#ifndef BIG16_BWD_W2T_PF
#define BIG16_BWD_W2T_PF 4              // (8 = ALL of W2^T in registers, requested before the tile's stash stores: measured equal -- 164 vs 161 us
#endif                                  //  at B = 2^17, 458 vs 442 registers -- so the refills behind the dz2 stash are not what the product waits for)
    BigLayer16<1, 1, true> LA; BigLayer16<HB / 32, BIG16_BWD_W2T_PF, false> LB;
    LA.init(bpack + d16.p_wht, ct0, lane); LA.prefetch();
    LB.init(bpack + d16.p_w2t, ct0, lane);
    if (BIG16_BWD_W2T_PF == HB / 32) LB.prefetch();
#pragma unroll
    for (int u = 0; u < (64 * S3 / 2) / 256; ++u) reinterpret_cast<unsigned *>(l3)[tid + 256 * u] = 0u;      // (no runtime loop: no vmcnt(0) in front of it)
    LDS_BARRIER();
    STAMP(1, 13);
    // the TD rule, one row per thread of wave 0; AM = the unrolled action bound (A <= 4 is the common case: four predicated
    // steps instead of sixteen -- the stamps had this block at 7 K cycles of select chains on the 16-wide form)
    auto td_block = [&](auto am_c) {
        constexpr int AM = decltype(am_c)::value;
        const int i = row0 + tid;
        float rowloss = 0.0f;
        if (i < B) {
            float qr[AM], tr[AM], nqr[AM], ntr[AM];
#pragma unroll
            for (int k2 = 0; k2 < AM; ++k2) qr[k2] = k2 < A ? lq[(pl * 64 + tid) * 16 + k2] : 0.0f;
            const int ai = td_a;
            if (bw.targets) {
#pragma unroll
                for (int k2 = 0; k2 < AM; ++k2) tr[k2] = k2 < A ? bw.targets[(long long)i * A + k2] : 0.0f;
            } else {
#pragma unroll
                for (int k2 = 0; k2 < AM; ++k2) {
                    nqr[k2] = ntr[k2] = 0.0f;
                    if (k2 < A) {
                        if (g.npass == 3) { nqr[k2] = lq[(0 * 64 + tid) * 16 + k2]; ntr[k2] = lq[(1 * 64 + tid) * 16 + k2]; }
                        else if (AM <= 4) { nqr[k2] = td_nq[k2 & 3]; ntr[k2] = td_nt[k2 & 3]; }              // (k_big_fwd16 wrote them)
                        else { nqr[k2] = bw.nq[(long long)i * A + k2]; ntr[k2] = bw.nt[(long long)i * A + k2]; }
                    }
                }
                // q_learning_functions.py:55-60 (td_row of dqn_net_common.h, unrolled: first max wins, quirks Q3 / Q4)
                float best = nqr[0], ntb = ntr[0], qa = qr[0];
#pragma unroll
                for (int k2 = 1; k2 < AM; ++k2) {
                    if (k2 < A && nqr[k2] > best) { best = nqr[k2]; ntb = ntr[k2]; }
                    if (k2 == ai) qa = qr[k2];
                }
                const float t1 = bw.gamma * ntb;
                const float t2 = t1 - qa;
                const float t3 = (1.0f - td_d) * t2;
                const float delta = td_r + t3;
#pragma unroll
                for (int k2 = 0; k2 < AM; ++k2) tr[k2] = qr[k2] + delta * (k2 == ai ? 1.0f : 0.0f);
                if (bw.td) bw.td[i] = delta;
                if (bw.td_abs) bw.td_abs[i] = fabsf(delta);
            }
            float w = 1.0f;
            if (bw.w_raw) { w = __fdiv_rn(td_w, td_wmax); if (bw.isw_out) bw.isw_out[i] = w; }
            else if (bw.isw) w = td_w;
            const float invB = __fdiv_rn(1.0f, (float)B);
            float gk[AM], gsum = 0.0f;
#pragma unroll
            for (int k2 = 0; k2 < AM; ++k2) {
                gk[k2] = 0.0f;
                if (k2 < A) {
                    const float e = qr[k2] - tr[k2];                   // pred - target, pred == q   (:35)
                    rowloss = rowloss + huber(e);                      // :36
                    const float cc = e > 1.0f ? 1.0f : (e < -1.0f ? -1.0f : e);
                    gk[k2] = (w * cc) * invB;                          // dL/dpred
                    gsum = gsum + gk[k2];
                    if (bw.dq) bw.dq[(long long)i * A + k2] = gk[k2];
                    if (bw.targets_out) bw.targets_out[(long long)i * A + k2] = tr[k2];
                }
            }
            if (bw.w_raw || bw.isw) rowloss = w * rowloss;
            const float gmean = __fdiv_rn(gsum, (float)A);             // dueling backward: dv = sum g ; dadv = g - mean g
            l3[tid * S3 + 0] = (__bf16)gsum;
#pragma unroll
            for (int k2 = 0; k2 < AM; ++k2) if (k2 < A) l3[tid * S3 + 1 + k2] = (__bf16)(gk[k2] - gmean);
        }
        lrow[tid] = rowloss;
    };
    if (tid < 64) { if (A <= 4) td_block(std::integral_constant<int, 4>{}); else td_block(std::integral_constant<int, 16>{}); }
    LDS_BARRIER();
    // per-tile loss (16-row sub-tiles in order, as the 16-row kernels; wave 1) beside the column sums of dz3 (the values the weight
    // gradient sees; wave 0: lane (col, quarter) adds 16 rows in order, quarters fold by two exchanges) and dz3 out (waves 2, 3)
    float *cs = g.colsum + (long long)tile * (2 * HB + 16);
    if (tid >= 64 && tid < 68) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s = s + lrow[16 * (tid - 64) + k];
        bw.loss_part[4 * tile + tid - 64] = s;
    }
    if (tid < 64) {
        const int cc = tid & 15, qt = tid >> 4;
        float s = 0.0f;
#pragma unroll
        for (int rl = 0; rl < 16; ++rl) s = s + (float)l3[(16 * qt + rl) * S3 + cc];
        s = s + __shfl_xor(s, 16, 64);
        s = s + __shfl_xor(s, 32, 64);
        if (tid < 16) cs[2 * HB + tid] = s;
    }
    if (tid >= 128) {                                  // dz3 out, k-packed [B/8][16][8]: thread (g8, cc) one 16-byte piece
        const int g8 = (tid - 128) >> 4, cc = tid & 15;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = l3[(8 * g8 + j) * S3 + cc];
        *reinterpret_cast<bf16x8 *>(reinterpret_cast<__bf16 *>(bw.pdz3) + ((long long)((row0 >> 3) + g8) * 16 + cc) * 8) = v;
    }
    STAMP(1, 15);
    auto colsum64 = [&](const f32x16 &x0, const f32x16 &x1) -> float {
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s = s + x0[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) s = s + x1[r];
        return s + __shfl_xor(s, 32, 64);
    };
    __bf16 *pdz2 = reinterpret_cast<__bf16 *>(bw.pdz2), *pdz1 = reinterpret_cast<__bf16 *>(bw.pdz1);
    // ---- dz2 = (dz3 . WH^T) * (h2 > 0)
    {
        f32x16 acc[2][2];
        zero_acc16(acc);
        if (BIG16_BWD_W2T_PF != HB / 32) LB.prefetch();                                 // W2^T's first k-blocks travel behind dz2
        STAMP(1, 16);
        LA.run(l3, S3, lane, acc);
        LDS_BARRIER();                                 // (heads / Q are long done with h2: dz2 replaces it)
        STAMP(1, 17);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = 32 * (ct0 + ct) + c;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    bf16x4 pk;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = 4 * gq + u;
                        const float v = ((m2 >> ((rt * 2 + ct) * 16 + r)) & 1ull) ? acc[rt][ct][r] : 0.0f;
                        acc[rt][ct][r] = v;
                        pk[u] = (__bf16)v;
                        la[(32 * rt + row32b(r, h)) * SA + col] = pk[u];
                    }
                    *reinterpret_cast<bf16x4 *>(pdz2 + ((long long)((row0 >> 3) + 4 * rt + gq) * HB + col) * 8 + 4 * h) = pk;
                }
            }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float s = colsum64(acc[0][ct], acc[1][ct]);
            if (h == 0) cs[HB + 32 * (ct0 + ct) + c] = s;
        }
        STAMP(1, 18);
    }
    LDS_BARRIER();
    STAMP(1, 19);
    // ---- dz1 = (dz2 . W2^T) * (h1 > 0)
    {
        f32x16 acc[2][2];
        zero_acc16(acc);
        LB.run(la, SA, lane, acc);
        STAMP(1, 20);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = 32 * (ct0 + ct) + c;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    bf16x4 pk;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = 4 * gq + u;
                        const float v = ((m1 >> ((rt * 2 + ct) * 16 + r)) & 1ull) ? acc[rt][ct][r] : 0.0f;
                        acc[rt][ct][r] = v;
                        pk[u] = (__bf16)v;
                    }
                    *reinterpret_cast<bf16x4 *>(pdz1 + ((long long)((row0 >> 3) + 4 * rt + gq) * HB + col) * 8 + 4 * h) = pk;
                }
            }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float s = colsum64(acc[0][ct], acc[1][ct]);
            if (h == 0) cs[32 * (ct0 + ct) + c] = s;
        }
        STAMP(1, 31);
    }
    LDS_BARRIER();                                     // the next tile's first pass rewrites the images
    }   // row tiles
}


static size_t big16_bwd_lds() {
    return 2 * (size_t)(64 * SX + 64 * SA + 64 * S3) + 4 * (size_t)(64 * 16 + 3 * 64 * 16 + 64);
}

// passes: update form = {online(s'), target(s'), online(s)}; grads form (bw.targets given) = {online(s)}
void launch_big16_rows_bwd(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, const BwdArgs &bw,
                           float *px, float *ph1, float *ph2, float *colsum, DqnState *st, int num_cus) {
    Big16Args g{};
    g.npass = npass;
    for (int i = 0; i < npass; ++i) g.p[i] = Big16Pass{passes[i].x, passes[i].params, reinterpret_cast<const __bf16 *>(passes[i].pack), passes[i].q};
    g.do_bwd = 1; g.g = bw;
    g.px = reinterpret_cast<__bf16 *>(px); g.ph1 = reinterpret_cast<__bf16 *>(ph1); g.ph2 = reinterpret_cast<__bf16 *>(ph2);
    g.colsum = colsum; g.st = st;
    const int tiles = (B + 63) / 64;
    const int grid = tiles < num_cus ? tiles : num_cus;
    const Dims16 d = make_dims16(m);
    if (m.D <= 16) DQN_LAUNCH((k_big_rows16_bwd<true>), dim3(grid), dim3(256), big16_bwd_lds(), s, m, d, g, B);
    else DQN_LAUNCH((k_big_rows16_bwd<false>), dim3(grid), dim3(256), big16_bwd_lds(), s, m, d, g, B);
}
