// csrc/dqn_net_big.hip -- the Q-network kernels for LARGE batches (B >= DQN_BIG_MIN rows, exact-f32 path) on
// v_mfma_f32_32x32x2_f32: 64 batch rows per workgroup instead of 16, so the 277 KB weight set is streamed once per 64 rows,
// each B fragment feeds two MFMAs of 64 cycles, and a workgroup keeps its rows ON CHIP from the sampled input to the row
// gradients:
//
//   k_big_rows   one workgroup = 64 rows of the batch. Update form: online(s') -> next_q, target(s') -> next_q_tm, online(s)
//                -> q (LunarLander/dddqn.py:24-31 three times, q_learning_functions.py:52-54), TD target / Huber gradient /
//                dueling backward (:55-60, :35-36), dz2, dz1 -- h1 / h2 / dz2 never leave LDS between the layers; what the
//                weight-gradient kernel needs (x, h1, h2, dz1, dz2, dz3) goes to HBM once, row-major. Forward form: 1..3
//                passes, Q (+ features) out.
//   k_big_dw     dW = H^T Z over the batch as a split-K GEMM: 128 x 128 output tiles (four 64 x 64 wave tiles), the batch
//                cut into KS slices; partial tiles go to a slab.
//   k_big_reduce sums the slices in slice order (deterministic), adds the bias gradients (per-row-tile column sums), applies
//                Adam / AdamW and refreshes the packed shadows (same element arithmetic as k_dw's epilogue).
//
// Arithmetic: every forward / row-backward dot product is the same k-ascending fmaf chain as in dqn_net.hip (an MFMA
// 32x32x2 adds k = 2s then k = 2s + 1; the heads keep their four interleaved chains on 16x16x4), so Q, targets and the row
// gradients are bit-identical to the 16-row kernels and to the CPU restatement; weight gradients differ in summation
// order only (batch slices), as k_dw's four wave slices already do.
//
// Weights are read from the SAME fragment-packed shadows (dqn_net.hip: packed(M)[(ct*KQ + kq)*256 + lane*4 + j] =
// M[16kq + 4j + (lane>>4)][16ct + (lane&15)]): for a 32-column tile, lane (h = lane>>5, c = lane&31) takes the two
// float4 of 16-column tile 2*ct32 + (c>>4) at lane slots g = h and g = h + 2: elements j of the first are k = 16kq + 4j + h
// (= 2s + h for s = 2j), of the second k = 16kq + 4j + 2 + h (s = 2j + 1) -- eight k-steps from two 16-byte loads.
//
// Only for hidden1 == hidden2 == 256, obs_dim <= 32, f32 (BASELINE configs[1] net); other shapes keep the 16-row kernels.
#include <type_traits>
#include "dqn_device.h"
#include "dqn_launch.h"
#include "dqn_net_common.h"
#include "dqn_big_reduce.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// position of k-in-block c (0..15) inside its 16-block of an LDS A-operand row: the lane half h = lane>>5 reads 8
// consecutive floats = k-steps s = 0..7 of k = 2s + h
__device__ __forceinline__ int perm2(int c) { return ((c & 1) << 3) | (c >> 1); }
// row of accumulator register r of a 32x32 tile for lane half h
__device__ __forceinline__ int row32(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

constexpr int BH = 256;                 // hidden width of the big path
constexpr int BS = BH + 4;              // LDS row stride of the activation image (== 4 mod 64: conflict-free ds_read_b128)
constexpr int BSX = 36;                 // LDS row stride of the input image (K1 <= 32)
constexpr int BS3 = 20;                 // LDS row stride of dz3 (K = 16)

// acc[rt][ct] (+)= A . W for the wave's 64 rows (rt = 0, 1) and two 32-column tiles ct32 = ct0, ct0 + 1.
// A: LDS image [64][S], perm2 order inside 16-blocks; W: fragment-packed, KQ 16-row k-blocks, as float4.
// The packed-weight loads run PF k-blocks ahead in a register ring (slot kq % PF is refilled right after it is consumed).
// Everything is straight-line code (KQ is a template parameter): inside a runtime loop the compiler's wait insertion
// drains the whole ring (s_waitcnt vmcnt(0)) at every use. prefetch() may be called long before run() -- before the
// barrier / epilogue in front of the layer -- so that the first blocks are on chip when the MFMAs start.
template <int KQ, int PF>
struct BigLayer {
    const float4 *pb[2];
    float4 blo[PF][2], bhi[PF][2];
    __device__ __forceinline__ void init(const float4 *wp, int ct0, int lane) {
        const int h = lane >> 5, c = lane & 31;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) pb[ct] = wp + (long long)(2 * (ct0 + ct) + (c >> 4)) * KQ * 64 + h * 16 + (c & 15);
    }
    __device__ __forceinline__ void prefetch() {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int kq = p < KQ ? p : KQ - 1;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) { blo[p][ct] = pb[ct][kq * 64]; bhi[p][ct] = pb[ct][kq * 64 + 32]; }
        }
    }
    __device__ __forceinline__ void run(const float *la, int S, int lane, f32x16 (&acc)[2][2]) {
        const int h = lane >> 5, c = lane & 31;
        const float *arow0 = la + c * S + 8 * h, *arow1 = arow0 + 32 * S;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const int p = kq % PF;
            const float4 a0l = *reinterpret_cast<const float4 *>(arow0 + 16 * kq), a0h = *reinterpret_cast<const float4 *>(arow0 + 16 * kq + 4);
            const float4 a1l = *reinterpret_cast<const float4 *>(arow1 + 16 * kq), a1h = *reinterpret_cast<const float4 *>(arow1 + 16 * kq + 4);
            const float a0[8] = {a0l.x, a0l.y, a0l.z, a0l.w, a0h.x, a0h.y, a0h.z, a0h.w};
            const float a1[8] = {a1l.x, a1l.y, a1l.z, a1l.w, a1h.x, a1h.y, a1h.z, a1h.w};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const float4 bb = (s & 1) ? bhi[p][ct] : blo[p][ct];
                    const float b = (s >> 1) == 0 ? bb.x : ((s >> 1) == 1 ? bb.y : ((s >> 1) == 2 ? bb.z : bb.w));
                    acc[0][ct] = MFMA32(a0[s], b, acc[0][ct]);
                    acc[1][ct] = MFMA32(a1[s], b, acc[1][ct]);
                }
            }
            if (kq + PF < KQ) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) { blo[p][ct] = pb[ct][(kq + PF) * 64]; bhi[p][ct] = pb[ct][(kq + PF) * 64 + 32]; }
            }
            __builtin_amdgcn_sched_barrier(0);                         // keep the k-blocks apart: hoisting every LDS read of the layer costs 256 registers
        }
    }
};

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
}

struct BigPass {
    const float *x;            // [B][D] rows of this pass
    const float *params;       // flat params (biases)
    const float *pack;         // fragment-packed weights
    float *q;                  // [B][A] out or NULL
};

struct BigArgs {
    BigPass p[3]; int npass;   // passes in order; in update / grads form the LAST pass is online(s)
    float *feat;               // [B][H2] features of the last pass (dddqn.py:32-33) or NULL
    int do_bwd;                // 0: forward only
    BwdArgs g;                 // TD inputs / outputs (q, nq, nt unused: they stay in LDS); pdz1/2/3, loss_part
    float *px, *ph1, *ph2;     // row-major stashes [B][K1], [B][H], [B][H]
    float *colsum;             // [tiles][2*BH + 16]: per-tile column sums of dz1 | dz2 | dz3 (bias gradients)
    DqnState *st;
};

template <int KQ1>                       // 16-row k-blocks of layer 1: obs_dim <= 16 -> 1, <= 32 -> 2
__global__ void __launch_bounds__(256)
k_big_rows(NetDims m, BigArgs g, int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    constexpr int K1 = KQ1 * 16, k1s = KQ1 == 1 ? 4 : 5, XV = 4 * KQ1;  // K1 = 16 or 32 columns of the staged input
    const int A = m.A;
    const int ntiles = (B + 63) >> 6;
    float *lx = lds;                                   // [64][BSX]
    float *la = lx + 64 * BSX;                         // [64][BS]   h1, then h2, then dz2
    float *lh = la + 64 * BS;                          // [64][16]   heads
    float *lq = lh + 64 * 16;                          // [3][64][16] Q rows of the passes
    float *l3 = lq + 3 * 64 * 16;                      // [64][BS3]  dz3
    float *lrow = l3 + 64 * BS3;                       // [64] per-row loss
    unsigned *lm1 = reinterpret_cast<unsigned *>(lrow + 64);   // [64][8] ReLU gates of h1 (bit = column within the 32-tile)
    unsigned *lm2 = lm1 + 64 * 8;                      // [64][8] of h2
    const int ct0 = 2 * wave;                          // this wave's two 32-column tiles

    // input rows of a pass: requested into registers one pass ahead (all loads of a thread in flight together), written to
    // the LDS image when the pass starts
    float xv[XV];
    auto x_request = [&](const BigPass &P, int rbase) {
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int t = tid0 + 256 * u;
            const int rl = t >> k1s, cc = t & (K1 - 1);
            xv[u] = 0.0f;
            if (rbase + rl < B && cc < m.D) xv[u] = P.x[(long long)(rbase + rl) * m.D + cc];
        }
    };
    STAMP(1, 0);
    // persistent over the row tiles (grid <= tiles): a workgroup requests its next tile's input rows while the current
    // tile is in its last layers, and pays the launch / argument / first-weights latency once
    x_request(g.p[0], (int)blockIdx.x * 64);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * 64;
    for (int ps = 0; ps < g.npass; ++ps) {
        // (the thread index is made opaque per pass: otherwise every address of the unrolled epilogues -- hundreds of values
        // that depend on it alone -- is hoisted out of the pass / tile loops and spilled to scratch)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, h = lane >> 5, c = lane & 31;
        const BigPass &P = g.p[ps];
        const bool last = ps == g.npass - 1;
        const bool stash = last && g.do_bwd;
        // ---- stage the 64 input rows (zero-padded to K1 columns, rows >= B zero) in A-operand order
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int t = tid + 256 * u;
            const int rl = t >> k1s, cc = t & (K1 - 1);
            lx[rl * BSX + (cc & ~15) + perm2(cc & 15)] = xv[u];
            if (stash) g.px[(long long)(row0 + rl) * K1 + cc] = xv[u];
        }
        float bias1[2], bias2[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) { bias1[ct] = P.params[m.o_b1 + 32 * (ct0 + ct) + c]; bias2[ct] = P.params[m.o_b2 + 32 * (ct0 + ct) + c]; }
        BigLayer<KQ1, KQ1> L1; BigLayer<BH / 16, 4> L2;
        L1.init(reinterpret_cast<const float4 *>(P.pack + m.p_w1), ct0, lane); L1.prefetch();
        L2.init(reinterpret_cast<const float4 *>(P.pack + m.p_w2), ct0, lane);
        L2.prefetch();                                                 // layer 2's first k-blocks travel behind layer 1
        LDS_BARRIER();
        if (ps == 0) STAMP(1, 1);
        // ---- layer 1: h1 = relu(x @ w1 + b1)                                  dddqn.py:25-26
        {
            f32x16 acc[2][2];
            zero_acc(acc);
            L1.run(lx, BSX, lane, acc);
            if (ps == 0) STAMP(1, 2);
            // the next pass's rows (or the next tile's first pass) travel behind layer 2 (the x image is rewritten at the top of
            // the next pass, behind two barriers)
            if (!last) x_request(g.p[ps + 1], row0);
            else if (tile + (int)gridDim.x < ntiles) x_request(g.p[0], (tile + (int)gridDim.x) * 64);
            int rb = row0;                                             // opaque per pass: keeps the 64 store addresses from being
            asm volatile("" : "+v"(rb));                               // hoisted out of the pass loop (and spilled)
            auto epi1 = [&](auto stash_tag) {                          // (the stash flag is compile-time inside: no branch per element)
                constexpr bool ST = decltype(stash_tag)::value;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const int col = 32 * (ct0 + ct) + c;
                        float *dst = la + (col & ~15) + perm2(col & 15);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = 32 * rt + row32(r, h);
                            float v = acc[rt][ct][r] + bias1[ct];
                            v = v > 0.0f ? v : 0.0f;
                            dst[rl * BS] = v;
                            if constexpr (ST) {
                                g.ph1[(long long)(rb + rl) * BH + col] = v;
                                const unsigned long long bits = __ballot(v > 0.0f);
                                if (c == 0) lm1[rl * 8 + ct0 + ct] = (unsigned)(bits >> (32 * h));
                            }
                        }
                    }
            };
            if (stash) epi1(std::true_type{}); else epi1(std::false_type{});
        }
        if (ps == 0) STAMP(1, 3);
        LDS_BARRIER();
        if (ps == 0) STAMP(1, 4);
        // ---- layer 2: h2 = relu(h1 @ w2 + b2)                                 dddqn.py:27-28
        float4 wv[BH / 16];                                            // the heads' weights (requested behind layer 2's MFMAs)
        {
            f32x16 acc[2][2];
            zero_acc(acc);
            L2.run(la, BS, lane, acc);
            if (ps == 0) STAMP(1, 5);
            {
                const float4 *wh = reinterpret_cast<const float4 *>(P.pack + m.p_wh) + lane;
#pragma unroll
                for (int kq = 0; kq < BH / 16; ++kq) wv[kq] = wh[kq * 64];
            }
            LDS_BARRIER();                                             // every wave has read all of h1: h2 may replace it
            if (ps == 0) STAMP(1, 6);
            int rb = row0;
            asm volatile("" : "+v"(rb));
            auto epi2 = [&](auto stash_tag, auto feat_tag) {
                constexpr bool ST = decltype(stash_tag)::value, FT = decltype(feat_tag)::value;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const int col = 32 * (ct0 + ct) + c;
                        float *dst = la + (col & ~15) + perm2(col & 15);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = 32 * rt + row32(r, h);
                            float v = acc[rt][ct][r] + bias2[ct];
                            v = v > 0.0f ? v : 0.0f;
                            dst[rl * BS] = v;
                            if constexpr (FT) { if (rb + rl < B) g.feat[(long long)(rb + rl) * BH + col] = v; }   // :32-33
                            if constexpr (ST) {
                                g.ph2[(long long)(rb + rl) * BH + col] = v;
                                const unsigned long long bits = __ballot(v > 0.0f);
                                if (c == 0) lm2[rl * 8 + ct0 + ct] = (unsigned)(bits >> (32 * h));
                            }
                        }
                    }
            };
            const bool feat_on = last && g.feat != nullptr;
            if (stash) { if (feat_on) epi2(std::true_type{}, std::true_type{}); else epi2(std::true_type{}, std::false_type{}); }
            else { if (feat_on) epi2(std::false_type{}, std::true_type{}); else epi2(std::false_type{}, std::false_type{}); }
        }
        if (ps == 0) STAMP(1, 7);
        LDS_BARRIER();
        if (ps == 0) STAMP(1, 8);
        // ---- heads (dddqn.py:29-30): wave w takes rows 16w .. 16w+15 on 16x16x4; the j-th MFMA of every k-block
        // accumulates into chain j, combined (c0 + c1) + (c2 + c3) (oracle: heads_row) -- as dqn_net.hip's finish_heads4
        {
            const int g4 = lane >> 4, r16 = lane & 15;
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
            const float *arow = la + (16 * wave + r16) * BS + 8 * (g4 & 1) + (g4 >> 1);    // + 16 kq + 2 j
            float biash = 0.0f;
            if (r16 == 0) biash = P.params[m.o_bv]; else if (r16 <= A) biash = P.params[m.o_ba + r16 - 1];
#pragma unroll
            for (int kq = 0; kq < BH / 16; ++kq) {
                const float *ak = arow + 16 * kq;
                c0 = MFMA4(ak[0], wv[kq].x, c0);
                c1 = MFMA4(ak[2], wv[kq].y, c1);
                c2 = MFMA4(ak[4], wv[kq].z, c2);
                c3 = MFMA4(ak[6], wv[kq].w, c3);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) lh[(16 * wave + 4 * g4 + r) * 16 + r16] = ((c0[r] + c1[r]) + (c2[r] + c3[r])) + biash;
        }
        if (ps == 0) STAMP(1, 9);
        LDS_BARRIER();
        if (ps == 0) STAMP(1, 10);
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
        if (ps == 0) STAMP(1, 11);
    }
    if (!g.do_bwd) continue;
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, h = lane >> 5, c = lane & 31;

    // ---- TD target / Huber gradient / dueling backward of this tile's rows (the row arithmetic of k_bwd_rows)
    const BwdArgs &bw = g.g;
    const int pl = g.npass - 1;                                        // pass that produced q = pred
    BigLayer<1, 1> LA; BigLayer<BH / 16, 4> LB;                         // the row backward's weights travel behind the TD arithmetic
    LA.init(reinterpret_cast<const float4 *>(bw.pack + m.p_wht), ct0, lane); LA.prefetch();
    LB.init(reinterpret_cast<const float4 *>(bw.pack + m.p_w2t), ct0, lane);
    for (int t = tid; t < 64 * BS3; t += 256) l3[t] = 0.0f;
    LDS_BARRIER();
    if (tid < 64) {
        const int i = row0 + tid;
        float rowloss = 0.0f;
        if (i < B) {
            float qr[16], tr[16], nqr[16], ntr[16];
            for (int k2 = 0; k2 < A; ++k2) qr[k2] = lq[(pl * 64 + tid) * 16 + k2];
            const int ai = bw.a ? bw.a[i] : 0;
            if (bw.targets) {
                for (int k2 = 0; k2 < A; ++k2) tr[k2] = bw.targets[(long long)i * A + k2];
            } else {
                for (int k2 = 0; k2 < A; ++k2) { nqr[k2] = lq[(0 * 64 + tid) * 16 + k2]; ntr[k2] = lq[(1 * 64 + tid) * 16 + k2]; }
                const float ri = bw.r[i];
                const float di = bw.d_f32 ? bw.d_f32[i] : (bw.d_u8[i] ? 1.0f : 0.0f);     // preprocessing :84
                const float delta = td_row(qr, nqr, ntr, ai, ri, di, bw.gamma, A, tr);
                if (bw.td) bw.td[i] = delta;
                if (bw.td_abs) bw.td_abs[i] = fabsf(delta);
            }
            float w = 1.0f;
            if (bw.w_raw) { w = __fdiv_rn(bw.w_raw[i], g.st->wmax); if (bw.isw_out) bw.isw_out[i] = w; }
            else if (bw.isw) w = bw.isw[i];
            const float invB = __fdiv_rn(1.0f, (float)B);
            float gk[16], gsum = 0.0f;
            for (int k2 = 0; k2 < A; ++k2) {
                const float e = qr[k2] - tr[k2];                       // pred - target, pred == q   (:35)
                rowloss = rowloss + huber(e);                          // :36
                const float cc = e > 1.0f ? 1.0f : (e < -1.0f ? -1.0f : e);
                gk[k2] = (w * cc) * invB;                              // dL/dpred
                gsum = gsum + gk[k2];
                if (bw.dq) bw.dq[(long long)i * A + k2] = gk[k2];
                if (bw.targets_out) bw.targets_out[(long long)i * A + k2] = tr[k2];
            }
            if (bw.w_raw || bw.isw) rowloss = w * rowloss;
            const float gmean = __fdiv_rn(gsum, (float)A);             // dueling backward: dv = sum g ; dadv = g - mean g
            l3[tid * BS3 + perm2(0)] = gsum;
            for (int k2 = 0; k2 < A; ++k2) l3[tid * BS3 + perm2(1 + k2)] = gk[k2] - gmean;
        }
        lrow[tid] = rowloss;
    }
    LDS_BARRIER();
    if (tid == 0) {                                                    // per-tile loss, 16-row sub-tiles in order (as the 16-row kernels)
        for (int q4 = 0; q4 < 4; ++q4) {
            float s = 0.0f;
            for (int k = 0; k < 16; ++k) s = s + lrow[16 * q4 + k];
            bw.loss_part[4 * tile + q4] = s;
        }
    }
    float *cs = g.colsum + (long long)tile * (2 * BH + 16);
    if (tid < 16) {                                                    // dz3 out (row-major [B][16]) + its column sums
        float s = 0.0f;
        for (int rl = 0; rl < 64; ++rl) s = s + l3[rl * BS3 + perm2(tid)];
        cs[2 * BH + tid] = s;
    }
    for (int t = tid; t < 64 * 16; t += 256) {
        const int rl = t >> 4, cc = t & 15;
        bw.pdz3[(long long)(row0 + rl) * 16 + cc] = l3[rl * BS3 + perm2(cc)];
    }
    // column sum over the wave's 64 rows of one accumulator column: registers, then the two lane halves
    auto colsum64 = [&](const f32x16 &x0, const f32x16 &x1) -> float {
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s = s + x0[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) s = s + x1[r];
        return s + __shfl_xor(s, 32, 64);
    };
    // ---- dz2 = (dz3 . WH^T) * (h2 > 0)
    {
        f32x16 acc[2][2];
        zero_acc(acc);
        LB.prefetch();                                                 // W2^T's first k-blocks travel behind dz2
        LA.run(l3, BS3, lane, acc);
        LDS_BARRIER();                                                 // (heads / Q are long done with h2: dz2 replaces it)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = 32 * (ct0 + ct) + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = 32 * rt + row32(r, h);
                    const float v = ((lm2[rl * 8 + ct0 + ct] >> c) & 1u) ? acc[rt][ct][r] : 0.0f;
                    acc[rt][ct][r] = v;
                    la[rl * BS + (col & ~15) + perm2(col & 15)] = v;
                    bw.pdz2[(long long)(row0 + rl) * BH + col] = v;
                }
            }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float s = colsum64(acc[0][ct], acc[1][ct]);
            if (h == 0) cs[BH + 32 * (ct0 + ct) + c] = s;
        }
    }
    LDS_BARRIER();
    // ---- dz1 = (dz2 . W2^T) * (h1 > 0)
    {
        f32x16 acc[2][2];
        zero_acc(acc);
        LB.run(la, BS, lane, acc);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = 32 * (ct0 + ct) + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = 32 * rt + row32(r, h);
                    const float v = ((lm1[rl * 8 + ct0 + ct] >> c) & 1u) ? acc[rt][ct][r] : 0.0f;
                    acc[rt][ct][r] = v;
                    bw.pdz1[(long long)(row0 + rl) * BH + col] = v;
                }
            }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float s = colsum64(acc[0][ct], acc[1][ct]);
            if (h == 0) cs[32 * (ct0 + ct) + c] = s;
        }
    }
    }   // row tiles
}

// ------------------------------------------------------------------ weight gradients, split-K
// Output tiles (128 rows x 128 columns of one weight block; 4 waves as 2 x 2, 64 x 64 each = 2 x 2 MFMA tiles):
//   0..3  dW2 = h1^T dz2   (256 x 256: tile (i, j) = rows 128i.., columns 128j..)
//   4, 5  dW1 = x^T  dz1   (K1 <= 32 rows used, columns 128 (t - 4)..)
//   6, 7  dWH = h2^T dz3   (rows 128 (t - 6).., 16 columns used)
// Operands are the row-major stashes: for k-step s the lane half h supplies batch row k0 + 2s + h, lanes c = 0..31 two
// adjacent columns 2c, 2c + 1 of the wave's 64 (one 8-byte load), i.e. MFMA tile q (q = 0, 1) holds columns 2c + q.
// blockIdx.y = batch slice; slab[slice][tile][128][128].
constexpr int DW_TILES = 8;
typedef float f32x2 __attribute__((ext_vector_type(2)));
// 8-byte load the compiler does not count (its destination is valid only behind a hand-placed s_waitcnt vmcnt)
__device__ __forceinline__ f32x2 gload2(const float *p) {
    f32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float gload1(const float *p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// One wave's share of a tile: acc[qa][qb] += sum over the batch rows [k_begin, k_end) of A(row, acol + qa) * B(row, bcol + qb).
// NA / NB = 2: the lane's two adjacent columns come with one 8-byte load (MFMA tile q holds columns 2c + q of the wave's
// 64); = 1: one 32-column tile, lane c = column c (a narrow operand: x or dz3), zero beyond `lim`.
// Operand ring, PF k-steps (2 batch rows each) deep. The loads are inline asm with hand-counted waits: inside a runtime
// loop the compiler's own wait insertion drains the whole ring at every use. Vector-memory results return in request
// order, so before slot p is used everything but the 2 (PF - 1) youngest requests must have landed. The row count is a
// multiple of 32, so the number of k-steps is a multiple of PF: no tail.
template <int NA, int NB>
__device__ __forceinline__ void dw_wave(const float *pa, int lda, int acol, int alim, const float *pb, int ldb, int bcol, int blim,
                                        int k_begin, int k_end, int lane, f32x16 (&acc)[2][2]) {
    const int h = lane >> 5, c = lane & 31;
    const int ca = NA == 2 ? acol + 2 * c : acol + c, cb = NB == 2 ? bcol + 2 * c : bcol + c;
    const bool a_on = ca < alim, b_on = cb < blim;                     // (widths are even: a column pair is in or out as a whole)
    const float *ra = pa + (long long)(k_begin + h) * lda + (a_on ? ca : 0);
    const float *rb = pb + (long long)(k_begin + h) * ldb + (b_on ? cb : 0);
    constexpr int PF = 16;
    f32x2 av[PF], bv[PF];
    const int nsteps = (k_end - k_begin) >> 1;
    const long long sa = 2ll * lda, sb = 2ll * ldb;
    auto lda_ = [&](const float *q) -> f32x2 { if constexpr (NA == 2) return gload2(q); else { f32x2 v; v.x = gload1(q); v.y = 0.0f; return v; } };
    auto ldb_ = [&](const float *q) -> f32x2 { if constexpr (NB == 2) return gload2(q); else { f32x2 v; v.x = gload1(q); v.y = 0.0f; return v; } };
#pragma unroll
    for (int p = 0; p < PF; ++p) { av[p] = lda_(ra + p * sa); bv[p] = ldb_(rb + p * sb); }
    for (int s0 = 0; s0 < nsteps; s0 += PF) {
        const int snext = s0 + PF < nsteps ? s0 + PF : s0;            // last round: re-request this round's rows (never used)
        const float *na = ra + snext * sa, *nb = rb + snext * sb;
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(av[p]), "+v"(bv[p]) : "n"(2 * (PF - 1)) : "memory");
            const float ax = a_on ? av[p].x : 0.0f, ay = a_on ? av[p].y : 0.0f;
            const float bx = b_on ? bv[p].x : 0.0f, by = b_on ? bv[p].y : 0.0f;
            acc[0][0] = MFMA32(ax, bx, acc[0][0]);
            if constexpr (NB == 2) acc[0][1] = MFMA32(ax, by, acc[0][1]);
            if constexpr (NA == 2) acc[1][0] = MFMA32(ay, bx, acc[1][0]);
            if constexpr (NA == 2 && NB == 2) acc[1][1] = MFMA32(ay, by, acc[1][1]);
            av[p] = lda_(na + p * sa); bv[p] = ldb_(nb + p * sb);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the re-requested rows land in dead registers: drain before they are reused
}

// blockIdx.x = batch slice, blockIdx.y = tile. Workgroups are dealt out in linear order, so the 4 x KS full-size dW2 tiles
// go round the CUs first and the 4 x KS light tiles (half the rows, half the MFMAs per step: a quarter of the work) fill
// up behind them: one heavy + one light workgroup per CU at KS = 64.
//   y = 0..3  dW2 tile (y>>1, y&1): 128 x 128, waves 2 x 2, 64 x 64 each                       slab slot: [128][128]
//   y = 4, 5  dW1 = x^T dz1, rows of the slice's half (y & 1): x as ONE 32-column tile, wave w = columns 64w..  [32][256]
//   y = 6, 7  dWH = h2^T dz3, half (y & 1): wave w = h2 columns 64w.., dz3 as ONE 32-column tile               [256][32]
__global__ void __launch_bounds__(256)
k_big_dw(NetDims m, const float *__restrict__ px, const float *__restrict__ ph1, const float *__restrict__ ph2,
         const float *__restrict__ pdz1, const float *__restrict__ pdz2, const float *__restrict__ pdz3,
         int Bp64, int rows_per_slice, float *slab) {
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = blockIdx.y, slice = blockIdx.x;
    const int K1 = m.KQ1 * 16;
    f32x16 acc[2][2];
    zero_acc(acc);
    int k_begin = slice * rows_per_slice;
    int k_end = k_begin + rows_per_slice; if (k_end > Bp64) k_end = Bp64;
    float *out = slab + ((long long)slice * DW_TILES + t) * 128 * 128;
    if (t < 4) {
        const int wi = wave >> 1, wj = wave & 1;
        if (k_begin < k_end)
            dw_wave<2, 2>(ph1, BH, 128 * (t >> 1) + 64 * wi, BH, pdz2, BH, 128 * (t & 1) + 64 * wj, BH, k_begin, k_end, lane, acc);
        // element (i32, j32) of MFMA tile (qa, qb) = dW2[.. + 2 i32 + qa][.. + 2 j32 + qb]; the two qb of a lane are adjacent
        // columns: one 8-byte store
#pragma unroll
        for (int qa = 0; qa < 2; ++qa)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 64 * wi + 2 * row32(r, h) + qa, col = 64 * wj + 2 * c;
                *reinterpret_cast<float2 *>(out + row * 128 + col) = float2{acc[qa][0][r], acc[qa][1][r]};
            }
        return;
    }
    // light tiles: the slice's rows in two halves (whole 32-row groups)
    const int half = (k_end - k_begin) >> 1;
    if (t & 1) k_begin += half; else k_end = k_begin + half;
    if (t < 6) {
        if (k_begin < k_end) dw_wave<1, 2>(px, K1, 0, K1, pdz1, BH, 64 * wave, BH, k_begin, k_end, lane, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r)                                   // row = x column (< 32), columns 64 wave + 2c, + 1
            *reinterpret_cast<float2 *>(out + row32(r, h) * 256 + 64 * wave + 2 * c) = float2{acc[0][0][r], acc[0][1][r]};
    } else {
        if (k_begin < k_end) dw_wave<2, 1>(ph2, BH, 64 * wave, BH, pdz3, 16, 0, 16, k_begin, k_end, lane, acc);
#pragma unroll
        for (int qa = 0; qa < 2; ++qa)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(64 * wave + 2 * row32(r, h) + qa) * 32 + c] = acc[qa][0][r];   // column c of dz3 (< 16 used)
    }
}

// ------------------------------------------------------------------------------- host side
bool big_supported(const NetDims &m, int B, bool any_size) {
    return B >= (any_size ? 64 : DQN_BIG_MIN) && m.H1 == BH && m.H2 == BH && m.D <= 32 && m.A <= 15;
}

// K slices of the weight-gradient GEMM: enough workgroups to fill the machine (8 tiles x KS), at least 128 rows each
int big_dw_slices(int B, int num_cus) {
    const int Bp64 = (B + 63) / 64 * 64;
    int ks = (2 * num_cus + DW_TILES - 1) / DW_TILES;
    if (ks > Bp64 / 128) ks = Bp64 / 128;
    if (ks < 1) ks = 1;
    if (ks > DQN_BIG_MAX_SLICES) ks = DQN_BIG_MAX_SLICES;
    return ks;
}
size_t big_slab_floats(int max_batch, int num_cus) { (void)max_batch; (void)num_cus; return (size_t)DQN_BIG_MAX_SLICES * DW_TILES * 128 * 128; }
size_t big_colsum_floats(int max_batch) { return (size_t)((max_batch + 63) / 64) * (2 * BH + 16); }

static size_t big_rows_lds() {
    return sizeof(float) * (64 * BSX + 64 * BS + 64 * 16 + 3 * 64 * 16 + 64 * BS3 + 64 + 2 * 64 * 8);
}

void launch_big_forward(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, int num_cus) {
    BigArgs g{};
    g.npass = npass;
    for (int i = 0; i < npass; ++i) g.p[i] = BigPass{passes[i].x, passes[i].params, passes[i].pack, passes[i].q};
    g.feat = passes[npass - 1].feat;
    const int tiles = (B + 63) / 64;
    if (m.KQ1 == 1) DQN_LAUNCH((k_big_rows<1>), dim3(tiles < num_cus ? tiles : num_cus), dim3(256), big_rows_lds(), s, m, g, B);
    else DQN_LAUNCH((k_big_rows<2>), dim3(tiles < num_cus ? tiles : num_cus), dim3(256), big_rows_lds(), s, m, g, B);
}

// passes: update form = {online(s'), target(s'), online(s)}; grads form (bw.targets given) = {online(s)}
void launch_big_rows_bwd(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, const BwdArgs &bw,
                         float *px, float *ph1, float *ph2, float *colsum, DqnState *st, int num_cus) {
    BigArgs g{};
    g.npass = npass;
    for (int i = 0; i < npass; ++i) g.p[i] = BigPass{passes[i].x, passes[i].params, passes[i].pack, passes[i].q};
    g.do_bwd = 1; g.g = bw; g.px = px; g.ph1 = ph1; g.ph2 = ph2; g.colsum = colsum; g.st = st;
    const int tiles = (B + 63) / 64;
    if (m.KQ1 == 1) DQN_LAUNCH((k_big_rows<1>), dim3(tiles < num_cus ? tiles : num_cus), dim3(256), big_rows_lds(), s, m, g, B);
    else DQN_LAUNCH((k_big_rows<2>), dim3(tiles < num_cus ? tiles : num_cus), dim3(256), big_rows_lds(), s, m, g, B);
}

void launch_big_dw(hipStream_t s, const NetDims &m, const float *px, const float *ph1, const float *ph2, const float *pdz1,
                   const float *pdz2, const float *pdz3, int B, float *slab, const float *colsum, float *grad,
                   const float *loss_part, float *loss_out, DqnState *st, int bump_ctr, const AdamArgs &adam, int num_cus) {
    const int Bp64 = (B + 63) / 64 * 64;
    const int KS = big_dw_slices(B, num_cus);
    int rps = (Bp64 + KS - 1) / KS;
    rps = (rps + 63) / 64 * 64;                                       // whole row tiles per slice (rows >= B carry zero gradients)
    const int ks_used = (Bp64 + rps - 1) / rps;
    DQN_LAUNCH(k_big_dw, dim3(ks_used, DW_TILES), dim3(256), 0, s, m, px, ph1, ph2, pdz1, pdz2, pdz3, Bp64, rps, slab);
    const int wblocks = (int)((4 * m.P + 255) / 256), bblocks = (2 * BH + 1 + m.A + 3) / 4;
    DQN_LAUNCH((k_big_reduce<false>), dim3(wblocks + bblocks), dim3(256), 0, s, m, slab, ks_used, colsum, Bp64 / 64, B, grad, loss_part, loss_out, st,
               bump_ctr, adam, wblocks);
}
