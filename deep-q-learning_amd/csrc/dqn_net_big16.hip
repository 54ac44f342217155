// csrc/dqn_net_big16.hip -- the Q-network kernels for LARGE batches in the bf16 precision mode (dqn_config.precision =
// DQN_PREC_BF16; bf16 operands, f32 accumulation, f32 master weights): 64 batch rows per workgroup on
// v_mfma_f32_32x32x16_bf16, the same decomposition as the exact-f32 path of dqn_net_big.hip:
//
//   k_big_rows16  one workgroup = 64 rows. Update form: online(s') -> next_q, target(s') -> next_q_tm, online(s) -> q
//                 (LunarLander/dddqn.py:24-31 three times, q_learning_functions.py:52-54), TD target / Huber gradient /
//                 dueling backward (:55-60, :35-36), dz2, dz1 -- h1 / h2 / dz2 never leave LDS between the layers (one bf16
//                 image, 33 KB); what the weight-gradient kernel needs (x, h1, h2, dz1, dz2, dz3) goes to HBM once.
//                 Forward form: 1..3 passes, Q (+ features) out. Two workgroups per CU (58 KB of LDS each): one's epilogue
//                 (bias / ReLU / bf16 rounding / LDS stores) runs beside the other's MFMAs.
//   k_big_dw16    dW = H^T Z over the batch as a split-K GEMM, 128 x 128 output tiles, partial tiles to the slab.
//   k_big_reduce<true> (dqn_big_reduce.h)  slab partials + column sums -> gradient, Adam / AdamW, bf16 shadow refresh.
//
// Weights come from the SAME bf16 fragment packs as the 16-row kernels (dqn_bf16_pack.h, 32-deep k-blocks for 16x16x32):
// for a 32-column MFMA tile, lane (h = lane>>5, c = lane&31) takes of 16-column pack tile 2*ct32 + (c>>4) the 16-byte
// fragments of pack lanes 16 h + (c&15) (k = 32 kq + 8 h + j: the block's first 16-deep k-step) and 16 (h + 2) + (c&15)
// (k = 32 kq + 16 + 8 h + j: its second) -- the operand map of 32x32x16: lane (r = l&31, h = l>>5) holds B[k = 8h + j][col r].
//
// Stashes for the weight gradients are K-PACKED: stash(C)[((b >> 3) * C + col) * 8 + (b & 7)] (bf16) -- eight consecutive
// batch rows of a column are 16 contiguous bytes. An accumulator lane (column c; registers 4g .. 4g+3 = rows 8g + 4h + 0..3)
// stores them as 8 bytes, the 64 lanes of a store instruction cover 512 contiguous bytes; the dW kernel, whose MFMA k index
// is the batch row, loads an operand fragment (8 consecutive rows of its column) as ONE 16-byte piece.
//
// Only for hidden1 == hidden2 == 256, obs_dim <= 32 (BASELINE configs[1] net); tolerance of the mode: 2e-2 of scale.
#include <type_traits>
#include "dqn_device.h"
#include "dqn_launch.h"
#include "dqn_net_common.h"
#include "dqn_bf16_pack.h"
#include "dqn_big_reduce.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
#define MFMA32B(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MFMA16B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// row of accumulator register r of a 32x32 tile for lane half h (C/D map of every 32x32 MFMA)
__device__ __forceinline__ int row32b(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

constexpr int HB = BIG_H;               // hidden width
constexpr int SA = HB + 8;              // LDS row stride (bf16 elements) of the activation image: 528 B
constexpr int SX = 32 + 8;              // of the input image (32 staged columns)
constexpr int S3 = 16 + 8;              // of dz3

// acc[rt][ct] (+)= A . W for the wave's 64 rows (rt = 0, 1) and two 32-column tiles ct0, ct0 + 1. A: row-major bf16 LDS image;
// W: bf16 fragment pack with KQ 32-deep k-blocks. Register ring PF k-blocks deep, straight-line code (see dqn_net_big.hip).
// ONE_STEP: only the first 16-deep k-step of the (single) block carries data (K <= 16).
template <int KQ, int PF, bool ONE_STEP>
struct BigLayer16 {
    const bf16x8 *pb[2];
    bf16x8 blo[PF][2], bhi[PF][2];
    __device__ __forceinline__ void init(const __bf16 *wp, int ct0, int lane) {
        const int h = lane >> 5, c = lane & 31;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
            pb[ct] = reinterpret_cast<const bf16x8 *>(wp) + (long long)(2 * (ct0 + ct) + (c >> 4)) * KQ * 64 + h * 16 + (c & 15);
    }
    __device__ __forceinline__ void prefetch() {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int kq = p < KQ ? p : KQ - 1;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                blo[p][ct] = pb[ct][kq * 64];
                if constexpr (!ONE_STEP) bhi[p][ct] = pb[ct][kq * 64 + 32];
            }
        }
    }
    __device__ __forceinline__ void run(const __bf16 *la, int S, int lane, f32x16 (&acc)[2][2]) {
        const int h = lane >> 5, c = lane & 31;
        const __bf16 *arow0 = la + c * S + 8 * h, *arow1 = arow0 + 32 * S;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const int p = kq % PF;
            const bf16x8 a00 = *reinterpret_cast<const bf16x8 *>(arow0 + 32 * kq), a10 = *reinterpret_cast<const bf16x8 *>(arow1 + 32 * kq);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                acc[0][ct] = MFMA32B(a00, blo[p][ct], acc[0][ct]);
                acc[1][ct] = MFMA32B(a10, blo[p][ct], acc[1][ct]);
            }
            if constexpr (!ONE_STEP) {
                const bf16x8 a01 = *reinterpret_cast<const bf16x8 *>(arow0 + 32 * kq + 16), a11 = *reinterpret_cast<const bf16x8 *>(arow1 + 32 * kq + 16);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc[0][ct] = MFMA32B(a01, bhi[p][ct], acc[0][ct]);
                    acc[1][ct] = MFMA32B(a11, bhi[p][ct], acc[1][ct]);
                }
            }
            if (kq + PF < KQ) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) { blo[p][ct] = pb[ct][(kq + PF) * 64]; bhi[p][ct] = pb[ct][(kq + PF) * 64 + 32]; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

__device__ __forceinline__ void zero_acc16(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
}

struct Big16Pass {
    const float *x;            // [B][D] f32 rows of this pass
    const float *params;       // flat f32 params (biases)
    const __bf16 *pack;        // bf16 fragment packs
    float *q;                  // [B][A] out or NULL
};

struct Big16Args {
    Big16Pass p[3]; int npass; // passes in order; in update / grads form the LAST pass is online(s)
    float *feat;               // [B][H2] f32 features of the last pass (dddqn.py:32-33) or NULL
    int do_bwd;
    BwdArgs g;                 // TD inputs / outputs; pdz1/2/3 (k-packed bf16), loss_part
    __bf16 *px, *ph1, *ph2;    // k-packed stashes, C = 32 / 256 / 256
    float *colsum;             // [tiles][2*HB + 16]: per-tile column sums of dz1 | dz2 | dz3 (bias gradients)
    DqnState *st;
};

template <bool X16>                      // obs_dim <= 16: layer 1 is one 16-deep k-step
__global__ void __launch_bounds__(256)
k_big_rows16(NetDims m, Dims16 d16, Big16Args g, int B) {
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
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, h = lane >> 5, c = lane & 31;

    // ---- TD target / Huber gradient / dueling backward of this tile's rows (the row arithmetic of k_bwd_rows, f32)
    const BwdArgs &bw = g.g;
    const __bf16 *bpack = reinterpret_cast<const __bf16 *>(bw.pack);
    const int pl = g.npass - 1;                        // pass that produced q = pred
    BigLayer16<1, 1, true> LA; BigLayer16<HB / 32, 4, false> LB;
    LA.init(bpack + d16.p_wht, ct0, lane); LA.prefetch();
    LB.init(bpack + d16.p_w2t, ct0, lane);
    for (int t = tid; t < 64 * S3; t += 256) l3[t] = (__bf16)0.0f;
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
            l3[tid * S3 + 0] = (__bf16)gsum;
            for (int k2 = 0; k2 < A; ++k2) l3[tid * S3 + 1 + k2] = (__bf16)(gk[k2] - gmean);
        }
        lrow[tid] = rowloss;
    }
    LDS_BARRIER();
    if (tid == 0) {                                    // per-tile loss, 16-row sub-tiles in order (as the 16-row kernels)
        for (int q4 = 0; q4 < 4; ++q4) {
            float s = 0.0f;
            for (int k = 0; k < 16; ++k) s = s + lrow[16 * q4 + k];
            bw.loss_part[4 * tile + q4] = s;
        }
    }
    float *cs = g.colsum + (long long)tile * (2 * HB + 16);
    if (tid < 16) {                                    // column sums of dz3 (the values the weight gradient sees)
        float s = 0.0f;
        for (int rl = 0; rl < 64; ++rl) s = s + (float)l3[rl * S3 + tid];
        cs[2 * HB + tid] = s;
    }
    if (tid < 128) {                                   // dz3 out, k-packed [B/8][16][8]: thread (g8, cc) one 16-byte piece
        const int g8 = tid >> 4, cc = tid & 15;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = l3[(8 * g8 + j) * S3 + cc];
        *reinterpret_cast<bf16x8 *>(reinterpret_cast<__bf16 *>(bw.pdz3) + ((long long)((row0 >> 3) + g8) * 16 + cc) * 8) = v;
    }
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
        LB.prefetch();                                 // W2^T's first k-blocks travel behind dz2
        LA.run(l3, S3, lane, acc);
        LDS_BARRIER();                                 // (heads / Q are long done with h2: dz2 replaces it)
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
    }
    LDS_BARRIER();
    // ---- dz1 = (dz2 . W2^T) * (h1 > 0)
    {
        f32x16 acc[2][2];
        zero_acc16(acc);
        LB.run(la, SA, lane, acc);
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
    }
    LDS_BARRIER();                                     // the next tile's first pass rewrites the images
    }   // row tiles
}

// ------------------------------------------------------------------ weight gradients, split-K
// Output tiles as in dqn_net_big.hip (128 x 128; 4 waves as 2 x 2, 64 x 64 each = 2 x 2 MFMA tiles):
//   0..3  dW2 = h1^T dz2;   4, 5  dW1 = x^T dz1 (32 x 256, the slice's rows in two halves);   6, 7  dWH = h2^T dz3 (256 x 32)
// Operands are the k-packed stashes: for the 16-deep k-step at batch row kb, lane (c, h) loads for column col the 16 bytes at
// stash[((kb/8 + h) * C + col) * 8] = rows kb + 8h .. + 7 -- the A / B fragment of 32x32x16 (A[row c][k = 8h + j] with row =
// weight row = operand column). 16-byte loads the compiler does not count (hand-placed s_waitcnt), ring PF k-steps deep.
__device__ __forceinline__ u32x4v gload4(const void *p) {
    u32x4v v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int NA, int NB>
__device__ __forceinline__ void dw_wave16(const __bf16 *pa, int Ca, int acol, int alim, const __bf16 *pb, int Cb, int bcol, int blim,
                                          int k_begin, int k_end, int lane, f32x16 (&acc)[2][2]) {
    const int h = lane >> 5, c = lane & 31;
    constexpr int PF = 8, NL = NA + NB;
    const int nsteps = (k_end - k_begin) >> 4;
    bool a_on[2], b_on[2];
    const __bf16 *ra[2], *rb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ca = acol + 32 * q + c, cb = bcol + 32 * q + c;
        a_on[q] = q < NA && ca < alim; b_on[q] = q < NB && cb < blim;
        ra[q] = pa + ((long long)((k_begin >> 3) + h) * Ca + (a_on[q] ? ca : 0)) * 8;
        rb[q] = pb + ((long long)((k_begin >> 3) + h) * Cb + (b_on[q] ? cb : 0)) * 8;
    }
    const long long sa = 2ll * Ca * 8, sb = 2ll * Cb * 8;               // elements per k-step (two 8-row groups)
    u32x4v av[PF][NA], bv[PF][NB];
    auto request = [&](int p, int step) {
        const int s = step < nsteps ? step : nsteps - 1;                // past the end: a valid address, never used
#pragma unroll
        for (int q = 0; q < NA; ++q) av[p][q] = gload4(ra[q] + s * sa);
#pragma unroll
        for (int q = 0; q < NB; ++q) bv[p][q] = gload4(rb[q] + s * sb);
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) request(p, p);
    const u32x4v zero = {0u, 0u, 0u, 0u};
    for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if constexpr (NA == 2 && NB == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(av[p][0]), "+v"(av[p][1]), "+v"(bv[p][0]), "+v"(bv[p][1]) : "n"(NL * (PF - 1)) : "memory");
            else if constexpr (NA == 1) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(av[p][0]), "+v"(bv[p][0]), "+v"(bv[p][1]) : "n"(NL * (PF - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(av[p][0]), "+v"(av[p][1]), "+v"(bv[p][0]) : "n"(NL * (PF - 1)) : "memory");
            if (s0 + p < nsteps) {
                bf16x8 fa[2], fb[2];
#pragma unroll
                for (int q = 0; q < NA; ++q) fa[q] = __builtin_bit_cast(bf16x8, a_on[q] ? av[p][q] : zero);
#pragma unroll
                for (int q = 0; q < NB; ++q) fb[q] = __builtin_bit_cast(bf16x8, b_on[q] ? bv[p][q] : zero);
#pragma unroll
                for (int qa = 0; qa < NA; ++qa)
#pragma unroll
                    for (int qb = 0; qb < NB; ++qb) acc[qa][qb] = MFMA32B(fa[qa], fb[qb], acc[qa][qb]);
            }
            request(p, s0 + PF + p);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the re-requested rows land in dead registers: drain before they are reused
}

__global__ void __launch_bounds__(256)
k_big_dw16(const __bf16 *__restrict__ px, const __bf16 *__restrict__ ph1, const __bf16 *__restrict__ ph2,
           const __bf16 *__restrict__ pdz1, const __bf16 *__restrict__ pdz2, const __bf16 *__restrict__ pdz3,
           int Bp64, int rows_per_slice, float *slab) {
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = blockIdx.y, slice = blockIdx.x;
    f32x16 acc[2][2];
    zero_acc16(acc);
    int k_begin = slice * rows_per_slice;
    int k_end = k_begin + rows_per_slice; if (k_end > Bp64) k_end = Bp64;
    float *out = slab + ((long long)slice * BIG_DW_TILES + t) * 128 * 128;
    if (t < 4) {
        const int wi = wave >> 1, wj = wave & 1;
        if (k_begin < k_end)
            dw_wave16<2, 2>(ph1, HB, 128 * (t >> 1) + 64 * wi, HB, pdz2, HB, 128 * (t & 1) + 64 * wj, HB, k_begin, k_end, lane, acc);
#pragma unroll
        for (int qa = 0; qa < 2; ++qa)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[(64 * wi + 32 * qa + row32b(r, h)) * 128 + 64 * wj + 32 * qb + c] = acc[qa][qb][r];
        return;
    }
    // light tiles: the slice's rows in two halves (whole 32-row groups)
    const int half = (k_end - k_begin) >> 1;
    if (t & 1) k_begin += half; else k_end = k_begin + half;
    if (t < 6) {
        if (k_begin < k_end) dw_wave16<1, 2>(px, 32, 0, 32, pdz1, HB, 64 * wave, HB, k_begin, k_end, lane, acc);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[row32b(r, h) * 256 + 64 * wave + 32 * qb + c] = acc[0][qb][r];   // row = x column (< 32)
    } else {
        if (k_begin < k_end) dw_wave16<2, 1>(ph2, HB, 64 * wave, HB, pdz3, 16, 0, 16, k_begin, k_end, lane, acc);
#pragma unroll
        for (int qa = 0; qa < 2; ++qa)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(64 * wave + 32 * qa + row32b(r, h)) * 32 + c] = acc[qa][0][r];   // column c of dz3 (< 16 used)
    }
}

// ------------------------------------------------------------------------------- host side
static size_t big16_rows_lds() {
    return 2 * (size_t)(64 * SX + 64 * SA + 64 * S3) + 4 * (size_t)(64 * 16 + 3 * 64 * 16 + 64);
}

static void launch_rows16(hipStream_t s, const NetDims &m, const Big16Args &g, int B, int num_cus) {
    const int tiles = (B + 63) / 64;
    const int grid = tiles < 2 * num_cus ? tiles : 2 * num_cus;          // two workgroups per CU
    const Dims16 d = make_dims16(m);
    if (m.D <= 16) DQN_LAUNCH((k_big_rows16<true>), dim3(grid), dim3(256), big16_rows_lds(), s, m, d, g, B);
    else DQN_LAUNCH((k_big_rows16<false>), dim3(grid), dim3(256), big16_rows_lds(), s, m, d, g, B);
}

void launch_big16_forward(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, int num_cus) {
    Big16Args g{};
    g.npass = npass;
    for (int i = 0; i < npass; ++i) g.p[i] = Big16Pass{passes[i].x, passes[i].params, reinterpret_cast<const __bf16 *>(passes[i].pack), passes[i].q};
    g.feat = passes[npass - 1].feat;
    launch_rows16(s, m, g, B, num_cus);
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
    launch_rows16(s, m, g, B, num_cus);
}

void launch_big16_dw(hipStream_t s, const NetDims &m, const float *px, const float *ph1, const float *ph2, const float *pdz1,
                     const float *pdz2, const float *pdz3, int B, float *slab, const float *colsum, float *grad,
                     const float *loss_part, float *loss_out, DqnState *st, int bump_ctr, const AdamArgs &adam, int num_cus) {
    const int Bp64 = (B + 63) / 64 * 64;
    const int KS = big_dw_slices(B, num_cus);
    int rps = (Bp64 + KS - 1) / KS;
    rps = (rps + 63) / 64 * 64;                                       // whole row tiles per slice (rows >= B carry zero gradients)
    const int ks_used = (Bp64 + rps - 1) / rps;
    auto bf = [](const float *p) { return reinterpret_cast<const __bf16 *>(p); };
    DQN_LAUNCH(k_big_dw16, dim3(ks_used, BIG_DW_TILES), dim3(256), 0, s, bf(px), bf(ph1), bf(ph2), bf(pdz1), bf(pdz2), bf(pdz3), Bp64, rps, slab);
    const int wblocks = (int)((m.P + 255) / 256), bblocks = (2 * HB + 1 + m.A + 3) / 4;
    DQN_LAUNCH((k_big_reduce<true>), dim3(wblocks + bblocks), dim3(256), 0, s, m, slab, ks_used, colsum, Bp64 / 64, B, grad, loss_part, loss_out, st,
               bump_ctr, adam, wblocks);
}
