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
#include "dqn_net_big16.h"

#define PSTAMP(S) do { if (ps == 0) STAMP(1, S); } while (0)
// Two workgroups per CU (58 KB of LDS, 243 registers each): one's epilogue / barriers / global round trips run beside the other's
// MFMAs -- measured r03 at B = 2^17, one forward pass: 43.8 us with one workgroup per CU, 30.8 us with two. The 256-register
// budget holds with the layer-2 ring two k-blocks deep (three: 2 scratch registers, 31.9 us; four: 19, no faster).
#ifndef BIG16_FWD_WGS_PER_CU
#define BIG16_FWD_WGS_PER_CU 2
#endif
#ifndef BIG16_FWD_PF
#define BIG16_FWD_PF 2
#endif
template <bool X16, bool FEAT>           // obs_dim <= 16: layer 1 is one 16-deep k-step; FEAT: the last pass also writes its features
__global__ void __launch_bounds__(256, FEAT ? 1 : BIG16_FWD_WGS_PER_CU)
k_big_fwd16(NetDims m, Dims16 d16, Big16Args g, int B) {
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
    STAMP(1, 0);
    x_request(g.p[0], (int)blockIdx.x * 64);
    // The operands of a pass -- layer-1 fragments, biases, the first BIG16_PF k-blocks of layer 2, the heads' fragments -- are
    // requested ONE PASS AHEAD, each set into the registers its predecessor has just freed. The pass after the last is the next
    // tile's first (same arrays again).
    // Two accumulator orientations (same operands, MFMA arguments swapped):
    //   T (passes that stash nothing): acc = (A W)^T -- lane = batch row, registers 4g..4g+3 = four consecutive OUTPUT COLUMNS:
    //     the bf16 LDS image takes them as ONE 8-byte store (16 per lane and layer instead of 64 two-byte ones); the bias is
    //     per register (16 values per column tile, wave-uniform per lane half);
    //   N (the stash pass online(s) and the row backward): lane = output column, registers = batch rows: four consecutive rows
    //     of a column are the 8-byte piece of the k-packed stash; the LDS image takes two-byte stores.
    const int lane0 = tid0 & 63, c0 = lane0 & 31, h0 = lane0 >> 5;
    BigLayer16<1, 1, X16> L1; BigLayer16<HB / 32, BIG16_FWD_PF, false> L2;
    float bn1[2], bn2[2];                              // N form: bias of the lane's column
    f32x4 bt1[2][4], bt2[2][4];                        // T form: biases of columns 8g + 4h + 0..3 of each column tile
    bf16x8 wv[HB / 32];                                // the heads' weights
    float biash = 0.0f;                                // head bias of column lane & 15 (value | advantages)
    auto req_l1 = [&](const Big16Pass &P) { L1.init(P.pack + d16.p_w1, ct0, lane0); L1.prefetch(); };
    auto req_l2 = [&](const Big16Pass &P) { L2.init(P.pack + d16.p_w2, ct0, lane0); L2.prefetch(); };
    auto req_b = [&](const Big16Pass &P, long long o_b, float (&bn)[2], f32x4 (&bt)[2][4]) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            bn[ct] = P.params[o_b + 32 * (ct0 + ct) + c0];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) bt[ct][gq] = *reinterpret_cast<const f32x4 *>(P.params + o_b + 32 * (ct0 + ct) + 8 * gq + 4 * h0);
        }
    };
    auto req_wv = [&](const Big16Pass &P) {
        const bf16x8 *wh = reinterpret_cast<const bf16x8 *>(P.pack + d16.p_wh) + lane0;
#pragma unroll
        for (int kq = 0; kq < HB / 32; ++kq) wv[kq] = wh[kq * 64];
        const int r16 = lane0 & 15;
        biash = 0.0f;
        if (r16 == 0) biash = P.params[m.o_bv]; else if (r16 <= A) biash = P.params[m.o_ba + r16 - 1];
    };
    req_l1(g.p[0]); req_b(g.p[0], m.o_b1, bn1, bt1); req_b(g.p[0], m.o_b2, bn2, bt2); req_l2(g.p[0]); req_wv(g.p[0]);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * 64;
    for (int ps = 0; ps < g.npass; ++ps) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));                  // opaque per pass (keeps the epilogue addresses out of the outer loops)
        const int lane = tid & 63, h = lane >> 5, c = lane & 31;
        const Big16Pass &P = g.p[ps];
        const bool last = ps == g.npass - 1;
        const bool feat_on = FEAT && last;
        const bool nform = feat_on;                    // (the features go out from the N form)
        const Big16Pass &NP = last ? g.p[0] : g.p[ps + 1];      // whose operands are requested during this pass

        {
            const int g8 = tid >> 5, cc = tid & 31;
            bf16x8 xb;
#pragma unroll
            for (int j = 0; j < 8; ++j) { xb[j] = (__bf16)xv[j]; lx[(8 * g8 + j) * SX + cc] = xb[j]; }
        }
        LDS_BARRIER();
        PSTAMP(1);
        int rb = row0;
        asm volatile("" : "+v"(rb));
        // N-form epilogue of a 256-wide layer: bias is in the accumulators already
        auto epi_n = [&](f32x16 (&acc)[2][2], auto feat_tag) {
            constexpr bool FT = decltype(feat_tag)::value;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int col = 32 * (ct0 + ct) + c;
                    unsigned short *dst = reinterpret_cast<unsigned short *>(la) + (32 * rt + 4 * h) * SA + col;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const unsigned w0 = relu_pack(acc[rt][ct][4 * gq], acc[rt][ct][4 * gq + 1]);
                        const unsigned w1 = relu_pack(acc[rt][ct][4 * gq + 2], acc[rt][ct][4 * gq + 3]);
                        dst[(8 * gq + 0) * SA] = (unsigned short)w0; dst[(8 * gq + 1) * SA] = (unsigned short)(w0 >> 16);
                        dst[(8 * gq + 2) * SA] = (unsigned short)w1; dst[(8 * gq + 3) * SA] = (unsigned short)(w1 >> 16);
                        if constexpr (FT) {                                                         // dddqn.py:32-33
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int rl = 32 * rt + 8 * gq + 4 * h + u;
                                const float v = acc[rt][ct][4 * gq + u];
                                if (rb + rl < B) g.feat[(long long)(rb + rl) * HB + col] = v > 0.0f ? v : 0.0f;
                            }
                        }
                    }
                }
        };
        // T-form epilogue: lane = batch row 32 rt + c, registers 4g .. 4g+3 = columns 8g + 4h + 0..3 of the column tile
        auto epi_t = [&](f32x16 (&acc)[2][2]) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    __bf16 *dst = la + (32 * rt + c) * SA + 32 * (ct0 + ct) + 4 * h;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<u32x2v *>(dst + 8 * gq) = u32x2v{relu_pack(acc[rt][ct][4 * gq], acc[rt][ct][4 * gq + 1]),
                                                                            relu_pack(acc[rt][ct][4 * gq + 2], acc[rt][ct][4 * gq + 3])};
                }
        };
        auto init_n = [&](f32x16 (&acc)[2][2], const float (&bn)[2]) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[rt][ct][r] = bn[ct];
        };
        auto init_t = [&](f32x16 (&acc)[2][2], const f32x4 (&bt)[2][4]) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[rt][ct][r] = bt[ct][r >> 2][r & 3];
        };
        // ---- layer 1: h1 = relu(x @ w1 + b1)                                  dddqn.py:25-26
        {
            f32x16 acc[2][2];
            if (nform) { init_n(acc, bn1); L1.run(lx, SX, lane, acc); } else { init_t(acc, bt1); L1.run_t(lx, SX, lane, acc); }
            PSTAMP(2);
            req_l1(NP);
            if (!last) x_request(g.p[ps + 1], row0);
            else if (tile + (int)gridDim.x < ntiles) x_request(g.p[0], (tile + (int)gridDim.x) * 64);
            if (nform) epi_n(acc, std::false_type{});
            else epi_t(acc);
            req_b(NP, m.o_b1, bn1, bt1);
            PSTAMP(3);
        }
        LDS_BARRIER();
        PSTAMP(4);
        // ---- layer 2: h2 = relu(h1 @ w2 + b2)                                 dddqn.py:27-28
        {
            f32x16 acc[2][2];
            if (nform) { init_n(acc, bn2); L2.run(la, SA, lane, acc); } else { init_t(acc, bt2); L2.run_t(la, SA, lane, acc); }
            PSTAMP(5);
            req_l2(NP);
            LDS_BARRIER();                             // every wave has read all of h1: h2 may replace it
            PSTAMP(6);
            if (nform) epi_n(acc, std::true_type{});
            else epi_t(acc);
            req_b(NP, m.o_b2, bn2, bt2);
            PSTAMP(7);
        }
        LDS_BARRIER();
        PSTAMP(8);
        // ---- heads (dddqn.py:29-30): wave w takes rows 16w .. 16w+15 on 16x16x32 (A: lane (row l&15, k = 8(l>>4) + j)), then
        // its own rows' Q = val + adv - mean(adv) (dddqn.py:31) from its slice of lh: no workgroup barrier in between
        {
            const int kg = lane >> 4, r16 = lane & 15;
            f32x4 hc = {0.f, 0.f, 0.f, 0.f};
            const __bf16 *arow = la + (16 * wave + r16) * SA + 8 * kg;
#pragma unroll
            for (int kq = 0; kq < HB / 32; ++kq) hc = MFMA16B(*reinterpret_cast<const bf16x8 *>(arow + 32 * kq), wv[kq], hc);
            const float bh = biash;
            req_wv(NP);
#pragma unroll
            for (int r = 0; r < 4; ++r) lh[(16 * wave + 4 * kg + r) * 16 + r16] = hc[r] + bh;      // C/D: col l&15, row 4(l>>4) + r
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (the wave's own LDS stores: wave-synchronous)
            if (lane < 16) {
                const int rl = 16 * wave + lane;
                const float *hr = lh + rl * 16;
                float sum = 0.0f;
                for (int a = 0; a < A; ++a) sum = sum + hr[1 + a];
                const float mean = __fdiv_rn(sum, (float)A);
                for (int a = 0; a < A; ++a) {
                    const float qv = (hr[0] + hr[1 + a]) - mean;
                    lq[(ps * 64 + rl) * 16 + a] = qv;
                    if (P.q && row0 + rl < B) P.q[(long long)(row0 + rl) * A + a] = qv;
                }
            }
            PSTAMP(9);
        }
        LDS_BARRIER();
        PSTAMP(10);
    }
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

void launch_big16_forward(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, int num_cus) {
    Big16Args g{};
    g.npass = npass;
    for (int i = 0; i < npass; ++i) g.p[i] = Big16Pass{passes[i].x, passes[i].params, reinterpret_cast<const __bf16 *>(passes[i].pack), passes[i].q};
    g.feat = passes[npass - 1].feat;
    const int tiles = (B + 63) / 64;
    const int grid = tiles < BIG16_FWD_WGS_PER_CU * num_cus ? tiles : BIG16_FWD_WGS_PER_CU * num_cus;
    const Dims16 d = make_dims16(m);
    if (g.feat) {
        const int grid1 = tiles < num_cus ? tiles : num_cus;
        if (m.D <= 16) DQN_LAUNCH((k_big_fwd16<true, true>), dim3(grid1), dim3(256), big16_rows_lds(), s, m, d, g, B);
        else DQN_LAUNCH((k_big_fwd16<false, true>), dim3(grid1), dim3(256), big16_rows_lds(), s, m, d, g, B);
    } else {
        if (m.D <= 16) DQN_LAUNCH((k_big_fwd16<true, false>), dim3(grid), dim3(256), big16_rows_lds(), s, m, d, g, B);
        else DQN_LAUNCH((k_big_fwd16<false, false>), dim3(grid), dim3(256), big16_rows_lds(), s, m, d, g, B);
    }
}

void launch_big16_dw(hipStream_t s, const NetDims &m, const float *px, const float *ph1, const float *ph2, const float *pdz1,
                     const float *pdz2, const float *pdz3, int B, float *slab, const float *colsum, float *grad,
                     const float *loss_part, float *loss_out, DqnState *st, int bump_ctr, const AdamArgs &adam, int num_cus) {
    const int Bp64 = (B + 63) / 64 * 64;
#ifndef BIG16_DW_SLICE_DIV
#define BIG16_DW_SLICE_DIV 2                                       // 32 slices: slab 16.7 MB -- reduce 44 -> 24 us, dW 73 -> 82 us at B = 2^17
#endif
    const int KS = big_dw_slices(B, num_cus / BIG16_DW_SLICE_DIV);
    int rps = (Bp64 + KS - 1) / KS;
    rps = (rps + 63) / 64 * 64;                                       // whole row tiles per slice (rows >= B carry zero gradients)
    const int ks_used = (Bp64 + rps - 1) / rps;
    auto bf = [](const float *p) { return reinterpret_cast<const __bf16 *>(p); };
    DQN_LAUNCH(k_big_dw16, dim3(ks_used, BIG_DW_TILES), dim3(256), 0, s, bf(px), bf(ph1), bf(ph2), bf(pdz1), bf(pdz2), bf(pdz3), Bp64, rps, slab);
    const int wblocks = (int)((4 * m.P + 255) / 256), bblocks = (2 * HB + 1 + m.A + 3) / 4;
    DQN_LAUNCH((k_big_reduce<true>), dim3(wblocks + bblocks), dim3(256), 0, s, m, slab, ks_used, colsum, Bp64 / 64, B, grad, loss_part, loss_out, st,
               bump_ctr, adam, wblocks);
}
