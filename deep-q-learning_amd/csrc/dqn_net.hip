// csrc/dqn_net.hip -- dueling Q-network forward / backward / optimizer on gfx950 (exact-f32 path).
//
// All contractions run on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate): bit-for-bit a
// k-ordered fmaf chain, i.e. the same arithmetic as the CPU restatement, at the f32 MFMA rate.
// Operands are kept in HBM in *fragment order* so that a wave fetches the B (or A) operand
// of four consecutive MFMAs with one coalesced 16-B-per-lane load and no LDS staging:
//
//   packed(M)[(ct*KQ + kq)*256 + lane*4 + j] = M[16*kq + 4*j + (lane>>4)][16*ct + (lane&15)]
//
// for a [K x C] matrix M contracted over its FIRST index (K = 16*KQ rows, C = 16*CT columns).
// Weights are packed this way (K = fan-in) for the forward pass, their transposes for the
// backward pass, and every [batch x C] activation / row-gradient is written in the same
// format (K = batch) so that dW = H^T Z needs no transposition either.
//
// Reference: LunarLander/dddqn.py:24-34 (forward), General/QLearning/q_learning_functions.py
// :52-61 (targets), :35-36 (loss), :23-25 (grad + optimizer).
#include <type_traits>
#include "dqn_device.h"
#include "dqn_launch.h"
#include "dqn_per_device.h"
#include "dqn_net_common.h"


#ifdef DQN_STAMPS
__device__ unsigned long long g_stamps[8][64][2];
extern "C" int dqn_debug_stamps(unsigned long long *out_host) {
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#endif

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// position of column c (0..15) inside its 16-block of an LDS A-operand row: the lane group
// g = lane>>4 reads 4 consecutive floats = k-steps j = 0..3 of k = 4*j + g
__device__ __forceinline__ int perm16(int c) { return ((c & 3) << 2) | (c >> 2); }


// position, in a batch-major packed [B x C] matrix (K = batch), of accumulator element r of this lane for the MFMA tile
// (16-row batch tile `tile`, 16-column tile ct): pidx(KQb, 16*tile + 4*(lane>>4) + r, 16*ct + (lane&15)) with the
// divisions and masks folded away
__device__ __forceinline__ int pfrag(int KQb, int tile, int ct, int r, int lane) {
    return (ct * KQb + tile) * 256 + r * 64 + ((lane & 15) << 2) + (lane >> 4);
}

NetDims make_dims(int D, int H1, int H2, int A) {
    NetDims m{};
    m.D = D; m.H1 = H1; m.H2 = H2; m.A = A;
    m.KQ1 = (D + 15) / 16;
    long long p = 0;
    m.o_w1 = p; p += (long long)D * H1;  m.o_b1 = p; p += H1;
    m.o_w2 = p; p += (long long)H1 * H2; m.o_b2 = p; p += H2;
    m.o_wv = p; p += H2;                 m.o_bv = p; p += 1;
    m.o_wa = p; p += (long long)H2 * A;  m.o_ba = p; p += A;
    m.P = p;
    long long q = 0;
    m.p_w1 = q;  q += (long long)m.KQ1 * 16 * H1;       // K = D (padded), C = H1
    m.p_w2 = q;  q += (long long)H1 * H2;               // K = H1, C = H2
    m.p_wh = q;  q += (long long)H2 * 16;               // K = H2, C = 1+A (padded to 16)
    m.p_w2t = q; q += (long long)H2 * H1;               // K = H2, C = H1   (W2 transposed)
    m.p_wht = q; q += (long long)16 * H2;               // K = 1+A (padded), C = H2 (heads transposed)
    m.p_w2k = q; q += (long long)H1 * H2;               // k-packed W2 (4 consecutive k per column)
    m.pack_floats = q;
    return m;
}

// ------------------------------------------------------------------------ weight packing
// canonical flat params -> the five fragment-ordered shadows. One thread per packed element.
__global__ void __launch_bounds__(256)
k_pack(NetDims m, const float *__restrict__ P, float *__restrict__ pack) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.pack_floats) return;
    // decode: which shadow, then (k, c) from the packed position
    long long base; int KQ; int which;
    if (t < m.p_w2)       { base = m.p_w1;  KQ = m.KQ1;     which = 0; }
    else if (t < m.p_wh)  { base = m.p_w2;  KQ = m.H1 / 16; which = 1; }
    else if (t < m.p_w2t) { base = m.p_wh;  KQ = m.H2 / 16; which = 2; }
    else if (t < m.p_wht) { base = m.p_w2t; KQ = m.H2 / 16; which = 3; }
    else if (t < m.p_w2k) { base = m.p_wht; KQ = 1;         which = 4; }
    else {                                              // k-packed W2: u = (kq*H2 + c)*4 + j
        const long long u = t - m.p_w2k;
        const int j = (int)(u & 3), c = (int)((u >> 2) % m.H2), kq = (int)((u >> 2) / m.H2);
        pack[t] = P[m.o_w2 + (long long)(4 * kq + j) * m.H2 + c];
        return;
    }
    const long long u = t - base;
    const int j = (int)(u & 3), lane = (int)((u >> 2) & 63);
    const long long blk = u >> 8;
    const int kq = (int)(blk % KQ), ct = (int)(blk / KQ);
    const int k = 16 * kq + 4 * j + (lane >> 4), c = 16 * ct + (lane & 15);
    float v = 0.0f;
    switch (which) {
    case 0: if (k < m.D && c < m.H1) v = P[m.o_w1 + (long long)k * m.H1 + c]; break;
    case 1: v = P[m.o_w2 + (long long)k * m.H2 + c]; break;
    case 2: if (c == 0) v = P[m.o_wv + k]; else if (c <= m.A) v = P[m.o_wa + (long long)k * m.A + (c - 1)]; break;
    case 3: v = P[m.o_w2 + (long long)c * m.H2 + k]; break;                       // W2^T[k][c] = W2[c][k]
    case 4: if (k == 0) v = P[m.o_wv + c]; else if (k <= m.A) v = P[m.o_wa + (long long)c * m.A + (k - 1)]; break;
    }
    pack[t] = v;
}

void launch_pack(hipStream_t s, const NetDims &m, const float *params, float *pack) {
    const int blocks = (int)((m.pack_floats + 255) / 256);
    hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, s, m, params, pack);
}

// ------------------------------------------------------------------- MFMA layer helper
// acc[t] (t-th 16-column tile of this wave: ct = wave + 4*t) = lds_a[16 x 16*KQ] . packed W
// The packed-weight loads (L2 / Infinity-Cache latency) are software-pipelined PF k-blocks
// ahead in a ping-pong register buffer. Loads are unconditional (tile / k-block indices are
// clamped, surplus data is never used) so that they sit in straight-line code and can be issued
// early -- start() may be called before the activations of the layer exist. The k order of
// every accumulation chain is unchanged.
// PF = k-blocks per register chunk. PP = ping-pong two chunks (any KQ); !PP = the whole layer is ONE
// chunk (requires KQ <= PF: hidden sizes <= 256), so every weight of the layer is in flight at once.
template <int TN, int PF, bool PP>
struct MmaLayer {
    static constexpr int AHEAD = PF >= 16 ? 6 : PF;          // k-blocks requested before finish() (!PP)
    const float4 *pk[TN];
    float4 b0[PF][TN], b1[PP ? PF : 1][PP ? TN : 1];
    int KQ, CT, wave;

    template <int R, int C>
    __device__ __forceinline__ void load(float4 (&b)[R][C], int kq0) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            int kq = kq0 + p;
            kq = kq < KQ ? kq : KQ - 1;
#pragma unroll
            for (int t = 0; t < TN; ++t) b[p][t] = pk[t][(long long)kq * 64];
        }
    }
    template <int R, int C>
    __device__ __forceinline__ void compute(const float4 (&b)[R][C], int kq0, const float *arow, f32x4 (&acc)[TN]) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int kq = kq0 + p;
            if (kq < KQ) {
                const float4 a4 = *reinterpret_cast<const float4 *>(arow + 16 * kq);
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    acc[t] = MFMA4(a4.x, b[p][t].x, acc[t]);
                    acc[t] = MFMA4(a4.y, b[p][t].y, acc[t]);
                    acc[t] = MFMA4(a4.z, b[p][t].z, acc[t]);
                    acc[t] = MFMA4(a4.w, b[p][t].w, acc[t]);
                }
            }
        }
    }
    __device__ __forceinline__ void init(const float *__restrict__ pack, int KQ_, int CT_, int wave_, int lane) {
        KQ = KQ_; CT = CT_; wave = wave_;
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            int ct = wave + 4 * t;
            ct = ct < CT ? ct : CT - 1;                      // surplus tiles recompute the last one
            pk[t] = reinterpret_cast<const float4 *>(pack) + (long long)ct * KQ * 64 + lane;
        }
    }
    // issue the loads of k-blocks [P0, P1) of the first chunk (the order of issue is the order of arrival)
    template <int P0, int P1>
    __device__ __forceinline__ void load_range() {
#pragma unroll
        for (int p = P0; p < P1; ++p) {
            int kq = p < KQ ? p : KQ - 1;
#pragma unroll
            for (int t = 0; t < TN; ++t) b0[p][t] = pk[t][(long long)kq * 64];
        }
    }
    // issue the whole first chunk
    __device__ __forceinline__ void start(const float *__restrict__ pack, int KQ_, int CT_, int wave_, int lane) {
        init(pack, KQ_, CT_, wave_, lane);
        load(b0, 0);
    }
    // the heads' layer (one column tile, whole layer preloaded): the j-th MFMA of every k-block accumulates into chain j
    // (k with (k / 4) % 4 == j, ascending), combined as (c0 + c1) + (c2 + c3) -- four independent chains a quarter as long
    // as the k-ordered one (oracle: heads_row)
    __device__ __forceinline__ void finish_heads4(const float *lds_a, int stride, int lane, f32x4 &out) {
        static_assert(TN == 1 && !PP, "heads: one tile, single chunk");
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
        const float *arow = lds_a + (lane & 15) * stride + 4 * (lane >> 4);
        float4 a_cur = *reinterpret_cast<const float4 *>(arow);
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if (p < KQ) {
                const float4 a4 = a_cur;
                if (p + 1 < PF) a_cur = *reinterpret_cast<const float4 *>(arow + (p + 1 < KQ ? 16 * (p + 1) : 0));
                c0 = MFMA4(a4.x, b0[p][0].x, c0);
                c1 = MFMA4(a4.y, b0[p][0].y, c1);
                c2 = MFMA4(a4.z, b0[p][0].z, c2);
                c3 = MFMA4(a4.w, b0[p][0].w, c3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r] = (c0[r] + c1[r]) + (c2[r] + c3[r]);
    }
    // preloaded = every k-block was already requested with load_range<0, PF>()
    __device__ __forceinline__ void finish(const float *lds_a, int stride, int lane, f32x4 (&acc)[TN], bool preloaded = false) {
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *arow = lds_a + (lane & 15) * stride + 4 * (lane >> 4);
        if constexpr (PP) {
            for (int kq0 = 0; kq0 < KQ; kq0 += 2 * PF) {
                if (kq0 + PF < KQ) load(b1, kq0 + PF);               // (nothing to prefetch past the end: layer 1 of a small obs_dim is one chunk)
                compute(b0, kq0, arow, acc);
                if (kq0 + 2 * PF < KQ) load(b0, kq0 + 2 * PF);
                compute(b1, kq0 + PF, arow, acc);
            }
        } else {
            // single chunk; k-blocks [0, AHEAD) were requested by load_range<0, AHEAD>() long ago, block
            // p + AHEAD is requested right before block p is multiplied: the address traffic of the layer
            // is spread between the MFMAs instead of blocking the wave's issue in one burst
            // the A fragment of k-block p + 1 is read from LDS before the MFMAs of block p (its latency hides behind them;
            // past the last block the read lands in the row's padding and is not used)
            float4 a_cur = *reinterpret_cast<const float4 *>(arow);
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                if (p < KQ) {
                    if (p + AHEAD < PF && !preloaded) {
                        const int kq = p + AHEAD < KQ ? p + AHEAD : KQ - 1;
#pragma unroll
                        for (int t = 0; t < TN; ++t) b0[p + AHEAD < PF ? p + AHEAD : 0][t] = pk[t][(long long)kq * 64];
                    }
                    const float4 a4 = a_cur;
                    if (p + 1 < PF) a_cur = *reinterpret_cast<const float4 *>(arow + (p + 1 < KQ ? 16 * (p + 1) : 0));
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        acc[t] = MFMA4(a4.x, b0[p][t].x, acc[t]);
                        acc[t] = MFMA4(a4.y, b0[p][t].y, acc[t]);
                        acc[t] = MFMA4(a4.z, b0[p][t].z, acc[t]);
                        acc[t] = MFMA4(a4.w, b0[p][t].w, acc[t]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
};

// ------------------------------------------------------------------------------ forward
// One workgroup (4 waves) = 16 batch rows of one pass. blockIdx.y selects the pass
// (online(s), online(s'), target(s') for compute_q_targets :52-54).
struct FwdPasses { FwdPass p[3]; };

// FUSE: the pass-0 workgroup of a tile goes on with the tile's row backward (what k_bwd_rows does as its own launch) once
// the tile's other two passes have handed over their Q rows: one launch boundary, one prologue and one drain less per
// update. The hand-over is 2 x 16 x A floats per tile: written with L1-bypassing stores and drained before the writer
// bumps the tile's counter, read with L1-bypassing loads after the reader has seen the counter (MI355X_MICROARCH.md,
// inter-workgroup visibility: the sc1 store / sc1 load form). Only for grids that are resident as a whole (3 * tiles <=
// 256 workgroups, one per CU): a waiting workgroup then never keeps its partners from being scheduled.

template <int TN1, int TN2, bool FUSE>
__global__ void __launch_bounds__(256)
k_qnet_fwd(NetDims m, FwdPasses passes, int B, SampleArgs smp, FuseBwd fb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FwdPass ps = passes.p[blockIdx.y];
    const int tile = blockIdx.x, row0 = tile * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KQb = (B + 15) / 16;
    const int K1 = m.KQ1 * 16;
    const int sx = K1 + 4, s1 = m.H1 + 4, s2 = m.H2 + 4;
    float *lx = lds, *l1 = lx + 16 * sx, *l2 = l1 + 16 * s1, *lh = l2 + 16 * s2;   // lh: [16][16]

    STAMP(0, 0);
    // Vector-memory returns are in order, so the order of issue is the order of arrival: the 16 input
    // rows first, then layer-1 weights + biases, then all of layer 2 and the heads (they stream in
    // while layer 1 runs; nothing below waits for more than it needs).
    const bool sampling = smp.st != nullptr;
    int *lidx = reinterpret_cast<int *>(lh + 256 + 16);            // leaves of this tile's rows (sampling mode)
    const float *ring = sampling ? (ps.src == 1 ? smp.states : smp.observations) : nullptr;
    auto xload = [&](int t) -> float {
        const int rl = t / K1, c = t - rl * K1;
        if (!(t < 16 * K1 && row0 + rl < B && c < m.D)) return 0.0f;
        return sampling ? ring[(long long)lidx[rl] * m.D + c] : ps.x[(long long)(row0 + rl) * m.D + c];
    };
    float xv[4];
    if (!sampling) {
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = xload(tid + 256 * u);
    }
    // batch drawn by the preceding actor launch (dqn_actor.hip): the index is the head of the dependent chain
    // idx -> ring row, so it is requested before anything else
    const bool presampled = sampling && smp.pre;
    int pre_leaf = 0;
    if (presampled && tid < 16) pre_leaf = smp.idx[row0 + tid < B ? row0 + tid : B - 1];
    MmaLayer<TN1, 2, true> L1; MmaLayer<TN2, 16, false> L2; MmaLayer<1, 16, false> LH;
    MmaLayer<TN2, 1, false> LA; MmaLayer<TN1, 16, false> LB;         // row backward (FUSE, pass 0)
    L1.start(ps.pack + m.p_w1, m.KQ1, m.H1 / 16, wave, lane);
    float bias1[TN1], bias2[TN2], biash = 0.0f;
#pragma unroll
    for (int t = 0; t < TN1; ++t) { int ct = wave + 4 * t; ct = ct < m.H1 / 16 ? ct : m.H1 / 16 - 1; bias1[t] = ps.params[m.o_b1 + 16 * ct + (lane & 15)]; }
#pragma unroll
    for (int t = 0; t < TN2; ++t) { int ct = wave + 4 * t; ct = ct < m.H2 / 16 ? ct : m.H2 / 16 - 1; bias2[t] = ps.params[m.o_b2 + 16 * ct + (lane & 15)]; }
    if (wave == 0) {
        const int c = lane & 15;
        if (c == 0) biash = ps.params[m.o_bv];
        else if (c <= m.A) biash = ps.params[m.o_ba + c - 1];
    }
    int row_a = 0; float row_r = 0.0f; int row_d = 0;               // a, r, d of this thread's row (FUSE, pass 0)
    if (presampled) {
        // rows of this tile: requested ahead of the layer-2 weights (in-order returns); pass 0 publishes a, r, d
        if (tid < 16) {
            lidx[tid] = pre_leaf;
            const int k = row0 + tid;
            if (blockIdx.y == 0 && k < B) {
                row_a = smp.actions[pre_leaf]; row_r = smp.rewards[pre_leaf]; row_d = smp.dones[pre_leaf];
                smp.a[k] = row_a; smp.r[k] = row_r; smp.d[k] = (uint8_t)row_d;
            }
        }
        LDS_BARRIER();
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = xload(tid + 256 * u);
    }
    L2.init(ps.pack + m.p_w2, m.H1 / 16, m.H2 / 16, wave, lane);
    // the first AHEAD k-blocks; the rest interleave with the MFMAs (requesting the whole layer here was measured equal:
    // the issue burst delays x by what it saves later -- the 256 KB weight stream per workgroup is the bound either way)
    L2.template load_range<0, 6>();
    if (wave == 0) LH.init(ps.pack + m.p_wh, m.H2 / 16, 1, 0, lane);
    if (sampling && !presampled) {
        // the weight requests above are in flight; now the dependent tree descent, then the gathered rows
        if (smp.tree) {
            float *lsub = reinterpret_cast<float *>(lidx + 16);
            sample_tile_coop(smp, row0, B, tid, blockIdx.y == 0, lidx, lsub, lsub + 512);     // ends with a barrier
        } else {
            if (tid < 16) sample_tile(smp, row0, B, tid, blockIdx.y == 0, lidx);
            LDS_BARRIER();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = xload(tid + 256 * u);
        if constexpr (FUSE) {
            if (blockIdx.y == 0 && tid < 16 && row0 + tid < B) {
                const int leaf = lidx[tid];
                row_a = smp.actions[leaf]; row_r = smp.rewards[leaf]; row_d = smp.dones[leaf];
            }
        }
    }

    // stage the 16 input rows (zero-padded) in A-operand order
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = tid + 256 * u;
        if (t < 16 * K1) {
            const int rl = t / K1, c = t - rl * K1;
            lx[rl * sx + (c & ~15) + perm16(c & 15)] = xv[u];
        }
    }
    for (int t = tid + 1024; t < 16 * K1; t += 256) {            // obs_dim > 64 only
        const int rl = t / K1, c = t - rl * K1;
        const float v = xload(t);
        lx[rl * sx + (c & ~15) + perm16(c & 15)] = v;
    }
    LDS_BARRIER();
    STAMP(0, 1);

    unsigned m1bits = 0u, m2bits = 0u;                               // ReLU gates of this lane's accumulator elements (FUSE)
    // layer 1: h1 = relu(x @ w1 + b1)                                      dddqn.py:25-26
    {
        f32x4 acc[TN1];
        L1.finish(lx, sx, lane, acc);
        STAMP(0, 11);
#pragma unroll
        for (int t = 0; t < TN1; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H1 / 16) {
                const int c = lane & 15;
                const float bias = bias1[t];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * (lane >> 4) + r;
                    float v = acc[t][r] + bias;
                    v = v > 0.0f ? v : 0.0f;
                    if constexpr (FUSE) m1bits |= (v > 0.0f ? 1u : 0u) << (4 * t + r);
                    l1[rl * s1 + 16 * ct + perm16(c)] = v;
                }
            }
        }
    }
    STAMP(0, 12);
    if (wave == 0) LH.template load_range<0, 16>();          // heads' weights: wave 0 only, 16 requests
    STAMP(0, 13);
    LDS_BARRIER();
    STAMP(0, 2);

    // layer 2: h2 = relu(h1 @ w2 + b2)                                     dddqn.py:27-28
    {
        f32x4 acc[TN2];
        L2.finish(l1, s1, lane, acc);
        STAMP(0, 3);
#pragma unroll
        for (int t = 0; t < TN2; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H2 / 16) {
                const int c = lane & 15, col = 16 * ct + c;
                const float bias = bias2[t];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * (lane >> 4) + r;
                    float v = acc[t][r] + bias;
                    v = v > 0.0f ? v : 0.0f;
                    if constexpr (FUSE) m2bits |= (v > 0.0f ? 1u : 0u) << (4 * t + r);
                    l2[rl * s2 + 16 * ct + perm16(c)] = v;
                    if (ps.feat && row0 + rl < B) ps.feat[(long long)(row0 + rl) * m.H2 + col] = v;   // :32-33
                }
            }
        }
    }
    LDS_BARRIER();
    STAMP(0, 4);

    // The batch-major stashes of x / h1 / h2 for k_dw (pass 0 of an update): written now, out of the LDS images, by the
    // three waves that would idle behind the heads -- one 16-B store per four rows of a column (the operand order of
    // k_dw: float4 number (ct*KQb + tile)*64 + lane' holds rows 4j + (lane'>>4), j = 0..3, of column 16ct + (lane'&15))
    // instead of one 4-B store per accumulator element inside the layer epilogues, where they queued behind the
    // weight prefetch.
    if (ps.px && wave != 0) {
        auto stash = [&](float *dst, const float *img, int stride, int CT) {
            float4 *out = reinterpret_cast<float4 *>(dst);
            for (int q = tid - 64; q < CT * 64; q += 192) {
                const int ct = q >> 6, lp = q & 63;
                const float *col = img + (lp >> 4) * stride + 16 * ct + perm16(lp & 15);
                out[(long long)(ct * KQb + tile) * 64 + lp] = float4{col[0], col[4 * stride], col[8 * stride], col[12 * stride]};
            }
        };
        stash(ps.px, lx, sx, m.KQ1);
        stash(ps.ph1, l1, s1, m.H1 / 16);
        stash(ps.ph2, l2, s2, m.H2 / 16);
    }
    // heads: column 0 = val (dddqn.py:29), columns 1..A = adv (:30); one 16-column tile
    if (wave == 0) {
        f32x4 acc;
        LH.finish_heads4(l2, s2, lane, acc);
        const int c = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) lh[(4 * (lane >> 4) + r) * 16 + c] = acc[r] + biash;
    }
    if constexpr (FUSE) {
        // the row backward's weights ([wv|wa]^T, then all of W2^T into the registers layer 2 has freed): waves 1..3 request
        // them while they would idle behind the heads, wave 0 right after its heads; they land behind the TD rows
        if (blockIdx.y == 0) {
            LA.start(fb.g.pack + m.p_wht, 1, m.H2 / 16, wave, lane);
            LB.init(fb.g.pack + m.p_w2t, m.H2 / 16, m.H1 / 16, wave, lane);
            LB.template load_range<0, 16>();
        }
    }
    LDS_BARRIER();
    STAMP(0, 5);

    // Q = val + adv - mean(adv)                                            dddqn.py:31
    float qrow[16];
    if (tid < 16 && row0 + tid < B) {
        const float *hr = lh + tid * 16;
        float sum = 0.0f;
        if constexpr (FUSE) {                                        // statically indexed (the row backward keeps qrow in registers)
            if (m.A <= 4) {                                          // (uniform branch: 4 select-guarded links instead of 15)
#pragma unroll
                for (int a = 0; a < 4; ++a) if (a < m.A) sum = sum + hr[1 + a];
            } else {
#pragma unroll
                for (int a = 0; a < 15; ++a) if (a < m.A) sum = sum + hr[1 + a];
            }
        } else {
            for (int a = 0; a < m.A; ++a) sum = sum + hr[1 + a];
        }
        const float mean = __fdiv_rn(sum, (float)m.A);
        if constexpr (FUSE) {
#pragma unroll
            for (int a = 0; a < 15; ++a) qrow[a] = a < m.A ? (hr[0] + hr[1 + a]) - mean : 0.0f;
        } else {
            for (int a = 0; a < m.A; ++a) qrow[a] = (hr[0] + hr[1 + a]) - mean;
        }
        if constexpr (FUSE) {
            if (blockIdx.y != 0) {
                // the partners' Q rows, all stores of a row back to back: written as a runtime loop the compiler puts a vmcnt(0)
                // in front of every store, i.e. each waits for the acknowledgement of the one before (stores count in vmcnt)
#pragma unroll
                for (int a = 0; a < 15; ++a) asm volatile("" : "+v"(qrow[a]));
#pragma unroll
                for (int a = 0; a < 15; ++a)
                    if (a < m.A) __hip_atomic_store(&ps.q[(long long)(row0 + tid) * m.A + a], qrow[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (ps.q) {
#pragma unroll
                for (int a = 0; a < 15; ++a) if (a < m.A) ps.q[(long long)(row0 + tid) * m.A + a] = qrow[a];
            }
        } else if (ps.q) for (int a = 0; a < m.A; ++a) ps.q[(long long)(row0 + tid) * m.A + a] = qrow[a];
        if (ps.act_out) {
            const float eps = ps.act_state ? ps.act_state->epsilon : ps.act_eps;
            const unsigned long long ctr = ps.act_state ? ps.act_state->env_ctr : ps.act_ctr;
            const int act = policy_row(qrow, m.A, eps, ps.act_seed, ctr, row0 + tid);
            ps.act_out[row0 + tid] = act;
        }
    }
    STAMP(0, 6);
    if constexpr (FUSE) {
        if (blockIdx.y != 0) {
            // passes 1, 2: the Q rows above were L1-bypassing stores of wave 0; drain them, then count this pass in
            if (wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0) atomicAdd(reinterpret_cast<unsigned *>(fb.tile_cnt) + tile, fb.withhold ? 0u : 1u);
            }
            return;
        }
        // ---- pass 0: TD target / Huber gradient / row backward of this tile (the body of k_bwd_rows)
        const BwdArgs &g = fb.g;
        const int A = m.A;
        const int s3 = 16 + 4;
        float *l3 = lx;                                              // [16][20] (x is dead), 16 * sx >= 16 * 20
        float *lz2 = lh + 256 + 32 + 528, *lrow = lz2 + 16 * s2;      // dz2 [16][s2], per-row loss [16]
        const int irow = row0 + tid;
        const bool rowt = tid < 16 && irow < B;
        float wi = 1.0f, wmax = 1.0f;
        if (rowt && g.w_raw) { wi = g.w_raw[irow]; wmax = fb.st->wmax; }
        for (int t = tid; t < 16 * s3; t += 256) l3[t] = 0.0f;
        bool gave_up = false;                                        // (thread 0)
        if (tid == 0) {
            // bounded wait (dqn_device.h): the launcher only fuses grids that are resident as a whole, so the partners are
            // running; should that ever not hold, the kernel still ends, the loss turns NaN and the error count goes up
            // instead of the GPU hanging. The counter only grows (2 per launch, wrapping); the tile's pass-0 workgroup keeps
            // what it has already consumed in seen[]: a late partner of a timed-out launch can never satisfy a later wait.
            unsigned *cnt = reinterpret_cast<unsigned *>(fb.tile_cnt) + tile, *seen = reinterpret_cast<unsigned *>(fb.tile_cnt) + fb.tiles + tile;
            const unsigned want = *seen + 2u;
            if (!wait_word_eq(cnt, want, 2)) { flag_wait_timeout(fb.st); gave_up = true; }
            *seen = want;
        }
        LDS_BARRIER();                                               // partners' rows are in L2; l3 is zeroed
        STAMP(0, 7);
        if (tid < 16) {
            float rowloss = 0.0f;
            if (rowt) {
                auto td_rows = [&](auto amax_tag) {
                constexpr int AMAX = decltype(amax_tag)::value;
                // q_learning_functions.py:55-60 + :35-36 for one row, written with statically indexed registers
                // (unrolled to AMAX = 4 or the 15-action maximum, predicated on k < A): same operations in the same order as
                // td_row() / k_bwd_rows
                float nqr[AMAX], ntr[AMAX];
#pragma unroll
                for (int k2 = 0; k2 < AMAX; ++k2) {
                    nqr[k2] = k2 < A ? __hip_atomic_load(&g.nq[(long long)irow * A + k2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
                    ntr[k2] = k2 < A ? __hip_atomic_load(&g.nt[(long long)irow * A + k2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
                }
                float w = 1.0f;
                if (g.w_raw) w = __fdiv_rn(wi, wmax);
                const float invB = __fdiv_rn(1.0f, (float)B);
                float best = nqr[0], nt_star = ntr[0], q_a = qrow[0];   // :55 argmax, first max wins; q[action]
#pragma unroll
                for (int k2 = 1; k2 < AMAX; ++k2) {
                    if (k2 < A && nqr[k2] > best) { best = nqr[k2]; nt_star = ntr[k2]; }
                    if (k2 == row_a) q_a = qrow[k2];
                }
                const float di = row_d ? 1.0f : 0.0f;                   // preprocessing :84
                const float t1 = g.gamma * nt_star;                      // :58, quirk Q3: (1-d) covers -q too
                const float t2 = t1 - q_a;
                const float t3 = (1.0f - di) * t2;
                float delta = row_r + t3, dabs = fabsf(delta);
                // the three per-row outputs, values first: each `if (ptr) store` block otherwise opens with a vmcnt(0) (for the
                // loads delta hangs on) that waits for the store of the block before it
                asm volatile("" : "+v"(delta), "+v"(dabs), "+v"(w));
                if (g.w_raw && g.isw_out) g.isw_out[irow] = w;
                if (g.td) g.td[irow] = delta;
                if (g.td_abs) g.td_abs[irow] = dabs;
                float gk[AMAX], gsum = 0.0f;
#pragma unroll
                for (int k2 = 0; k2 < AMAX; ++k2) {
                    gk[k2] = 0.0f;
                    if (k2 < A) {
                        const float trk = qrow[k2] + delta * (k2 == row_a ? 1.0f : 0.0f);   // :59, quirk Q4
                        const float e = qrow[k2] - trk;                 // pred - target, pred == q   (:35)
                        rowloss = rowloss + huber(e);                   // :36
                        const float c = e > 1.0f ? 1.0f : (e < -1.0f ? -1.0f : e);
                        gk[k2] = (w * c) * invB;                        // dL/dpred
                        gsum = gsum + gk[k2];
                        if (g.dq) g.dq[(long long)irow * A + k2] = gk[k2];
                        if (g.targets_out) g.targets_out[(long long)irow * A + k2] = trk;
                    }
                }
                if (g.w_raw) rowloss = w * rowloss;
                const float gmean = __fdiv_rn(gsum, (float)A);           // dueling backward: dv = sum_a g_a ; dadv_j = g_j - mean
                l3[tid * s3 + perm16(0)] = gsum;
#pragma unroll
                for (int k2 = 0; k2 < AMAX; ++k2) if (k2 < A) l3[tid * s3 + perm16(1 + k2)] = gk[k2] - gmean;
                };
                if (A <= 4) td_rows(std::integral_constant<int, 4>{}); else td_rows(std::integral_constant<int, 15>{});
            }
            lrow[tid] = rowloss;
        }
        LDS_BARRIER();
        STAMP(0, 8);
        if (tid == 0) {
            float sl = 0.0f;
            for (int k = 0; k < 16; ++k) sl = sl + lrow[k];
            g.loss_part[tile] = gave_up ? __int_as_float(0x7fc00000) : sl;   // a wait that gave up poisons THIS launch's loss (r02 kept
                                                                             // the NaN by re-reading the slot: sticky across launches)
        }
        {   // stash dz3 (packed, K = batch, C = 16)
            const int rl = tid >> 4, c = tid & 15;
            g.pdz3[pidx(KQb, row0 + rl, c)] = l3[rl * s3 + perm16(c)];
        }
        {   // dz2 = (dz3 . WH^T) * (h2 > 0)
            f32x4 acc[TN2];
            LA.finish(l3, s3, lane, acc);
#pragma unroll
            for (int t = 0; t < TN2; ++t) {
                const int ct = wave + 4 * t;
                if (ct < m.H2 / 16) {
                    const int c = lane & 15;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rl = 4 * (lane >> 4) + r;
                        const float v = ((m2bits >> (4 * t + r)) & 1u) ? acc[t][r] : 0.0f;
                        lz2[rl * s2 + 16 * ct + perm16(c)] = v;
                        g.pdz2[pfrag(KQb, tile, ct, r, lane)] = v;
                    }
                }
            }
        }
        LDS_BARRIER();
        STAMP(0, 9);
        {   // dz1 = (dz2 . W2^T) * (h1 > 0)
            f32x4 acc[TN1];
            LB.finish(lz2, s2, lane, acc, true);
#pragma unroll
            for (int t = 0; t < TN1; ++t) {
                const int ct = wave + 4 * t;
                if (ct < m.H1 / 16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) g.pdz1[pfrag(KQb, tile, ct, r, lane)] = ((m1bits >> (4 * t + r)) & 1u) ? acc[t][r] : 0.0f;
                }
            }
        }
        STAMP(0, 10);
    }
}

static inline int tn_of(int H) { const int ct = H / 16; return ct <= 4 ? 1 : (ct <= 8 ? 2 : 4); }

void launch_qnet_fwd(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, const SampleArgs *smp,
                     const BwdArgs *fuse, int *tile_cnt, DqnState *st, int tile_stride, int withhold) {
    FwdPasses ps{};
    for (int i = 0; i < npass; ++i) ps.p[i] = passes[i];
    const SampleArgs sa = smp ? *smp : SampleArgs{};
    const dim3 grid((B + 15) / 16, npass), block(256);
    size_t lds = sizeof(float) * (16 * (m.KQ1 * 16 + 4) + 16 * (m.H1 + 4) + 16 * (m.H2 + 4) + 256 + 32 + 528);
    const int t1 = tn_of(m.H1), t2 = tn_of(m.H2);
    if (fuse) {
        FuseBwd fb{*fuse, tile_cnt, st, tile_stride, withhold};
        lds += sizeof(float) * (16 * (m.H2 + 4) + 16);
#define FWD_CASE(A1, A2) if (t1 == A1 && t2 == A2) { DQN_LAUNCH((k_qnet_fwd<A1, A2, true>), grid, block, lds, s, m, ps, B, sa, fb); return; }
        FWD_CASE(1, 1) FWD_CASE(1, 2) FWD_CASE(1, 4) FWD_CASE(2, 1) FWD_CASE(2, 2) FWD_CASE(2, 4)
        FWD_CASE(4, 1) FWD_CASE(4, 2) FWD_CASE(4, 4)
#undef FWD_CASE
    }
    const FuseBwd fb{};
#define FWD_CASE(A1, A2) if (t1 == A1 && t2 == A2) { DQN_LAUNCH((k_qnet_fwd<A1, A2, false>), grid, block, lds, s, m, ps, B, sa, fb); return; }
    FWD_CASE(1, 1) FWD_CASE(1, 2) FWD_CASE(1, 4) FWD_CASE(2, 1) FWD_CASE(2, 2) FWD_CASE(2, 4)
    FWD_CASE(4, 1) FWD_CASE(4, 2) FWD_CASE(4, 4)
#undef FWD_CASE
}

// --------------------------------------------------------------- per-sample TD arithmetic
// standalone elementwise kernel behind dqn_td_targets (one thread per row). loss: per-row
// weighted Huber sums go to scratch[B]; a single-block pass reduces them in a fixed order.
__global__ void __launch_bounds__(256)
k_td(const float *__restrict__ q, const float *__restrict__ nq, const float *__restrict__ nt,
     const int32_t *__restrict__ a, const float *__restrict__ r, const float *__restrict__ d,
     const float *__restrict__ isw, float gamma, int B, int A, float *targets, float *td, float *dq,
     float *scratch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float qr[16], nqr[16], ntr[16], tr[16];
    for (int k = 0; k < A; ++k) { qr[k] = q[(long long)i * A + k]; nqr[k] = nq[(long long)i * A + k]; ntr[k] = nt[(long long)i * A + k]; }
    const float delta = td_row(qr, nqr, ntr, a[i], r[i], d[i], gamma, A, tr);
    const float w = isw ? isw[i] : 1.0f;
    const float invB = __fdiv_rn(1.0f, (float)B);
    float row = 0.0f;
    for (int k = 0; k < A; ++k) {
        const float e = qr[k] - tr[k];
        row = row + huber(e);
        const float c = e > 1.0f ? 1.0f : (e < -1.0f ? -1.0f : e);
        if (dq) dq[(long long)i * A + k] = (w * c) * invB;
        if (targets) targets[(long long)i * A + k] = tr[k];
    }
    if (td) td[i] = delta;
    if (scratch) scratch[i] = isw ? w * row : row;
}

// deterministic mean: one block, each thread sums a strided slice in order, then a tree
__global__ void __launch_bounds__(256)
k_mean(const float *__restrict__ v, int n, float denom, float *out) {
    __shared__ float red[256];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) s = s + v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = __fdiv_rn(red[0], denom);
}

void launch_td(hipStream_t s, const float *q, const float *nq, const float *nt, const int32_t *a,
               const float *r, const float *d, const float *isw, float gamma, int B, int A,
               float *targets, float *td, float *dq, float *loss, float *scratch) {
    hipLaunchKernelGGL(k_td, dim3((B + 255) / 256), dim3(256), 0, s, q, nq, nt, a, r, d, isw, gamma, B, A,
                       targets, td, dq, loss ? scratch : nullptr);
    if (loss) hipLaunchKernelGGL(k_mean, dim3(1), dim3(256), 0, s, scratch, B, (float)B, loss);
}

// compute_loss (:35-36) given pred: mean_i w_i sum_a huber(pred - target)
__global__ void __launch_bounds__(256)
k_loss(const float *__restrict__ pred, const float *__restrict__ targets, const float *__restrict__ isw,
       int B, int A, float *out) {
    __shared__ float red[256];
    float s = 0.0f;
    for (int i = threadIdx.x; i < B; i += 256) {
        float row = 0.0f;
        for (int k = 0; k < A; ++k) row = row + huber(pred[(long long)i * A + k] - targets[(long long)i * A + k]);
        s = s + (isw ? isw[i] * row : row);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = __fdiv_rn(red[0], (float)B);
}

void launch_loss(hipStream_t s, const float *pred, const float *targets, const float *isw, int B, int A,
                 float *loss) {
    hipLaunchKernelGGL(k_loss, dim3(1), dim3(256), 0, s, pred, targets, isw, B, A, loss);
}

// -------------------------------------------------------------- row-wise backward pass
// One workgroup = 16 batch rows: TD / Huber gradient (fused mode) or pred - target (parity
// mode), dueling backward, then dz2 = (dz3 . WH^T) * relu'(h2), dz1 = (dz2 . W2^T) * relu'(h1).
// Row gradients leave in batch-major packed form for k_dw.
template <int TN1, int TN2>
__global__ void __launch_bounds__(256)
k_bwd_rows(NetDims m, BwdArgs g, int B, DqnState *st) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tile = blockIdx.x, row0 = tile * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KQb = (B + 15) / 16;
    const int A = m.A;
    const int s3 = 16 + 4, s2 = m.H2 + 4;
    float *l3 = lds, *l2 = l3 + 16 * s3, *lrow = l2 + 16 * s2;     // lrow[16]: per-row loss

    STAMP(5, 0);
    // In-order returns again: the few scattered TD inputs of the 16 row-threads are issued FIRST,
    // then the transposed-weight chunks and the post-ReLU activations that gate the ReLU derivatives.
    const int irow = row0 + tid;
    const bool rowt = tid < 16 && irow < B;
    float qr[16], tr[16], nqr[16], ntr[16];
    int ai = 0; float ri = 0.0f, di = 0.0f, wi = 1.0f, wmax = 1.0f;
    if (rowt) {
        for (int k2 = 0; k2 < A; ++k2) qr[k2] = g.q[(long long)irow * A + k2];
        if (g.a) ai = g.a[irow];
        if (g.targets) {
            for (int k2 = 0; k2 < A; ++k2) tr[k2] = g.targets[(long long)irow * A + k2];
        } else {
            for (int k2 = 0; k2 < A; ++k2) { nqr[k2] = g.nq[(long long)irow * A + k2]; ntr[k2] = g.nt[(long long)irow * A + k2]; }
            ri = g.r[irow];
            di = g.d_f32 ? g.d_f32[irow] : (g.d_u8[irow] ? 1.0f : 0.0f);       // preprocessing :84
        }
        if (g.w_raw) { wi = g.w_raw[irow]; wmax = st->wmax; }     // batch max: atomicMax'ed by the sampler
        else if (g.isw) wi = g.isw[irow];
    }
    MmaLayer<TN2, 1, false> LA; MmaLayer<TN1, 16, false> LB;
    LA.start(g.pack + m.p_wht, 1, m.H2 / 16, wave, lane);
    float m2[TN2][4], m1[TN1][4];
#pragma unroll
    for (int t = 0; t < TN2; ++t) {
        int ct = wave + 4 * t; ct = ct < m.H2 / 16 ? ct : m.H2 / 16 - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) m2[t][r] = g.ph2[pfrag(KQb, tile, ct, r, lane)];
    }
    // the layer-1 ReLU gates and the first k-blocks of W2^T are requested now: they stream in behind the TD arithmetic
#pragma unroll
    for (int t = 0; t < TN1; ++t) {
        int ct = wave + 4 * t; ct = ct < m.H1 / 16 ? ct : m.H1 / 16 - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) m1[t][r] = g.ph1[pfrag(KQb, tile, ct, r, lane)];
    }
    LB.init(g.pack + m.p_w2t, m.H2 / 16, m.H1 / 16, wave, lane);
    LB.template load_range<0, 6>();
    STAMP(5, 1);
    for (int t = tid; t < 16 * s3; t += 256) l3[t] = 0.0f;
    LDS_BARRIER();

    if (tid < 16) {
        float rowloss = 0.0f;
        if (rowt) {
            const int i = irow;
            if (!g.targets) {
                const float delta = td_row(qr, nqr, ntr, ai, ri, di, g.gamma, A, tr);
                if (g.td) g.td[i] = delta;
                if (g.td_abs) g.td_abs[i] = fabsf(delta);
            }
            float w = 1.0f;
            if (g.w_raw) { w = __fdiv_rn(wi, wmax); if (g.isw_out) g.isw_out[i] = w; }
            else if (g.isw) w = wi;
            const float invB = __fdiv_rn(1.0f, (float)B);
            float gk[16], gsum = 0.0f;
            for (int k2 = 0; k2 < A; ++k2) {
                const float e = qr[k2] - tr[k2];               // pred - target, pred == q   (:35)
                rowloss = rowloss + huber(e);                  // :36
                const float c = e > 1.0f ? 1.0f : (e < -1.0f ? -1.0f : e);
                gk[k2] = (w * c) * invB;                       // dL/dpred
                gsum = gsum + gk[k2];
                if (g.dq) g.dq[(long long)i * A + k2] = gk[k2];
                if (g.targets_out) g.targets_out[(long long)i * A + k2] = tr[k2];
            }
            if (g.w_raw || g.isw) rowloss = w * rowloss;
            // dueling backward: dv = sum_a g_a ; dadv_j = g_j - (1/A) sum_a g_a
            const float gmean = __fdiv_rn(gsum, (float)A);
            l3[tid * s3 + perm16(0)] = gsum;
            for (int k2 = 0; k2 < A; ++k2) l3[tid * s3 + perm16(1 + k2)] = gk[k2] - gmean;
        }
        lrow[tid] = rowloss;
    }
    LDS_BARRIER();
    STAMP(5, 2);
    if (tid == 0) {
        float s = 0.0f;
        for (int k = 0; k < 16; ++k) s = s + lrow[k];
        g.loss_part[tile] = s;
    }
    // stash dz3 (packed, K = batch, C = 16)
    for (int t = tid; t < 256; t += 256) {
        const int rl = t >> 4, c = t & 15;
        g.pdz3[pidx(KQb, row0 + rl, c)] = l3[rl * s3 + perm16(c)];
    }

    // dz2 = (dz3 . WH^T) * (h2 > 0)
    {
        f32x4 acc[TN2];
        LA.finish(l3, s3, lane, acc);
#pragma unroll
        for (int t = 0; t < TN2; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H2 / 16) {
                const int c = lane & 15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * (lane >> 4) + r;
                    const float v = m2[t][r] > 0.0f ? acc[t][r] : 0.0f;
                    l2[rl * s2 + 16 * ct + perm16(c)] = v;
                    g.pdz2[pfrag(KQb, tile, ct, r, lane)] = v;
                }
            }
        }
    }
    LDS_BARRIER();
    STAMP(5, 3);

    // dz1 = (dz2 . W2^T) * (h1 > 0)
    {
        f32x4 acc[TN1];
        LB.finish(l2, s2, lane, acc);
#pragma unroll
        for (int t = 0; t < TN1; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H1 / 16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    g.pdz1[pfrag(KQb, tile, ct, r, lane)] = m1[t][r] > 0.0f ? acc[t][r] : 0.0f;
                }
            }
        }
    }
    STAMP(5, 4);
}

void launch_bwd_rows(hipStream_t s, const NetDims &m, const BwdArgs &g, int B, DqnState *st) {
    const dim3 grid((B + 15) / 16), block(256);
    const size_t lds = sizeof(float) * (16 * 20 + 16 * (m.H2 + 4) + 16);
    const int t1 = tn_of(m.H1), t2 = tn_of(m.H2);
#define BWD_CASE(A1, A2) if (t1 == A1 && t2 == A2) { DQN_LAUNCH((k_bwd_rows<A1, A2>), grid, block, lds, s, m, g, B, st); return; }
    BWD_CASE(1, 1) BWD_CASE(1, 2) BWD_CASE(1, 4) BWD_CASE(2, 1) BWD_CASE(2, 2) BWD_CASE(2, 4)
    BWD_CASE(4, 1) BWD_CASE(4, 2) BWD_CASE(4, 4)
#undef BWD_CASE
}

// ---------------------------------------------------------------------------- optimizer
// optax scale_by_adam -> add_decayed_weights (adamw) -> scale(-lr) -> apply_updates for ONE element,
// plus the refresh of that element's slots in the fragment-packed shadows. One IEEE rounding per
// written operation: bit-exact against the CPU restatement.
__device__ __forceinline__ void scatter_packs(const NetDims &m, int i, float v, float *pack) {
    const int o_b1 = (int)m.o_b1, o_w2 = (int)m.o_w2, o_b2 = (int)m.o_b2, o_wv = (int)m.o_wv, o_bv = (int)m.o_bv,
              o_wa = (int)m.o_wa, o_ba = (int)m.o_ba;
    if (i < o_b1) {                                     // w1[k][n]
        const int k = i / m.H1, n = i - k * m.H1;
        pack[m.p_w1 + pidx(m.KQ1, k, n)] = v;
    } else if (i >= o_w2 && i < o_b2) {                 // w2[k][n]
        const int u = i - o_w2;
        const int k = u / m.H2, n = u - k * m.H2;
        pack[m.p_w2 + pidx(m.H1 / 16, k, n)] = v;
        pack[m.p_w2t + pidx(m.H2 / 16, n, k)] = v;
        pack[m.p_w2k + ((long long)(k >> 2) * m.H2 + n) * 4 + (k & 3)] = v;
    } else if (i >= o_wv && i < o_bv) {                 // wv[k]
        const int k = i - o_wv;
        pack[m.p_wh + pidx(m.H2 / 16, k, 0)] = v;
        pack[m.p_wht + pidx(1, 0, k)] = v;
    } else if (i >= o_wa && i < o_ba) {                 // wa[k][a]
        const int u = i - o_wa;
        const int k = u / m.A, a = u - k * m.A;
        pack[m.p_wh + pidx(m.H2 / 16, k, 1 + a)] = v;
        pack[m.p_wht + pidx(1, 1 + a, k)] = v;
    }
}

// ------------------------------------------------------------------ weight gradients
// dW = H^T . Z over the batch: one workgroup per 16x16 tile of one weight block, the batch
// (K) split over its 4 waves and combined through LDS in a fixed order (deterministic).
// The tiles with mt == 0 also produce the bias gradient (column sums of Z).
__global__ void __launch_bounds__(256)
k_dw(NetDims m, const float *__restrict__ px, const float *__restrict__ ph1, const float *__restrict__ ph2,
     const float *__restrict__ pdz1, const float *__restrict__ pdz2, const float *__restrict__ pdz3, int B,
     float *grad, const float *loss_part, float *loss_out, DqnState *st, int bump_ctr, AdamArgs ad, PwArgs pw,
     int tiles) {
    __shared__ float red[4][64][4];
    __shared__ float redb[4][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if ((int)blockIdx.x >= tiles) {
        // surplus workgroups: the PER priority write-back of this batch (independent of the weight gradients; sharing
        // the launch hides it behind the dW tiles on other CUs). The dense top of the tree is rebuilt by the next
        // launch (k_per_top): doing it here behind a release -> counter -> acquire hand-off was measured slower
        // (21 us vs 10.6 + 5.4 us), as the guide predicts for an all-to-all seam.
        if (wave == 0)
            per_write_sorted_wave(st, pw.tree, pw.N, pw.L, pw.idx, pw.td_abs, pw.B, 1, pw.alpha, pw.eps, (int)blockIdx.x - tiles);
        return;
    }
    const int KQb = (B + 15) / 16;
    const int MT2 = m.H1 / 16, NT2 = m.H2 / 16, MT1 = m.KQ1, NT1 = m.H1 / 16, MTH = m.H2 / 16;
    int b = blockIdx.x;
    const float *pa, *pb; int mt, nt, which;
    if (b < MT2 * NT2)                  { which = 2; mt = b / NT2; nt = b % NT2; pa = ph1; pb = pdz2; }
    else if ((b -= MT2 * NT2) < MT1 * NT1) { which = 1; mt = b / NT1; nt = b % NT1; pa = px;  pb = pdz1; }
    else                                 { b -= MT1 * NT1; which = 3; mt = b; nt = 0; pa = ph2; pb = pdz3; (void)MTH; }

    STAMP(6, 0);
    // Which parameter(s) will this thread update? One tile element (row tid/16, col tid%16), and for the
    // mt == 0 tiles threads 0..15 also own a bias element. Their P / mu / nu and the step coefficients do
    // not depend on the gradient: fetch them now, the MFMA loop hides the latency.
    int ei[2] = {-1, -1};
    {
        const int rr = tid >> 4, c = tid & 15, n = 16 * nt + c, mrow = 16 * mt + rr;
        if (which == 2) ei[0] = (int)m.o_w2 + mrow * m.H2 + n;
        else if (which == 1) { if (mrow < m.D) ei[0] = (int)m.o_w1 + mrow * m.H1 + n; }
        else { if (n == 0) ei[0] = (int)m.o_wv + mrow; else if (n <= m.A) ei[0] = (int)m.o_wa + mrow * m.A + (n - 1); }
        if (mt == 0 && tid < 16) {
            if (which == 2) ei[1] = (int)m.o_b2 + n;
            else if (which == 1) ei[1] = (int)m.o_b1 + n;
            else { if (n == 0) ei[1] = (int)m.o_bv; else if (n <= m.A) ei[1] = (int)m.o_ba + n - 1; }
        }
    }
    double b1pow = 0.0, b2pow = 0.0;
    AdamCoef co{};
    float pP[2] = {0.f, 0.f}, pM[2] = {0.f, 0.f}, pV[2] = {0.f, 0.f};
    if (ad.P) {
        co = adam_coef(st, ad.b1, ad.b2, &b1pow, &b2pow);
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (ei[e] >= 0) { pP[e] = ad.P[ei[e]]; pM[e] = ad.mu[ei[e]]; pV[e] = ad.nu[ei[e]]; }
    }
    const int per = (KQb + 3) / 4;
    const int k0 = wave * per, k1 = (k0 + per < KQb) ? k0 + per : KQb;
    const float4 *A4 = reinterpret_cast<const float4 *>(pa) + ((long long)mt * KQb) * 64 + lane;
    const float4 *B4 = reinterpret_cast<const float4 *>(pb) + ((long long)nt * KQb) * 64 + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    // operand loads pipelined 8 k-blocks ahead (ping-pong); chain order over the batch unchanged
    constexpr int PF = 8;
    float4 a0[PF], c0[PF], a1[PF], c1[PF];
    auto load = [&](float4 (&a)[PF], float4 (&c)[PF], int kq0) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            int kq = kq0 + p;
            kq = kq < KQb ? kq : KQb - 1;                     // unconditional, clamped: surplus never used
            a[p] = A4[(long long)kq * 64]; c[p] = B4[(long long)kq * 64];
        }
    };
    auto compute = [&](const float4 (&a)[PF], const float4 (&c)[PF], int kq0) {
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (kq0 + p < k1) {
                acc = MFMA4(a[p].x, c[p].x, acc);
                acc = MFMA4(a[p].y, c[p].y, acc);
                acc = MFMA4(a[p].z, c[p].z, acc);
                acc = MFMA4(a[p].w, c[p].w, acc);
                bsum = (((bsum + c[p].x) + c[p].y) + c[p].z) + c[p].w;
            }
    };
    load(a0, c0, k0);
    if (per <= 2 * PF) {
        // the wave's whole batch slice in one round trip (B <= 1024): both chunks requested before the first MFMA
        load(a1, c1, k0 + PF);
        compute(a0, c0, k0);
        compute(a1, c1, k0 + PF);
    } else {
        for (int kq0 = k0; kq0 < k1; kq0 += 2 * PF) {
            load(a1, c1, kq0 + PF);
            compute(a0, c0, kq0);
            load(a0, c0, kq0 + 2 * PF);
            compute(a1, c1, kq0 + PF);
        }
    }
    STAMP(6, 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc[r];
    redb[wave][lane] = bsum;
    LDS_BARRIER();
    // arrival ticket for the optimizer-counter commit at the end, taken HERE: every wave of the block is past its operand
    // waits, hence has the coefficients it requested before them; the returning atomic's round trip then hides behind the
    // epilogue instead of being the tail of the launch
    unsigned int ticket = 0u;
    if (ad.P && tid == 0) {
        unsigned int one = 1u;
        asm volatile("" : "+v"(one) : "v"(co.c1), "v"(co.c2), "v"(co.neglr));
        ticket = ticket_take_early(&st->arrive, one);
    }
    // reduce the four K-slices in wave order (fixed => deterministic), then write / apply
    {
        const int rr = tid >> 4, c = tid & 15;
        const int sl = ((rr >> 2) << 4) | c, sr = rr & 3;           // accumulator lane / register holding it
        float gv[2];
        gv[0] = ((red[0][sl][sr] + red[1][sl][sr]) + red[2][sl][sr]) + red[3][sl][sr];
        gv[1] = 0.0f;
        if (ei[1] >= 0)
            for (int w = 0; w < 4; ++w)
                for (int gq = 0; gq < 4; ++gq) gv[1] = gv[1] + redb[w][16 * gq + tid];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = ei[e];
            if (i < 0) continue;
            grad[i] = gv[e];
            if (ad.P) {
                // optax scale_by_adam -> add_decayed_weights -> scale(-lr) -> apply_updates (as adam_elem)
                const float gi = gv[e] * ad.grad_scale;
                const float mm = (ad.b1 * pM[e]) + (co.omb1 * gi);
                const float vv = (ad.b2 * pV[e]) + (co.omb2 * (gi * gi));
                ad.mu[i] = mm; ad.nu[i] = vv;
                const float mhat = __fdiv_rn(mm, co.c1), vhat = __fdiv_rn(vv, co.c2);
                float u = __fdiv_rn(mhat, sqrtf(vhat) + ad.eps);
                float pnew = pP[e];
                if (ad.adamw) u = u + (ad.wd * pnew);
                pnew = pnew + (co.neglr * u);
                ad.P[i] = pnew;
                if (e == 0 && which == 2) {
                    // a W2 tile element knows its (k, n): the three shadows without the index divisions of scatter_packs
                    const int k = 16 * mt + (tid >> 4), n = 16 * nt + (tid & 15);
                    ad.pack[m.p_w2 + pidx(m.H1 / 16, k, n)] = pnew;
                    ad.pack[m.p_w2t + pidx(m.H2 / 16, n, k)] = pnew;
                    ad.pack[m.p_w2k + ((long long)(k >> 2) * m.H2 + n) * 4 + (k & 3)] = pnew;
                } else scatter_packs(m, i, pnew, ad.pack);
            }
        }
    }
    STAMP(6, 2);
    if (blockIdx.x == 0 && wave == 1) {
        // loss = (sum of the per-tile partial sums) / B: lanes load in parallel, fixed shuffle tree
        float s = 0.0f;
        for (int t = lane; t < KQb; t += 64) s = s + loss_part[t];
        for (int o = 32; o > 0; o >>= 1) s = s + __shfl_xor(s, o, 64);
        if (lane == 0) {
            float Lv = __fdiv_rn(s, (float)B);
            if (st->err_count != 0u) Lv = __int_as_float(0x7fc00000);     // a hand-over wait gave up: dqn_device_errors_host
            st->loss = Lv;
            if (loss_out) *loss_out = Lv;
            if (bump_ctr) { st->sample_ctr += 1ull; st->wmax = 0.0f; }
        }
    }
    if (ad.P) {
        // commit the optimizer counters once every block has read them: the block that drew the last ticket (tickets are
        // taken after the operand waits, above) writes them
        if (tid == 0 && ticket == (unsigned)tiles - 1u) { st->b1pow = b1pow; st->b2pow = b2pow; st->adam_count += 1; st->arrive = 0; }
    }
    STAMP(6, 3);
}

void launch_dw(hipStream_t s, const NetDims &m, const float *px, const float *ph1, const float *ph2,
               const float *pdz1, const float *pdz2, const float *pdz3, int B, float *grad,
               const float *loss_part, float *loss_out, DqnState *st, int bump_ctr, const AdamArgs &adam,
               const PwArgs &pw) {
    const int tiles = (m.H1 / 16) * (m.H2 / 16) + m.KQ1 * (m.H1 / 16) + m.H2 / 16;
    const int extra = pw.tree ? (pw.B + 63) / 64 : 0;
    DQN_LAUNCH(k_dw, dim3(tiles + extra), dim3(256), 0, s, m, px, ph1, ph2, pdz1, pdz2, pdz3, B, grad,
                       loss_part, loss_out, st, bump_ctr, adam, pw, tiles);
}

// ---------------------------------------------------------------------------- optimizer
// optax scale_by_adam -> add_decayed_weights (adamw) -> scale(-lr) -> apply_updates, and the
// refresh of the fragment-packed shadows in the same pass. Bit-exact vs the CPU restatement.
__global__ void __launch_bounds__(256)
k_adam(NetDims m, DqnState *st, float *P, const float *__restrict__ g, float *mu, float *nu, float *pack,
       int adamw, float b1, float b2, float eps, float wd, float grad_scale) {
    double b1pow, b2pow;
    const AdamCoef co = adam_coef(st, b1, b2, &b1pow, &b2pow);
    const int nP = (int)m.P;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nP; i += gridDim.x * blockDim.x) {
        const float p = adam_elem(co, g[i], P, mu, nu, i, adamw, b1, b2, eps, wd, grad_scale);
        scatter_packs(m, i, p, pack);
    }
    // commit the step counters once every block has read them: a thread's stores above depend on
    // the coefficients, so passing this barrier implies its reads of the state are complete.
    LDS_BARRIER();                                           // (a barrier only: no need to drain this block's stores first)
    if (threadIdx.x == 0) {
        const unsigned int ticket = atomicAdd(&st->arrive, 1u);
        if (ticket == gridDim.x - 1) { st->b1pow = b1pow; st->b2pow = b2pow; st->adam_count += 1; st->arrive = 0; }
    }
}

void launch_adam(hipStream_t s, const NetDims &m, DqnState *st, float *params, const float *grad, float *mu,
                 float *nu, float *pack, int adamw, float b1, float b2, float eps, float wd, float grad_scale) {
    int blocks = (int)((m.P + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    DQN_LAUNCH(k_adam, dim3(blocks), dim3(256), 0, s, m, st, params, grad, mu, nu, pack, adamw, b1, b2,
                       eps, wd, grad_scale);
}

// ----------------------------------------------------------------------- epsilon-greedy
__global__ void __launch_bounds__(256)
k_policy(const float *__restrict__ q, int n, int A, float epsilon, unsigned long long seed,
         unsigned long long ctr, int32_t *actions, const DqnState *st_from) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (st_from) { epsilon = st_from->epsilon; ctr = st_from->env_ctr; }
    float qrow[16];
    for (int k = 0; k < A; ++k) qrow[k] = q[(long long)i * A + k];
    actions[i] = policy_row(qrow, A, epsilon, seed, ctr, i);
}

void launch_policy(hipStream_t s, const float *q, int n, int A, float epsilon, unsigned long long seed,
                   unsigned long long ctr, int32_t *actions, const DqnState *st_from) {
    hipLaunchKernelGGL(k_policy, dim3((n + 255) / 256), dim3(256), 0, s, q, n, A, epsilon, seed, ctr, actions, st_from);
}

__global__ void __launch_bounds__(256) k_u8_to_f32(const uint8_t *__restrict__ in, float *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] ? 1.0f : 0.0f;            // preprocessing: dones.astype(float32) (:84)
}

void launch_u8_to_f32(hipStream_t s, const uint8_t *in, float *out, int n) {
    hipLaunchKernelGGL(k_u8_to_f32, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n);
}
