// csrc/dqn_net_bf16.hip -- the Q-network kernels on bf16 MFMA (v_mfma_f32_16x16x32_bf16): bf16 operands,
// f32 accumulation, f32 master weights / optimizer state. Throughput path (dqn_config.precision =
// DQN_PREC_BF16); the f32 path in dqn_net.hip is the 1e-5 parity path. Same decomposition, same
// fragment-packed operand idea, with 32-deep k-blocks and 16-B-per-lane operands:
//
//   packed16(M)[((ct*KQ + kq)*64 + lane)*8 + j] = bf16( M[32*kq + 8*(lane>>4) + j][16*ct + (lane&15)] )
//
// LDS activations are plain row-major bf16 (a lane's A fragment is 8 consecutive k of one row).
// Pointer fields of FwdPass / BwdArgs typed `float*` carry bf16 data here (pack, px, ph*, pdz*).
#include <type_traits>
#include "dqn_net_common.h"
#include "dqn_bf16_pack.h"
#include "dqn_per_device.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

long long bf16_pack_elems(const NetDims &m) { return make_dims16(m).pack_elems; }

// ------------------------------------------------------------------------ weight packing
__global__ void __launch_bounds__(256)
k_pack16(NetDims m, Dims16 d, const float *__restrict__ P, __bf16 *__restrict__ pack) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d.pack_elems) return;
    long long base; int KQ; int which;
    if (t < d.p_w2)       { base = d.p_w1;  KQ = d.KQ1; which = 0; }
    else if (t < d.p_wh)  { base = d.p_w2;  KQ = d.KQ2; which = 1; }
    else if (t < d.p_w2t) { base = d.p_wh;  KQ = d.KQH; which = 2; }
    else if (t < d.p_wht) { base = d.p_w2t; KQ = d.KQH; which = 3; }
    else                  { base = d.p_wht; KQ = 1;     which = 4; }
    const long long u = t - base;
    const int j = (int)(u & 7), lane = (int)((u >> 3) & 63);
    const long long blk = u >> 9;
    const int kq = (int)(blk % KQ), ct = (int)(blk / KQ);
    const int k = 32 * kq + 8 * (lane >> 4) + j, c = 16 * ct + (lane & 15);
    float v = 0.0f;
    switch (which) {
    case 0: if (k < m.D && c < m.H1) v = P[m.o_w1 + (long long)k * m.H1 + c]; break;
    case 1: if (k < m.H1) v = P[m.o_w2 + (long long)k * m.H2 + c]; break;
    case 2: if (k < m.H2) { if (c == 0) v = P[m.o_wv + k]; else if (c <= m.A) v = P[m.o_wa + (long long)k * m.A + (c - 1)]; } break;
    case 3: if (k < m.H2) v = P[m.o_w2 + (long long)c * m.H2 + k]; break;
    case 4: if (k == 0) v = P[m.o_wv + c]; else if (k <= m.A) v = P[m.o_wa + (long long)c * m.A + (k - 1)]; break;
    }
    pack[t] = (__bf16)v;
}

void launch_pack_bf16(hipStream_t s, const NetDims &m, const float *params, float *pack) {
    const Dims16 d = make_dims16(m);
    hipLaunchKernelGGL(k_pack16, dim3((unsigned)((d.pack_elems + 255) / 256)), dim3(256), 0, s, m, d, params,
                       reinterpret_cast<__bf16 *>(pack));
}

// ------------------------------------------------------------------- MFMA layer helper
// Whole layer in one register chunk (hidden sizes <= 256 => at most 8 k-blocks): the MFMA phase is a few
// hundred cycles, so everything is requested up front and the kernel is bounded by one memory latency.
template <int TN, int PF>
struct MmaLayer16 {
    const bf16x8 *pk[TN];
    bf16x8 b0[PF][TN];
    int KQ;

    __device__ __forceinline__ void start(const __bf16 *__restrict__ pack, int KQ_, int CT, int wave, int lane) {
        KQ = KQ_;
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            int ct = wave + 4 * t;
            ct = ct < CT ? ct : CT - 1;
            pk[t] = reinterpret_cast<const bf16x8 *>(pack) + (long long)ct * KQ * 64 + lane;
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int kq = p < KQ ? p : KQ - 1;
#pragma unroll
            for (int t = 0; t < TN; ++t) b0[p][t] = pk[t][(long long)kq * 64];
        }
    }
    __device__ __forceinline__ void finish(const __bf16 *lds_a, int stride, int lane, f32x4 (&acc)[TN]) {
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const __bf16 *arow = lds_a + (lane & 15) * stride + 8 * (lane >> 4);
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if (p < KQ) {
                const bf16x8 a8 = *reinterpret_cast<const bf16x8 *>(arow + 32 * p);
#pragma unroll
                for (int t = 0; t < TN; ++t) acc[t] = MFMA16(a8, b0[p][t], acc[t]);
            }
        }
    }
};

static inline int tn_of(int H) { const int ct = H / 16; return ct <= 4 ? 1 : (ct <= 8 ? 2 : 4); }

// ------------------------------------------------------------------------------ forward
struct FwdPasses16 { FwdPass p[3]; };

// FUSE: as k_qnet_fwd<.., true> (dqn_net.hip) -- the pass-0 workgroup of a tile goes on with the tile's row backward
// once the other two passes have handed over their Q rows (L1-bypassing stores, drained, then the tile's counter)
template <int TN1, int TN2, bool FUSE>
__global__ void __launch_bounds__(256)
k_qnet_fwd16(NetDims m, Dims16 d, FwdPasses16 passes, int B, SampleArgs smp, FuseBwd fb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FwdPass ps = passes.p[blockIdx.y];
    const __bf16 *pack = reinterpret_cast<const __bf16 *>(ps.pack);
    __bf16 *px = reinterpret_cast<__bf16 *>(ps.px), *ph1 = reinterpret_cast<__bf16 *>(ps.ph1), *ph2 = reinterpret_cast<__bf16 *>(ps.ph2);
    const int tile = blockIdx.x, row0 = tile * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KQb = (B + 31) / 32;
    const int K1 = d.KQ1 * 32, K2 = d.KQ2 * 32, KH = d.KQH * 32;
    const int sx = K1 + 8, s1 = K2 + 8, s2 = KH + 8;
    __bf16 *lx = reinterpret_cast<__bf16 *>(smem), *l1 = lx + 16 * sx, *l2 = l1 + 16 * s1;
    float *lh = reinterpret_cast<float *>(l2 + 16 * s2);

    // order of issue = order of arrival: input rows, layer 1, biases, layer 2, heads
    const bool sampling = smp.st != nullptr;
    int *lidx = reinterpret_cast<int *>(lh + 256 + 16);
    const float *ring = sampling ? (ps.src == 1 ? smp.states : smp.observations) : nullptr;
    auto xload = [&](int t) -> float {
        const int rl = t / K1, c = t - rl * K1;
        if (!(t < 16 * K1 && row0 + rl < B && c < m.D)) return 0.0f;
        return sampling ? ring[(long long)lidx[rl] * m.D + c] : ps.x[(long long)(row0 + rl) * m.D + c];
    };
    float xv[2];
    if (!sampling) {
#pragma unroll
        for (int u = 0; u < 2; ++u) xv[u] = xload(tid + 256 * u);
    }
    // batch drawn by the preceding actor launch (dqn_actor.hip): idx -> ring row is the dependent chain, requested first
    const bool presampled = sampling && smp.pre;
    int pre_leaf = 0;
    if (presampled && tid < 16) pre_leaf = smp.idx[row0 + tid < B ? row0 + tid : B - 1];
    MmaLayer16<TN1, 8> L1; MmaLayer16<TN2, 8> L2; MmaLayer16<1, 8> LH;
    L1.start(pack + d.p_w1, d.KQ1, m.H1 / 16, wave, lane);
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
    unsigned m1bits = 0u, m2bits = 0u;                               // ReLU gates of this lane's accumulator elements (FUSE)
    if (presampled) {
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
        for (int u = 0; u < 2; ++u) xv[u] = xload(tid + 256 * u);
    }
    L2.start(pack + d.p_w2, d.KQ2, m.H2 / 16, wave, lane);
    if (wave == 0) LH.start(pack + d.p_wh, d.KQH, 1, 0, lane);
    if (sampling && !presampled) {
        if (smp.tree) {
            float *lsub = reinterpret_cast<float *>(lidx + 16);
            sample_tile_coop(smp, row0, B, tid, blockIdx.y == 0, lidx, lsub, lsub + 512);     // ends with a barrier
        } else {
            if (tid < 16) sample_tile(smp, row0, B, tid, blockIdx.y == 0, lidx);
            LDS_BARRIER();
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) xv[u] = xload(tid + 256 * u);
        if constexpr (FUSE) {
            if (blockIdx.y == 0 && tid < 16 && row0 + tid < B) {
                const int leaf = lidx[tid];
                row_a = smp.actions[leaf]; row_r = smp.rewards[leaf]; row_d = smp.dones[leaf];
            }
        }
    }

    // zero the k-padding columns of the hidden activations (hidden % 32 == 16 only)
    for (int t = tid; t < 16 * (K2 - m.H1); t += 256) l1[(t / (K2 - m.H1)) * s1 + m.H1 + t % (K2 - m.H1)] = (__bf16)0.0f;
    for (int t = tid; t < 16 * (KH - m.H2); t += 256) l2[(t / (KH - m.H2)) * s2 + m.H2 + t % (KH - m.H2)] = (__bf16)0.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + 256 * u;
        if (t < 16 * K1) {
            const int rl = t / K1, c = t - rl * K1;
            const __bf16 b = (__bf16)xv[u];
            lx[rl * sx + c] = b;
            if (px && c < m.KQ1 * 16) px[pidx16(KQb, row0 + rl, c)] = b;
        }
    }
    for (int t = tid + 512; t < 16 * K1; t += 256) {             // obs_dim > 32 only
        const int rl = t / K1, c = t - rl * K1;
        const float v = xload(t);
        lx[rl * sx + c] = (__bf16)v;
        if (px && c < m.KQ1 * 16) px[pidx16(KQb, row0 + rl, c)] = (__bf16)v;
    }
    LDS_BARRIER();

    {   // layer 1                                                           dddqn.py:25-26
        f32x4 acc[TN1];
        L1.finish(lx, sx, lane, acc);
#pragma unroll
        for (int t = 0; t < TN1; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H1 / 16) {
                const int c = lane & 15, col = 16 * ct + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * (lane >> 4) + r;
                    float v = acc[t][r] + bias1[t];
                    v = v > 0.0f ? v : 0.0f;
                    const __bf16 b = (__bf16)v;
                    if constexpr (FUSE) m1bits |= ((float)b > 0.0f ? 1u : 0u) << (4 * t + r);
                    l1[rl * s1 + col] = b;
                    if (ph1) ph1[pidx16(KQb, row0 + rl, col)] = b;
                }
            }
        }
    }
    LDS_BARRIER();

    {   // layer 2                                                           dddqn.py:27-28
        f32x4 acc[TN2];
        L2.finish(l1, s1, lane, acc);
#pragma unroll
        for (int t = 0; t < TN2; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H2 / 16) {
                const int c = lane & 15, col = 16 * ct + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * (lane >> 4) + r;
                    float v = acc[t][r] + bias2[t];
                    v = v > 0.0f ? v : 0.0f;
                    const __bf16 b = (__bf16)v;
                    if constexpr (FUSE) m2bits |= ((float)b > 0.0f ? 1u : 0u) << (4 * t + r);
                    l2[rl * s2 + col] = b;
                    if (ph2) ph2[pidx16(KQb, row0 + rl, col)] = b;
                    if (ps.feat && row0 + rl < B) ps.feat[(long long)(row0 + rl) * m.H2 + col] = (float)b;
                }
            }
        }
    }
    LDS_BARRIER();

    if (wave == 0) {                                                        // heads: dddqn.py:29-30
        f32x4 acc[1];
        LH.finish(l2, s2, lane, acc);
        const int c = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) lh[(4 * (lane >> 4) + r) * 16 + c] = acc[0][r] + biash;
    }
    MmaLayer16<TN2, 1> LA; MmaLayer16<TN1, 8> LB;                             // row backward (FUSE, pass 0)
    if constexpr (FUSE) {
        if (blockIdx.y == 0) {
            LA.start(reinterpret_cast<const __bf16 *>(fb.g.pack) + d.p_wht, 1, m.H2 / 16, wave, lane);
            LB.start(reinterpret_cast<const __bf16 *>(fb.g.pack) + d.p_w2t, d.KQH, m.H1 / 16, wave, lane);
        }
    }
    LDS_BARRIER();

    float qrow[16];
    if (tid < 16 && row0 + tid < B) {                                       // dddqn.py:31 (+ policy)
        const float *hr = lh + tid * 16;
        float sum = 0.0f;
        if constexpr (FUSE) {
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
        if (FUSE && blockIdx.y != 0) {
            for (int a = 0; a < m.A; ++a) __hip_atomic_store(&ps.q[(long long)(row0 + tid) * m.A + a], qrow[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (ps.q) for (int a = 0; a < m.A; ++a) ps.q[(long long)(row0 + tid) * m.A + a] = qrow[a];
        if (ps.act_out) {
            const float eps = ps.act_state ? ps.act_state->epsilon : ps.act_eps;
            const unsigned long long ctr = ps.act_state ? ps.act_state->env_ctr : ps.act_ctr;
            const int act = policy_row(qrow, m.A, eps, ps.act_seed, ctr, row0 + tid);
            ps.act_out[row0 + tid] = act;
        }
    }
    if constexpr (FUSE) {
        if (blockIdx.y != 0) {
            if (wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the Q rows above have left this CU
                if (tid == 0) atomicAdd(reinterpret_cast<unsigned *>(fb.tile_cnt) + tile, fb.withhold ? 0u : 1u);
            }
            return;
        }
        // ---- pass 0: TD target / Huber gradient / row backward of this tile (the body of k_bwd_rows16)
        const BwdArgs &g = fb.g;
        __bf16 *pdz1 = reinterpret_cast<__bf16 *>(g.pdz1), *pdz2 = reinterpret_cast<__bf16 *>(g.pdz2), *pdz3 = reinterpret_cast<__bf16 *>(g.pdz3);
        const int A = m.A;
        const int s3 = 32 + 8;
        __bf16 *l3 = lx;                                             // [16][40] (x is dead; 16 * sx >= 16 * 40)
        __bf16 *lz2 = reinterpret_cast<__bf16 *>(lh + 256 + 32 + 528);   // dz2 [16][s2]
        float *lrow = reinterpret_cast<float *>(lz2 + 16 * s2);
        const int irow = row0 + tid;
        const bool rowt = tid < 16 && irow < B;
        float wi = 1.0f, wmax = 1.0f;
        if (rowt && g.w_raw) { wi = g.w_raw[irow]; wmax = fb.st->wmax; }
        for (int t = tid; t < 16 * s3; t += 256) l3[t] = (__bf16)0.0f;
        for (int t = tid; t < 16 * (KH - m.H2); t += 256) lz2[(t / (KH - m.H2)) * s2 + m.H2 + t % (KH - m.H2)] = (__bf16)0.0f;
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
        LDS_BARRIER();
        if (tid < 16) {
            float rowloss = 0.0f;
            if (rowt) {
                auto td_rows = [&](auto amax_tag) {                    // AMAX = 4 or the 15-action maximum (see k_qnet_fwd)
                constexpr int AMAX = decltype(amax_tag)::value;
                float nqr[AMAX], ntr[AMAX];
#pragma unroll
                for (int k2 = 0; k2 < AMAX; ++k2) {
                    nqr[k2] = k2 < A ? __hip_atomic_load(&g.nq[(long long)irow * A + k2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
                    ntr[k2] = k2 < A ? __hip_atomic_load(&g.nt[(long long)irow * A + k2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
                }
                float w = 1.0f;
                if (g.w_raw) { w = __fdiv_rn(wi, wmax); if (g.isw_out) g.isw_out[irow] = w; }
                const float invB = __fdiv_rn(1.0f, (float)B);
                float best = nqr[0], nt_star = ntr[0], q_a = qrow[0];   // q_learning_functions.py:55 argmax, first max wins
#pragma unroll
                for (int k2 = 1; k2 < AMAX; ++k2) {
                    if (k2 < A && nqr[k2] > best) { best = nqr[k2]; nt_star = ntr[k2]; }
                    if (k2 == row_a) q_a = qrow[k2];
                }
                const float di = row_d ? 1.0f : 0.0f;
                const float t1 = g.gamma * nt_star;                      // :58 (quirk Q3)
                const float t2 = t1 - q_a;
                const float t3 = (1.0f - di) * t2;
                const float delta = row_r + t3;
                if (g.td) g.td[irow] = delta;
                if (g.td_abs) g.td_abs[irow] = fabsf(delta);
                float gk[AMAX], gsum = 0.0f;
#pragma unroll
                for (int k2 = 0; k2 < AMAX; ++k2) {
                    gk[k2] = 0.0f;
                    if (k2 < A) {
                        const float trk = qrow[k2] + delta * (k2 == row_a ? 1.0f : 0.0f);   // :59 (quirk Q4)
                        const float e = qrow[k2] - trk;
                        rowloss = rowloss + huber(e);
                        const float c = e > 1.0f ? 1.0f : (e < -1.0f ? -1.0f : e);
                        gk[k2] = (w * c) * invB;
                        gsum = gsum + gk[k2];
                        if (g.dq) g.dq[(long long)irow * A + k2] = gk[k2];
                        if (g.targets_out) g.targets_out[(long long)irow * A + k2] = trk;
                    }
                }
                if (g.w_raw) rowloss = w * rowloss;
                const float gmean = __fdiv_rn(gsum, (float)A);
                l3[tid * s3 + 0] = (__bf16)gsum;
#pragma unroll
                for (int k2 = 0; k2 < AMAX; ++k2) if (k2 < A) l3[tid * s3 + 1 + k2] = (__bf16)(gk[k2] - gmean);
                };
                if (A <= 4) td_rows(std::integral_constant<int, 4>{}); else td_rows(std::integral_constant<int, 15>{});
            }
            lrow[tid] = rowloss;
        }
        LDS_BARRIER();
        if (tid == 0 && tile < (B + 15) / 16) {
            float sl = 0.0f;
            for (int k = 0; k < 16; ++k) sl = sl + lrow[k];
            g.loss_part[tile] = gave_up ? __int_as_float(0x7fc00000) : sl;
        }
        pdz3[pidx16(KQb, row0 + (tid >> 4), tid & 15)] = l3[(tid >> 4) * s3 + (tid & 15)];
        {   // dz2 = (dz3 . WH^T) * (h2 > 0)
            f32x4 acc[TN2];
            LA.finish(l3, s3, lane, acc);
#pragma unroll
            for (int t = 0; t < TN2; ++t) {
                const int ct = wave + 4 * t;
                if (ct < m.H2 / 16) {
                    const int c = lane & 15, col = 16 * ct + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rl = 4 * (lane >> 4) + r;
                        const __bf16 b = (__bf16)(((m2bits >> (4 * t + r)) & 1u) ? acc[t][r] : 0.0f);
                        lz2[rl * s2 + col] = b;
                        pdz2[pidx16(KQb, row0 + rl, col)] = b;
                    }
                }
            }
        }
        LDS_BARRIER();
        {   // dz1 = (dz2 . W2^T) * (h1 > 0)
            f32x4 acc[TN1];
            LB.finish(lz2, s2, lane, acc);
#pragma unroll
            for (int t = 0; t < TN1; ++t) {
                const int ct = wave + 4 * t;
                if (ct < m.H1 / 16) {
                    const int c = lane & 15, col = 16 * ct + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rl = 4 * (lane >> 4) + r;
                        pdz1[pidx16(KQb, row0 + rl, col)] = (__bf16)(((m1bits >> (4 * t + r)) & 1u) ? acc[t][r] : 0.0f);
                    }
                }
            }
        }
    }
}

// tiles are launched in pairs so that every 32-row k-block of the batch-major stashes is fully written
static inline int tiles16(int B) { return 2 * ((B + 31) / 32); }

void launch_qnet_fwd_bf16(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, const SampleArgs *smp,
                          const BwdArgs *fuse, int *tile_cnt, DqnState *st, int tile_stride, int withhold) {
    FwdPasses16 ps{};
    bool stash = false;
    for (int i = 0; i < npass; ++i) { ps.p[i] = passes[i]; stash |= passes[i].px != nullptr; }
    const Dims16 d = make_dims16(m);
    const SampleArgs sa = smp ? *smp : SampleArgs{};
    const dim3 grid(stash ? tiles16(B) : (B + 15) / 16, npass), block(256);
    size_t lds = 2 * (16 * (d.KQ1 * 32 + 8) + 16 * (d.KQ2 * 32 + 8) + 16 * (d.KQH * 32 + 8)) + 4 * (256 + 32 + 528);
    const int t1 = tn_of(m.H1), t2 = tn_of(m.H2);
    if (fuse) {
        const FuseBwd fb{*fuse, tile_cnt, st, tile_stride, withhold};
        lds += 2 * 16 * (d.KQH * 32 + 8) + 4 * 16;
#define FWD_CASE(A1, A2) if (t1 == A1 && t2 == A2) { DQN_LAUNCH((k_qnet_fwd16<A1, A2, true>), grid, block, lds, s, m, d, ps, B, sa, fb); return; }
        FWD_CASE(1, 1) FWD_CASE(1, 2) FWD_CASE(1, 4) FWD_CASE(2, 1) FWD_CASE(2, 2) FWD_CASE(2, 4)
        FWD_CASE(4, 1) FWD_CASE(4, 2) FWD_CASE(4, 4)
#undef FWD_CASE
    }
    const FuseBwd fb{};
#define FWD_CASE(A1, A2) if (t1 == A1 && t2 == A2) { DQN_LAUNCH((k_qnet_fwd16<A1, A2, false>), grid, block, lds, s, m, d, ps, B, sa, fb); return; }
    FWD_CASE(1, 1) FWD_CASE(1, 2) FWD_CASE(1, 4) FWD_CASE(2, 1) FWD_CASE(2, 2) FWD_CASE(2, 4)
    FWD_CASE(4, 1) FWD_CASE(4, 2) FWD_CASE(4, 4)
#undef FWD_CASE
}

// -------------------------------------------------------------- row-wise backward pass
template <int TN1, int TN2>
__global__ void __launch_bounds__(256)
k_bwd_rows16(NetDims m, Dims16 d, BwdArgs g, int B, DqnState *st) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const __bf16 *pack = reinterpret_cast<const __bf16 *>(g.pack);
    const __bf16 *ph1 = reinterpret_cast<const __bf16 *>(g.ph1), *ph2 = reinterpret_cast<const __bf16 *>(g.ph2);
    __bf16 *pdz1 = reinterpret_cast<__bf16 *>(g.pdz1), *pdz2 = reinterpret_cast<__bf16 *>(g.pdz2), *pdz3 = reinterpret_cast<__bf16 *>(g.pdz3);
    const int tile = blockIdx.x, row0 = tile * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KQb = (B + 31) / 32;
    const int A = m.A;
    const int KH = d.KQH * 32;
    const int s3 = 32 + 8, s2 = KH + 8;
    __bf16 *l3 = reinterpret_cast<__bf16 *>(smem), *l2 = l3 + 16 * s3;
    float *lrow = reinterpret_cast<float *>(l2 + 16 * s2);

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
            di = g.d_f32 ? g.d_f32[irow] : (g.d_u8[irow] ? 1.0f : 0.0f);
        }
        if (g.w_raw) { wi = g.w_raw[irow]; wmax = st->wmax; }
        else if (g.isw) wi = g.isw[irow];
    }
    MmaLayer16<TN2, 1> LA; MmaLayer16<TN1, 8> LB;
    LA.start(pack + d.p_wht, 1, m.H2 / 16, wave, lane);
    float m2[TN2][4], m1[TN1][4];
#pragma unroll
    for (int t = 0; t < TN2; ++t) {
        int ct = wave + 4 * t; ct = ct < m.H2 / 16 ? ct : m.H2 / 16 - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) m2[t][r] = (float)ph2[pidx16(KQb, row0 + 4 * (lane >> 4) + r, 16 * ct + (lane & 15))];
    }
    LB.start(pack + d.p_w2t, d.KQH, m.H1 / 16, wave, lane);
#pragma unroll
    for (int t = 0; t < TN1; ++t) {
        int ct = wave + 4 * t; ct = ct < m.H1 / 16 ? ct : m.H1 / 16 - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) m1[t][r] = (float)ph1[pidx16(KQb, row0 + 4 * (lane >> 4) + r, 16 * ct + (lane & 15))];
    }
    for (int t = tid; t < 16 * s3; t += 256) l3[t] = (__bf16)0.0f;
    for (int t = tid; t < 16 * (KH - m.H2); t += 256) l2[(t / (KH - m.H2)) * s2 + m.H2 + t % (KH - m.H2)] = (__bf16)0.0f;
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
                const float e = qr[k2] - tr[k2];
                rowloss = rowloss + huber(e);
                const float c = e > 1.0f ? 1.0f : (e < -1.0f ? -1.0f : e);
                gk[k2] = (w * c) * invB;
                gsum = gsum + gk[k2];
                if (g.dq) g.dq[(long long)i * A + k2] = gk[k2];
                if (g.targets_out) g.targets_out[(long long)i * A + k2] = tr[k2];
            }
            if (g.w_raw || g.isw) rowloss = w * rowloss;
            const float gmean = __fdiv_rn(gsum, (float)A);
            l3[tid * s3 + 0] = (__bf16)gsum;
            for (int k2 = 0; k2 < A; ++k2) l3[tid * s3 + 1 + k2] = (__bf16)(gk[k2] - gmean);
        }
        lrow[tid] = rowloss;
    }
    LDS_BARRIER();
    if (tid == 0) {
        float s = 0.0f;
        for (int k = 0; k < 16; ++k) s = s + lrow[k];
        if (tile < (B + 15) / 16) g.loss_part[tile] = s;
    }
    for (int t = tid; t < 256; t += 256) pdz3[pidx16(KQb, row0 + (t >> 4), t & 15)] = l3[(t >> 4) * s3 + (t & 15)];

    {   // dz2 = (dz3 . WH^T) * (h2 > 0)
        f32x4 acc[TN2];
        LA.finish(l3, s3, lane, acc);
#pragma unroll
        for (int t = 0; t < TN2; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H2 / 16) {
                const int c = lane & 15, col = 16 * ct + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * (lane >> 4) + r;
                    const __bf16 b = (__bf16)(m2[t][r] > 0.0f ? acc[t][r] : 0.0f);
                    l2[rl * s2 + col] = b;
                    pdz2[pidx16(KQb, row0 + rl, col)] = b;
                }
            }
        }
    }
    LDS_BARRIER();

    {   // dz1 = (dz2 . W2^T) * (h1 > 0)
        f32x4 acc[TN1];
        LB.finish(l2, s2, lane, acc);
#pragma unroll
        for (int t = 0; t < TN1; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H1 / 16) {
                const int c = lane & 15, col = 16 * ct + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * (lane >> 4) + r;
                    pdz1[pidx16(KQb, row0 + rl, col)] = (__bf16)(m1[t][r] > 0.0f ? acc[t][r] : 0.0f);
                }
            }
        }
    }
}

void launch_bwd_rows_bf16(hipStream_t s, const NetDims &m, const BwdArgs &g, int B, DqnState *st) {
    const Dims16 d = make_dims16(m);
    const dim3 grid(tiles16(B)), block(256);
    const size_t lds = 2 * (16 * 40 + 16 * (d.KQH * 32 + 8)) + 4 * 16;
    const int t1 = tn_of(m.H1), t2 = tn_of(m.H2);
#define BWD_CASE(A1, A2) if (t1 == A1 && t2 == A2) { DQN_LAUNCH((k_bwd_rows16<A1, A2>), grid, block, lds, s, m, d, g, B, st); return; }
    BWD_CASE(1, 1) BWD_CASE(1, 2) BWD_CASE(1, 4) BWD_CASE(2, 1) BWD_CASE(2, 2) BWD_CASE(2, 4)
    BWD_CASE(4, 1) BWD_CASE(4, 2) BWD_CASE(4, 4)
#undef BWD_CASE
}

// ------------------------------------------------------------------ weight gradients
__global__ void __launch_bounds__(256)
k_dw16(NetDims m, Dims16 d, const __bf16 *__restrict__ px, const __bf16 *__restrict__ ph1, const __bf16 *__restrict__ ph2,
       const __bf16 *__restrict__ pdz1, const __bf16 *__restrict__ pdz2, const __bf16 *__restrict__ pdz3, int B,
       float *grad, const float *loss_part, float *loss_out, DqnState *st, int bump_ctr, AdamArgs ad, PwArgs pw, int tiles) {
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
    const int KQb = (B + 31) / 32;
    const int NT2 = m.H2 / 16, MT2 = m.H1 / 16, MT1 = m.KQ1, NT1 = m.H1 / 16;
    int b = blockIdx.x;
    const __bf16 *pa, *pb; int mt, nt, which;
    if (b < MT2 * NT2)                      { which = 2; mt = b / NT2; nt = b % NT2; pa = ph1; pb = pdz2; }
    else if ((b -= MT2 * NT2) < MT1 * NT1)  { which = 1; mt = b / NT1; nt = b % NT1; pa = px;  pb = pdz1; }
    else                                    { b -= MT1 * NT1; which = 3; mt = b; nt = 0; pa = ph2; pb = pdz3; }

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
    const bf16x8 *A8 = reinterpret_cast<const bf16x8 *>(pa) + ((long long)mt * KQb) * 64 + lane;
    const bf16x8 *B8 = reinterpret_cast<const bf16x8 *>(pb) + ((long long)nt * KQb) * 64 + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    constexpr int PF = 8;
    for (int kq0 = k0; kq0 < k1; kq0 += PF) {
        bf16x8 a[PF], c[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            int kq = kq0 + p; kq = kq < KQb ? kq : KQb - 1;
            a[p] = A8[(long long)kq * 64]; c[p] = B8[(long long)kq * 64];
        }
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (kq0 + p < k1) {
                acc = MFMA16(a[p], c[p], acc);
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum = bsum + (float)c[p][j];
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc[r];
    redb[wave][lane] = bsum;
    LDS_BARRIER();
    unsigned int ticket = 0u;                                    // arrival ticket for the commit at the end, taken early (see k_dw)
    if (ad.P && tid == 0) {
        unsigned int one = 1u;
        asm volatile("" : "+v"(one) : "v"(co.c1), "v"(co.c2), "v"(co.neglr));
        ticket = ticket_take_early(&st->arrive, one);
    }
    {
        const int rr = tid >> 4, c = tid & 15;
        const int sl = ((rr >> 2) << 4) | c, sr = rr & 3;
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
                scatter_packs16(m, d, i, pnew, reinterpret_cast<__bf16 *>(ad.pack));
                if (ad.pack_act) scatter_actor_packs(m, i, pnew, ad.pack_act);
            }
        }
    }
    if (blockIdx.x == 0 && wave == 1) {
        float s = 0.0f;
        for (int t = lane; t < (B + 15) / 16; t += 64) s = s + loss_part[t];
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
        if (tid == 0 && ticket == (unsigned)tiles - 1u) { st->b1pow = b1pow; st->b2pow = b2pow; st->adam_count += 1; st->arrive = 0; }
    }
}

void launch_dw_bf16(hipStream_t s, const NetDims &m, const float *px, const float *ph1, const float *ph2,
                    const float *pdz1, const float *pdz2, const float *pdz3, int B, float *grad,
                    const float *loss_part, float *loss_out, DqnState *st, int bump_ctr, const AdamArgs &adam,
                    const PwArgs &pw) {
    const Dims16 d = make_dims16(m);
    const int tiles = (m.H1 / 16) * (m.H2 / 16) + m.KQ1 * (m.H1 / 16) + m.H2 / 16;
    const int extra = pw.tree ? (pw.B + 63) / 64 : 0;
    auto bf = [](const float *p) { return reinterpret_cast<const __bf16 *>(p); };
    DQN_LAUNCH(k_dw16, dim3(tiles + extra), dim3(256), 0, s, m, d, bf(px), bf(ph1), bf(ph2), bf(pdz1), bf(pdz2),
                       bf(pdz3), B, grad, loss_part, loss_out, st, bump_ctr, adam, pw, tiles);
}

// ---------------------------------------------------------------------------- optimizer
__global__ void __launch_bounds__(256)
k_adam16(NetDims m, Dims16 d, DqnState *st, float *P, const float *__restrict__ g, float *mu, float *nu, __bf16 *pack,
         int adamw, float b1, float b2, float eps, float wd, float grad_scale, float *pack_act) {
    double b1pow, b2pow;
    const AdamCoef co = adam_coef(st, b1, b2, &b1pow, &b2pow);
    const int nP = (int)m.P;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nP; i += gridDim.x * blockDim.x) {
        const float p = adam_elem(co, g[i], P, mu, nu, i, adamw, b1, b2, eps, wd, grad_scale);
        scatter_packs16(m, d, i, p, pack);
        if (pack_act) scatter_actor_packs(m, i, p, pack_act);
    }
    LDS_BARRIER();                                           // (a barrier only: no need to drain this block's stores first)
    if (threadIdx.x == 0) {
        const unsigned int ticket = atomicAdd(&st->arrive, 1u);
        if (ticket == gridDim.x - 1) { st->b1pow = b1pow; st->b2pow = b2pow; st->adam_count += 1; st->arrive = 0; }
    }
}

void launch_adam_bf16(hipStream_t s, const NetDims &m, DqnState *st, float *params, const float *grad, float *mu,
                      float *nu, float *pack, int adamw, float b1, float b2, float eps, float wd, float grad_scale,
                      float *pack_act) {
    int blocks = (int)((m.P + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    DQN_LAUNCH(k_adam16, dim3(blocks), dim3(256), 0, s, m, make_dims16(m), st, params, grad, mu, nu,
                       reinterpret_cast<__bf16 *>(pack), adamw, b1, b2, eps, wd, grad_scale, pack_act);
}

// bf16 k-packed W2 behind the f32 actor shadows (scatter_actor_packs keeps it current afterwards)
__global__ void k_pack_w2k16(NetDims m, const float *__restrict__ params, float *__restrict__ pack_act) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= m.H1 * m.H2) return;
    const int k = u / m.H2, n = u - k * m.H2;
    reinterpret_cast<__bf16 *>(pack_act + m.pack_floats)[((long long)(k >> 3) * m.H2 + n) * 8 + (k & 7)] = (__bf16)params[m.o_w2 + u];
}
void launch_pack_w2k16(hipStream_t s, const NetDims &m, const float *params, float *pack_act) {
    hipLaunchKernelGGL(k_pack_w2k16, dim3((m.H1 * m.H2 + 255) / 256), dim3(256), 0, s, m, params, pack_act);
}

