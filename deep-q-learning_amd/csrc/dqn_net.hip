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
#include "dqn_device.h"
#include "dqn_launch.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// position of column c (0..15) inside its 16-block of an LDS A-operand row: the lane group
// g = lane>>4 reads 4 consecutive floats = k-steps j = 0..3 of k = 4*j + g
__device__ __forceinline__ int perm16(int c) { return ((c & 3) << 2) | (c >> 2); }

// index into a packed matrix of element [k][c]; KQ = number of 16-row k-blocks
__device__ __host__ __forceinline__ long long pidx(int KQ, int k, int c) {
    const int kq = k >> 4, kk = k & 15;
    return ((long long)((c >> 4) * KQ + kq)) * 256 + ((((kk & 3) << 4) | (c & 15)) << 2) + (kk >> 2);
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
    else                  { base = m.p_wht; KQ = 1;         which = 4; }
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
template <int TN>
__device__ __forceinline__ void mma_layer(const float *lds_a, int stride, int KQ,
                                          const float *__restrict__ pack, int CT, int wave, int lane,
                                          f32x4 (&acc)[TN]) {
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *arow = lds_a + (lane & 15) * stride + 4 * (lane >> 4);
    const float4 *pk = reinterpret_cast<const float4 *>(pack) + lane;
#pragma unroll 2
    for (int kq = 0; kq < KQ; ++kq) {
        const float4 a4 = *reinterpret_cast<const float4 *>(arow + 16 * kq);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int ct = wave + 4 * t;
            if (ct < CT) {
                const float4 b4 = pk[((long long)ct * KQ + kq) * 64];
                acc[t] = MFMA4(a4.x, b4.x, acc[t]);
                acc[t] = MFMA4(a4.y, b4.y, acc[t]);
                acc[t] = MFMA4(a4.z, b4.z, acc[t]);
                acc[t] = MFMA4(a4.w, b4.w, acc[t]);
            }
        }
    }
}

// ------------------------------------------------------------------------------ forward
// One workgroup (4 waves) = 16 batch rows of one pass. blockIdx.y selects the pass
// (online(s), online(s'), target(s') for compute_q_targets :52-54).
struct FwdPasses { FwdPass p[3]; };

template <int TN1, int TN2>
__global__ void __launch_bounds__(256)
k_qnet_fwd(NetDims m, FwdPasses passes, int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FwdPass ps = passes.p[blockIdx.y];
    const int tile = blockIdx.x, row0 = tile * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KQb = (B + 15) / 16;
    const int K1 = m.KQ1 * 16;
    const int sx = K1 + 4, s1 = m.H1 + 4, s2 = m.H2 + 4;
    float *lx = lds, *l1 = lx + 16 * sx, *l2 = l1 + 16 * s1, *lh = l2 + 16 * s2;   // lh: [16][16]

    // stage the 16 input rows (zero-padded) in A-operand order
    for (int t = tid; t < 16 * K1; t += 256) {
        const int rl = t / K1, c = t - rl * K1;
        float v = 0.0f;
        if (row0 + rl < B && c < m.D) v = ps.x[(long long)(row0 + rl) * m.D + c];
        lx[rl * sx + (c & ~15) + perm16(c & 15)] = v;
        if (ps.px) ps.px[pidx(KQb, row0 + rl, c)] = v;
    }
    __syncthreads();

    // layer 1: h1 = relu(x @ w1 + b1)                                      dddqn.py:25-26
    {
        f32x4 acc[TN1];
        mma_layer<TN1>(lx, sx, m.KQ1, ps.pack + m.p_w1, m.H1 / 16, wave, lane, acc);
#pragma unroll
        for (int t = 0; t < TN1; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H1 / 16) {
                const int c = lane & 15, col = 16 * ct + c;
                const float bias = ps.params[m.o_b1 + col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * (lane >> 4) + r;
                    float v = acc[t][r] + bias;
                    v = v > 0.0f ? v : 0.0f;
                    l1[rl * s1 + 16 * ct + perm16(c)] = v;
                    if (ps.ph1) ps.ph1[pidx(KQb, row0 + rl, col)] = v;
                }
            }
        }
    }
    __syncthreads();

    // layer 2: h2 = relu(h1 @ w2 + b2)                                     dddqn.py:27-28
    {
        f32x4 acc[TN2];
        mma_layer<TN2>(l1, s1, m.H1 / 16, ps.pack + m.p_w2, m.H2 / 16, wave, lane, acc);
#pragma unroll
        for (int t = 0; t < TN2; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H2 / 16) {
                const int c = lane & 15, col = 16 * ct + c;
                const float bias = ps.params[m.o_b2 + col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * (lane >> 4) + r;
                    float v = acc[t][r] + bias;
                    v = v > 0.0f ? v : 0.0f;
                    l2[rl * s2 + 16 * ct + perm16(c)] = v;
                    if (ps.ph2) ps.ph2[pidx(KQb, row0 + rl, col)] = v;
                    if (ps.feat && row0 + rl < B) ps.feat[(long long)(row0 + rl) * m.H2 + col] = v;   // :32-33
                }
            }
        }
    }
    __syncthreads();

    // heads: column 0 = val (dddqn.py:29), columns 1..A = adv (:30); one 16-column tile
    if (wave == 0) {
        f32x4 acc[1];
        mma_layer<1>(l2, s2, m.H2 / 16, ps.pack + m.p_wh, 1, 0, lane, acc);
        const int c = lane & 15;
        float bias = 0.0f;
        if (c == 0) bias = ps.params[m.o_bv];
        else if (c <= m.A) bias = ps.params[m.o_ba + c - 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) lh[(4 * (lane >> 4) + r) * 16 + c] = acc[0][r] + bias;
    }
    __syncthreads();

    // Q = val + adv - mean(adv)                                            dddqn.py:31
    if (tid < 16 && row0 + tid < B && ps.q) {
        const float *hr = lh + tid * 16;
        float sum = 0.0f;
        for (int a = 0; a < m.A; ++a) sum = sum + hr[1 + a];
        const float mean = __fdiv_rn(sum, (float)m.A);
        for (int a = 0; a < m.A; ++a) ps.q[(long long)(row0 + tid) * m.A + a] = (hr[0] + hr[1 + a]) - mean;
    }
}

static inline int tn_of(int H) { const int ct = H / 16; return ct <= 4 ? 1 : (ct <= 8 ? 2 : 4); }

void launch_qnet_fwd(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B) {
    FwdPasses ps{};
    for (int i = 0; i < npass; ++i) ps.p[i] = passes[i];
    const dim3 grid((B + 15) / 16, npass), block(256);
    const size_t lds = sizeof(float) * (16 * (m.KQ1 * 16 + 4) + 16 * (m.H1 + 4) + 16 * (m.H2 + 4) + 256);
    const int t1 = tn_of(m.H1), t2 = tn_of(m.H2);
#define FWD_CASE(A1, A2) if (t1 == A1 && t2 == A2) { hipLaunchKernelGGL((k_qnet_fwd<A1, A2>), grid, block, lds, s, m, ps, B); return; }
    FWD_CASE(1, 1) FWD_CASE(1, 2) FWD_CASE(1, 4) FWD_CASE(2, 1) FWD_CASE(2, 2) FWD_CASE(2, 4)
    FWD_CASE(4, 1) FWD_CASE(4, 2) FWD_CASE(4, 4)
#undef FWD_CASE
}

// --------------------------------------------------------------- per-sample TD arithmetic
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
    __shared__ float red[256];
    const int tile = blockIdx.x, row0 = tile * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KQb = (B + 15) / 16;
    const int A = m.A;
    const int s3 = 16 + 4, s2 = m.H2 + 4;
    float *l3 = lds, *l2 = l3 + 16 * s3, *lrow = l2 + 16 * s2;     // lrow[16]: per-row loss

    // max raw IS weight over the batch (order-independent), for w_i / max_j w_j
    float wmax = 1.0f;
    if (g.w_raw) {
        float mx = 0.0f;
        for (int j = tid; j < B; j += 256) mx = fmaxf(mx, g.w_raw[j]);
        red[tid] = mx;
        __syncthreads();
        for (int sft = 128; sft > 0; sft >>= 1) {
            if (tid < sft) red[tid] = fmaxf(red[tid], red[tid + sft]);
            __syncthreads();
        }
        wmax = red[0];
        if (tile == 0 && tid == 0) st->wmax = wmax;
    }

    for (int t = tid; t < 16 * s3; t += 256) l3[t] = 0.0f;
    __syncthreads();

    if (tid < 16) {
        const int i = row0 + tid;
        float rowloss = 0.0f;
        if (i < B) {
            float qr[16], tr[16];
            for (int k = 0; k < A; ++k) qr[k] = g.q[(long long)i * A + k];
            float delta = 0.0f;
            const int ai = g.a ? g.a[i] : 0;
            if (g.targets) {
                for (int k = 0; k < A; ++k) tr[k] = g.targets[(long long)i * A + k];
            } else {
                float nqr[16], ntr[16];
                for (int k = 0; k < A; ++k) { nqr[k] = g.nq[(long long)i * A + k]; ntr[k] = g.nt[(long long)i * A + k]; }
                const float df = g.d_f32 ? g.d_f32[i] : (g.d_u8[i] ? 1.0f : 0.0f);     // preprocessing :84
                delta = td_row(qr, nqr, ntr, ai, g.r[i], df, g.gamma, A, tr);
                if (g.td) g.td[i] = delta;
                if (g.td_abs) g.td_abs[i] = fabsf(delta);
            }
            float w = 1.0f;
            if (g.w_raw) { w = __fdiv_rn(g.w_raw[i], wmax); if (g.isw_out) g.isw_out[i] = w; }
            else if (g.isw) w = g.isw[i];
            const float invB = __fdiv_rn(1.0f, (float)B);
            float gk[16], gsum = 0.0f;
            for (int k = 0; k < A; ++k) {
                const float e = qr[k] - tr[k];                 // pred - target, pred == q   (:35)
                rowloss = rowloss + huber(e);                  // :36
                const float c = e > 1.0f ? 1.0f : (e < -1.0f ? -1.0f : e);
                gk[k] = (w * c) * invB;                        // dL/dpred
                gsum = gsum + gk[k];
                if (g.dq) g.dq[(long long)i * A + k] = gk[k];
                if (g.targets_out) g.targets_out[(long long)i * A + k] = tr[k];
            }
            if (g.w_raw || g.isw) rowloss = w * rowloss;
            // dueling backward: dv = sum_a g_a ; dadv_j = g_j - (1/A) sum_a g_a
            const float gmean = __fdiv_rn(gsum, (float)A);
            l3[tid * s3 + perm16(0)] = gsum;
            for (int k = 0; k < A; ++k) l3[tid * s3 + perm16(1 + k)] = gk[k] - gmean;
        }
        lrow[tid] = rowloss;
    }
    __syncthreads();
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
        mma_layer<TN2>(l3, s3, 1, g.pack + m.p_wht, m.H2 / 16, wave, lane, acc);
#pragma unroll
        for (int t = 0; t < TN2; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H2 / 16) {
                const int c = lane & 15, col = 16 * ct + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * (lane >> 4) + r;
                    const long long pi = pidx(KQb, row0 + rl, col);
                    const float v = g.ph2[pi] > 0.0f ? acc[t][r] : 0.0f;
                    l2[rl * s2 + 16 * ct + perm16(c)] = v;
                    g.pdz2[pi] = v;
                }
            }
        }
    }
    __syncthreads();

    // dz1 = (dz2 . W2^T) * (h1 > 0)
    {
        f32x4 acc[TN1];
        mma_layer<TN1>(l2, s2, m.H2 / 16, g.pack + m.p_w2t, m.H1 / 16, wave, lane, acc);
#pragma unroll
        for (int t = 0; t < TN1; ++t) {
            const int ct = wave + 4 * t;
            if (ct < m.H1 / 16) {
                const int c = lane & 15, col = 16 * ct + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * (lane >> 4) + r;
                    const long long pi = pidx(KQb, row0 + rl, col);
                    g.pdz1[pi] = g.ph1[pi] > 0.0f ? acc[t][r] : 0.0f;
                }
            }
        }
    }
}

void launch_bwd_rows(hipStream_t s, const NetDims &m, const BwdArgs &g, int B, DqnState *st) {
    const dim3 grid((B + 15) / 16), block(256);
    const size_t lds = sizeof(float) * (16 * 20 + 16 * (m.H2 + 4) + 16);
    const int t1 = tn_of(m.H1), t2 = tn_of(m.H2);
#define BWD_CASE(A1, A2) if (t1 == A1 && t2 == A2) { hipLaunchKernelGGL((k_bwd_rows<A1, A2>), grid, block, lds, s, m, g, B, st); return; }
    BWD_CASE(1, 1) BWD_CASE(1, 2) BWD_CASE(1, 4) BWD_CASE(2, 1) BWD_CASE(2, 2) BWD_CASE(2, 4)
    BWD_CASE(4, 1) BWD_CASE(4, 2) BWD_CASE(4, 4)
#undef BWD_CASE
}

// ------------------------------------------------------------------ weight gradients
// dW = H^T . Z over the batch: one workgroup per 16x16 tile of one weight block, the batch
// (K) split over its 4 waves and combined through LDS in a fixed order (deterministic).
// The tiles with mt == 0 also produce the bias gradient (column sums of Z).
__global__ void __launch_bounds__(256)
k_dw(NetDims m, const float *__restrict__ px, const float *__restrict__ ph1, const float *__restrict__ ph2,
     const float *__restrict__ pdz1, const float *__restrict__ pdz2, const float *__restrict__ pdz3, int B,
     float *grad, const float *loss_part, float *loss_out, DqnState *st, int bump_ctr) {
    __shared__ float red[4][64][4];
    __shared__ float redb[4][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KQb = (B + 15) / 16;
    const int MT2 = m.H1 / 16, NT2 = m.H2 / 16, MT1 = m.KQ1, NT1 = m.H1 / 16, MTH = m.H2 / 16;
    int b = blockIdx.x;
    const float *pa, *pb; int mt, nt, which;
    if (b < MT2 * NT2)                  { which = 2; mt = b / NT2; nt = b % NT2; pa = ph1; pb = pdz2; }
    else if ((b -= MT2 * NT2) < MT1 * NT1) { which = 1; mt = b / NT1; nt = b % NT1; pa = px;  pb = pdz1; }
    else                                 { b -= MT1 * NT1; which = 3; mt = b; nt = 0; pa = ph2; pb = pdz3; (void)MTH; }

    const int per = (KQb + 3) / 4;
    const int k0 = wave * per, k1 = (k0 + per < KQb) ? k0 + per : KQb;
    const float4 *A4 = reinterpret_cast<const float4 *>(pa) + ((long long)mt * KQb) * 64 + lane;
    const float4 *B4 = reinterpret_cast<const float4 *>(pb) + ((long long)nt * KQb) * 64 + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
#pragma unroll 4
    for (int kq = k0; kq < k1; ++kq) {
        const float4 a4 = A4[(long long)kq * 64], b4 = B4[(long long)kq * 64];
        acc = MFMA4(a4.x, b4.x, acc);
        acc = MFMA4(a4.y, b4.y, acc);
        acc = MFMA4(a4.z, b4.z, acc);
        acc = MFMA4(a4.w, b4.w, acc);
        bsum = (((bsum + b4.x) + b4.y) + b4.z) + b4.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc[r];
    redb[wave][lane] = bsum;
    __syncthreads();
    if (wave == 0) {
        const int c = lane & 15, n = 16 * nt + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = ((red[0][lane][r] + red[1][lane][r]) + red[2][lane][r]) + red[3][lane][r];
            const int mrow = 16 * mt + 4 * (lane >> 4) + r;
            if (which == 2) grad[m.o_w2 + (long long)mrow * m.H2 + n] = v;
            else if (which == 1) { if (mrow < m.D) grad[m.o_w1 + (long long)mrow * m.H1 + n] = v; }
            else { if (n == 0) grad[m.o_wv + mrow] = v; else if (n <= m.A) grad[m.o_wa + (long long)mrow * m.A + (n - 1)] = v; }
        }
        if (mt == 0 && lane < 16) {
            float s = 0.0f;
            for (int w = 0; w < 4; ++w)
                for (int gq = 0; gq < 4; ++gq) s = s + redb[w][16 * gq + lane];
            if (which == 2) grad[m.o_b2 + n] = s;
            else if (which == 1) grad[m.o_b1 + n] = s;
            else { if (n == 0) grad[m.o_bv] = s; else if (n <= m.A) grad[m.o_ba + n - 1] = s; }
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        float s = 0.0f;
        for (int t = 0; t < KQb; ++t) s = s + loss_part[t];
        const float L = __fdiv_rn(s, (float)B);
        st->loss = L;
        if (loss_out) *loss_out = L;
        if (bump_ctr) st->sample_ctr += 1ull;
    }
}

void launch_dw(hipStream_t s, const NetDims &m, const float *px, const float *ph1, const float *ph2,
               const float *pdz1, const float *pdz2, const float *pdz3, int B, float *grad,
               const float *loss_part, float *loss_out, DqnState *st, int bump_ctr) {
    const int tiles = (m.H1 / 16) * (m.H2 / 16) + m.KQ1 * (m.H1 / 16) + m.H2 / 16;
    hipLaunchKernelGGL(k_dw, dim3(tiles), dim3(256), 0, s, m, px, ph1, ph2, pdz1, pdz2, pdz3, B, grad,
                       loss_part, loss_out, st, bump_ctr);
}

// ---------------------------------------------------------------------------- optimizer
// optax scale_by_adam -> add_decayed_weights (adamw) -> scale(-lr) -> apply_updates, and the
// refresh of the fragment-packed shadows in the same pass. Bit-exact vs the CPU restatement.
__device__ __forceinline__ void scatter_packs(const NetDims &m, long long i, float v, float *pack) {
    if (i < m.o_b1) {                                   // w1[k][n]
        const int k = (int)(i / m.H1), n = (int)(i % m.H1);
        pack[m.p_w1 + pidx(m.KQ1, k, n)] = v;
    } else if (i >= m.o_w2 && i < m.o_b2) {             // w2[k][n]
        const long long u = i - m.o_w2;
        const int k = (int)(u / m.H2), n = (int)(u % m.H2);
        pack[m.p_w2 + pidx(m.H1 / 16, k, n)] = v;
        pack[m.p_w2t + pidx(m.H2 / 16, n, k)] = v;
    } else if (i >= m.o_wv && i < m.o_bv) {             // wv[k]
        const int k = (int)(i - m.o_wv);
        pack[m.p_wh + pidx(m.H2 / 16, k, 0)] = v;
        pack[m.p_wht + pidx(1, 0, k)] = v;
    } else if (i >= m.o_wa && i < m.o_ba) {             // wa[k][a]
        const long long u = i - m.o_wa;
        const int k = (int)(u / m.A), a = (int)(u % m.A);
        pack[m.p_wh + pidx(m.H2 / 16, k, 1 + a)] = v;
        pack[m.p_wht + pidx(1, 1 + a, k)] = v;
    }
}

__global__ void __launch_bounds__(256)
k_adam(NetDims m, DqnState *st, float *P, const float *__restrict__ g, float *mu, float *nu, float *pack,
       int adamw, float b1, float b2, float eps, float wd, float grad_scale) {
    const double b1pow = st->b1pow * (double)b1, b2pow = st->b2pow * (double)b2;
    const float c1 = (float)(1.0 - b1pow), c2 = (float)(1.0 - b2pow);
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2, neglr = -st->lr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m.P; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        const float mm = (b1 * mu[i]) + (omb1 * gi);
        const float vv = (b2 * nu[i]) + (omb2 * (gi * gi));
        mu[i] = mm; nu[i] = vv;
        const float mhat = __fdiv_rn(mm, c1), vhat = __fdiv_rn(vv, c2);
        float u = __fdiv_rn(mhat, __fsqrt_rn(vhat) + eps);
        float p = P[i];
        if (adamw) u = u + (wd * p);
        p = p + (neglr * u);
        P[i] = p;
        scatter_packs(m, i, p, pack);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int ticket = atomicAdd(&st->arrive, 1u);
        if (ticket == gridDim.x - 1) {
            st->b1pow = b1pow; st->b2pow = b2pow; st->adam_count += 1; st->arrive = 0;
            __threadfence();
        }
    }
}

void launch_adam(hipStream_t s, const NetDims &m, DqnState *st, float *params, const float *grad, float *mu,
                 float *nu, float *pack, int adamw, float b1, float b2, float eps, float wd, float grad_scale) {
    int blocks = (int)((m.P + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, m, st, params, grad, mu, nu, pack, adamw, b1, b2,
                       eps, wd, grad_scale);
}

// ----------------------------------------------------------------------- epsilon-greedy
// Agent._policy (q_agent.py:137-141): greedy iff eps < U(0,1) else randint(0, A);
// compute_action (q_learning_functions.py:70): argmax, first max wins.
__global__ void __launch_bounds__(256)
k_policy(const float *__restrict__ q, int n, int A, float epsilon, unsigned long long seed,
         unsigned long long ctr, int32_t *actions, const DqnState *st_from) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (st_from) { epsilon = st_from->epsilon; ctr = st_from->env_ctr; }
    const u32x4 o = philox_draw(seed, ctr, (uint32_t)i, DQN_STREAM_POLICY);
    int act;
    if (epsilon < u01(o.x)) {
        act = 0;
        for (int k = 1; k < A; ++k) if (q[(long long)i * A + k] > q[(long long)i * A + act]) act = k;
    } else {
        act = (int)(((unsigned long long)o.y * (unsigned long long)A) >> 32);
    }
    actions[i] = act;
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
