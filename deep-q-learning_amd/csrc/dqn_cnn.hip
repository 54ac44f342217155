// csrc/dqn_cnn.hip -- Nature-CNN dueling Q-network for BASELINE configs[4] (PongNoFrameskip-v4 shape):
// forward, loss gradient, AdamW step, Agent._step on a given minibatch or from a ring of u8 frame stacks, epsilon-greedy acting.
// Not in the reference, which has only the MLP of LunarLander/dddqn.py:19-22; the trunk ends in the reference's dueling head
// (dddqn.py:29-31: Q = val + adv - mean(adv)), its Q values feed the reference's TD rule unchanged
// (General/QLearning/q_learning_functions.py:55-60, td_row), its loss is the reference's (:31-39).
//
//   frames u8 [B][84][84][4] (NHWC, 4 stacked frames) / 255
//   conv1 32 x 8x8 / 4 -> [B][20][20][32]   conv2 64 x 4x4 / 2 -> [B][9][9][64]   conv3 64 x 3x3 / 1 -> [B][7][7][64]
//   fc 3136 -> 512, ReLU after each; val 512 -> 1, adv 512 -> A
//
// Forward: every layer is ONE implicit GEMM kernel, Out[M][N] = relu(Patch[M][K] . W[K][N] + b): row m = output position
// (b, oh, ow), k = (kh, kw, c) -- in NHWC a patch row (kh fixed) is KW*IC contiguous elements, a multiple of 8 for every layer,
// so 8-element pieces are gathered straight from the activation tensor into an LDS image, no im2col buffer. Weights are kept
// transposed ([N][K], k contiguous) so that both MFMA operands are 16-byte LDS reads of consecutive k.
//   precision bf16: v_mfma_f32_32x32x16_bf16 / 16x16x32, activations / weights bf16, f32 accumulate (tolerance 2e-2 of scale)
//   precision f32 : v_mfma_f32_32x32x2_f32 / 16x16x4, exact: every output is the k-ascending fmaf chain of the CPU restatement
// Backward: see the section "backward (loss gradient)" below; host side and C ABI at the end of the file.
#include "../../include/dqn_hip.h"
#include "dqn_device.h"
#include "dqn_launch.h"
#include "dqn_net_common.h"

#include <new>
#include <string>

typedef float f32x16c __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));

// Geometry of the four GEMM layers, compile-time (every index division below folds into shifts / multiplies), with the tile
// shape chosen per layer (CnnTile below).
//   conv1  M = 400 B, K = 256 (4 chunks),  N = 32   128 x 32 tile, everything requested up front
//   conv2  M =  81 B, K = 512 (8),         N = 64    64 x 64
//   conv3  M =  49 B, K = 576 (9),         N = 64    64 x 64
//   fc     M =      B, K = 3136 (49),      N = 512   32 x 32 from 16 x 16 MFMA tiles: at B = 512 that is 256 workgroups, and in
//          f32 mode the k-ascending chain of 16x16x4 is 784 MFMAs x 32 cycles (the 32x32x2 chain would be 1568 x 64)
template <int L> struct CnnGeo;
template <> struct CnnGeo<0> { static constexpr int IH = 84, IW = 84, IC = 4, OH = 20, OW = 20, OC = 32, KH = 8, KW = 8, S = 4; };
template <> struct CnnGeo<1> { static constexpr int IH = 20, IW = 20, IC = 32, OH = 9, OW = 9, OC = 64, KH = 4, KW = 4, S = 2; };
template <> struct CnnGeo<2> { static constexpr int IH = 9, IW = 9, IC = 64, OH = 7, OW = 7, OC = 64, KH = 3, KW = 3, S = 1; };
template <> struct CnnGeo<3> { static constexpr int IH = 1, IW = 1, IC = 3136, OH = 1, OW = 1, OC = 512, KH = 1, KW = 1, S = 1; };
// Tile shape of the forward kernel, variant V. Workgroup tile = (MT WM TM) rows x (MT WN TN) columns: four waves as WM x WN,
// each TM x TN MFMA tiles of MT x MT; R = depth of the register prefetch ring.
//   V = 0 (small batches: many workgroups, one MFMA tile per wave -- the shapes in the table above)
//   V = 1 (fc layer, bf16 mode, B >= 8192 = 256 workgroups): 2 x 2 MFMA tiles per wave, 128 x 128 workgroup tile -- 88 -> 45 us at B = 8192.
//         The same idea on the convolutions (256-row tiles, 2 x 2 tiles per wave, one workgroup per CU) was 30-80 % SLOWER than
//         V = 0 at B = 8192: those kernels are bound by staging (L2 -> registers -> LDS, 11 TB/s of L2 reads at B = 8192, the
//         4x im2col expansion included), and four small workgroups per CU overlap it where one large one serialises it
//         (128 x 64 tiles, two workgroups per CU: 123 -> 145 us for conv2, 61 -> 83 us for conv3 -- slower as well)
//         and smaller ones at B = 512 (32 x 32 from 16 x 16 MFMA tiles: conv2 10.3 -> 12.6 us, conv3 8.2 -> 9.3 us): 64 x 64 is the optimum
template <int L, int V> struct CnnTile;
template <> struct CnnTile<0, 0> { static constexpr int MT = 32, WM = 4, WN = 1, TM = 1, TN = 1, R = 4; };
template <> struct CnnTile<1, 0> { static constexpr int MT = 32, WM = 2, WN = 2, TM = 1, TN = 1, R = 4; };
template <> struct CnnTile<2, 0> { static constexpr int MT = 32, WM = 2, WN = 2, TM = 1, TN = 1, R = 3; };
template <> struct CnnTile<3, 0> { static constexpr int MT = 16, WM = 2, WN = 2, TM = 1, TN = 1, R = 7; };
template <> struct CnnTile<3, 1> { static constexpr int MT = 32, WM = 2, WN = 2, TM = 2, TN = 2, R = 4; };

constexpr int KC = 64;        // k-chunk (a multiple of 8: an 8-element piece never crosses a patch row)
constexpr int CNN_F = 512;    // fc width = the heads' K

// 8 consecutive k of one image row, as loaded (raw) and as committed to LDS (compute type)
template <typename T> struct RawPiece;
template <> struct RawPiece<uint8_t> { typedef uint2 t; };
template <> struct RawPiece<__bf16> { typedef bf16x8c t; };
struct f32x8raw { float4 a, b; };
template <> struct RawPiece<float> { typedef f32x8raw t; };

template <typename T> __device__ __forceinline__ typename RawPiece<T>::t load_piece(const T *p) {
    return *reinterpret_cast<const typename RawPiece<T>::t *>(p);
}
// u8 pixels: v = (float)u8 / 255.0f, read from a 256-entry LDS table of exactly those quotients
__device__ __forceinline__ void piece_f32(const uint2 &raw, const float *lut, float (&v)[8]) {
    const uint32_t w[2] = {raw.x, raw.y};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = lut[(w[j >> 2] >> (8 * (j & 3))) & 0xffu];
}
__device__ __forceinline__ void piece_f32(const f32x8raw &raw, const float *, float (&v)[8]) {
    v[0] = raw.a.x; v[1] = raw.a.y; v[2] = raw.a.z; v[3] = raw.a.w; v[4] = raw.b.x; v[5] = raw.b.y; v[6] = raw.b.z; v[7] = raw.b.w;
}
__device__ __forceinline__ void piece_f32(const bf16x8c &raw, const float *, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)raw[j];
}
// LDS image of a 64-deep chunk of one row. bf16: k in order (a lane's MFMA operand = 8 consecutive k = one 16-byte read).
// f32: k permuted so that the k a lane feeds to consecutive MFMAs are consecutive words (16-byte reads again):
//   32x32x2 (lane half h takes k = 2s + h):  position (k & 1) * 32 + (k >> 1)
//   16x16x4 (lane group g takes k = 4s + g): position (k & 3) * 16 + (k >> 2)
// bf16 mode keeps the pixels as the integers 0..255 (exact in bf16: v_cvt_f32_ubyte + pack, no table) and folds the 1/255
// into conv1's weight shadow (k_cnn_pack with scale 255).
template <int MT, typename TR>
__device__ __forceinline__ void commit_piece(__bf16 *row, int j, const TR &raw, const float *) {
    if constexpr (__is_same(TR, bf16x8c)) *reinterpret_cast<bf16x8c *>(row + 8 * j) = raw;
    else {
        static_assert(__is_same(TR, uint2), "bf16 layers read u8 frames or bf16 activations");
        const uint32_t w[2] = {raw.x, raw.y};
        bf16x8c a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
        *reinterpret_cast<bf16x8c *>(row + 8 * j) = a;
    }
}
template <int MT, typename TR>
__device__ __forceinline__ void commit_piece(float *row, int j, const TR &raw, const float *lut) {
    float v[8]; piece_f32(raw, lut, v);
    if constexpr (MT == 32) {
        *reinterpret_cast<float4 *>(row + 4 * j) = float4{v[0], v[2], v[4], v[6]};
        *reinterpret_cast<float4 *>(row + 32 + 4 * j) = float4{v[1], v[3], v[5], v[7]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float2 *>(row + 16 * i + 2 * j) = float2{v[i], v[4 + i]};
    }
}

// TI: element type of the input tensor (uint8_t frames, or the compute type); TC: compute / weight / output type.
// The loop over k-chunks is a software pipeline: chunk kc + R is requested into the register slot chunk kc has just left (R
// chunks of global latency in flight per thread), the LDS image is double-buffered (one barrier per chunk; conv1, four chunks all requested
// up front, keeps a single buffer and more workgroups per CU). All slot indices are compile-time, so every s_waitcnt counts the loads of
// the newer slots instead of draining the queue. Workgroups are numbered so that the ones an XCD receives (id mod 8) are
// neighbours in m (shared patch rows / weight columns stay in that XCD's L2).
template <typename TI, typename TC, int L, int V>
__global__ void __launch_bounds__(256)
k_cnn_layer(int M, const TI *__restrict__ in, const TI *__restrict__ in2, int images1, const TC *__restrict__ wt, const float *__restrict__ bias, TC *__restrict__ out) {
    // images 0 .. images1-1 come from `in`, the rest from `in2` (two frame batches through one pass; layers above conv1: one tensor)
    typedef CnnGeo<L> G; typedef CnnTile<L, V> T;
    constexpr int MT = T::MT, WM = T::WM, WN = T::WN, TM = T::TM, TN = T::TN, R = T::R;
    constexpr int BM = MT * WM * TM, BN = MT * WN * TN, K = G::KH * G::KW * G::IC, ROWLEN = G::KW * G::IC, NCH = K / KC;
    constexpr int LS = KC + (sizeof(TC) == 2 ? 8 : 4);
    constexpr int LBUF = L == 0 ? 1 : 2;
    constexpr int APT = BM * 8 / 256, BPT = BN * 8 / 256, NACC = MT == 32 ? 16 : 4;
    static_assert(WM * WN == 4 && K % KC == 0 && R <= NCH && G::OC % BN == 0 && ROWLEN % 8 == 0 && APT >= 1 && BPT >= 1, "tile shape");
    typedef typename RawPiece<TI>::t RA;
    typedef typename RawPiece<TC>::t RB;
    __shared__ __attribute__((aligned(16))) TC lA[LBUF * BM * LS];
    __shared__ __attribute__((aligned(16))) TC lB[LBUF * BN * LS];
    __shared__ float lut[(sizeof(TI) == 1 && sizeof(TC) == 4) ? 256 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    if constexpr (sizeof(TI) == 1 && sizeof(TC) == 4) lut[tid] = __fdiv_rn((float)tid, 255.0f);          // barrier below, behind the first requests
    constexpr int NT = G::OC / BN;
    const int mtiles = (M + BM - 1) / BM, total = mtiles * NT;
    int t = blockIdx.x;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
    const int m0 = (t / NT) * BM, n0 = (t % NT) * BN;
    const int pj = tid & 7;                                                            // this thread's k-piece within a row
    const TI *ap[APT]; const TC *bp[BPT];
#pragma unroll
    for (int u = 0; u < APT; ++u) {
        int mm = m0 + (tid >> 3) + 32 * u;
        mm = mm < M ? mm : M - 1;                                                      // rows past the end: a valid row, never stored
        const unsigned ow = (unsigned)mm % G::OW, t2 = (unsigned)mm / G::OW, oh = t2 % G::OH, b = t2 / G::OH;
        const bool second = (int)b >= images1;
        ap[u] = (second ? in2 : in) + ((long long)((second ? b - images1 : b) * G::IH + oh * G::S) * G::IW + ow * G::S) * G::IC;
    }
#pragma unroll
    for (int u = 0; u < BPT; ++u) bp[u] = wt + (long long)(n0 + (tid >> 3) + 32 * u) * K + 8 * pj;
    RA ra[R][APT]; RB rb[R][BPT];
    auto request = [&](int slot, int kc) {
        const int kp = kc * KC + 8 * pj, kh = kp / ROWLEN, rem = kp - kh * ROWLEN, koff = kh * (G::IW * G::IC) + rem;
#pragma unroll
        for (int u = 0; u < APT; ++u) ra[slot][u] = load_piece<TI>(ap[u] + koff);
#pragma unroll
        for (int u = 0; u < BPT; ++u) rb[slot][u] = load_piece<TC>(bp[u] + kc * KC);
    };
    auto commit = [&](int slot, int buf) {
        TC *a = lA + buf * (BM * LS) + (tid >> 3) * LS, *b = lB + buf * (BN * LS) + (tid >> 3) * LS;
#pragma unroll
        for (int u = 0; u < APT; ++u) commit_piece<MT>(a + 32 * u * LS, pj, ra[slot][u], lut);
#pragma unroll
        for (int u = 0; u < BPT; ++u) commit_piece<MT>(b + 32 * u * LS, pj, rb[slot][u], lut);
    };
    typedef float accv __attribute__((ext_vector_type(NACC)));
    accv acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[i][j][r] = 0.0f;
    const int hi = lane / MT, c = lane % MT;                                           // MT = 32: half h; MT = 16: k-group g
    auto multiply = [&](int buf) {
        const TC *ar = lA + buf * (BM * LS) + (MT * TM * wm + c) * LS, *br = lB + buf * (BN * LS) + (MT * TN * wn + c) * LS;
        if constexpr (sizeof(TC) == 2) {
            constexpr int KS = MT == 32 ? 16 : 32;                                     // k per MFMA
#pragma unroll
            for (int ks = 0; ks < KC / KS; ++ks) {
                bf16x8c fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8c *>(ar + MT * i * LS + KS * ks + 8 * hi);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8c *>(br + MT * j * LS + KS * ks + 8 * hi);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (MT == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            constexpr int QN = MT == 32 ? 8 : 4, HO = MT == 32 ? 32 : 16;              // 16-byte reads per chunk; offset of the lane's k-class
#pragma unroll
            for (int q = 0; q < QN; ++q) {                                             // k ascending within every accumulator's chain
                float4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4 *>(ar + MT * i * LS + HO * hi + 4 * q);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4 *>(br + MT * j * LS + HO * hi + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float av = e == 0 ? fa[i].x : e == 1 ? fa[i].y : e == 2 ? fa[i].z : fa[i].w;
                            const float bv = e == 0 ? fb[j].x : e == 1 ? fb[j].y : e == 2 ? fb[j].z : fb[j].w;
                            if constexpr (MT == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i][j], 0, 0, 0);
                        }
            }
        }
    };
#pragma unroll
    for (int r = 0; r < R; ++r) request(r, r);
    if constexpr (sizeof(TI) == 1 && sizeof(TC) == 4) __syncthreads();                 // the table, before the first commit reads it
#pragma unroll                      // fully: a rolled loop turns the ring into register copies that wait for the loads just issued
    for (int kc = 0; kc < NCH; ++kc) {
        const int buf = LBUF == 2 ? (kc & 1) : 0;
        commit(kc % R, buf);
        if (kc + R < NCH) request(kc % R, kc + R);
        __syncthreads();
        multiply(buf);
        if constexpr (LBUF == 1) __syncthreads();
    }
    // outputs first, then the (predicated) stores: with the bias add inside the row test every store block starts with a
    // vmcnt(0) for the bias load -- which then waits for the PREVIOUS STORE (stores count in vmcnt on gfx9): 16 serial round trips
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + MT * (TN * wn + j) + c;
        const float bv = bias[n];
        float o[TM][NACC];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                const float v = acc[i][j][r] + bv;
                o[i][r] = v > 0.0f ? v : 0.0f;
                asm volatile("" : "+v"(o[i][r]));                   // materialised here: nothing of the bias load is left for the store blocks
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                const int mm = m0 + MT * (TM * wm + i) + (MT == 32 ? (r & 3) + 8 * (r >> 2) + 4 * hi : 4 * hi + r);
                if (mm < M) out[(long long)mm * G::OC + n] = (TC)o[i][r];
            }
    }
}

#include "dqn_cnn_trunk.h"
#include "dqn_cnn_btrunk.h"

// dueling head (dddqn.py:29-31) on the fc features [B][512]: 16 rows per workgroup, thread (row, j) runs the k-ascending fmaf
// chain of output j (0 = val, 1.. = adv) from LDS images of the rows and of the j-major head weights [16][512];
// Q = val + adv - mean(adv) with the restatement's order of additions.
template <typename TC>
__global__ void __launch_bounds__(256)
k_cnn_head(const TC *__restrict__ feat, const float *__restrict__ wht, const float *__restrict__ bh, int A, int B, float *q) {
    constexpr int LS = CNN_F + 4;
    __shared__ __attribute__((aligned(16))) float lx[16 * LS];
    __shared__ __attribute__((aligned(16))) float lw[16 * LS];
    __shared__ float lo[16 * 16];
    const int tid = threadIdx.x, r0 = blockIdx.x * 16;
#pragma unroll
    for (int u = 0; u < 4; ++u) {                                                       // 16 rows x 64 pieces of 8
        const int p = tid + 256 * u, row = p >> 6, j = p & 63;
        int i = r0 + row; i = i < B ? i : B - 1;
        float v[8]; piece_f32(load_piece<TC>(feat + (long long)i * CNN_F + 8 * j), nullptr, v);
        *reinterpret_cast<float4 *>(lx + row * LS + 8 * j) = float4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<float4 *>(lx + row * LS + 8 * j + 4) = float4{v[4], v[5], v[6], v[7]};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {                                                       // 16 outputs x 128 float4
        const int p = tid + 256 * u, j = p >> 7, k4 = p & 127;
        *reinterpret_cast<float4 *>(lw + j * LS + 4 * k4) = *reinterpret_cast<const float4 *>(wht + j * CNN_F + 4 * k4);
    }
    __syncthreads();
    const int row = tid >> 4, j = tid & 15;
    const float *x = lx + row * LS, *w = lw + j * LS;
    float acc = 0.0f;
#pragma unroll 8
    for (int k4 = 0; k4 < CNN_F / 4; ++k4) {
        const float4 xv = *reinterpret_cast<const float4 *>(x + 4 * k4), wv = *reinterpret_cast<const float4 *>(w + 4 * k4);
        acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
    }
    lo[tid] = j <= A ? acc + bh[j] : 0.0f;
    __syncthreads();
    const int i = r0 + row;
    if (j < A && i < B) {
        float sum = 0.0f;
        for (int a = 1; a <= A; ++a) sum = sum + lo[16 * row + a];
        const float mean = __fdiv_rn(sum, (float)A);
        q[(long long)i * A + j] = (lo[16 * row] + lo[16 * row + 1 + j]) - mean;
    }
}

// ------------------------------------------------------------------------------------ backward (loss gradient)
// jax.grad(compute_loss) (General/QLearning/q_learning_functions.py:23, :31-39) through the CNN, hand-derived like the MLP's
// (dqn_net.hip). Three kinds of kernels:
//   k_cnn_head_bwd                   loss, dL/dQ, dueling backward, gradient at the fc pre-activations, head-leaf partials
//   k_cnn_bwd_data<L>                gradient at layer L's INPUT pre-activations: dZ_L . W_L^T as an implicit GEMM in gather form
//   k_cnn_dw<L>                      dW_L = Patch_L^T . dZ_L (reduction over the output positions), cut into row slices
// Sums run in another order than the restatement's sample-by-sample accumulation: the bar is 1e-5 of the leaf scale against
// the f64 form (f32 mode), 2e-2 in bf16 mode.

template <typename TC, int MT>
__device__ __forceinline__ void mfma_chunk(const TC *ar, const TC *br, int hi, f32x16c &acc) {
    static_assert(MT == 32, "the backward kernels use 32 x 32 tiles");
    if constexpr (sizeof(TC) == 2) {
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8c *>(ar + 16 * ks + 8 * hi),
                                                          *reinterpret_cast<const bf16x8c *>(br + 16 * ks + 8 * hi), acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 a = *reinterpret_cast<const float4 *>(ar + 32 * hi + 4 * q), b = *reinterpret_cast<const float4 *>(br + 32 * hi + 4 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
}
template <typename TR> __device__ __forceinline__ TR zero_piece();
template <> __device__ __forceinline__ bf16x8c zero_piece<bf16x8c>() { bf16x8c z; for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.0f; return z; }
template <> __device__ __forceinline__ f32x8raw zero_piece<f32x8raw>() { return f32x8raw{float4{0, 0, 0, 0}, float4{0, 0, 0, 0}}; }

// Backward-data geometry of layer L (dZ_L -> gradient at a_{L-1}, masked by a_{L-1} > 0 = dZ_{L-1}):
//   L = 3 (fc)     rows b,             K = 512 (n of the fc),            N = 3136, weights as stored ([3136][512])
//   L = 2 (conv3)  rows (b, ih, iw) of the 9 x 9 input, K = 9 taps x 64 oc, N = 64 ic: tap (kh, kw) reads dZ[b][ih-kh][iw-kw] or 0
//   L = 1 (conv2)  stride 2: an input pixel (ih, iw) is reached by the taps kh = (ih & 1) + 2 th, kw = (iw & 1) + 2 tw only, so
//                  the rows are grouped into the four parity classes (each with its own 32 x 256 weight matrix): rows (b, i2, j2)
//                  of a class, K = 4 taps x 64 oc, tap (th, tw) reads dZ[b][i2-th][j2-tw] or 0
template <int L> struct BwdGeo;
template <> struct BwdGeo<3> { static constexpr int K = 512, N = 3136, WM = 2, WN = 2, R = 4, CLASSES = 1, ROWS = 1; };
template <> struct BwdGeo<2> { static constexpr int K = 576, N = 64, WM = 2, WN = 2, R = 3, CLASSES = 1, ROWS = 81; };
template <> struct BwdGeo<1> { static constexpr int K = 256, N = 32, WM = 4, WN = 1, R = 4, CLASSES = 4, ROWS = 100; };

template <typename TC, int L>
__global__ void __launch_bounds__(256)
k_cnn_bwd_data(int B, const TC *__restrict__ dz, const TC *__restrict__ wb, const TC *__restrict__ act, TC *__restrict__ out) {
    typedef BwdGeo<L> G;
    constexpr int MT = 32, WM = G::WM, WN = G::WN, R = G::R, K = G::K;
    constexpr int BM = MT * WM, BN = MT * WN, NCH = K / KC, NIT = NCH / R;
    constexpr int LS = KC + (sizeof(TC) == 2 ? 8 : 4);
    constexpr int APT = BM * 8 / 256, BPT = BN * 8 / 256, NT = G::N / BN;
    static_assert(WM * WN == 4 && K % KC == 0 && NCH % R == 0 && G::N % BN == 0, "tile shape");
    typedef typename RawPiece<TC>::t RP;
    __shared__ __attribute__((aligned(16))) TC lA[2 * BM * LS];
    __shared__ __attribute__((aligned(16))) TC lB[2 * BN * LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    const int M = B * G::ROWS;                                   // rows per class
    const int mtiles = (M + BM - 1) / BM, per_class = mtiles * NT;
    const int cl = blockIdx.x / per_class;
    int t = blockIdx.x - cl * per_class;
    if ((per_class & 7) == 0) t = (t & 7) * (per_class >> 3) + (t >> 3);
    const int m0 = (t / NT) * BM, n0 = (t % NT) * BN, ph = cl >> 1, pw = cl & 1;
    const int pj = tid & 7;
    // a row's (b, y, x): for L = 2 the input pixel, for L = 1 the pixel's half coordinates inside its parity class
    int rb[APT], ry[APT], rx[APT];
#pragma unroll
    for (int u = 0; u < APT; ++u) {
        int mm = m0 + (tid >> 3) + 32 * u;
        mm = mm < M ? mm : M - 1;
        if constexpr (L == 3) { rb[u] = mm; ry[u] = 0; rx[u] = 0; }
        else if constexpr (L == 2) { const unsigned b = (unsigned)mm / 81u, p = (unsigned)mm - b * 81u; rb[u] = b; ry[u] = p / 9u; rx[u] = p - (p / 9u) * 9u; }
        else { const unsigned b = (unsigned)mm / 100u, p = (unsigned)mm - b * 100u; rb[u] = b; ry[u] = p / 10u; rx[u] = p - (p / 10u) * 10u; }
    }
    const TC *bp[BPT];
#pragma unroll
    for (int u = 0; u < BPT; ++u) bp[u] = wb + ((long long)cl * G::N + n0 + (tid >> 3) + 32 * u) * K + 8 * pj;
    RP ra[R][APT], rbw[R][BPT];
    auto request = [&](int slot, int kc) {
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            bool ok = true; long long off;
            if constexpr (L == 3) off = (long long)rb[u] * 512 + kc * KC;
            else if constexpr (L == 2) {
                const int oh = ry[u] - kc / 3, ow = rx[u] - kc % 3;
                ok = oh >= 0 && oh < 7 && ow >= 0 && ow < 7;
                off = ok ? ((long long)(rb[u] * 7 + oh) * 7 + ow) * 64 : 0;
            } else {
                const int oh = ry[u] - kc / 2, ow = rx[u] - kc % 2;
                ok = oh >= 0 && oh < 9 && ow >= 0 && ow < 9;
                off = ok ? ((long long)(rb[u] * 9 + oh) * 9 + ow) * 64 : 0;
            }
            const RP v = load_piece<TC>(dz + off + 8 * pj);      // always loaded (a tap outside the map reads a valid address, then 0)
            ra[slot][u] = ok ? v : zero_piece<RP>();
        }
#pragma unroll
        for (int u = 0; u < BPT; ++u) rbw[slot][u] = load_piece<TC>(bp[u] + kc * KC);
    };
    auto commit = [&](int slot, int buf) {
        TC *a = lA + buf * (BM * LS) + (tid >> 3) * LS, *b = lB + buf * (BN * LS) + (tid >> 3) * LS;
#pragma unroll
        for (int u = 0; u < APT; ++u) commit_piece<MT>(a + 32 * u * LS, pj, ra[slot][u], nullptr);
#pragma unroll
        for (int u = 0; u < BPT; ++u) commit_piece<MT>(b + 32 * u * LS, pj, rbw[slot][u], nullptr);
    };
    f32x16c acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int hi = lane >> 5, c = lane & 31;
#pragma unroll
    for (int r = 0; r < R; ++r) request(r, r);
    // the ReLU gates of this lane's 16 outputs: requested here, behind the first operand loads, consumed in the epilogue (read
    // inside the store loop they were 16 serial round trips per workgroup: 3.4x the time of the forward kernel of the same shape)
    const int n = n0 + MT * wn + c;
    auto out_off = [&](int r) -> long long {
        int mm = m0 + MT * wm + (r & 3) + 8 * (r >> 2) + 4 * hi;
        mm = mm < M ? mm : M - 1;
        if constexpr (L == 3) return (long long)mm * 3136 + n;
        else if constexpr (L == 2) return (long long)mm * 64 + n;
        else {
            const unsigned b = (unsigned)mm / 100u, p = (unsigned)mm - b * 100u, i2 = p / 10u, j2 = p - i2 * 10u;
            return ((long long)(b * 20 + 2 * i2 + ph) * 20 + 2 * j2 + pw) * 32 + n;
        }
    };
    TC gate_raw[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) gate_raw[r] = act[out_off(r)];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int kc = it * R + r, buf = kc & 1;
            commit(r, buf);
            if (it + 1 < NIT) request(r, kc + R);
            __syncthreads();
            mfma_chunk<TC, MT>(lA + buf * (BM * LS) + (MT * wm + c) * LS, lB + buf * (BN * LS) + (MT * wn + c) * LS, hi, acc);
        }
    }
    float o[16];                                                      // values first, stores after (see k_cnn_layer's epilogue)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[r] = (float)gate_raw[r] > 0.0f ? acc[r] : 0.0f; asm volatile("" : "+v"(o[r])); }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mm = m0 + MT * wm + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (mm < M) out[out_off(r)] = (TC)o[r];
    }
}

// dW geometry of layer L: workgroup = (k-tile of KT rows of dW) x (n-tile of NT columns) x (slice of the rows m); four waves as
// WK x WN, each (KT / WK) x (NT / WN) = TK x TN MFMA tiles. The rows of a slice are walked in super-chunks of NSC chunks of MR
// rows: inside a super-chunk the pipeline is straight-line code (compile-time ring slots), between two the queue drains.
// Both operands have the reduction index m as their ROW index in memory, MFMA wants it along the lane's registers:
//   f32 : row-major LDS images [m][k], [m][n]; a lane feeds ONE element per 32x32x2 MFMA: scalar reads, consecutive lanes =
//         consecutive words, conflict-free. MR = 32 (64 for conv1).
//   bf16: TRANSPOSED images [k][m], [n][m], written two bytes at a time -- the pieces are dealt so that the lanes of a wave hold
//         64 consecutive rows m of the same 8 columns, i.e. every 2-byte store instruction covers 128 contiguous bytes -- and
//         read as 16-byte fragments (8 consecutive m of one k) like the forward kernel's. MR = 64. (The first version
//         gathered the 8 rows with ds_read_u16 from row-major images: 24 LDS instructions per two MFMAs.)
template <int L> struct DwGeo;
template <> struct DwGeo<0> { static constexpr int KT = 256, NT = 32, WK = 4, WN = 1, MRF = 64, R = 4, NSC = 8; };
template <> struct DwGeo<1> { static constexpr int KT = 256, NT = 64, WK = 4, WN = 1, MRF = 32, R = 4, NSC = 8; };
template <> struct DwGeo<2> { static constexpr int KT = 192, NT = 64, WK = 2, WN = 2, MRF = 32, R = 4, NSC = 8; };
template <> struct DwGeo<3> { static constexpr int KT = 64, NT = 128, WK = 2, WN = 2, MRF = 32, R = 4, NSC = 8; };
template <typename TC, int L> struct DwRows { static constexpr int MR = sizeof(TC) == 2 ? 64 : DwGeo<L>::MRF; };

// f32: natural-order row images
template <typename TR>
__device__ __forceinline__ void commit_row_piece(float *dst, const TR &raw, const float *) {
    float v[8];
    if constexpr (__is_same(TR, uint2)) {      // conv1's dW multiplies the INTEGER pixels (exact in f32) and divides the sum by 255 in the reduction
        const uint32_t w[2] = {raw.x, raw.y};
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
    } else piece_f32(raw, nullptr, v);
    *reinterpret_cast<float4 *>(dst) = float4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<float4 *>(dst + 4) = float4{v[4], v[5], v[6], v[7]};
}
// bf16: the 8 elements of a piece go to 8 rows of the transposed image, column m
template <typename TR>
__device__ __forceinline__ void commit_col_piece(__bf16 *dst, int ls, const TR &raw) {
    bf16x8c a;
    if constexpr (__is_same(TR, bf16x8c)) a = raw;
    else {
        const uint32_t w[2] = {raw.x, raw.y};
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i * ls] = a[i];
}

template <typename TI, typename TC, int L>
__global__ void __launch_bounds__(256)
k_cnn_dw(int M, int rows_per_slice, const TI *__restrict__ in, const TC *__restrict__ dz, float *__restrict__ slab, float *__restrict__ bslab) {
    typedef CnnGeo<L> G; typedef DwGeo<L> D;
    constexpr bool TR = sizeof(TC) == 2;                              // transposed images
    constexpr int KT = D::KT, NT = D::NT, WK = D::WK, WN = D::WN, MR = DwRows<TC, L>::MR, R = D::R, NSC = D::NSC;
    constexpr int K = G::KH * G::KW * G::IC, N = G::OC, ROWLEN = G::KW * G::IC, KTILES = K / KT, NTILES = N / NT;
    constexpr int TK = KT / WK / 32, TN = NT / WN / 32;
    constexpr int LSA = TR ? MR + 8 : KT + 4, LSB = TR ? MR + 8 : NT + 4;      // row length of the images
    constexpr int APR = KT / 8, BPR = NT / 8;                         // pieces per row m
    constexpr int APT = MR * APR / 256, BPT = MR * BPR / 256;
    constexpr int LBUF = (TR && L != 0) ? 2 : 1;       // conv1: one 42 KB image set, three workgroups per CU
    static_assert(WK * WN == 4 && K % KT == 0 && N % NT == 0 && (KT % ROWLEN == 0 || ROWLEN % KT == 0), "tile shape");
    static_assert(MR * APR % 256 == 0 && MR * BPR % 256 == 0 && NSC % R == 0 && TK >= 1 && TN >= 1 && 256 % MR == 0, "piece split");
    typedef typename RawPiece<TI>::t RA;
    typedef typename RawPiece<TC>::t RB;
    __shared__ __attribute__((aligned(16))) TC lA[LBUF * (TR ? KT * LSA : MR * LSA)];
    __shared__ __attribute__((aligned(16))) TC lB[LBUF * (TR ? NT * LSB : MR * LSB)];
    constexpr int ASZ = TR ? KT * LSA : MR * LSA, BSZ = TR ? NT * LSB : MR * LSB;
    const float *lut = nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wk = wave / WN, wn = wave % WN, hi = lane >> 5, c = lane & 31;
    const int tile = blockIdx.x % (KTILES * NTILES), slice = blockIdx.x / (KTILES * NTILES);
    const int kt0 = (tile / NTILES) * KT, n0 = (tile % NTILES) * NT;
    const int m_begin = slice * rows_per_slice, m_end = min(M, m_begin + rows_per_slice);
    // this thread's pieces: image row m and 8-column group. f32: consecutive threads = consecutive column groups of a row (one
    // row = one contiguous run); bf16: consecutive threads = consecutive rows of the same column group (see above)
    int arow[APT], akp[APT], akoff[APT], brow[BPT], bkp[BPT];
#pragma unroll
    for (int u = 0; u < APT; ++u) {
        const int q = tid + 256 * u;
        arow[u] = TR ? q % MR : q / APR; akp[u] = TR ? q / MR : q % APR;
        const int kp = kt0 + 8 * akp[u], kh = kp / ROWLEN;
        akoff[u] = kh * (G::IW * G::IC) + (kp - kh * ROWLEN);
    }
#pragma unroll
    for (int u = 0; u < BPT; ++u) { const int q = tid + 256 * u; brow[u] = TR ? q % MR : q / BPR; bkp[u] = TR ? q / MR : q % BPR; }
    RA ra[R][APT]; RB rb[R][BPT];
    auto request = [&](int slot, int mc) {                             // mc: first row of the chunk
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            int mm = mc + arow[u]; mm = mm < M ? mm : M - 1;
            const unsigned ow = (unsigned)mm % G::OW, t2 = (unsigned)mm / G::OW, oh = t2 % G::OH, b = t2 / G::OH;
            ra[slot][u] = load_piece<TI>(in + ((long long)(b * G::IH + oh * G::S) * G::IW + ow * G::S) * G::IC + akoff[u]);
        }
#pragma unroll
        for (int u = 0; u < BPT; ++u) {
            const int mm = mc + brow[u];
            const RB v = load_piece<TC>(dz + (long long)(mm < M ? mm : M - 1) * N + n0 + 8 * bkp[u]);
            rb[slot][u] = mm < m_end ? v : zero_piece<RB>();           // rows past the slice contribute nothing
        }
    };
    auto commit = [&](int slot, int buf) {
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            if constexpr (TR) commit_col_piece(lA + buf * ASZ + 8 * akp[u] * LSA + arow[u], LSA, ra[slot][u]);
            else commit_row_piece(lA + buf * ASZ + arow[u] * LSA + 8 * akp[u], ra[slot][u], lut);
        }
#pragma unroll
        for (int u = 0; u < BPT; ++u) {
            if constexpr (TR) commit_col_piece(lB + buf * BSZ + 8 * bkp[u] * LSB + brow[u], LSB, rb[slot][u]);
            else commit_row_piece(lB + buf * BSZ + brow[u] * LSB + 8 * bkp[u], rb[slot][u], nullptr);
        }
    };
    f32x16c acc[TK][TN];
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bsum = 0.0f;
    auto multiply = [&](int buf) {
        if (tile / NTILES == 0 && tid < NT) {                          // bias gradient: column sums of dZ (k-tile 0 only)
            if constexpr (TR) {
                const TC *row = lB + buf * BSZ + tid * LSB;
#pragma unroll
                for (int r = 0; r < MR; r += 8) {
                    const bf16x8c v = *reinterpret_cast<const bf16x8c *>(row + r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum = bsum + (float)v[e];
                }
            } else {
                const TC *col = lB + buf * BSZ + tid;
#pragma unroll 8
                for (int r = 0; r < MR; ++r) bsum = bsum + (float)col[r * LSB];
            }
        }
        if constexpr (TR) {
            const TC *a = lA + buf * ASZ + (wk * TK * 32 + c) * LSA + 8 * hi, *b = lB + buf * BSZ + (wn * TN * 32 + c) * LSB + 8 * hi;
#pragma unroll
            for (int s = 0; s < MR / 16; ++s) {
                bf16x8c fa[TK], fb[TN];
#pragma unroll
                for (int i = 0; i < TK; ++i) fa[i] = *reinterpret_cast<const bf16x8c *>(a + 32 * i * LSA + 16 * s);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8c *>(b + 32 * j * LSB + 16 * s);
#pragma unroll
                for (int i = 0; i < TK; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        } else {
            const TC *a = lA + buf * ASZ + (wk * TK * 32 + c), *b = lB + buf * BSZ + (wn * TN * 32 + c);
#pragma unroll 4
            for (int s = 0; s < MR / 2; ++s) {
                float fa[TK], fb[TN];
#pragma unroll
                for (int i = 0; i < TK; ++i) fa[i] = a[(2 * s + hi) * LSA + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = b[(2 * s + hi) * LSB + 32 * j];
#pragma unroll
                for (int i = 0; i < TK; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    for (int ms = m_begin; ms < m_end; ms += NSC * MR) {               // one super-chunk: straight-line pipeline
#pragma unroll
        for (int r = 0; r < R; ++r) request(r, ms + r * MR);
#pragma unroll
        for (int j = 0; j < NSC; ++j) {
            const int buf = LBUF == 2 ? (j & 1) : 0;
            commit(j % R, buf);
            if (j + R < NSC) request(j % R, ms + (j + R) * MR);
            __syncthreads();
            multiply(buf);
            if constexpr (LBUF == 1) __syncthreads();
        }
        if constexpr (LBUF == 2 && (NSC & 1)) __syncthreads();
    }
    float *o = slab + (long long)slice * K * N;
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kt0 + (wk * TK + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, n = n0 + (wn * TN + j) * 32 + c;
                o[(long long)k * N + n] = acc[i][j][r];
            }
    if (tile / NTILES == 0 && tid < NT) bslab[(long long)slice * N + n0 + tid] = bsum;
}

// Loss, dL/dQ and the dueling backward (as k_bwd_rows in dqn_net.hip: g = w clip(q - target, -1, 1) / B, dv = sum g,
// dadv = g - mean g), the gradient at the fc pre-activations dz4[i][k] = a3 > 0 ? wv[k] dv + sum_j wa[k][j] dadv_j : 0, and this
// workgroup's share of the head leaves: 16 rows per workgroup = one slice of the head slabs
//   hslab[blk][j][k] = sum_{its rows} a3[i][k] gd[i][j]   (j = 0: val, 1..A: adv),   hbslab[blk][j] = sum gd[i][j]
// (k_cnn_reduce adds the slices in order); one loss partial per workgroup.
struct CnnTdArgs { const float *nq, *nt; const int32_t *a; const float *r, *d; float gamma; float *td_abs; };     // nq == nullptr: targets are given
template <typename TC>
__global__ void __launch_bounds__(256)
k_cnn_head_bwd(const float *__restrict__ q, const float *__restrict__ targets, const float *__restrict__ isw, const TC *__restrict__ feat,
               const float *__restrict__ wht, int A, int B, TC *__restrict__ dz4, float *__restrict__ hslab, float *__restrict__ hbslab,
               float *__restrict__ loss_part, CnnTdArgs td) {
    __shared__ float lg[16 * 16], lrow[16], ltg[16 * 16];
    const int tid = threadIdx.x, r0 = blockIdx.x * 16;
    if (td.nq) {                                                   // the TD rule (q_learning_functions.py:55-60, td_row) instead of given targets
        if (tid < 16 && r0 + tid < B) {
            const int i = r0 + tid;
            float qr[16], nqr[16], ntr[16], tr[16];
            for (int k = 0; k < A; ++k) { qr[k] = q[(long long)i * A + k]; nqr[k] = td.nq[(long long)i * A + k]; ntr[k] = td.nt[(long long)i * A + k]; }
            const float delta = td_row(qr, nqr, ntr, td.a[i], td.r[i], td.d[i], td.gamma, A, tr);
            for (int k = 0; k < A; ++k) ltg[16 * tid + k] = tr[k];
            if (td.td_abs) td.td_abs[i] = fabsf(delta);
        }
        __syncthreads();
    }
    {                                                              // thread (row, j): one Q entry
        const int row = tid >> 4, j = tid & 15, i = r0 + row;
        const bool on = i < B && j < A;
        float e = 0.0f, w = 1.0f;
        if (on) { e = q[(long long)i * A + j] - (td.nq ? ltg[16 * row + j] : targets[(long long)i * A + j]); if (isw) w = isw[i]; }
        const float cpd = e > 1.0f ? 1.0f : (e < -1.0f ? -1.0f : e);
        const float g = on ? (w * cpd) * __fdiv_rn(1.0f, (float)B) : 0.0f;
        float hub = on ? huber(e) : 0.0f, gsum = g;
        // sums over the 16 lanes of a row in ascending j (the order of the restatement): a serial walk through LDS
        lg[tid] = g; __syncthreads();
        if (j == 0) {
            float s = 0.0f;
            for (int a = 0; a < A; ++a) s = s + lg[16 * row + a];
            gsum = s;
        }
        __syncthreads();
        lg[tid] = hub; __syncthreads();
        if (j == 0) {
            float s = 0.0f;
            for (int a = 0; a < A; ++a) s = s + lg[16 * row + a];
            lrow[row] = (i < B && isw) ? w * s : s;
        }
        __syncthreads();
        gsum = __shfl(gsum, (tid & 48), 64);                       // lane j = 0 of this row (16 lanes per row, 4 rows per wave)
        const float gmean = __fdiv_rn(gsum, (float)A);
        if (j < 15) lg[16 * row + 1 + j] = on ? g - gmean : 0.0f;  // gd[row][1 + j] = dadv_j, slots above A stay 0 (all Huber sums are read)
        if (j == 0) lg[16 * row] = i < B ? gsum : 0.0f;            // gd[row][0] = dv
        __syncthreads();
    }
    if (tid == 0) { float s = 0.0f; for (int r = 0; r < 16; ++r) s = s + lrow[r]; loss_part[blockIdx.x] = s; }
    if (tid < 16) {
        float s = 0.0f;
        for (int r = 0; r < 16; ++r) s = s + lg[16 * r + tid];
        hbslab[(long long)blockIdx.x * 16 + tid] = s;
    }
    // thread: 2 columns k of all 16 rows
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = tid + 256 * kk;
        float w[16], acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { w[j] = wht[j * CNN_F + k]; acc[j] = 0.0f; }          // rows above A are 0 in the shadow
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int i = r0 + r; x[r] = (float)feat[(long long)(i < B ? i : B - 1) * CNN_F + k]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = r0 + r;
            float t = w[0] * lg[16 * r];
#pragma unroll
            for (int j = 1; j < 16; ++j) t = fmaf(w[j], lg[16 * r + j], t);
            if (i < B) dz4[(long long)i * CNN_F + k] = x[r] > 0.0f ? (TC)t : (TC)0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fmaf(x[r], lg[16 * r + j], acc[j]);      // rows past B: gd = 0
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) hslab[((long long)blockIdx.x * 16 + j) * CNN_F + k] = acc[j];
    }
}

// grad leaf = sum over the slices of its slab, in slice order (deterministic); loss = sum of the head partials / B.
// Where a segment's size and offsets allow, four elements are summed by 8 lanes (each an eighth of the slices, partials added in
// lane order: 200 dependent slices of conv1's slab were 17 of the kernel's 21 us); adv > 0 marks the adv-head leaf, whose slab is
// output-major ([1 + j][k]) while the leaf is [k][j]. units = threads of the segment (vector segments: 8 per float4).
struct CnnSeg { const float *slab; long long n, dst, stride, units; int S, adv; float div; int vec; };      // units: multiples of 8 (whole lane groups)
struct CnnSegs { CnnSeg s[12]; int count; };
__global__ void __launch_bounds__(256)
k_cnn_reduce(CnnSegs segs, long long total_units, float *__restrict__ grad, const float *__restrict__ loss_part, int loss_parts, int B, float *__restrict__ loss) {
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t == 0 && loss) { float s = 0.0f; for (int i = 0; i < loss_parts; ++i) s = s + loss_part[i]; *loss = __fdiv_rn(s, (float)B); }
    if (t >= total_units) return;
    int g = 0;
    for (int i = 0; i + 1 < segs.count; ++i) if (g == i && t >= segs.s[i].units) { t -= segs.s[i].units; ++g; }
    const CnnSeg sg = segs.s[g];
    if (sg.vec) {                // four elements per 8 lanes: lane g sums its eighth of the slices (in order), the eight partials are added in lane order
        const int g = (int)(t & 7);
        const long long unit = t >> 3;
        const float4 *p = reinterpret_cast<const float4 *>(sg.slab + 4 * unit);
        const long long st4 = sg.stride / 4;
        const int per = (sg.S + 7) / 8, s0 = g * per, s1 = min(sg.S, s0 + per);
        float4 acc = float4{0.0f, 0.0f, 0.0f, 0.0f};
        int s = s0;
        for (; s + 8 <= s1; s += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long long)(s + u) * st4];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x = acc.x + v[u].x; acc.y = acc.y + v[u].y; acc.z = acc.z + v[u].z; acc.w = acc.w + v[u].w; }
        }
        for (; s < s1; ++s) { const float4 v = p[(long long)s * st4]; acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w; }
        float4 tot = float4{0.0f, 0.0f, 0.0f, 0.0f};
        const int base = (threadIdx.x & 63) & ~7;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            tot.x = tot.x + __shfl(acc.x, base + j, 64); tot.y = tot.y + __shfl(acc.y, base + j, 64);
            tot.z = tot.z + __shfl(acc.z, base + j, 64); tot.w = tot.w + __shfl(acc.w, base + j, 64);
        }
        if (g == 0) {
            if (sg.div != 1.0f) { tot.x = __fdiv_rn(tot.x, sg.div); tot.y = __fdiv_rn(tot.y, sg.div); tot.z = __fdiv_rn(tot.z, sg.div); tot.w = __fdiv_rn(tot.w, sg.div); }
            *reinterpret_cast<float4 *>(grad + sg.dst + 4 * unit) = tot;
        }
        return;
    }
    if (t >= sg.n) return;       // padding of a scalar segment
    const long long src = sg.adv ? (long long)(1 + t % sg.adv) * CNN_F + t / sg.adv : t;
    const float *p = sg.slab + src;
    float acc = 0.0f;
    int s = 0;
    for (; s + 8 <= sg.S; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(s + u) * sg.stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = acc + v[u];
    }
    for (; s < sg.S; ++s) acc = acc + p[(long long)s * sg.stride];
    grad[sg.dst + t] = sg.div != 1.0f ? __fdiv_rn(acc, sg.div) : acc;
}

// every compute-type shadow of flat parameter element i (value v): forward [N][K] (conv1's scaled by 1/255 in bf16 mode),
// backward-data (BwdGeo), output-major heads
struct CnnOffs { long long o_w[4], o_b[4], o_wv, o_bv, o_wa, o_ba, P; int A; };
template <typename TC>
struct CnnShadows { TC *wt[4]; TC *wb[4]; float *wh, *bh; TC *wp, *wpb; };
// wp (bf16 mode): the three convolutions' weights once more, packed by MFMA fragment for k_cnn_trunk16 -- element (n, k) of layer l at
// CNN_WP_OFF[l] + ((((n >> 5) * (K / 16) + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (n & 31)) * 8 + (k & 7): the 64 lanes' 16-byte fragments of
// (column tile, k-step) are 1 KB contiguous (as [N][K] they are 64 pieces of 64 different cache lines: the trunk kernel's 84
// fragment loads per lane took longer than its three convolutions)
constexpr int CNN_WP_OFF1 = 32 * 256, CNN_WP_OFF2 = CNN_WP_OFF1 + 64 * 512, CNN_WP_ELEMS = CNN_WP_OFF2 + 64 * 576;
// wpb: the backward-data matrices of conv3 ([64 ic][576]) and conv2 ([4 classes][32 ic][256]) packed the same way for k_cnn_btrunk16
constexpr int CNN_WPB_OFF1 = 64 * 576, CNN_WPB_ELEMS = CNN_WPB_OFF1 + 4 * 32 * 256;
template <typename TC>
__device__ __forceinline__ void scatter_shadows(const CnnOffs &o, const CnnShadows<TC> &sh, long long i, float v) {
    if (i >= o.o_wv) {
        if (i < o.o_bv) sh.wh[i - o.o_wv] = v;
        else if (i < o.o_wa) sh.bh[0] = v;
        else if (i < o.o_ba) { const long long e = i - o.o_wa; const int k = (int)(e / o.A), j = (int)(e - (long long)k * o.A); sh.wh[(1 + j) * CNN_F + k] = v; }
        else sh.bh[1 + (i - o.o_ba)] = v;
        return;
    }
    int l = 0;
#pragma unroll
    for (int u = 1; u < 4; ++u) if (i >= o.o_w[u]) l = u;
    if (i >= o.o_b[l]) return;                                         // conv / fc biases are read from the flat vector
    const int N = l == 0 ? 32 : (l == 3 ? 512 : 64), K = l == 0 ? 256 : (l == 1 ? 512 : (l == 2 ? 576 : 3136));
    const long long e = i - o.o_w[l];
    const int k = (int)(e / N), n = (int)(e - (long long)k * N);
    const float vf = (l == 0 && sizeof(TC) == 2) ? __fdiv_rn(v, 255.0f) : v;
    sh.wt[l][(long long)n * K + k] = (TC)vf;
    if (sizeof(TC) == 2 && l < 3)
        sh.wp[(l == 0 ? 0 : (l == 1 ? CNN_WP_OFF1 : CNN_WP_OFF2)) + ((((n >> 5) * (K >> 4) + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (n & 31)) << 3) + (k & 7)] = (TC)vf;
    if (l == 3) sh.wb[3][e] = (TC)v;
    else if (l == 2) {
        const int tap = k >> 6, ic = k & 63, kk = tap * 64 + n;       // row ic, column kk of conv3^T
        sh.wb[2][(long long)ic * 576 + kk] = (TC)v;
        if (sizeof(TC) == 2) sh.wpb[((((ic >> 5) * 36 + (kk >> 4)) * 64 + ((kk >> 3) & 1) * 32 + (ic & 31)) << 3) + (kk & 7)] = (TC)v;
    }
    else if (l == 1) {
        const int tap = k >> 5, ic = k & 31, kh = tap >> 2, kw = tap & 3, cl = (kh & 1) * 2 + (kw & 1), tt = (kh >> 1) * 2 + (kw >> 1);
        sh.wb[1][((long long)cl * 32 + ic) * 256 + tt * 64 + n] = (TC)v;
        if (sizeof(TC) == 2) { const int kk = tt * 64 + n; sh.wpb[CNN_WPB_OFF1 + (((cl * 16 + (kk >> 4)) * 64 + ((kk >> 3) & 1) * 32 + ic) << 3) + (kk & 7)] = (TC)v; }
    }
}
template <typename TC>
__global__ void __launch_bounds__(256)
k_cnn_scatter(CnnOffs o, CnnShadows<TC> sh, const float *__restrict__ P) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < o.P && !(i >= o.o_w[3] && i < o.o_b[3])) scatter_shadows<TC>(o, sh, i, P[i]);      // fc weights: k_cnn_fc_leaf
}

// optax adam / adamw element (adam_elem, dqn_net_common.h) + shadow refresh; the step counters live in CnnOptState and are
// advanced by k_cnn_opt_prep BEFORE the step's Adam kernel (dqn_cnn_update: on the side stream, beside the backward)
struct CnnOptState { double b1pow, b2pow; int count; float lr; };
__global__ void k_cnn_opt_prep(CnnOptState *st, float b1, float b2) {
    st->b1pow *= (double)b1; st->b2pow *= (double)b2; st->count += 1;
}
template <typename TC>
__global__ void __launch_bounds__(256)
k_cnn_adam(CnnOffs o, CnnShadows<TC> sh, const CnnOptState *__restrict__ st, float *P, const float *__restrict__ grad, float *mu, float *nu,
           int adamw, float b1, float b2, float eps, float wd, float grad_scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= o.P || (i >= o.o_w[3] && i < o.o_b[3])) return;          // fc weights: k_cnn_fc_leaf
    const AdamCoef c{(float)(1.0 - st->b1pow), (float)(1.0 - st->b2pow), 1.0f - b1, 1.0f - b2, -st->lr};
    const float gi = grad[i] * grad_scale;
    const float mm = (b1 * mu[i]) + (c.omb1 * gi);
    const float vv = (b2 * nu[i]) + (c.omb2 * (gi * gi));
    mu[i] = mm; nu[i] = vv;
    const float mhat = __fdiv_rn(mm, c.c1), vhat = __fdiv_rn(vv, c.c2);
    float u = __fdiv_rn(mhat, sqrtf(vhat) + eps);
    float p = P[i];
    if (adamw) u = u + (wd * p);
    p = p + (c.neglr * u);
    P[i] = p;
    scatter_shadows<TC>(o, sh, i, p);
}

// The fc weight leaf [3136][512] (95 % of the parameters): its forward shadow is the transpose, and written element by element
// (k_cnn_scatter / k_cnn_adam's first form) that is 1.6 M scattered 2-byte stores. Here a workgroup takes a 64 x 64 tile: Adam
// (ADAM) or a plain copy on coalesced rows, the new values through an LDS tile, the transposed shadow as 64-element runs.
template <typename TC, bool ADAM>
__global__ void __launch_bounds__(256)
k_cnn_fc_leaf(CnnOffs o, TC *__restrict__ wt, TC *__restrict__ wb, const CnnOptState *__restrict__ st, float *P, const float *__restrict__ grad, float *mu, float *nu,
              int adamw, float b1, float b2, float eps, float wd, float grad_scale) {
    constexpr int K = 3136, N = 512;
    __shared__ float tile[64][65];
    const int k0 = (blockIdx.x / (N / 64)) * 64, n0 = (blockIdx.x % (N / 64)) * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    AdamCoef c{};
    if constexpr (ADAM) c = AdamCoef{(float)(1.0 - st->b1pow), (float)(1.0 - st->b2pow), 1.0f - b1, 1.0f - b2, -st->lr};
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const int kk = ty + 4 * u;
        const long long e = (long long)(k0 + kk) * N + n0 + tx, i = o.o_w[3] + e;
        float p = P[i];
        if constexpr (ADAM) {
            const float gi = grad[i] * grad_scale;
            const float mm = (b1 * mu[i]) + (c.omb1 * gi);
            const float vv = (b2 * nu[i]) + (c.omb2 * (gi * gi));
            mu[i] = mm; nu[i] = vv;
            const float mhat = __fdiv_rn(mm, c.c1), vhat = __fdiv_rn(vv, c.c2);
            float upd = __fdiv_rn(mhat, sqrtf(vhat) + eps);
            if (adamw) upd = upd + (wd * p);
            p = p + (c.neglr * upd);
            P[i] = p;
        }
        wb[e] = (TC)p;
        tile[kk][tx] = p;
    }
    __syncthreads();
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const int nn = ty + 4 * u;
        wt[(long long)(n0 + nn) * K + k0 + tx] = (TC)tile[tx][nn];
    }
}

// ------------------------------------------------------------------------------------ frame replay ring (configs[4] loop)
// ReplayBuffer (General/Base/replay_buffer.py:20-85) for u8 frame stacks: rows of 84*84*4 = 28 224 bytes for s and for s'
// (the reference stores both, :28-31), actions i32, rewards f32, dones f32. add = contiguous copies at the ring head (:58-65);
// the sampled indices come from a PER tree kept by a dqn_handle of the same capacity (host mirror: CnnVectorAgent).
constexpr int CNN_FRAME_BYTES = 84 * 84 * 4;
// one workgroup per (sample, s | s'): 1764 16-byte pieces of a row; the scalars ride along in the first workgroups.
// n_step > 1 (SURVEY 8(f) rank 3 for the frame ring): rows are stored one env step each, step-major (row = step * n_envs + env),
// and the n-step transition that STARTS at the sampled row is put together here from the n rows row + k n_envs: its action, R =
// r_0 + gamma (r_1 + gamma (...)) cut after the first done (Horner form, the arithmetic of nstep_row in dqn_actor.hip), that done
// flag, s of the first and s' of the last row. The caller samples only rows whose n - 1 successors are in the ring.
__global__ void __launch_bounds__(256)
k_cnn_gather(const uint8_t *__restrict__ ring_s, const uint8_t *__restrict__ ring_s2, const int32_t *__restrict__ ring_a, const float *__restrict__ ring_r,
             const float *__restrict__ ring_d, const int32_t *__restrict__ idx, int B, long long cap, int n_step, int n_envs, float gamma,
             uint8_t *__restrict__ s, uint8_t *__restrict__ s2, int32_t *__restrict__ a, float *__restrict__ r, float *__restrict__ d) {
    const int i = blockIdx.x, part = blockIdx.y;
    long long row = idx[i];
    row = row < 0 ? 0 : (row >= cap ? cap - 1 : row);
    const long long row_last = (row + (long long)(n_step - 1) * n_envs) % cap;
    const uint4 *src = reinterpret_cast<const uint4 *>(part ? ring_s2 + row_last * CNN_FRAME_BYTES : ring_s + row * CNN_FRAME_BYTES);
    uint4 *dst = reinterpret_cast<uint4 *>((part ? s2 : s) + (long long)i * CNN_FRAME_BYTES);
    uint4 v[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) { const int p = threadIdx.x + 256 * u; if (p < CNN_FRAME_BYTES / 16) v[u] = src[p]; }
#pragma unroll
    for (int u = 0; u < 7; ++u) { const int p = threadIdx.x + 256 * u; if (p < CNN_FRAME_BYTES / 16) dst[p] = v[u]; }
    if (part == 0 && threadIdx.x == 0) {
        a[i] = ring_a[row];
        float rr[8], dd[8];
        for (int k = 0; k < n_step; ++k) { const long long q = (row + (long long)k * n_envs) % cap; rr[k] = ring_r[q]; dd[k] = ring_d[q]; }
        int last = n_step - 1;
        for (int k = n_step - 2; k >= 0; --k) if (dd[k] != 0.0f) last = k;      // first done in the window
        float acc = rr[last];
        for (int k = last - 1; k >= 0; --k) acc = rr[k] + gamma * acc;
        r[i] = acc; d[i] = last == n_step - 1 ? dd[n_step - 1] : 1.0f;
    }
}

// ------------------------------------------------------------------------------------ C ABI
#define CNN_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return dqn_set_error(DQN_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); } while (0)
#define CNN_REQ(cond, msg) do { if (!(cond)) return dqn_set_error(DQN_ERR_INVALID, msg); } while (0)

struct CnnLayer { int K, N; long long o_w, o_b; };     // offsets into the flat parameter vector
struct dqn_cnn_handle {
    int A = 0, max_batch = 0, num_cus = 256; bool bf16 = false;
    CnnLayer L[4]; long long o_wv = 0, o_bv = 0, o_wa = 0, o_ba = 0, P = 0;
    void *arena = nullptr;
    float *params[2] = {nullptr, nullptr};             // online, target (flat f32, HWIO leaf order)
    void *wt[2][4] = {{nullptr}};                      // forward shadows [N][K] of the four GEMM layers
    void *wb[2][4] = {{nullptr}};                      // backward-data shadows (BwdGeo; [1..3])
    void *wp[2] = {nullptr, nullptr};                  // bf16 mode: conv weights packed by MFMA fragment (k_cnn_trunk16)
    void *wpb[2] = {nullptr, nullptr};                 // ... and the backward-data matrices of conv3 / conv2 (k_cnn_btrunk16)
    float *wh[2] = {nullptr, nullptr}, *bh[2] = {nullptr, nullptr};   // heads, output-major [16][512] (0 = val, 1.. = adv), biases [1 + A]
    void *act[4] = {nullptr};                          // layer outputs (kept for the backward)
    void *act_t[4] = {nullptr};                        // the target pass of dqn_cnn_update (runs beside the online pass)
    void *dz[4] = {nullptr};                           // gradients at the layers' pre-activations (same shapes)
    float *q[3] = {nullptr, nullptr, nullptr};         // Q of the three passes of compute_q_targets
    float *scratch = nullptr;
    float *grad = nullptr, *mu = nullptr, *nu = nullptr, *loss_part = nullptr, *loss = nullptr, *targets = nullptr;
    float *slab[4] = {nullptr}, *bslab[4] = {nullptr}, *hslab = nullptr, *hbslab = nullptr; int smax[4] = {1, 1, 1, 1};
    CnnOptState *opt = nullptr;
    // frame replay ring (dqn_cnn_replay_init)
    void *ring_arena = nullptr; long long ring_cap = 0, ring_counter = 0;
    uint8_t *ring_s = nullptr, *ring_s2 = nullptr, *stage_s = nullptr, *stage_s2 = nullptr; int32_t *ring_a = nullptr, *stage_a = nullptr;
    float *ring_r = nullptr, *ring_d = nullptr, *stage_r = nullptr, *stage_d = nullptr;
    hipStream_t side = nullptr; hipEvent_t ev_dz[4] = {nullptr}, ev_side = nullptr, ev_fork = nullptr, ev_tgt = nullptr;    // the dW kernels of layers 1..3 run beside the backward-data chain
    int adamw = 1; float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, wd = 1e-4f;
    int flags = 0; hipStream_t side_kept = nullptr;    // dqn_cnn_set_flags (diagnostics)
    hipStream_t side2 = nullptr; hipEvent_t ev_side2 = nullptr;        // r03: conv3's / conv2's dW on a second side stream (beside the fc leaf's step on the first)
    int env_n = 0; uint64_t env_seed = 0; long long env_steps = 0; const uint8_t *env_cur = nullptr;   // dqn_cnn_env_*_synth
    void *comm = nullptr; int rank = 0, world = 1;     // dqn_cnn_comm_init: per-GPU learners, one gradient all-reduce per update
    hipStream_t comm_st = nullptr; hipEvent_t ev_fc = nullptr, ev_fc_done = nullptr;   // the fc leaf's all-reduce on a stream of its own (data-parallel update)
};

struct LayerShape { int K, OC, positions; };            // host view of CnnGeo<l>: K = KH*KW*IC, output positions per frame stack
template <int L> static LayerShape shape_of() { typedef CnnGeo<L> G; return LayerShape{G::KH * G::KW * G::IC, G::OC, G::OH * G::OW}; }
static LayerShape cnn_shape(int layer) {
    switch (layer) { case 0: return shape_of<0>(); case 1: return shape_of<1>(); case 2: return shape_of<2>(); default: return shape_of<3>(); }
}
struct DwShape { int tiles, unit; };                    // (k-tiles x n-tiles) of DwGeo<l>; rows of one super-chunk
template <int L> static DwShape dw_shape_of(bool bf16) { typedef CnnGeo<L> G; typedef DwGeo<L> D; return DwShape{(G::KH * G::KW * G::IC / D::KT) * (G::OC / D::NT), D::NSC * (bf16 ? DwRows<__bf16, L>::MR : DwRows<float, L>::MR)}; }
static DwShape dw_shape(int layer, bool bf16) {
    switch (layer) { case 0: return dw_shape_of<0>(bf16); case 1: return dw_shape_of<1>(bf16); case 2: return dw_shape_of<2>(bf16); default: return dw_shape_of<3>(bf16); }
}
// rows of layer l's dW are cut into at most smax[l] slices of whole super-chunks
static int dw_rows_per_slice(const dqn_cnn_handle *h, int l, long long M) {
    const DwShape d = dw_shape(l, h->bf16);
    const long long per = (M + (long long)h->smax[l] * d.unit - 1) / ((long long)h->smax[l] * d.unit);
    return (int)(per < 1 ? 1 : per) * d.unit;
}

static int dw_slices(const dqn_cnn_handle *h, int l, int B) {
    const long long M = (long long)B * cnn_shape(l).positions;
    const int rows = dw_rows_per_slice(h, l, M);
    return (int)((M + rows - 1) / rows);
}

static CnnOffs cnn_offs(const dqn_cnn_handle *h) {
    CnnOffs o{};
    for (int l = 0; l < 4; ++l) { o.o_w[l] = h->L[l].o_w; o.o_b[l] = h->L[l].o_b; }
    o.o_wv = h->o_wv; o.o_bv = h->o_bv; o.o_wa = h->o_wa; o.o_ba = h->o_ba; o.P = h->P; o.A = h->A;
    return o;
}
template <typename TC> static CnnShadows<TC> cnn_shadows(const dqn_cnn_handle *h, int which) {
    CnnShadows<TC> s{};
    for (int l = 0; l < 4; ++l) { s.wt[l] = (TC *)h->wt[which][l]; s.wb[l] = (TC *)h->wb[which][l]; }
    s.wh = h->wh[which]; s.bh = h->bh[which]; s.wp = (TC *)h->wp[which]; s.wpb = (TC *)h->wpb[which];
    return s;
}

extern "C" int dqn_cnn_create(int32_t num_actions, int32_t max_batch, int32_t precision, dqn_cnn_handle **out) {
    CNN_REQ(out && num_actions >= 1 && num_actions <= 15 && max_batch >= 1 && max_batch <= (1 << 16), "dqn_cnn_create: bad argument");
    CNN_REQ(precision == DQN_PREC_F32 || precision == DQN_PREC_BF16, "unknown precision");
    dqn_cnn_handle *h = new (std::nothrow) dqn_cnn_handle();
    if (!h) return dqn_set_error(DQN_ERR_NOMEM, "host allocation failed");
    h->A = num_actions; h->max_batch = max_batch; h->bf16 = precision == DQN_PREC_BF16;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) h->num_cus = cus;
    long long p = 0;
    for (int l = 0; l < 4; ++l) {
        const LayerShape g = cnn_shape(l);
        h->L[l] = CnnLayer{g.K, g.OC, p, p + (long long)g.K * g.OC};
        p += (long long)g.K * g.OC + g.OC;
        const int t = dw_shape(l, h->bf16).tiles;
        h->smax[l] = t >= h->num_cus ? 1 : (t > h->num_cus / 2 ? 2 : (l == 0 ? 2 : 1) * h->num_cus / t);
    }
    h->o_wv = p; p += 512; h->o_bv = p; p += 1; h->o_wa = p; p += 512ll * h->A; h->o_ba = p; p += h->A; h->P = p;
    const size_t esz = h->bf16 ? 2 : 4;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t sz_params = al(h->P * 4), sz_wh = al(16 * CNN_F * 4), sz_bh = al(16 * 4), sz_q = al((size_t)max_batch * 16 * 4);
    size_t sz_wt[4], sz_act[4], sz_slab[4], sz_bslab[4];
    for (int l = 0; l < 4; ++l) {
        sz_wt[l] = al((size_t)h->L[l].K * h->L[l].N * esz);
        sz_act[l] = al((size_t)max_batch * cnn_shape(l).positions * h->L[l].N * esz);          // activations hold 2 x this (the paired pass), gradients 1 x
        sz_slab[l] = al((size_t)h->smax[l] * h->L[l].K * h->L[l].N * 4); sz_bslab[l] = al((size_t)h->smax[l] * h->L[l].N * 4);
    }
    const size_t hblocks = ((size_t)max_batch + 15) / 16, sz_hslab = al(hblocks * 16 * CNN_F * 4), sz_hbslab = al(hblocks * 16 * 4);
    const size_t sz_wp = al((size_t)CNN_WP_ELEMS * 2) + al((size_t)CNN_WPB_ELEMS * 2);
    size_t total = 5 * sz_params + 4 * (sz_wt[0] + sz_wt[1] + sz_wt[2] + sz_wt[3]) + 2 * (sz_wh + sz_bh + sz_wp) + 6 * sz_q + 4 * al((size_t)max_batch * 4) + 1024 + sz_hslab + sz_hbslab;
    for (int l = 0; l < 4; ++l) total += 4 * sz_act[l] + sz_slab[l] + sz_bslab[l];
    hipError_t e = hipMalloc(&h->arena, total);
    if (e != hipSuccess) { delete h; return dqn_set_error(DQN_ERR_NOMEM, (std::string("hipMalloc: ") + hipGetErrorString(e)).c_str()); }
    char *c = (char *)h->arena;
    auto take = [&](size_t n) { char *r = c; c += n; return r; };
    for (int w = 0; w < 2; ++w) {
        h->params[w] = (float *)take(sz_params);
        for (int l = 0; l < 4; ++l) { h->wt[w][l] = take(sz_wt[l]); h->wb[w][l] = take(sz_wt[l]); }
        h->wh[w] = (float *)take(sz_wh); h->bh[w] = (float *)take(sz_bh); h->wp[w] = take(sz_wp); h->wpb[w] = (char *)h->wp[w] + al((size_t)CNN_WP_ELEMS * 2);
    }
    h->grad = (float *)take(sz_params); h->mu = (float *)take(sz_params); h->nu = (float *)take(sz_params);
    for (int l = 0; l < 4; ++l) { h->act[l] = take(2 * sz_act[l]); h->act_t[l] = take(sz_act[l]); h->dz[l] = take(sz_act[l]); h->slab[l] = (float *)take(sz_slab[l]); h->bslab[l] = (float *)take(sz_bslab[l]); }
    h->q[0] = (float *)take(2 * sz_q); h->q[1] = nullptr; h->q[2] = (float *)take(sz_q);      // q[0]: [2 B][A] of the paired online pass
    h->targets = (float *)take(sz_q);
    h->scratch = (float *)take(al((size_t)max_batch * 4)); h->loss_part = (float *)take(al((size_t)max_batch * 4));
    h->loss = (float *)take(256); h->opt = (CnnOptState *)take(256);
    h->hslab = (float *)take(sz_hslab); h->hbslab = (float *)take(sz_hbslab);
    (void)hipMemset(h->arena, 0, total);
    if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) == hipSuccess) {
        bool ok = hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess
                  && hipEventCreateWithFlags(&h->ev_tgt, hipEventDisableTiming) == hipSuccess;
        for (int l = 1; l < 4; ++l) ok = ok && hipEventCreateWithFlags(&h->ev_dz[l], hipEventDisableTiming) == hipSuccess;
        if (!ok) { (void)hipStreamDestroy(h->side); h->side = nullptr; }
        else if (hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_side2, hipEventDisableTiming) != hipSuccess) {
            if (h->side2) (void)hipStreamDestroy(h->side2);
            h->side2 = nullptr;
        }
    } else h->side = nullptr;
    const CnnOptState st0{1.0, 1.0, 0, 3e-4f};
    (void)hipMemcpy(h->opt, &st0, sizeof(st0), hipMemcpyHostToDevice);
    *out = h;
    return DQN_OK;
}

extern "C" int dqn_cnn_destroy(dqn_cnn_handle *h) {
    if (!h) return DQN_OK;
    (void)hipDeviceSynchronize();
    if (h->side_kept) h->side = h->side_kept;
    dqn_rccl_comm_destroy(h->comm);
    if (h->ev_fc) (void)hipEventDestroy(h->ev_fc);
    if (h->ev_fc_done) (void)hipEventDestroy(h->ev_fc_done);
    if (h->comm_st) (void)hipStreamDestroy(h->comm_st);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->side2) (void)hipStreamDestroy(h->side2);
    if (h->ev_side2) (void)hipEventDestroy(h->ev_side2);
    for (int l = 1; l < 4; ++l) if (h->ev_dz[l]) (void)hipEventDestroy(h->ev_dz[l]);
    if (h->ev_side) (void)hipEventDestroy(h->ev_side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_tgt) (void)hipEventDestroy(h->ev_tgt);
    if (h->arena) (void)hipFree(h->arena);
    if (h->ring_arena) (void)hipFree(h->ring_arena);
    delete h;
    return DQN_OK;
}

extern "C" int dqn_cnn_param_count(const dqn_cnn_handle *h, int64_t *n) {
    CNN_REQ(h && n, "null argument");
    *n = h->P;
    return DQN_OK;
}

static void cnn_refresh_shadows(dqn_cnn_handle *h, int which, hipStream_t s) {
    const unsigned blocks = (unsigned)((h->P + 255) / 256);
    const unsigned fcb = (3136 / 64) * (512 / 64);
    if (h->bf16) {
        hipLaunchKernelGGL((k_cnn_scatter<__bf16>), dim3(blocks), dim3(256), 0, s, cnn_offs(h), cnn_shadows<__bf16>(h, which), h->params[which]);
        hipLaunchKernelGGL((k_cnn_fc_leaf<__bf16, false>), dim3(fcb), dim3(256), 0, s, cnn_offs(h), (__bf16 *)h->wt[which][3], (__bf16 *)h->wb[which][3], nullptr, h->params[which],
                           nullptr, nullptr, nullptr, 0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f);
    } else {
        hipLaunchKernelGGL((k_cnn_scatter<float>), dim3(blocks), dim3(256), 0, s, cnn_offs(h), cnn_shadows<float>(h, which), h->params[which]);
        hipLaunchKernelGGL((k_cnn_fc_leaf<float, false>), dim3(fcb), dim3(256), 0, s, cnn_offs(h), (float *)h->wt[which][3], (float *)h->wb[which][3], nullptr, h->params[which],
                           nullptr, nullptr, nullptr, 0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// which: DQN_NET_ONLINE / DQN_NET_TARGET. Flat f32: conv1 w[8,8,4,32] b[32] conv2 w[4,4,32,64] b[64] conv3 w[3,3,64,64] b[64]
// fc w[3136,512] b[512] val w[512,1] b[1] adv w[512,A] b[A]  (HWIO; fc rows in [7][7][64] order)
extern "C" int dqn_cnn_set_params(dqn_cnn_handle *h, int which, const float *src, int src_is_host, void *stream) {
    CNN_REQ(h && src && (which == DQN_NET_ONLINE || which == DQN_NET_TARGET), "bad argument");
    hipStream_t s = (hipStream_t)stream;
    CNN_TRY(hipMemcpyAsync(h->params[which], src, h->P * 4, src_is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    cnn_refresh_shadows(h, which, s);
    CNN_TRY(hipGetLastError());
    if (src_is_host) CNN_TRY(hipStreamSynchronize(s));
    return DQN_OK;
}

/* which_buf: DQN_BUF_PARAMS / _TARGET / _GRAD / _MU / _NU (P floats each) */
extern "C" int dqn_cnn_get_buffer(dqn_cnn_handle *h, int which_buf, float *dst, int dst_is_host, void *stream) {
    CNN_REQ(h && dst, "null argument");
    const float *src = which_buf == DQN_BUF_PARAMS ? h->params[0] : which_buf == DQN_BUF_TARGET ? h->params[1] : which_buf == DQN_BUF_GRAD ? h->grad
                     : which_buf == DQN_BUF_MU ? h->mu : which_buf == DQN_BUF_NU ? h->nu : nullptr;
    CNN_REQ(src, "unknown buffer");
    CNN_TRY(hipMemcpyAsync(dst, src, h->P * 4, dst_is_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (dst_is_host) CNN_TRY(hipStreamSynchronize((hipStream_t)stream));
    return DQN_OK;
}

/* device pointer of a handle-owned buffer (no copy): the gradient all-reduce of per-GPU learners runs in place on DQN_BUF_GRAD */
extern "C" int dqn_cnn_buffer(dqn_cnn_handle *h, int which_buf, void **ptr, int64_t *bytes) {
    CNN_REQ(h && ptr && bytes, "null argument");
    float *src = which_buf == DQN_BUF_PARAMS ? h->params[0] : which_buf == DQN_BUF_TARGET ? h->params[1] : which_buf == DQN_BUF_GRAD ? h->grad
               : which_buf == DQN_BUF_MU ? h->mu : which_buf == DQN_BUF_NU ? h->nu : nullptr;
    CNN_REQ(src, "unknown buffer");
    *ptr = src; *bytes = h->P * 4;
    return DQN_OK;
}

/* Agent._update_target_model (General/QLearning/q_agent.py:143-144) */
extern "C" int dqn_cnn_sync_target(dqn_cnn_handle *h, void *stream) {
    CNN_REQ(h, "null handle");
    CNN_TRY(hipMemcpyAsync(h->params[1], h->params[0], h->P * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    cnn_refresh_shadows(h, 1, (hipStream_t)stream);
    CNN_TRY(hipGetLastError());
    return DQN_OK;
}

/* diagnostics (tests): DQN_CNN_FLAG_FC_WIDE_TILE takes the 128 x 128 fc tile of the bf16 mode (normally from 8 192 rows) at every
 * batch size; DQN_CNN_FLAG_NO_SIDE_STREAM runs the target pass, the dW kernels and the fc leaf's optimizer step in stream order
 * on the caller's stream instead of the handle's side stream. Both leave every result bit-identical. */
extern "C" int dqn_cnn_set_flags(dqn_cnn_handle *h, int32_t flags) {
    CNN_REQ(h, "null handle");
    CNN_REQ((flags & ~(DQN_CNN_FLAG_FC_WIDE_TILE | DQN_CNN_FLAG_NO_SIDE_STREAM | DQN_CNN_FLAG_LAYERWISE_CONV)) == 0, "unknown dqn_cnn flag");
    (void)hipDeviceSynchronize();
    if (h->side && !h->side_kept) h->side_kept = h->side;
    h->side = (flags & DQN_CNN_FLAG_NO_SIDE_STREAM) ? nullptr : h->side_kept;
    h->flags = flags;
    return DQN_OK;
}

template <typename TI, typename TC, int L>
static void launch_layer(hipStream_t s, int B, const TI *in, const TI *in2, int images1, const TC *wt, const float *bias, TC *out, bool wide = false) {
    typedef CnnGeo<L> G;
    const int M = B * G::OH * G::OW;
    if constexpr (sizeof(TC) == 2 && L == 3) {
        if (B >= 8192 || wide) {
            typedef CnnTile<L, 1> T;
            const int BM = T::MT * T::WM * T::TM, BN = T::MT * T::WN * T::TN;
            DQN_LAUNCH((k_cnn_layer<TI, TC, L, 1>), dim3((unsigned)((M + BM - 1) / BM * (G::OC / BN))), dim3(256), 0, s, M, in, in2, images1, wt, bias, out);
            return;
        }
    }
    {
        typedef CnnTile<L, 0> T;
        const int BM = T::MT * T::WM * T::TM, BN = T::MT * T::WN * T::TN;
        DQN_LAUNCH((k_cnn_layer<TI, TC, L, 0>), dim3((unsigned)((M + BM - 1) / BM * (G::OC / BN))), dim3(256), 0, s, M, in, in2, images1, wt, bias, out);
    }
}

// k_cnn_trunk16's description of one pass (bf16 mode)
static TrunkJob trunk_job(const dqn_cnn_handle *h, int which, const uint8_t *frames, int B1, const uint8_t *frames2, int B2, void *const *act, bool want_maps) {
    const float *P = h->params[which];
    const __bf16 *wp = (const __bf16 *)h->wp[which];
    return TrunkJob{frames, frames2 ? frames2 : frames, B1, B1 + B2, wp, wp + CNN_WP_OFF1, wp + CNN_WP_OFF2, P + h->L[0].o_b, P + h->L[1].o_b, P + h->L[2].o_b,
                    want_maps ? (__bf16 *)act[0] : nullptr, want_maps ? (__bf16 *)act[1] : nullptr, (__bf16 *)act[2], nullptr, 1, 0, 0};
}
// Agent._step's two passes in ONE trunk launch (bf16 mode): online over s | s' (maps kept for the backward), target over s'
// ring_idx != nullptr: s / s2 are the frame ring's two arrays and transition b is row ring_idx[b] of s, row (ring_idx[b] + ring_off2)
// mod capacity of s2 (dqn_cnn_update_replay: the last row of the n-step window)
static bool cnn_trunk_fused(const dqn_cnn_handle *h) { return h->bf16 && !(h->flags & DQN_CNN_FLAG_LAYERWISE_CONV); }
static bool cnn_trunk_both(dqn_cnn_handle *h, const uint8_t *s, const uint8_t *s2, int B, hipStream_t st, const int32_t *ring_idx = nullptr, int ring_off2 = 0) {
    if (!cnn_trunk_fused(h)) return false;
    TrunkJob on = trunk_job(h, DQN_NET_ONLINE, s, B, s2, B, h->act, true), tg = trunk_job(h, DQN_NET_TARGET, s2, B, nullptr, 0, h->act_t, false);
    if (ring_idx) {
        on.idx = tg.idx = ring_idx; on.cap = tg.cap = h->ring_cap;
        on.off1 = 0; on.off2 = ring_off2; tg.off1 = ring_off2; tg.off2 = 0;
    }
    const int p0 = B, p1 = (B + 1) / 2, total = p0 + p1;
    const TrunkArgs ta{{on, tg}, p0, total};
    DQN_LAUNCH(k_cnn_trunk16, dim3((unsigned)(total < h->num_cus ? total : h->num_cus)), dim3(256), 0, st, ta);
    return true;
}
// one pass over B1 frame stacks from `frames` followed by B2 from `frames2` (B2 = 0: a plain forward); q: [B1 + B2][A]
template <typename TC>
static void cnn_forward_t(dqn_cnn_handle *h, int which, const uint8_t *frames, int B1, const uint8_t *frames2, int B2, float *q, hipStream_t s, void *const *act, bool want_maps, bool trunk_done = false) {
    const float *P = h->params[which];
    const int B = B1 + B2;
    TC *a0 = (TC *)act[0], *a1 = (TC *)act[1], *a2 = (TC *)act[2], *a3 = (TC *)act[3];
    bool fused = false;
    if constexpr (sizeof(TC) == 2) {
        if (!(h->flags & DQN_CNN_FLAG_LAYERWISE_CONV)) {           // r03: frames -> conv3's map in one persistent kernel
            if (!trunk_done) {
                const TrunkJob job = trunk_job(h, which, frames, B1, frames2, B2, act, want_maps);
                const int npairs = (B + 1) / 2;
                const TrunkArgs ta{{job, job}, npairs, npairs};
                DQN_LAUNCH(k_cnn_trunk16, dim3((unsigned)(npairs < h->num_cus ? npairs : h->num_cus)), dim3(256), 0, s, ta);
            }
            fused = true;
        }
    }
    if (!fused) {
        launch_layer<uint8_t, TC, 0>(s, B, frames, frames2 ? frames2 : frames, B1, (const TC *)h->wt[which][0], P + h->L[0].o_b, a0);
        launch_layer<TC, TC, 1>(s, B, a0, a0, B, (const TC *)h->wt[which][1], P + h->L[1].o_b, a1);
        launch_layer<TC, TC, 2>(s, B, a1, a1, B, (const TC *)h->wt[which][2], P + h->L[2].o_b, a2);
    }
    launch_layer<TC, TC, 3>(s, B, a2, a2, B, (const TC *)h->wt[which][3], P + h->L[3].o_b, a3, (h->flags & DQN_CNN_FLAG_FC_WIDE_TILE) != 0);
    hipLaunchKernelGGL((k_cnn_head<TC>), dim3((B + 15) / 16), dim3(256), 0, s, a3, h->wh[which], h->bh[which], h->A, B, q);
}
static void cnn_forward_pair(dqn_cnn_handle *h, int which, const uint8_t *f1, int B1, const uint8_t *f2, int B2, float *q, hipStream_t s, void *const *act = nullptr, bool want_maps = true, bool trunk_done = false) {
    if (!act) act = h->act;
    if (h->bf16) cnn_forward_t<__bf16>(h, which, f1, B1, f2, B2, q, s, act, want_maps, trunk_done); else cnn_forward_t<float>(h, which, f1, B1, f2, B2, q, s, act, want_maps);
}

/* Q[B][A] of the Nature-CNN dueling net for B stacks of four 84x84 u8 frames (NHWC). */
extern "C" int dqn_cnn_forward(dqn_cnn_handle *h, int which, const uint8_t *frames, int32_t B, float *q, void *stream) {
    CNN_REQ(h && frames && q && (which == DQN_NET_ONLINE || which == DQN_NET_TARGET), "bad argument");
    CNN_REQ(B >= 1 && B <= h->max_batch, "B exceeds max_batch");
    cnn_forward_pair(h, which, frames, B, nullptr, 0, q, (hipStream_t)stream);
    CNN_TRY(hipGetLastError());
    return DQN_OK;
}

/* compute_q_targets (q_learning_functions.py:42-64) with the CNN as the model: three forwards + the TD rule (k_td).
 * targets: [B][A] (the predictions with the taken action's entry replaced, :61-63). The target pass runs first; the two online
 * passes (s, s') are ONE pass over 2 B frame stacks, s first, so its activations are the ones the backward finds at rows [0, B). */
extern "C" int dqn_cnn_q_targets(dqn_cnn_handle *h, const uint8_t *s, const int32_t *a, const float *r, const uint8_t *s2,
                                 const float *d, float gamma, int32_t B, float *targets, void *stream) {
    CNN_REQ(h && s && a && r && s2 && d && targets, "null argument");
    CNN_REQ(B >= 1 && B <= h->max_batch, "B exceeds max_batch");
    hipStream_t st = (hipStream_t)stream;
    cnn_forward_pair(h, DQN_NET_TARGET, s2, B, nullptr, 0, h->q[2], st, nullptr, false);                             // :54
    cnn_forward_pair(h, DQN_NET_ONLINE, s, B, s2, B, h->q[0], st);                                   // :52, :53 in one pass: q rows [0, B) | [B, 2B)
    launch_td(st, h->q[0], h->q[0] + (size_t)B * h->A, h->q[2], a, r, d, nullptr, gamma, B, h->A, targets, nullptr, nullptr, nullptr, h->scratch);
    CNN_TRY(hipGetLastError());
    return DQN_OK;
}

template <typename TI, typename TC, int L>
static void launch_dw(dqn_cnn_handle *h, hipStream_t s, int B, const TI *in, const TC *dz, CnnSegs &segs, int &nseg, float div) {
    typedef CnnGeo<L> G;
    const int M = B * G::OH * G::OW, rows = dw_rows_per_slice(h, L, M), S = (M + rows - 1) / rows, tiles = dw_shape(L, h->bf16).tiles;
    if (S == 1 && div == 1.0f) {          // one slice: the kernel writes the leaf itself (the fc layer at B = 512: 6.4 MB less to write and re-read)
        DQN_LAUNCH((k_cnn_dw<TI, TC, L>), dim3((unsigned)tiles), dim3(256), 0, s, M, rows, in, dz, h->grad + h->L[L].o_w, h->grad + h->L[L].o_b);
        return;
    }
    DQN_LAUNCH((k_cnn_dw<TI, TC, L>), dim3((unsigned)(tiles * S)), dim3(256), 0, s, M, rows, in, dz, h->slab[L], h->bslab[L]);
    const long long nw = (long long)h->L[L].K * h->L[L].N, nb = h->L[L].N;
    segs.s[nseg++] = CnnSeg{h->slab[L], nw, h->L[L].o_w, nw, nw / 4 * 8, S, 0, div, 1};              // vector segments: 8 lanes per float4
    segs.s[nseg++] = CnnSeg{h->bslab[L], nb, h->L[L].o_b, nb, nb / 4 * 8, S, 0, 1.0f, 1};
}
template <typename TC, int L>
static void launch_bwd_data(dqn_cnn_handle *h, hipStream_t s, int B) {
    typedef BwdGeo<L> G;
    const int BM = 32 * G::WM, BN = 32 * G::WN, M = B * G::ROWS;
    DQN_LAUNCH((k_cnn_bwd_data<TC, L>), dim3((unsigned)(G::CLASSES * ((M + BM - 1) / BM) * (G::N / BN))), dim3(256), 0, s, B, (const TC *)h->dz[L],
               (const TC *)h->wb[0][L], (const TC *)h->act[L - 1], (TC *)h->dz[L - 1]);
}

// backward from the activations the last online forward left in the handle and its predictions q
// returns true when the fc weight leaf has been stepped already (prep_opt: dqn_cnn_update; only then)
template <typename TC>
static bool cnn_backward_t(dqn_cnn_handle *h, const uint8_t *frames, const float *q, const float *targets, const float *isw, int B, hipStream_t s, bool prep_opt,
                           const CnnTdArgs &td = CnnTdArgs{}, int *fc_reduced = nullptr) {
    const int blocks = (B + 15) / 16;
    hipLaunchKernelGGL((k_cnn_head_bwd<TC>), dim3(blocks), dim3(256), 0, s, q, targets, isw, (const TC *)h->act[3], h->wh[0], h->A, B, (TC *)h->dz[3], h->hslab, h->hbslab,
                       h->loss_part, td);
    CnnSegs segs{}; int nseg = 0;
    segs.s[nseg++] = CnnSeg{h->hslab, CNN_F, h->o_wv, 16 * CNN_F, CNN_F / 4 * 8, blocks, 0, 1.0f, 1};
    segs.s[nseg++] = CnnSeg{h->hbslab, 1, h->o_bv, 16, 8, blocks, 0, 1.0f, 0};
    segs.s[nseg++] = CnnSeg{h->hslab, (long long)CNN_F * h->A, h->o_wa, 16 * CNN_F, ((long long)CNN_F * h->A + 7) / 8 * 8, blocks, h->A, 1.0f, 0};
    segs.s[nseg++] = CnnSeg{h->hbslab + 1, h->A, h->o_ba, 16, (h->A + 7) / 8 * 8, blocks, 0, 1.0f, 0};
    // dW_l needs dZ_l only: layers 3..1 go to the side stream as soon as their dZ exists, the chain dZ4 -> dZ3 -> dZ2 -> dZ1 -> dW_conv1
    // stays on the caller's stream (fork / join by events: also valid inside a stream capture)
    hipStream_t sd = h->side ? h->side : s;
    if (h->side) { (void)hipEventRecord(h->ev_dz[3], s); (void)hipStreamWaitEvent(sd, h->ev_dz[3], 0); }
    if (prep_opt) hipLaunchKernelGGL(k_cnn_opt_prep, dim3(1), dim3(1), 0, sd, h->opt, h->b1, h->b2);
    launch_dw<TC, TC, 3>(h, sd, B, (const TC *)h->act[2], (const TC *)h->dz[3], segs, nseg, 1.0f);
    launch_bwd_data<TC, 3>(h, s, B);
    // (r03: conv3's and conv2's dW on a SECOND side stream: behind the fc leaf's 20 us optimizer step on the first they were the
    //  update's critical path -- dW3 + leaf step + dW2 + dW1 = 80 us in a row against 60 on the caller's stream)
#ifndef CNN_SD2_MAX_B
#define CNN_SD2_MAX_B 1024
#endif
    hipStream_t sd2 = (h->side && h->side2 && cnn_trunk_fused(h) && B <= CNN_SD2_MAX_B) ? h->side2 : sd;      // (exact-f32 mode, large batches: the dW kernels fill the chip, measured slower)
    if (h->side) { (void)hipEventRecord(h->ev_dz[2], s); (void)hipStreamWaitEvent(sd, h->ev_dz[2], 0); if (sd2 != sd) (void)hipStreamWaitEvent(sd2, h->ev_dz[2], 0); }
    // the fc leaf (95 % of the parameters) is complete when its single-slice dW is (written straight into the gradient) and
    // backward-data has read the fc weights for the last time: its Adam step goes here, beside the rest of the backward
    bool fc_done = false;
    if (prep_opt && h->side && dw_slices(h, 3, B) == 1) {
        hipLaunchKernelGGL((k_cnn_fc_leaf<TC, true>), dim3((3136 / 64) * (512 / 64)), dim3(256), 0, sd, cnn_offs(h), (TC *)h->wt[0][3], (TC *)h->wb[0][3], h->opt, h->params[0], h->grad,
                           h->mu, h->nu, h->adamw, h->b1, h->b2, h->eps, h->wd, 1.0f);
        fc_done = true;
    }
    // data-parallel learners: the fc weight leaf is 95 % of the gradient (6.4 MB) and final here (one slice: its dW kernel wrote it
    // straight into the gradient buffer) -- its all-reduce is issued NOW on the communicator's own stream (behind an event of the
    // side stream, so neither the dW kernels that follow there nor the backward-data chain wait for it): the message whose xGMI
    // bandwidth matters travels beside conv3 / conv2 / conv1's backward; the small leaves follow after the reduction
    if (fc_reduced && h->comm && h->world > 1 && h->side && h->comm_st && dw_slices(h, 3, B) == 1) {
        (void)hipEventRecord(h->ev_fc, sd); (void)hipStreamWaitEvent(h->comm_st, h->ev_fc, 0);
        if (dqn_rccl_allreduce_sum_f32(h->comm, h->grad + h->L[3].o_w, (size_t)h->L[3].K * h->L[3].N, h->comm_st) == DQN_OK) {
            *fc_reduced = 1;
            (void)hipEventRecord(h->ev_fc_done, h->comm_st);
        }
    }
    launch_dw<TC, TC, 2>(h, sd2, B, (const TC *)h->act[1], (const TC *)h->dz[2], segs, nseg, 1.0f);
    bool bfused = false;
    if constexpr (sizeof(TC) == 2) {
        if (cnn_trunk_fused(h)) {                      // r03: conv3^T and conv2^T in one persistent kernel (dqn_cnn_btrunk.h)
            const BTrunkArgs ba{(const __bf16 *)h->dz[2], (const __bf16 *)h->act[1], (const __bf16 *)h->act[0], (__bf16 *)h->dz[1], (__bf16 *)h->dz[0],
                                (const __bf16 *)h->wpb[0], (const __bf16 *)h->wpb[0] + CNN_WPB_OFF1, B};
            const int npairs = (B + 1) / 2;
            DQN_LAUNCH(k_cnn_btrunk16, dim3((unsigned)(npairs < h->num_cus ? npairs : h->num_cus)), dim3(256), 0, s, ba);
            if (h->side) { (void)hipEventRecord(h->ev_dz[1], s); (void)hipStreamWaitEvent(sd2, h->ev_dz[1], 0); }
            launch_dw<TC, TC, 1>(h, sd2, B, (const TC *)h->act[0], (const TC *)h->dz[1], segs, nseg, 1.0f);
            bfused = true;
        }
    }
    if (!bfused) {
        launch_bwd_data<TC, 2>(h, s, B);
        if (h->side) { (void)hipEventRecord(h->ev_dz[1], s); (void)hipStreamWaitEvent(sd2, h->ev_dz[1], 0); }
        launch_dw<TC, TC, 1>(h, sd2, B, (const TC *)h->act[0], (const TC *)h->dz[1], segs, nseg, 1.0f);
        launch_bwd_data<TC, 1>(h, s, B);
    }
    launch_dw<uint8_t, TC, 0>(h, s, B, frames, (const TC *)h->dz[0], segs, nseg, 255.0f);      // the kernel multiplies integer pixels (both modes)
    if (h->side) { (void)hipEventRecord(h->ev_side, sd); (void)hipStreamWaitEvent(s, h->ev_side, 0); }
    if (sd2 != sd) { (void)hipEventRecord(h->ev_side2, sd2); (void)hipStreamWaitEvent(s, h->ev_side2, 0); }
    long long total = 0;
    for (int i = 0; i < nseg; ++i) total += segs.s[i].units;
    segs.count = nseg;
    hipLaunchKernelGGL(k_cnn_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, segs, total, h->grad, h->loss_part, blocks, B, h->loss);
    return fc_done;
}

/* gradient of compute_loss (q_learning_functions.py:31-39: mean_i w_i sum_a huber(model(s_i)[a] - targets[i][a])) w.r.t. every
 * leaf, into the handle's gradient buffer (dqn_cnn_get_buffer(DQN_BUF_GRAD)); loss_host (optional) receives the loss. */
extern "C" int dqn_cnn_grads(dqn_cnn_handle *h, const uint8_t *frames, const float *targets, const float *isw, int32_t B, float *loss_host, void *stream) {
    CNN_REQ(h && frames && targets, "null argument");
    CNN_REQ(B >= 1 && B <= h->max_batch, "B exceeds max_batch");
    hipStream_t s = (hipStream_t)stream;
    int rc = dqn_cnn_forward(h, DQN_NET_ONLINE, frames, B, h->q[0], stream); if (rc) return rc;
    if (h->bf16) cnn_backward_t<__bf16>(h, frames, h->q[0], targets, isw, B, s, false); else cnn_backward_t<float>(h, frames, h->q[0], targets, isw, B, s, false);
    CNN_TRY(hipGetLastError());
    if (loss_host) { CNN_TRY(hipMemcpyAsync(loss_host, h->loss, 4, hipMemcpyDeviceToHost, s)); CNN_TRY(hipStreamSynchronize(s)); }
    return DQN_OK;
}

/* optimizer = optax.adam / adamw(lr, b1, b2, eps, weight_decay) (Test/lunar_lander.py:48); resets moments and step count */
extern "C" int dqn_cnn_set_optimizer(dqn_cnn_handle *h, int32_t adamw, float lr, float b1, float b2, float eps, float weight_decay, void *stream) {
    CNN_REQ(h && lr >= 0.0f && b1 >= 0.0f && b1 < 1.0f && b2 >= 0.0f && b2 < 1.0f && eps >= 0.0f, "bad optimizer argument");
    hipStream_t s = (hipStream_t)stream;
    h->adamw = adamw != 0; h->b1 = b1; h->b2 = b2; h->eps = eps; h->wd = weight_decay;
    const CnnOptState st0{1.0, 1.0, 0, lr};
    CNN_TRY(hipMemcpyAsync(h->opt, &st0, sizeof(st0), hipMemcpyHostToDevice, s));
    CNN_TRY(hipMemsetAsync(h->mu, 0, h->P * 4, s)); CNN_TRY(hipMemsetAsync(h->nu, 0, h->P * 4, s));
    CNN_TRY(hipStreamSynchronize(s));
    return DQN_OK;
}

static int cnn_adam(dqn_cnn_handle *h, float grad_scale, hipStream_t s, bool fc_done = false) {
    const unsigned blocks = (unsigned)((h->P + 255) / 256);
    const unsigned fcb = (3136 / 64) * (512 / 64);
    if (h->bf16) {
        if (!fc_done) hipLaunchKernelGGL((k_cnn_fc_leaf<__bf16, true>), dim3(fcb), dim3(256), 0, s, cnn_offs(h), (__bf16 *)h->wt[0][3], (__bf16 *)h->wb[0][3], h->opt, h->params[0], h->grad, h->mu, h->nu,
                           h->adamw, h->b1, h->b2, h->eps, h->wd, grad_scale);
        hipLaunchKernelGGL((k_cnn_adam<__bf16>), dim3(blocks), dim3(256), 0, s, cnn_offs(h), cnn_shadows<__bf16>(h, 0), h->opt, h->params[0], h->grad, h->mu, h->nu,
                           h->adamw, h->b1, h->b2, h->eps, h->wd, grad_scale);
    } else {
        if (!fc_done) hipLaunchKernelGGL((k_cnn_fc_leaf<float, true>), dim3(fcb), dim3(256), 0, s, cnn_offs(h), (float *)h->wt[0][3], (float *)h->wb[0][3], h->opt, h->params[0], h->grad, h->mu, h->nu,
                           h->adamw, h->b1, h->b2, h->eps, h->wd, grad_scale);
        hipLaunchKernelGGL((k_cnn_adam<float>), dim3(blocks), dim3(256), 0, s, cnn_offs(h), cnn_shadows<float>(h, 0), h->opt, h->params[0], h->grad, h->mu, h->nu,
                           h->adamw, h->b1, h->b2, h->eps, h->wd, grad_scale);
    }
    CNN_TRY(hipGetLastError());
    return DQN_OK;
}

/* optimizer.update + optax.apply_updates (q_learning_functions.py:24-25) on the gradient buffer; refreshes the shadows */
extern "C" int dqn_cnn_optimizer_step(dqn_cnn_handle *h, float grad_scale, void *stream) {
    CNN_REQ(h, "null handle");
    hipLaunchKernelGGL(k_cnn_opt_prep, dim3(1), dim3(1), 0, (hipStream_t)stream, h->opt, h->b1, h->b2);
    return cnn_adam(h, grad_scale, (hipStream_t)stream);
}

/* train_step (q_learning_functions.py:14-28) */
extern "C" int dqn_cnn_train_step(dqn_cnn_handle *h, const uint8_t *frames, const float *targets, const float *isw, int32_t B, void *stream) {
    int rc = dqn_cnn_grads(h, frames, targets, isw, B, nullptr, stream); if (rc) return rc;
    return dqn_cnn_optimizer_step(h, 1.0f, stream);
}

/* Agent._step (q_agent.py:146-169) on a given minibatch: compute_q_targets (three forwards), then train_step's backward from
 * the activations of the online pass over s (the fourth forward of the reference's two separate jits is the same numbers),
 * Adam / AdamW. loss_host optional; td_abs_out (device, B floats, optional) = |delta| of q_learning_functions.py:58 per sample
 * (the priorities a PER write-back wants). */
/* Per-GPU learners: the handle's own RCCL communicator (one process per GPU; unique id from dqn_comm_unique_id on rank 0, carried
 * to the other ranks by the caller, e.g. torch.distributed). With it dqn_cnn_update / dqn_cnn_update_replay all-reduce the gradient
 * (sum) before the optimizer and scale it by 1 / world. */
extern "C" int dqn_cnn_comm_init(dqn_cnn_handle *h, const void *unique_id_128, int32_t rank, int32_t world) {
    CNN_REQ(h && unique_id_128 && world >= 1 && rank >= 0 && rank < world, "dqn_cnn_comm_init: bad argument");
    CNN_REQ(!h->comm, "the handle has a communicator already");
    int rc = dqn_rccl_comm_init(&h->comm, unique_id_128, rank, world); if (rc) return rc;
    h->rank = rank; h->world = world;
    if (!h->ev_fc) (void)hipEventCreateWithFlags(&h->ev_fc, hipEventDisableTiming);
    if (!h->ev_fc_done) (void)hipEventCreateWithFlags(&h->ev_fc_done, hipEventDisableTiming);
    if (!h->comm_st && hipStreamCreateWithFlags(&h->comm_st, hipStreamNonBlocking) != hipSuccess) h->comm_st = nullptr;
    return DQN_OK;
}
extern "C" int dqn_cnn_comm_count_host(dqn_cnn_handle *h, int32_t *ranks) {
    CNN_REQ(h && ranks, "null argument");
    *ranks = 0;
    if (!h->comm) return DQN_OK;
    int n = 0;
    int rc = dqn_rccl_comm_count(h->comm, &n); if (rc) return rc;
    *ranks = n;
    return DQN_OK;
}
/* In-place SUM all-reduce of the gradient buffer (DQN_BUF_GRAD) over the communicator's ranks. fc_leaf_done != 0: the fc weight
 * leaf has been reduced already (the data-parallel update issues it beside the backward): the two ranges around it follow here.
 * Between dqn_cnn_grads and dqn_cnn_optimizer_step(grad_scale = 1 / world) call it with fc_leaf_done = 0. */
extern "C" int dqn_cnn_allreduce_grads(dqn_cnn_handle *h, int32_t fc_leaf_done, void *stream) {
    CNN_REQ(h, "null handle");
    if (!h->comm) return dqn_set_error(DQN_ERR_STATE, "dqn_cnn_allreduce_grads before dqn_cnn_comm_init");
    hipStream_t st = (hipStream_t)stream;
    if (!fc_leaf_done) return dqn_rccl_allreduce_sum_f32(h->comm, h->grad, (size_t)h->P, st);
    const long long lo = h->L[3].o_w, hi = lo + (long long)h->L[3].K * h->L[3].N;
    int rc = dqn_rccl_allreduce_sum_f32(h->comm, h->grad, (size_t)lo, st); if (rc) return rc;
    return dqn_rccl_allreduce_sum_f32(h->comm, h->grad + hi, (size_t)(h->P - hi), st);
}

static int cnn_update_impl(dqn_cnn_handle *h, const uint8_t *s, const int32_t *a, const float *r, const uint8_t *s2, const float *d,
                           const float *isw, float gamma, int32_t B, float *td_abs_out, float *loss_host, void *stream,
                           const int32_t *ring_idx = nullptr, int ring_off2 = 0) {
    CNN_REQ(h && s && a && r && s2 && d, "null argument");
    CNN_REQ(B >= 1 && B <= h->max_batch, "B exceeds max_batch");
    hipStream_t st = (hipStream_t)stream;
    int rc = DQN_OK;
    // the target pass (:54) runs on the side stream in its own activation buffers, beside the online pass over s and s' (:52, :53)
    // (bf16 mode, r03: the convolutions of both passes are one launch; only the target's fc + heads go to the side stream)
    const bool both = ring_idx ? cnn_trunk_both(h, h->ring_s, h->ring_s2, B, st, ring_idx, ring_off2) : cnn_trunk_both(h, s, s2, B, st);
    if (h->side) { (void)hipEventRecord(h->ev_fork, st); (void)hipStreamWaitEvent(h->side, h->ev_fork, 0); }
    cnn_forward_pair(h, DQN_NET_TARGET, s2, B, nullptr, 0, h->q[2], h->side ? h->side : st, h->act_t, false, both);
    if (h->side) (void)hipEventRecord(h->ev_tgt, h->side);
    cnn_forward_pair(h, DQN_NET_ONLINE, s, B, s2, B, h->q[0], st, nullptr, true, both);              // one pass; s's activations are rows [0, B)
    if (h->side) (void)hipStreamWaitEvent(st, h->ev_tgt, 0);
    // the TD rule (:55-60) is the first thing the head-backward kernel does with the three Q tensors
    const CnnTdArgs td{h->q[0] + (size_t)B * h->A, h->q[2], a, r, d, gamma, td_abs_out};
    if (h->comm) {
        // per-GPU learners: backward without the early fc-leaf step, the gradient summed over the ranks (fc leaf beside
        // the backward, the rest behind it), then the optimizer with grad_scale = 1 / world
        int fc_reduced = 0;
        if (h->bf16) cnn_backward_t<__bf16>(h, s, h->q[0], nullptr, isw, B, st, false, td, &fc_reduced); else cnn_backward_t<float>(h, s, h->q[0], nullptr, isw, B, st, false, td, &fc_reduced);
        CNN_TRY(hipGetLastError());
        if (fc_reduced) (void)hipStreamWaitEvent(st, h->ev_fc_done, 0);                              // the fc leaf's sum has landed
        rc = dqn_cnn_allreduce_grads(h, fc_reduced, stream); if (rc) return rc;
        hipLaunchKernelGGL(k_cnn_opt_prep, dim3(1), dim3(1), 0, st, h->opt, h->b1, h->b2);      // step counters (the single-learner route does it beside the backward)
        rc = cnn_adam(h, 1.0f / (float)h->world, st, false); if (rc) return rc;
        if (loss_host) { CNN_TRY(hipMemcpyAsync(loss_host, h->loss, 4, hipMemcpyDeviceToHost, st)); CNN_TRY(hipStreamSynchronize(st)); }
        return DQN_OK;
    }
    const bool fc_done = h->bf16 ? cnn_backward_t<__bf16>(h, s, h->q[0], nullptr, isw, B, st, true, td) : cnn_backward_t<float>(h, s, h->q[0], nullptr, isw, B, st, true, td);
    CNN_TRY(hipGetLastError());
    rc = cnn_adam(h, 1.0f, st, fc_done); if (rc) return rc;
    if (loss_host) { CNN_TRY(hipMemcpyAsync(loss_host, h->loss, 4, hipMemcpyDeviceToHost, st)); CNN_TRY(hipStreamSynchronize(st)); }
    return DQN_OK;
}
extern "C" int dqn_cnn_update(dqn_cnn_handle *h, const uint8_t *s, const int32_t *a, const float *r, const uint8_t *s2, const float *d,
                              const float *isw, float gamma, int32_t B, float *loss_host, void *stream) {
    return cnn_update_impl(h, s, a, r, s2, d, isw, gamma, B, nullptr, loss_host, stream);
}

/* Agent._policy (q_agent.py:137-141) with the CNN: epsilon-greedy actions for n frame stacks; Philox (seed, ctr, i, policy
 * stream) as dqn_act */
extern "C" int dqn_cnn_act(dqn_cnn_handle *h, const uint8_t *frames, int32_t n, float epsilon, uint64_t seed, uint64_t ctr, int32_t *actions, void *stream) {
    CNN_REQ(h && frames && actions, "null argument");
    CNN_REQ(n >= 1 && n <= h->max_batch, "n exceeds max_batch");
    cnn_forward_pair(h, DQN_NET_ONLINE, frames, n, nullptr, 0, h->q[0], (hipStream_t)stream, nullptr, false);      // (acting keeps no maps)
    launch_policy((hipStream_t)stream, h->q[0], n, h->A, epsilon, seed, ctr, actions, nullptr);
    CNN_TRY(hipGetLastError());
    return DQN_OK;
}

/* ReplayBuffer.__init__ (replay_buffer.py:20-34) for frame stacks: capacity rows of s, s' (u8), a, r, d; zeroed */
extern "C" int dqn_cnn_replay_init(dqn_cnn_handle *h, int64_t capacity) {
    CNN_REQ(h && capacity >= 1 && capacity <= ((int64_t)1 << 24), "dqn_cnn_replay_init: bad capacity");
    CNN_REQ(!h->ring_arena, "the ring exists already");
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t fr = al((size_t)capacity * CNN_FRAME_BYTES), sc = al((size_t)capacity * 4), sfr = al((size_t)h->max_batch * CNN_FRAME_BYTES), ssc = al((size_t)h->max_batch * 4);
    const size_t total = 2 * fr + 3 * sc + 2 * sfr + 3 * ssc;
    hipError_t e = hipMalloc(&h->ring_arena, total);
    if (e != hipSuccess) { h->ring_arena = nullptr; return dqn_set_error(DQN_ERR_NOMEM, (std::string("hipMalloc (frame ring): ") + hipGetErrorString(e)).c_str()); }
    char *c = (char *)h->ring_arena;
    h->ring_s = (uint8_t *)c; c += fr; h->ring_s2 = (uint8_t *)c; c += fr;
    h->ring_a = (int32_t *)c; c += sc; h->ring_r = (float *)c; c += sc; h->ring_d = (float *)c; c += sc;
    h->stage_s = (uint8_t *)c; c += sfr; h->stage_s2 = (uint8_t *)c; c += sfr;
    h->stage_a = (int32_t *)c; c += ssc; h->stage_r = (float *)c; c += ssc; h->stage_d = (float *)c; c += ssc;
    CNN_TRY(hipMemset(h->ring_arena, 0, total));
    h->ring_cap = capacity; h->ring_counter = 0;
    return DQN_OK;
}

/* ReplayBuffer.add (replay_buffer.py:58-65), n transitions at once: rows counter .. counter + n - 1 (mod capacity).
 * first_index (optional) receives the ring position of the first one. */
extern "C" int dqn_cnn_replay_add(dqn_cnn_handle *h, const uint8_t *s, const int32_t *a, const float *r, const uint8_t *s2, const float *d,
                                  int32_t n, int64_t *first_index, void *stream) {
    CNN_REQ(h && h->ring_arena, "no ring: call dqn_cnn_replay_init");
    CNN_REQ(s && a && r && s2 && d && n >= 1 && n <= h->ring_cap, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long long pos = h->ring_counter % h->ring_cap, n1 = (pos + n <= h->ring_cap) ? n : h->ring_cap - pos, n2 = n - n1;
    auto put = [&](void *ring, const void *src, size_t row) -> hipError_t {
        hipError_t e = hipMemcpyAsync((char *)ring + pos * row, src, n1 * row, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && n2) e = hipMemcpyAsync(ring, (const char *)src + n1 * row, n2 * row, hipMemcpyDeviceToDevice, st);
        return e;
    };
    CNN_TRY(put(h->ring_s, s, CNN_FRAME_BYTES)); CNN_TRY(put(h->ring_s2, s2, CNN_FRAME_BYTES));
    CNN_TRY(put(h->ring_a, a, 4)); CNN_TRY(put(h->ring_r, r, 4)); CNN_TRY(put(h->ring_d, d, 4));
    if (first_index) *first_index = pos;
    h->ring_counter += n;
    return DQN_OK;
}

// ---- the synthetic vector env of BASELINE configs[4]'s shape ON THE DEVICE (no physics; frames are uniform u8,
// rewards Irwin-Hall normals, dones Bernoulli(p_done)): one kernel files a whole vector step in the frame ring --
//   ring_s[row]  = the env's current frame stack (copied from where the previous step left it: the ring's own s' row),
//   ring_s2[row] = the next frame stack, drawn here: 16 bytes per Philox4x32-10 call, counter (step, 16-byte piece index),
//   ring_r / ring_d from one more draw per env; ring_a was written by the policy kernel in front of it.
// The next step's "current frames" ARE the s' rows just written (no copy back to a staging buffer).
__global__ void __launch_bounds__(256)
k_cnn_synth_step(const uint8_t *__restrict__ cur, uint8_t *__restrict__ ring_s, uint8_t *__restrict__ ring_s2, float *__restrict__ ring_r,
                 float *__restrict__ ring_d, long long pos, int n, unsigned long long seed, unsigned long long step, float p_done,
                 const float *__restrict__ q, int A, float epsilon, int32_t *__restrict__ ring_a) {
    const long long pieces = (long long)n * (CNN_FRAME_BYTES / 16);
    const uint4 *src = reinterpret_cast<const uint4 *>(cur);
    uint4 *ds = reinterpret_cast<uint4 *>(ring_s + pos * CNN_FRAME_BYTES), *d2 = reinterpret_cast<uint4 *>(ring_s2 + pos * CNN_FRAME_BYTES);
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < pieces; p += (long long)gridDim.x * blockDim.x) {
        const uint4 v = src[p];
        const u32x4 o = philox_draw(seed, step, (uint32_t)p, DQN_STREAM_ENV);
        ds[p] = v;
        d2[p] = uint4{o.x, o.y, o.z, o.w};
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const u32x4 o = philox_draw(seed, step, 0x80000000u + (uint32_t)i, DQN_STREAM_ENV);        // (piece indices stay below 2^31)
        const u32x4 o2 = philox_draw(seed, step, 0xC0000000u + (uint32_t)i, DQN_STREAM_ENV);
        ring_r[pos + i] = (((u01(o.x) + u01(o.y)) + (u01(o.z) + u01(o.w))) - 2.0f) * 1.73205078f;     // Irwin-Hall normal (as ih_normal)
        ring_d[pos + i] = u01(o2.x) < p_done ? 1.0f : 0.0f;
        ring_a[pos + i] = policy_row(q + (long long)i * A, A, epsilon, seed, step, i);             // q_agent.py:137-141, as dqn_cnn_act
    }
}
__global__ void __launch_bounds__(256)
k_cnn_synth_frames(uint8_t *__restrict__ dst, int n, unsigned long long seed, unsigned long long step) {
    const long long pieces = (long long)n * (CNN_FRAME_BYTES / 16);
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < pieces; p += (long long)gridDim.x * blockDim.x) {
        const u32x4 o = philox_draw(seed, step, (uint32_t)p, DQN_STREAM_ENV);
        reinterpret_cast<uint4 *>(dst)[p] = uint4{o.x, o.y, o.z, o.w};
    }
}

/* reset() of n synthetic frame-stack envs: the first observations are the Philox frames of "step" 2^63 + seed-independent constant;
 * needs the ring (dqn_cnn_replay_init) with capacity % n == 0 so that a vector step never wraps inside the ring */
extern "C" int dqn_cnn_env_reset_synth(dqn_cnn_handle *h, int32_t n, uint64_t seed, void *stream) {
    CNN_REQ(h && h->ring_arena, "no ring: call dqn_cnn_replay_init");
    CNN_REQ(n >= 1 && n <= h->max_batch && h->ring_cap % n == 0, "dqn_cnn_env_reset_synth: n must divide the ring capacity and fit max_batch");
    h->env_n = n; h->env_seed = seed; h->env_steps = 0;
    h->env_cur = h->stage_s2;                                           // (free until the first update stages a batch: by then the envs live in the ring)
    hipLaunchKernelGGL(k_cnn_synth_frames, dim3(1024), dim3(256), 0, (hipStream_t)stream, h->stage_s2, n, seed, 0x8000000000000000ull);
    CNN_TRY(hipGetLastError());
    CNN_TRY(hipStreamSynchronize((hipStream_t)stream));                 // (the staging buffer must not be reused before the first step has read it)
    return DQN_OK;
}

/* One vector env step, everything on the device (q_agent.py:176-183 for n envs): epsilon-greedy actions of the CNN on the current
 * frame stacks (Philox policy stream (seed, step, env), as dqn_cnn_act), the synthetic transition, ReplayBuffer.add of the n
 * transitions at ring rows [first, first + n). first_index (optional): the row of env 0. */
extern "C" int dqn_cnn_env_step_synth(dqn_cnn_handle *h, float epsilon, float p_done, int64_t *first_index, void *stream) {
    CNN_REQ(h && h->ring_arena && h->env_n > 0, "no envs: call dqn_cnn_env_reset_synth");
    hipStream_t st = (hipStream_t)stream;
    const int n = h->env_n;
    const long long pos = h->ring_counter % h->ring_cap;
    cnn_forward_pair(h, DQN_NET_ONLINE, h->env_cur, n, nullptr, 0, h->q[0], (hipStream_t)stream, nullptr, false);
    hipLaunchKernelGGL(k_cnn_synth_step, dim3(2048), dim3(256), 0, st, h->env_cur, h->ring_s, h->ring_s2, h->ring_r, h->ring_d, pos, n,
                       h->env_seed, (unsigned long long)h->env_steps, p_done, h->q[0], h->A, epsilon, h->ring_a);
    CNN_TRY(hipGetLastError());
    h->env_cur = h->ring_s2 + pos * CNN_FRAME_BYTES;
    if (first_index) *first_index = pos;
    h->ring_counter += n; h->env_steps += 1;
    return DQN_OK;
}

extern "C" int dqn_cnn_replay_size_host(const dqn_cnn_handle *h, int64_t *size, int64_t *counter) {
    CNN_REQ(h && size && counter, "null argument");
    *counter = h->ring_counter; *size = h->ring_counter < h->ring_cap ? h->ring_counter : h->ring_cap;
    return DQN_OK;
}

/* sample_batch's gather (replay_buffer.py:79-85) for given indices. n_step = 1: the stored rows. n_step 2..8 (rows stored
 * step-major, n_envs per step): the n-step transition that starts at each row -- (s, a, R, s' of the last step, done_n) */
extern "C" int dqn_cnn_replay_gather(dqn_cnn_handle *h, const int32_t *idx, int32_t B, int32_t n_step, int32_t n_envs, float gamma,
                                     uint8_t *s, int32_t *a, float *r, uint8_t *s2, float *d, void *stream) {
    CNN_REQ(h && h->ring_arena, "no ring: call dqn_cnn_replay_init");
    CNN_REQ(idx && s && a && r && s2 && d && B >= 1, "bad argument");
    CNN_REQ(n_step >= 1 && n_step <= 8 && (n_step == 1 || (n_envs >= 1 && h->ring_cap % n_envs == 0 && (long long)n_step * n_envs <= h->ring_cap)), "bad n_step / n_envs");
    hipLaunchKernelGGL(k_cnn_gather, dim3(B, 2), dim3(256), 0, (hipStream_t)stream, h->ring_s, h->ring_s2, h->ring_a, h->ring_r, h->ring_d, idx, B, h->ring_cap,
                       n_step, n_step == 1 ? 0 : n_envs, gamma, s, s2, a, r, d);
    CNN_TRY(hipGetLastError());
    return DQN_OK;
}

/* Agent._step (q_agent.py:146-169) from the frame ring: gather the rows idx (sampled by the caller's PER tree; n-step
 * transitions when n_step > 1, bootstrapped with gamma^n), update with the importance weights isw (optional), |delta| per
 * sample into td_abs_out (optional) for the priority write-back */
extern "C" int dqn_cnn_update_replay(dqn_cnn_handle *h, const int32_t *idx, const float *isw, float gamma, int32_t n_step, int32_t n_envs, int32_t B,
                                     float *td_abs_out, float *loss_host, void *stream) {
    CNN_REQ(h && h->ring_arena, "no ring: call dqn_cnn_replay_init");
    CNN_REQ(idx && B >= 1 && B <= h->max_batch, "bad argument");
    CNN_REQ(h->env_cur != h->stage_s2, "the synthetic envs' reset frames still sit in the staging buffer: take one dqn_cnn_env_step_synth before the first update");
    float gn = gamma;
    for (int k = 1; k < n_step; ++k) gn = gn * gamma;                  // gamma^n as n - 1 f32 products
    if (cnn_trunk_fused(h)) {
        // r03 (bf16 mode): the trunk kernel reads the sampled rows where they lie in the ring (its DMA source address). What is left
        // of the gather -- action / n-step return / done, and the copy of s that conv1's weight gradient reads much later -- goes to
        // the side stream, which the update joins before its backward anyway (14.4 MB instead of 2 x 14.4 MB, off the critical path)
        CNN_REQ(n_step >= 1 && n_step <= 8 && (n_step == 1 || (n_envs >= 1 && h->ring_cap % n_envs == 0 && (long long)n_step * n_envs <= h->ring_cap)), "bad n_step / n_envs");
        hipStream_t st = (hipStream_t)stream, sd = h->side ? h->side : st;
        if (h->side) { (void)hipEventRecord(h->ev_fork, st); (void)hipStreamWaitEvent(sd, h->ev_fork, 0); }
        hipLaunchKernelGGL(k_cnn_gather, dim3(B, 1), dim3(256), 0, sd, h->ring_s, h->ring_s2, h->ring_a, h->ring_r, h->ring_d, idx, B, h->ring_cap,
                           n_step, n_step == 1 ? 0 : n_envs, gamma, h->stage_s, h->stage_s2, h->stage_a, h->stage_r, h->stage_d);
        CNN_TRY(hipGetLastError());
        return cnn_update_impl(h, h->stage_s, h->stage_a, h->stage_r, h->stage_s2, h->stage_d, isw, gn, B, td_abs_out, loss_host, stream,
                               idx, n_step == 1 ? 0 : (n_step - 1) * n_envs);
    }
    int rc = dqn_cnn_replay_gather(h, idx, B, n_step, n_envs, gamma, h->stage_s, h->stage_a, h->stage_r, h->stage_s2, h->stage_d, stream); if (rc) return rc;
    return cnn_update_impl(h, h->stage_s, h->stage_a, h->stage_r, h->stage_s2, h->stage_d, isw, gn, B, td_abs_out, loss_host, stream);
}
