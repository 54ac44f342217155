// csrc/dqn_cnn.hip -- Nature-CNN dueling Q-network, forward (BASELINE configs[4]: PongNoFrameskip-v4 shape, SURVEY.md 8(f)
// rank 4). Not in the reference, which has only the MLP of LunarLander/dddqn.py:19-22; the trunk ends in the reference's
// dueling head (dddqn.py:29-31: Q = val + adv - mean(adv)) and its Q values feed the reference's TD rule unchanged
// (General/QLearning/q_learning_functions.py:55-60, k_td).
//
//   frames u8 [B][84][84][4] (NHWC, 4 stacked frames) / 255
//   conv1 32 x 8x8 / 4 -> [B][20][20][32]   conv2 64 x 4x4 / 2 -> [B][9][9][64]   conv3 64 x 3x3 / 1 -> [B][7][7][64]
//   fc 3136 -> 512, ReLU after each; val 512 -> 1, adv 512 -> A
//
// Every layer is ONE implicit GEMM kernel, Out[M][N] = relu(Patch[M][K] . W[K][N] + b): row m = output position
// (b, oh, ow), k = (kh, kw, c) -- in NHWC a patch row (kh fixed) is KW*IC contiguous elements, a multiple of the 32-deep
// k-chunk for every layer, so a chunk of a row is one contiguous run: gathered straight from the activation tensor into
// an LDS image, no im2col buffer. Weights are kept transposed ([N][K], k contiguous) so that both MFMA operands are
// 16-byte LDS reads of consecutive k.
//   precision bf16: v_mfma_f32_32x32x16_bf16, activations / weights bf16, f32 accumulate (tolerance 2e-2 of scale)
//   precision f32 : v_mfma_f32_32x32x2_f32, exact: every output is the k-ascending fmaf chain of the CPU restatement
#include "../../include/dqn_hip.h"
#include "dqn_device.h"
#include "dqn_launch.h"
#include "dqn_net_common.h"

#include <new>
#include <string>

typedef float f32x16c __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));

// Geometry of the four GEMM layers, compile-time (every index division below folds into shifts / multiplies), with the tile
// shape chosen per layer: workgroup tile = (MT*WM) rows x (MT*WN) columns, one MT x MT MFMA tile per wave (4 waves), and R =
// depth of the register prefetch ring (R divides the chunk count K / 64).
//   conv1  M = 400 B, K = 256 (4 chunks),  N = 32   128 x 32 tile, everything requested up front
//   conv2  M =  81 B, K = 512 (8),         N = 64    64 x 64
//   conv3  M =  49 B, K = 576 (9),         N = 64    64 x 64
//   fc     M =      B, K = 3136 (49),      N = 512   32 x 32 from 16 x 16 MFMA tiles: at B = 512 that is 256 workgroups, and in
//          f32 mode the k-ascending chain of 16x16x4 is 784 MFMAs x 32 cycles (the 32x32x2 chain would be 1568 x 64)
template <int L> struct CnnGeo;
template <> struct CnnGeo<0> { static constexpr int IH = 84, IW = 84, IC = 4, OH = 20, OW = 20, OC = 32, KH = 8, KW = 8, S = 4, MT = 32, WM = 4, WN = 1, R = 4; };
template <> struct CnnGeo<1> { static constexpr int IH = 20, IW = 20, IC = 32, OH = 9, OW = 9, OC = 64, KH = 4, KW = 4, S = 2, MT = 32, WM = 2, WN = 2, R = 4; };
template <> struct CnnGeo<2> { static constexpr int IH = 9, IW = 9, IC = 64, OH = 7, OW = 7, OC = 64, KH = 3, KW = 3, S = 1, MT = 32, WM = 2, WN = 2, R = 3; };
template <> struct CnnGeo<3> { static constexpr int IH = 1, IW = 1, IC = 3136, OH = 1, OW = 1, OC = 512, KH = 1, KW = 1, S = 1, MT = 16, WM = 2, WN = 2, R = 7; };

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
template <typename TI, typename TC, int L>
__global__ void __launch_bounds__(256)
k_cnn_layer(int M, const TI *__restrict__ in, const TC *__restrict__ wt, const float *__restrict__ bias, TC *__restrict__ out) {
    typedef CnnGeo<L> G;
    constexpr int MT = G::MT, WM = G::WM, WN = G::WN, R = G::R;
    constexpr int BM = MT * WM, BN = MT * WN, K = G::KH * G::KW * G::IC, ROWLEN = G::KW * G::IC, NCH = K / KC, NIT = NCH / R;
    constexpr int LS = KC + (sizeof(TC) == 2 ? 8 : 4);
    constexpr int LBUF = L == 0 ? 1 : 2;
    constexpr int APT = BM * 8 / 256, BPT = BN * 8 / 256;
    static_assert(WM * WN == 4 && K % KC == 0 && NCH % R == 0 && G::OC % BN == 0 && ROWLEN % 8 == 0 && APT >= 1 && BPT >= 1, "tile shape");
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
        ap[u] = in + ((long long)(b * G::IH + oh * G::S) * G::IW + ow * G::S) * G::IC;
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
    typedef float accv __attribute__((ext_vector_type(MT == 32 ? 16 : 4)));
    accv acc;
#pragma unroll
    for (int r = 0; r < (MT == 32 ? 16 : 4); ++r) acc[r] = 0.0f;
    const int hi = lane / MT, c = lane % MT;                                           // MT = 32: half h; MT = 16: k-group g
    auto multiply = [&](int buf) {
        const TC *ar = lA + buf * (BM * LS) + (MT * wm + c) * LS, *br = lB + buf * (BN * LS) + (MT * wn + c) * LS;
        if constexpr (sizeof(TC) == 2 && MT == 32) {
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8c *>(ar + 16 * ks + 8 * hi),
                                                              *reinterpret_cast<const bf16x8c *>(br + 16 * ks + 8 * hi), acc, 0, 0, 0);
        } else if constexpr (sizeof(TC) == 2) {
#pragma unroll
            for (int ks = 0; ks < KC / 32; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8c *>(ar + 32 * ks + 8 * hi),
                                                              *reinterpret_cast<const bf16x8c *>(br + 32 * ks + 8 * hi), acc, 0, 0, 0);
        } else if constexpr (MT == 32) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {                                               // k = 2 (4q + i) + h: ascending chain
                const float4 a = *reinterpret_cast<const float4 *>(ar + 32 * hi + 4 * q), b = *reinterpret_cast<const float4 *>(br + 32 * hi + 4 * q);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                               // k = 4 (4q + i) + g: ascending chain
                const float4 a = *reinterpret_cast<const float4 *>(ar + 16 * hi + 4 * q), b = *reinterpret_cast<const float4 *>(br + 16 * hi + 4 * q);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
            }
        }
    };
#pragma unroll
    for (int r = 0; r < R; ++r) request(r, r);
    if constexpr (sizeof(TI) == 1 && sizeof(TC) == 4) __syncthreads();                 // the table, before the first commit reads it
#pragma unroll                      // fully: a rolled loop turns the ring into register copies that wait for the loads just issued
    for (int it = 0; it < NIT - 1; ++it) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int kc = it * R + r, buf = LBUF == 2 ? (kc & 1) : 0;
            commit(r, buf);
            request(r, kc + R);
            __syncthreads();
            multiply(buf);
            if constexpr (LBUF == 1) __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int kc = (NIT - 1) * R + r, buf = LBUF == 2 ? (kc & 1) : 0;
        commit(r, buf);
        __syncthreads();
        multiply(buf);
        if constexpr (LBUF == 1) __syncthreads();
    }
    const int n = n0 + MT * wn + c;
    const float bv = bias[n];
#pragma unroll
    for (int r = 0; r < (MT == 32 ? 16 : 4); ++r) {
        const int mm = m0 + MT * wm + (MT == 32 ? (r & 3) + 8 * (r >> 2) + 4 * hi : 4 * hi + r);
        if (mm < M) {
            float v = acc[r] + bv;
            v = v > 0.0f ? v : 0.0f;
            out[(long long)mm * G::OC + n] = (TC)v;
        }
    }
}

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

// flat f32 parameters -> transposed compute-type shadow [N][K]
template <typename TC>
__global__ void __launch_bounds__(256)
k_cnn_pack(const float *__restrict__ w, int K, int N, float div, TC *__restrict__ wt) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)K * N) return;
    const int n = (int)(t / K), k = (int)(t - (long long)n * K);
    const float v = w[(long long)k * N + n];
    wt[t] = (TC)(div != 1.0f ? __fdiv_rn(v, div) : v);
}

// ------------------------------------------------------------------------------------ C ABI
#define CNN_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return dqn_set_error(DQN_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); } while (0)
#define CNN_REQ(cond, msg) do { if (!(cond)) return dqn_set_error(DQN_ERR_INVALID, msg); } while (0)

struct CnnLayer { int K, N; long long o_w, o_b; };     // offsets into the flat parameter vector
struct dqn_cnn_handle {
    int A = 0, max_batch = 0; bool bf16 = false;
    CnnLayer L[4]; long long o_wv = 0, o_bv = 0, o_wa = 0, o_ba = 0, P = 0;
    void *arena = nullptr;
    float *params[2] = {nullptr, nullptr};             // online, target (flat f32, HWIO leaf order)
    void *wt[2][4] = {{nullptr}};                      // transposed shadows of the four GEMM layers
    float *wh[2] = {nullptr, nullptr}, *bh[2] = {nullptr, nullptr};   // heads, output-major [16][512] (0 = val, 1.. = adv), biases [1 + A]
    void *act[4] = {nullptr};                          // layer outputs
    float *q[3] = {nullptr, nullptr, nullptr};         // Q of the three passes of compute_q_targets
    float *scratch = nullptr;
};

struct LayerShape { int K, OC, positions; };            // host view of CnnGeo<l>: K = KH*KW*IC, output positions per frame stack
template <int L> static LayerShape shape_of() { typedef CnnGeo<L> G; return LayerShape{G::KH * G::KW * G::IC, G::OC, G::OH * G::OW}; }
static LayerShape cnn_shape(int layer) {
    switch (layer) { case 0: return shape_of<0>(); case 1: return shape_of<1>(); case 2: return shape_of<2>(); default: return shape_of<3>(); }
}

extern "C" int dqn_cnn_create(int32_t num_actions, int32_t max_batch, int32_t precision, dqn_cnn_handle **out) {
    CNN_REQ(out && num_actions >= 1 && num_actions <= 15 && max_batch >= 1 && max_batch <= (1 << 16), "dqn_cnn_create: bad argument");
    CNN_REQ(precision == DQN_PREC_F32 || precision == DQN_PREC_BF16, "unknown precision");
    dqn_cnn_handle *h = new (std::nothrow) dqn_cnn_handle();
    if (!h) return dqn_set_error(DQN_ERR_NOMEM, "host allocation failed");
    h->A = num_actions; h->max_batch = max_batch; h->bf16 = precision == DQN_PREC_BF16;
    long long p = 0;
    for (int l = 0; l < 4; ++l) {
        const LayerShape g = cnn_shape(l);
        h->L[l] = CnnLayer{g.K, g.OC, p, p + (long long)g.K * g.OC};
        p += (long long)g.K * g.OC + g.OC;
    }
    h->o_wv = p; p += 512; h->o_bv = p; p += 1; h->o_wa = p; p += 512ll * h->A; h->o_ba = p; p += h->A; h->P = p;
    const size_t esz = h->bf16 ? 2 : 4;
    size_t total = 0;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t sz_params = al(h->P * 4), sz_wt[4], sz_act[4], sz_wh = al(16 * CNN_F * 4), sz_bh = al((h->A + 1) * 4), sz_q = al((size_t)max_batch * h->A * 4);
    for (int l = 0; l < 4; ++l) { sz_wt[l] = al((size_t)h->L[l].K * h->L[l].N * esz); sz_act[l] = al((size_t)max_batch * cnn_shape(l).positions * h->L[l].N * esz); }
    total = 2 * sz_params + 2 * (sz_wt[0] + sz_wt[1] + sz_wt[2] + sz_wt[3]) + 2 * (sz_wh + sz_bh) + sz_act[0] + sz_act[1] + sz_act[2] + sz_act[3] + 3 * sz_q + al((size_t)max_batch * 4);
    hipError_t e = hipMalloc(&h->arena, total);
    if (e != hipSuccess) { delete h; return dqn_set_error(DQN_ERR_NOMEM, (std::string("hipMalloc: ") + hipGetErrorString(e)).c_str()); }
    char *c = (char *)h->arena;
    for (int w = 0; w < 2; ++w) {
        h->params[w] = (float *)c; c += sz_params;
        for (int l = 0; l < 4; ++l) { h->wt[w][l] = c; c += sz_wt[l]; }
        h->wh[w] = (float *)c; c += sz_wh; h->bh[w] = (float *)c; c += sz_bh;
    }
    for (int l = 0; l < 4; ++l) { h->act[l] = c; c += sz_act[l]; }
    for (int i = 0; i < 3; ++i) { h->q[i] = (float *)c; c += sz_q; }
    h->scratch = (float *)c;
    (void)hipMemset(h->arena, 0, total);
    *out = h;
    return DQN_OK;
}

extern "C" int dqn_cnn_destroy(dqn_cnn_handle *h) {
    if (!h) return DQN_OK;
    (void)hipDeviceSynchronize();
    if (h->arena) (void)hipFree(h->arena);
    delete h;
    return DQN_OK;
}

extern "C" int dqn_cnn_param_count(const dqn_cnn_handle *h, int64_t *n) {
    CNN_REQ(h && n, "null argument");
    *n = h->P;
    return DQN_OK;
}

__global__ void __launch_bounds__(256)
k_cnn_pack_head(const float *__restrict__ P, long long o_wv, long long o_bv, long long o_wa, long long o_ba, int A, float *wh, float *bh) {
    const int t = blockIdx.x * 256 + threadIdx.x;                  // grid covers 16 * 512
    const int j = t / CNN_F, k = t - j * CNN_F;
    wh[t] = j == 0 ? P[o_wv + k] : (j <= A ? P[o_wa + (long long)k * A + (j - 1)] : 0.0f);
    if (t <= A) bh[t] = t == 0 ? P[o_bv] : P[o_ba + t - 1];
}

// which: DQN_NET_ONLINE / DQN_NET_TARGET. Flat f32: conv1 w[8,8,4,32] b[32] conv2 w[4,4,32,64] b[64] conv3 w[3,3,64,64] b[64]
// fc w[3136,512] b[512] val w[512,1] b[1] adv w[512,A] b[A]  (HWIO; fc rows in [7][7][64] order)
extern "C" int dqn_cnn_set_params(dqn_cnn_handle *h, int which, const float *src, int src_is_host, void *stream) {
    CNN_REQ(h && src && (which == DQN_NET_ONLINE || which == DQN_NET_TARGET), "bad argument");
    hipStream_t s = (hipStream_t)stream;
    CNN_TRY(hipMemcpyAsync(h->params[which], src, h->P * 4, src_is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < 4; ++l) {
        const long long n = (long long)h->L[l].K * h->L[l].N;
        const int blocks = (int)((n + 255) / 256);
        if (h->bf16) hipLaunchKernelGGL((k_cnn_pack<__bf16>), dim3(blocks), dim3(256), 0, s, h->params[which] + h->L[l].o_w, h->L[l].K, h->L[l].N, l == 0 ? 255.0f : 1.0f, (__bf16 *)h->wt[which][l]);
        else hipLaunchKernelGGL((k_cnn_pack<float>), dim3(blocks), dim3(256), 0, s, h->params[which] + h->L[l].o_w, h->L[l].K, h->L[l].N, 1.0f, (float *)h->wt[which][l]);
    }
    hipLaunchKernelGGL(k_cnn_pack_head, dim3(16 * CNN_F / 256), dim3(256), 0, s, h->params[which], h->o_wv, h->o_bv, h->o_wa, h->o_ba, h->A, h->wh[which], h->bh[which]);
    CNN_TRY(hipGetLastError());
    if (src_is_host) CNN_TRY(hipStreamSynchronize(s));
    return DQN_OK;
}

template <typename TI, typename TC, int L>
static void launch_layer(hipStream_t s, int B, const TI *in, const TC *wt, const float *bias, TC *out) {
    typedef CnnGeo<L> G;
    const int M = B * G::OH * G::OW, BM = G::MT * G::WM, BN = G::MT * G::WN;
    DQN_LAUNCH((k_cnn_layer<TI, TC, L>), dim3((unsigned)((M + BM - 1) / BM * (G::OC / BN))), dim3(256), 0, s, M, in, wt, bias, out);
}

template <typename TC>
static void cnn_forward_t(dqn_cnn_handle *h, int which, const uint8_t *frames, int B, float *q, hipStream_t s) {
    const float *P = h->params[which];
    TC *a0 = (TC *)h->act[0], *a1 = (TC *)h->act[1], *a2 = (TC *)h->act[2], *a3 = (TC *)h->act[3];
    launch_layer<uint8_t, TC, 0>(s, B, frames, (const TC *)h->wt[which][0], P + h->L[0].o_b, a0);
    launch_layer<TC, TC, 1>(s, B, a0, (const TC *)h->wt[which][1], P + h->L[1].o_b, a1);
    launch_layer<TC, TC, 2>(s, B, a1, (const TC *)h->wt[which][2], P + h->L[2].o_b, a2);
    launch_layer<TC, TC, 3>(s, B, a2, (const TC *)h->wt[which][3], P + h->L[3].o_b, a3);
    hipLaunchKernelGGL((k_cnn_head<TC>), dim3((B + 15) / 16), dim3(256), 0, s, a3, h->wh[which], h->bh[which], h->A, B, q);
}

/* Q[B][A] of the Nature-CNN dueling net for B stacks of four 84x84 u8 frames (NHWC). */
extern "C" int dqn_cnn_forward(dqn_cnn_handle *h, int which, const uint8_t *frames, int32_t B, float *q, void *stream) {
    CNN_REQ(h && frames && q && (which == DQN_NET_ONLINE || which == DQN_NET_TARGET), "bad argument");
    CNN_REQ(B >= 1 && B <= h->max_batch, "B exceeds max_batch");
    if (h->bf16) cnn_forward_t<__bf16>(h, which, frames, B, q, (hipStream_t)stream);
    else cnn_forward_t<float>(h, which, frames, B, q, (hipStream_t)stream);
    CNN_TRY(hipGetLastError());
    return DQN_OK;
}

/* compute_q_targets (q_learning_functions.py:42-64) with the CNN as the model: three forwards + the TD rule (k_td). */
extern "C" int dqn_cnn_q_targets(dqn_cnn_handle *h, const uint8_t *s, const int32_t *a, const float *r, const uint8_t *s2,
                                 const float *d, float gamma, int32_t B, float *targets, void *stream) {
    CNN_REQ(h && s && a && r && s2 && d && targets, "null argument");
    CNN_REQ(B >= 1 && B <= h->max_batch, "B exceeds max_batch");
    hipStream_t st = (hipStream_t)stream;
    int rc = dqn_cnn_forward(h, DQN_NET_ONLINE, s, B, h->q[0], stream); if (rc) return rc;      // :52
    rc = dqn_cnn_forward(h, DQN_NET_ONLINE, s2, B, h->q[1], stream); if (rc) return rc;          // :53
    rc = dqn_cnn_forward(h, DQN_NET_TARGET, s2, B, h->q[2], stream); if (rc) return rc;          // :54
    launch_td(st, h->q[0], h->q[1], h->q[2], a, r, d, nullptr, gamma, B, h->A, targets, nullptr, nullptr, nullptr, h->scratch);
    CNN_TRY(hipGetLastError());
    return DQN_OK;
}
