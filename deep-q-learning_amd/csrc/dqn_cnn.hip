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

struct ConvGeom {
    int IH, IW, IC, OH, OW, OC, KH, KW, S;
    int K, rowlen;            // K = KH*KW*IC, rowlen = KW*IC (contiguous elements of a patch row)
    long long M;              // B*OH*OW
    float in_scale;           // u8 input: divide by this (255); else unused
};

constexpr int KC = 64;        // k-chunk (a multiple of 8: an 8-element piece never crosses a patch row)
template <typename T> struct Pad { static constexpr int v = 8; };            // LDS row = KC + pad elements

// u8 pixels: v = (float)u8 / 255.0f, read from a 256-entry table of exactly those quotients (a division per element would
// be ten instructions on 52 M elements per forward)
__device__ __forceinline__ void ld8(const uint8_t *p, const float *lut, float (&v)[8]) {
    const uint2 raw = *reinterpret_cast<const uint2 *>(p);
    const uint32_t w[2] = {raw.x, raw.y};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = lut[(w[j >> 2] >> (8 * (j & 3))) & 0xffu];
}
__device__ __forceinline__ void ld8(const float *p, const float *, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const __bf16 *p, const float *, float (&v)[8]) {
    const bf16x8c a = *reinterpret_cast<const bf16x8c *>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
}
__device__ __forceinline__ void st8(float *q, const float (&v)[8]) {
    *reinterpret_cast<float4 *>(q) = float4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<float4 *>(q + 4) = float4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void st8(__bf16 *q, const float (&v)[8]) {
    bf16x8c a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8c *>(q) = a;
}

// TI: element type of the input tensor (uint8_t frames, or the compute type); TC: compute / weight / output type.
// Workgroup tile: BM = 32 WM rows x BN = 32 WN columns (WM * WN = 4 waves, one 32 x 32 MFMA tile each). The pieces of chunk
// kc + 1 are requested into registers before chunk kc is multiplied and written to LDS behind the barrier that follows it
// (one chunk of global latency hidden per chunk of MFMAs; several workgroups per CU hide the rest).
template <typename TI, typename TC, int WM, int WN>
__global__ void __launch_bounds__(256)
k_igemm(ConvGeom g, const TI *__restrict__ in, const TC *__restrict__ wt, const float *__restrict__ bias, TC *__restrict__ out, int relu) {
    constexpr int BM = 32 * WM, BN = 32 * WN, LS = KC + Pad<TC>::v, PPR = KC / 8;      // pieces per row
    __shared__ __attribute__((aligned(16))) TC lA[BM * LS];
    __shared__ __attribute__((aligned(16))) TC lB[BN * LS];
    __shared__ float lut[sizeof(TI) == 1 ? 256 : 1];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    if constexpr (sizeof(TI) == 1) { lut[tid] = __fdiv_rn((float)tid, g.in_scale); __syncthreads(); }
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    // this thread's share of the images: 8-element pieces, piece q = (row q / PPR, k-offset 8 (q % PPR))
    constexpr int APT = BM * PPR / 256, BPT = (BN * PPR + 255) / 256;
    long long abase[APT]; bool aon[APT];
#pragma unroll
    for (int u = 0; u < APT; ++u) {
        const int q = tid + 256 * u, rl = q / PPR;
        const long long mm = m0 + rl;
        aon[u] = mm < g.M;
        const long long m2 = aon[u] ? mm : 0;
        const int ow = (int)(m2 % g.OW); const long long t2 = m2 / g.OW;
        const int oh = (int)(t2 % g.OH); const long long b = t2 / g.OH;
        abase[u] = ((b * g.IH + (long long)oh * g.S) * g.IW + (long long)ow * g.S) * g.IC;
    }
    float va[APT][8], vb[BPT][8];
    auto request = [&](int kc) {
        const int k0 = kc * KC;
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            const int q = tid + 256 * u, kp = k0 + 8 * (q % PPR);
            const int kh = kp / g.rowlen, rem = kp - kh * g.rowlen;
            if (aon[u]) ld8(in + abase[u] + (long long)kh * g.IW * g.IC + rem, lut, va[u]);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) va[u][j] = 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < BPT; ++u) {
            const int q = tid + 256 * u, nl = q / PPR, ko = 8 * (q % PPR);
            if (q < BN * PPR && n0 + nl < g.OC) ld8(wt + (long long)(n0 + nl) * g.K + k0 + ko, lut, vb[u]);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) vb[u][j] = 0.0f;
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < APT; ++u) { const int q = tid + 256 * u; st8(lA + (q / PPR) * LS + 8 * (q % PPR), va[u]); }
#pragma unroll
        for (int u = 0; u < BPT; ++u) { const int q = tid + 256 * u; if (q < BN * PPR) st8(lB + (q / PPR) * LS + 8 * (q % PPR), vb[u]); }
    };
    f32x16c acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nchunks = g.K / KC;
    request(0);
    for (int kc = 0; kc < nchunks; ++kc) {
        commit();
        __syncthreads();
        if (kc + 1 < nchunks) request(kc + 1);
        const TC *ar = lA + (32 * wm + c) * LS, *br = lB + (32 * wn + c) * LS;
        if constexpr (sizeof(TC) == 2) {
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                const bf16x8c a = *reinterpret_cast<const bf16x8c *>(ar + 16 * ks + 8 * h);
                const bf16x8c b = *reinterpret_cast<const bf16x8c *>(br + 16 * ks + 8 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < KC / 2; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * s + h], br[2 * s + h], acc, 0, 0, 0);   // k = 2s, then 2s + 1: ascending chain
        }
        __syncthreads();
    }
    const int n = n0 + 32 * wn + c;
    if (n < g.OC) {
        const float bv = bias[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long mm = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mm < g.M) {
                float v = acc[r] + bv;
                if (relu) v = v > 0.0f ? v : 0.0f;
                out[mm * g.OC + n] = (TC)v;
            }
        }
    }
}

// dueling head (dddqn.py:29-31) on the fc features [B][F]: one thread per row; val / adv are k-ascending fmaf chains
// (f32 weights [F][1 + A]: column 0 = val, 1.. = adv), Q = val + adv - mean(adv)
template <typename TC>
__global__ void __launch_bounds__(64)
k_cnn_head(const TC *__restrict__ feat, int F, const float *__restrict__ wh, const float *__restrict__ bh, int A, int B, float *q) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= B) return;
    float acc[16];
    for (int j = 0; j <= A; ++j) acc[j] = 0.0f;
    const TC *x = feat + (long long)i * F;
    for (int k = 0; k < F; ++k) {
        const float xv = (float)x[k];
        for (int j = 0; j <= A; ++j) acc[j] = fmaf(xv, wh[(long long)k * (A + 1) + j], acc[j]);
    }
    float sum = 0.0f;
    for (int j = 1; j <= A; ++j) { acc[j] = acc[j] + bh[j]; sum = sum + acc[j]; }
    const float v = acc[0] + bh[0], mean = __fdiv_rn(sum, (float)A);
    for (int j = 0; j < A; ++j) q[(long long)i * A + j] = (v + acc[1 + j]) - mean;
}

// flat f32 parameters -> transposed compute-type shadow [N][K]
template <typename TC>
__global__ void __launch_bounds__(256)
k_cnn_pack(const float *__restrict__ w, int K, int N, TC *__restrict__ wt) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)K * N) return;
    const int n = (int)(t / K), k = (int)(t - (long long)n * K);
    wt[t] = (TC)w[(long long)k * N + n];
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
    float *wh[2] = {nullptr, nullptr}, *bh[2] = {nullptr, nullptr};   // heads [512][1 + A], biases [1 + A]
    void *act[4] = {nullptr};                          // layer outputs
    float *q[3] = {nullptr, nullptr, nullptr};         // Q of the three passes of compute_q_targets
    float *scratch = nullptr;
};

static const int CNN_IH = 84, CNN_IC = 4;
static ConvGeom cnn_geom(int layer, long long B) {
    ConvGeom g{};
    switch (layer) {
    case 0: g = ConvGeom{84, 84, 4, 20, 20, 32, 8, 8, 4, 0, 0, 0, 255.0f}; break;
    case 1: g = ConvGeom{20, 20, 32, 9, 9, 64, 4, 4, 2, 0, 0, 0, 1.0f}; break;
    case 2: g = ConvGeom{9, 9, 64, 7, 7, 64, 3, 3, 1, 0, 0, 0, 1.0f}; break;
    default: g = ConvGeom{1, 1, 3136, 1, 1, 512, 1, 1, 1, 0, 0, 0, 1.0f}; break;
    }
    g.K = g.KH * g.KW * g.IC; g.rowlen = g.KW * g.IC; g.M = B * g.OH * g.OW;
    return g;
}

extern "C" int dqn_cnn_create(int32_t num_actions, int32_t max_batch, int32_t precision, dqn_cnn_handle **out) {
    CNN_REQ(out && num_actions >= 1 && num_actions <= 15 && max_batch >= 1 && max_batch <= (1 << 16), "dqn_cnn_create: bad argument");
    CNN_REQ(precision == DQN_PREC_F32 || precision == DQN_PREC_BF16, "unknown precision");
    dqn_cnn_handle *h = new (std::nothrow) dqn_cnn_handle();
    if (!h) return dqn_set_error(DQN_ERR_NOMEM, "host allocation failed");
    h->A = num_actions; h->max_batch = max_batch; h->bf16 = precision == DQN_PREC_BF16;
    long long p = 0;
    for (int l = 0; l < 4; ++l) {
        const ConvGeom g = cnn_geom(l, 1);
        h->L[l] = CnnLayer{g.K, g.OC, p, p + (long long)g.K * g.OC};
        p += (long long)g.K * g.OC + g.OC;
    }
    h->o_wv = p; p += 512; h->o_bv = p; p += 1; h->o_wa = p; p += 512ll * h->A; h->o_ba = p; p += h->A; h->P = p;
    const size_t esz = h->bf16 ? 2 : 4;
    size_t total = 0;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t sz_params = al(h->P * 4), sz_wt[4], sz_act[4], sz_wh = al(512 * (h->A + 1) * 4), sz_bh = al((h->A + 1) * 4), sz_q = al((size_t)max_batch * h->A * 4);
    for (int l = 0; l < 4; ++l) { sz_wt[l] = al((size_t)h->L[l].K * h->L[l].N * esz); sz_act[l] = al((size_t)cnn_geom(l, max_batch).M * h->L[l].N * esz); }
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
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < 512 * (A + 1)) {
        const int k = t / (A + 1), j = t - k * (A + 1);
        wh[t] = j == 0 ? P[o_wv + k] : P[o_wa + (long long)k * A + (j - 1)];
    }
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
        if (h->bf16) hipLaunchKernelGGL((k_cnn_pack<__bf16>), dim3(blocks), dim3(256), 0, s, h->params[which] + h->L[l].o_w, h->L[l].K, h->L[l].N, (__bf16 *)h->wt[which][l]);
        else hipLaunchKernelGGL((k_cnn_pack<float>), dim3(blocks), dim3(256), 0, s, h->params[which] + h->L[l].o_w, h->L[l].K, h->L[l].N, (float *)h->wt[which][l]);
    }
    hipLaunchKernelGGL(k_cnn_pack_head, dim3((512 * (h->A + 1) + 255) / 256), dim3(256), 0, s, h->params[which], h->o_wv, h->o_bv, h->o_wa, h->o_ba, h->A, h->wh[which], h->bh[which]);
    CNN_TRY(hipGetLastError());
    if (src_is_host) CNN_TRY(hipStreamSynchronize(s));
    return DQN_OK;
}

template <typename TI, typename TC, int WM, int WN>
static void launch_igemm(hipStream_t s, const ConvGeom &g, const TI *in, const TC *wt, const float *bias, TC *out, int relu) {
    const dim3 grid((unsigned)((g.M + 32 * WM - 1) / (32 * WM)), (unsigned)((g.OC + 32 * WN - 1) / (32 * WN)));
    DQN_LAUNCH((k_igemm<TI, TC, WM, WN>), grid, dim3(256), 0, s, g, in, wt, bias, out, relu);
}

template <typename TC>
static void cnn_forward_t(dqn_cnn_handle *h, int which, const uint8_t *frames, int B, float *q, hipStream_t s) {
    const float *P = h->params[which];
    TC *a0 = (TC *)h->act[0], *a1 = (TC *)h->act[1], *a2 = (TC *)h->act[2], *a3 = (TC *)h->act[3];
    launch_igemm<uint8_t, TC, 4, 1>(s, cnn_geom(0, B), frames, (const TC *)h->wt[which][0], P + h->L[0].o_b, a0, 1);
    launch_igemm<TC, TC, 2, 2>(s, cnn_geom(1, B), a0, (const TC *)h->wt[which][1], P + h->L[1].o_b, a1, 1);
    launch_igemm<TC, TC, 2, 2>(s, cnn_geom(2, B), a1, (const TC *)h->wt[which][2], P + h->L[2].o_b, a2, 1);
    launch_igemm<TC, TC, 2, 2>(s, cnn_geom(3, B), a2, (const TC *)h->wt[which][3], P + h->L[3].o_b, a3, 1);
    hipLaunchKernelGGL((k_cnn_head<TC>), dim3((B + 63) / 64), dim3(64), 0, s, a3, 512, h->wh[which], h->bh[which], h->A, B, q);
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
