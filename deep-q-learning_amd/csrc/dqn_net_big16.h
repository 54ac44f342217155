// csrc/dqn_net_big16.h -- shared by the 64-row bf16 kernels: the forward kernel + the weight-gradient kernel
// (dqn_net_big16.hip) and the update / gradient kernel (dqn_net_big16_bwd.hip). See dqn_net_big16.hip for the layouts.
#pragma once
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

#ifndef BIG16_PF
#define BIG16_PF 4                      // k-blocks (32 deep) of the 256-deep layers held ahead in registers (8 = the whole layer)
#endif
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
    // the transposed product: acc[rt][ct] (+)= (A . W)^T -- the same fragments with the MFMA's operands swapped (the weight
    // fragment as the A operand: lane (r, h) holds W[k = 8h + j][col r] = A-operand element [row r][k]); lane = batch row of
    // the 32-row tile, register r = output column (r&3) + 8 (r>>2) + 4h of the 32-column tile
    __device__ __forceinline__ void run_t(const __bf16 *la, int S, int lane, f32x16 (&acc)[2][2]) {
        const int h = lane >> 5, c = lane & 31;
        const __bf16 *arow0 = la + c * S + 8 * h, *arow1 = arow0 + 32 * S;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const int p = kq % PF;
            const bf16x8 a00 = *reinterpret_cast<const bf16x8 *>(arow0 + 32 * kq), a10 = *reinterpret_cast<const bf16x8 *>(arow1 + 32 * kq);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                acc[0][ct] = MFMA32B(blo[p][ct], a00, acc[0][ct]);
                acc[1][ct] = MFMA32B(blo[p][ct], a10, acc[1][ct]);
            }
            if constexpr (!ONE_STEP) {
                const bf16x8 a01 = *reinterpret_cast<const bf16x8 *>(arow0 + 32 * kq + 16), a11 = *reinterpret_cast<const bf16x8 *>(arow1 + 32 * kq + 16);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc[0][ct] = MFMA32B(bhi[p][ct], a01, acc[0][ct]);
                    acc[1][ct] = MFMA32B(bhi[p][ct], a11, acc[1][ct]);
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

// ---- epilogue arithmetic on PACKED bf16 pairs: one v_cvt_pk_bf16_f32 per two accumulator elements, ReLU as a packed signed
// 16-bit max with 0 on the bf16 bit patterns (negative bf16 <=> negative int16; relu(round(x)) == round(relu(x))). Measured
// r03: with one instruction stream per element (add bias, max, convert, store, gate bit) the kernel spent 43 % of its cycles
// issuing non-MFMA instructions -- 6.7 K issue cycles per tile and pass against 2.3 K of MFMA.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2));
}
__device__ __forceinline__ unsigned relu_pack(float a, float b) {
    const s16x2 q = __builtin_bit_cast(s16x2, __builtin_convertvector(f32x2v{a, b}, bf16x2));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(q, s16x2{0, 0}));
}

