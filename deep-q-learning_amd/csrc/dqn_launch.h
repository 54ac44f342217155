// csrc/dqn_launch.h -- host-side launcher prototypes shared by dqn_api.hip and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dqn_device.h"

#include <hip/hip_ext.h>

int dqn_set_error(int code, const char *msg);   // dqn_api.hip: message returned by dqn_last_error()

struct EnvArgs;   // dqn_per_device.h
struct SampleArgs;

// Kernel launch used by the hot-path launchers. In profiling mode (dqn_profile_begin) the API arms a pair of
// events before each launch; the launch is then issued with hipExtLaunchKernelGGL so that the two events carry the
// dispatch's own start / stop timestamps (what rocprofv3 reports), not host-side bracketing.
extern thread_local hipEvent_t g_prof_ev0, g_prof_ev1;
#define DQN_LAUNCH(kernel, grid, block, lds, stream, ...)                                                    \
    do {                                                                                                      \
        if (g_prof_ev0) {                                                                                     \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, g_prof_ev0, g_prof_ev1, 0, __VA_ARGS__); \
            g_prof_ev0 = nullptr; g_prof_ev1 = nullptr;                                                       \
        } else {                                                                                              \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                \
        }                                                                                                     \
    } while (0)

// ----- network geometry -----------------------------------------------------------------
// Flat parameter layout (haiku leaf order, w is [in,out] row-major; LunarLander/dddqn.py:19-22):
//   w1[D*H1] b1[H1] w2[H1*H2] b2[H2] wv[H2] bv[1] wa[H2*A] ba[A]
struct NetDims {
    int D, H1, H2, A;
    int KQ1;            // ceil(D/16): 16-wide k-blocks of layer 1
    long long o_w1, o_b1, o_w2, o_b2, o_wv, o_bv, o_wa, o_ba, P;
    // fragment-packed shadows of the weights, as offsets (floats) into one pack buffer:
    long long p_w1, p_w2, p_wh, p_w2t, p_wht, pack_floats;
    // k-packed W2 for the actor kernel (dqn_actor.hip): w2k[(kq*H2 + c)*4 + j] = W2[4*kq + j][c], so that a lane's
    // 16-B load is four consecutive k of its own column; lives behind the five fragment-ordered shadows
    long long p_w2k;
};
NetDims make_dims(int D, int H1, int H2, int A);

// Per-call description of one forward pass handled by k_qnet_fwd
struct FwdPass {
    const float *x;        // [B, D] row-major input rows
    const float *params;   // flat params (biases are read from here)
    const float *pack;     // fragment-packed weights of the same net
    float *q;              // [B, A] out (may be NULL)
    float *feat;           // [B, H2] row-major out (may be NULL)
    float *px, *ph1, *ph2; // batch-major packed stashes for backward (may be NULL)
    int src;               // with fused sampling: 1 = ring states of the sampled rows, 2 = their observations
    // optional fused epsilon-greedy policy (q_agent.py:137-141) on the rows of this pass
    int32_t *act_out;            // [B] or NULL
    const DqnState *act_state;   // non-NULL: epsilon / counter from the device state
    float act_eps; unsigned long long act_seed, act_ctr;
};

struct BwdArgs {
    // inputs
    const float *q, *nq, *nt;          // [B,A] (fused mode) ; q is also "pred"
    const float *targets;              // [B,A] (parity mode) or NULL
    const int32_t *a; const float *r; const uint8_t *d_u8; const float *d_f32;
    const float *w_raw;                // raw IS weights (fused: normalised in-kernel) or NULL
    const float *isw;                  // already-normalised weights (parity mode) or NULL
    float gamma;
    const float *ph1, *ph2;            // packed post-ReLU activations of the s-pass
    const float *pack;                 // online net packs (p_w2t, p_wht used)
    // outputs
    float *pdz1, *pdz2, *pdz3;         // packed row-gradients for k_dw
    float *td, *td_abs, *dq, *targets_out, *isw_out;   // optional [B] / [B,A]
    float *loss_part;                  // [ceil(B/16)]
};

struct FuseBwd { BwdArgs g; int *tile_cnt; DqnState *st; int tiles; int withhold; };   // withhold (diagnostic, dqn_debug_withhold_handover): the partner passes do not count themselves in   // tile_cnt: [tiles] arrival counters + [tiles] consumed counts   // row backward fused into the forward launch (k_qnet_fwd<.., FUSE>)

struct PwArgs {             // sorted PER write-back run by surplus workgroups of k_dw when tree != NULL
    float *tree; long long N; int L; const int32_t *idx; const float *td_abs; int B; float alpha, eps;
};
struct AdamArgs {           // optimizer applied in the dW epilogue (single-GPU fused path) when P != NULL
    float *P, *mu, *nu, *pack; int adamw; float b1, b2, eps, wd, grad_scale;
    float *pack_act;        // bf16 mode: f32 shadows read by the actor kernel (k-packed W2, heads), else NULL
};

void launch_pack(hipStream_t s, const NetDims &m, const float *params, float *pack);
void launch_pack_w2k16(hipStream_t s, const NetDims &m, const float *params, float *pack_act);   // bf16 mode (dqn_net_bf16.hip)
// fuse != NULL (f32, three passes, 3 * ceil(B/16) <= 256, batch weights final before the launch): the pass-0 workgroups
// also run the row backward of their tiles (k_bwd_rows' work); tile_cnt = ceil(B/16) zeroed counters
void launch_qnet_fwd(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, const SampleArgs *smp = nullptr,
                     const BwdArgs *fuse = nullptr, int *tile_cnt = nullptr, DqnState *st = nullptr, int tile_stride = 0, int withhold = 0);
// T vector env steps of n envs in one launch (+ leaf insert, + presampling of the next PER batch); dqn_actor.hip.
// num_cus: the device's CU count -- tree, sampler and actor workgroups wait for each other inside the launch, so the grid
// is sized to be resident as a whole; no_wide: never the 16-env small-net kernel (diagnostic). Returns whether the PER
// batch of B rows was drawn by the launch (false: B == 0, no tree, or no room for sampler workgroups on this device --
// the update that follows then samples for itself).
bool actor_multi_supported(const NetDims &m, int n_envs, int T);
bool launch_actor_multi(hipStream_t s, const NetDims &m, const EnvArgs &env, int T, const float *params, const float *pack,
                        int32_t *act_out, int B, const SampleArgs *smp, bool bf16, int num_cus, bool no_wide);
void launch_td(hipStream_t s, const float *q, const float *nq, const float *nt, const int32_t *a,
               const float *r, const float *d, const float *isw, float gamma, int B, int A,
               float *targets, float *td, float *dq, float *loss, float *scratch);
void launch_loss(hipStream_t s, const float *pred, const float *targets, const float *isw, int B, int A,
                 float *loss);
void launch_bwd_rows(hipStream_t s, const NetDims &m, const BwdArgs &g, int B, DqnState *st);
void launch_dw(hipStream_t s, const NetDims &m, const float *px, const float *ph1, const float *ph2,
               const float *pdz1, const float *pdz2, const float *pdz3, int B, float *grad,
               const float *loss_part, float *loss_out, DqnState *st, int bump_ctr, const AdamArgs &adam,
               const PwArgs &pw = PwArgs{});
void launch_per_top(hipStream_t st_, DqnState *st, float *tree, int L);
void launch_adam(hipStream_t s, const NetDims &m, DqnState *st, float *params, const float *grad, float *mu,
                 float *nu, float *pack, int adamw, float b1, float b2, float eps, float wd, float grad_scale);
void launch_policy(hipStream_t s, const float *q, int n, int A, float epsilon, unsigned long long seed,
                   unsigned long long ctr, int32_t *actions, const DqnState *st_from);
void launch_u8_to_f32(hipStream_t s, const uint8_t *in, float *out, int n);

// ----- large-batch f32 kernels (dqn_net_big.hip): 64 rows per workgroup on v_mfma_f32_32x32x2_f32 -------------
#define DQN_BIG_MIN 16384           // batch rows from which the 64-row kernels replace the 16-row ones (measured crossover: tools/sweep.py)
#define DQN_BIG_MIN_BF16 8192       // the same crossover in the bf16 mode (r03 sweep: update 96 vs 120 us at 8 192 rows, 96 vs 75 at 4 096)
#define DQN_BIG_MAX_SLICES 64       // batch slices of the split-K weight-gradient GEMM (slab size)
bool big_supported(const NetDims &m, int B, bool any_size = false);   // any_size: from 64 rows (DQN_FLAG_BIG_ROWS)
size_t big_slab_floats(int max_batch, int num_cus);
size_t big_colsum_floats(int max_batch);
int big_dw_slices(int B, int num_cus);
// RCCL through the one dlopen'd library of dqn_api.hip (used by the CNN handle's communicator, dqn_cnn.hip)
int dqn_rccl_comm_init(void **comm, const void *unique_id_128, int rank, int world);
int dqn_rccl_allreduce_sum_f32(void *comm, float *buf, size_t n, hipStream_t st);
void dqn_rccl_comm_destroy(void *comm);
int dqn_rccl_comm_count(void *comm, int *n);
void launch_big_forward(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, int num_cus);
// the same three entry points in the bf16 precision mode (dqn_net_big16.hip: v_mfma_f32_32x32x16_bf16, k-packed bf16 stashes)
void launch_big16_forward(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, int num_cus);
void launch_big16_rows_bwd(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, const BwdArgs &bw,
                           float *px, float *ph1, float *ph2, float *colsum, DqnState *st, int num_cus);
void launch_big16_dw(hipStream_t s, const NetDims &m, const float *px, const float *ph1, const float *ph2, const float *pdz1,
                     const float *pdz2, const float *pdz3, int B, float *slab, const float *colsum, float *grad,
                     const float *loss_part, float *loss_out, DqnState *st, int bump_ctr, const AdamArgs &adam, int num_cus);
void launch_big_rows_bwd(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, const BwdArgs &bw,
                         float *px, float *ph1, float *ph2, float *colsum, DqnState *st, int num_cus);
void launch_big_dw(hipStream_t s, const NetDims &m, const float *px, const float *ph1, const float *ph2, const float *pdz1,
                   const float *pdz2, const float *pdz3, int B, float *slab, const float *colsum, float *grad,
                   const float *loss_part, float *loss_out, DqnState *st, int bump_ctr, const AdamArgs &adam, int num_cus);

// ----- replay / PER ---------------------------------------------------------------------
void launch_replay_add(hipStream_t st_, DqnState *st, float *states, int32_t *actions, float *rewards,
                       float *observations, uint8_t *dones, long long N, int D, const float *s,
                       const int32_t *a, const float *r, const float *s2, const uint8_t *d, int n,
                       float *s_advance, int bump_env);
void launch_sample_uniform(hipStream_t st_, const DqnState *st, const float *states, const int32_t *actions,
                           const float *rewards, const float *observations, const uint8_t *dones, int D,
                           int B, unsigned long long seed, unsigned long long ctr, int from_state,
                           const int32_t *idx_in, float *s, int32_t *a, float *r, float *s2, uint8_t *d,
                           int32_t *idx_out);
void launch_per_sample(hipStream_t st_, const DqnState *st, const float *tree, long long N, int L,
                       const float *states, const int32_t *actions, const float *rewards,
                       const float *observations, const uint8_t *dones, int D, int B, float beta,
                       unsigned long long seed, unsigned long long ctr, int from_state,
                       float *s, int32_t *a, float *r, float *s2, uint8_t *d, int32_t *idx, float *w_raw,
                       unsigned int *wmax_bits, int num_cus, bool nt_out);
// pw_part: 8 192 floats of scratch for the segment kernel (k_per_write_seg, large batches); force: 0 = by batch size,
// 1 = always the wave-per-chunk kernel, 2 = always the segment kernel (tests)
void launch_per_write_sorted(hipStream_t st_, DqnState *st, float *tree, long long N, int L, const int32_t *idx,
                             const float *val, int B, int mode, float alpha, float eps, float *pw_part = nullptr, int force = 0);
void launch_per_add(hipStream_t st_, const DqnState *st, float *tree, long long Nt, int L, int n, long long cap, int advance = 0, long long zero_first = 0, int zero_n = 0);
void launch_isw_normalize(hipStream_t st_, const float *w_raw, int B, float *isw, DqnState *st, const unsigned int *wmax_bits);
void launch_per_write(hipStream_t st_, DqnState *st, float *tree, unsigned long long *stamp, long long N,
                      int L, const int32_t *idx, const float *val, int B, int mode, float alpha, float eps,
                      long long ring_capacity);

// ----- bf16 MFMA variants (dqn_net_bf16.hip); pointer fields typed float* carry bf16 data -------------
long long bf16_pack_elems(const NetDims &m);
void launch_pack_bf16(hipStream_t s, const NetDims &m, const float *params, float *pack);
void launch_qnet_fwd_bf16(hipStream_t s, const NetDims &m, const FwdPass *passes, int npass, int B, const SampleArgs *smp = nullptr,
                          const BwdArgs *fuse = nullptr, int *tile_cnt = nullptr, DqnState *st = nullptr, int tile_stride = 0, int withhold = 0);
void launch_bwd_rows_bf16(hipStream_t s, const NetDims &m, const BwdArgs &g, int B, DqnState *st);
void launch_dw_bf16(hipStream_t s, const NetDims &m, const float *px, const float *ph1, const float *ph2,
                    const float *pdz1, const float *pdz2, const float *pdz3, int B, float *grad,
                    const float *loss_part, float *loss_out, DqnState *st, int bump_ctr, const AdamArgs &adam,
                    const PwArgs &pw);
void launch_adam_bf16(hipStream_t s, const NetDims &m, DqnState *st, float *params, const float *grad, float *mu,
                      float *nu, float *pack, int adamw, float b1, float b2, float eps, float wd, float grad_scale,
                      float *pack_act);
