/* include/dqn_hip.h -- C ABI of libdqn_hip.so: the MI355X (gfx950) DDDQN inner loop.
 *
 * The reference (hal9000universe/deep-q-learning) has no FFI/plugin boundary: its seam
 * is a set of Python function factories returning jitted closures, wired in
 * General/QLearning/q_agent.py:110-113 and called only from Agent._step (:146-169) and
 * Agent._policy (:137-141). Each entry point below names the reference function it
 * replaces; INTEGRATION.md shows the ctypes binding a maintainer would add there.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - every call returns 0 on success or a negative dqn_status; dqn_last_error() gives
 *     the message of the calling thread's last failure. No exceptions, no abort().
 *   - all pointers are DEVICE pointers unless the name ends in _host; buffers are
 *     caller-owned, contiguous, f32 / i32 / u8. The handle owns its own device state
 *     (parameters, optimizer moments, replay ring, sum-tree, workspaces).
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream. Work is only
 *     enqueued; nothing synchronises except the *_host getters.
 *   - one host thread per handle; one process per GPU.
 */
#ifndef DQN_HIP_H
#define DQN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DQN_ABI_VERSION 4

typedef enum {
    DQN_OK = 0,
    DQN_ERR_INVALID = -1,     /* bad argument / unsupported shape */
    DQN_ERR_HIP = -2,         /* HIP runtime error */
    DQN_ERR_NOMEM = -3,
    DQN_ERR_STATE = -4,       /* call sequence error (e.g. PER call on a uniform handle) */
    DQN_ERR_COMM = -5         /* RCCL error / RCCL not available */
} dqn_status;

typedef enum { DQN_OPT_ADAM = 0, DQN_OPT_ADAMW = 1 } dqn_optimizer;        /* optax.adam / optax.adamw */
typedef enum { DQN_PREC_F32 = 0, DQN_PREC_BF16 = 1 } dqn_precision;        /* exact f32 MFMA / bf16 MFMA */
typedef enum { DQN_NET_ONLINE = 0, DQN_NET_TARGET = 1 } dqn_net;
typedef enum {                                                              /* dqn_buffer() selectors */
    DQN_BUF_PARAMS = 0, DQN_BUF_TARGET = 1, DQN_BUF_MU = 2, DQN_BUF_NU = 3, DQN_BUF_GRAD = 4,
    DQN_BUF_TREE = 5, DQN_BUF_STATES = 6, DQN_BUF_ACTIONS = 7, DQN_BUF_REWARDS = 8,
    DQN_BUF_OBSERVATIONS = 9, DQN_BUF_DONES = 10,
    DQN_BUF_BATCH_IDX = 11, DQN_BUF_BATCH_ISW = 12, DQN_BUF_BATCH_TD = 13, DQN_BUF_LOSS = 14,
    DQN_BUF_ENV_OBS = 15, DQN_BUF_ENV_ACTIONS = 16
} dqn_buffer_id;

typedef enum { DQN_ENV_SYNTHETIC = 0, DQN_ENV_CARTPOLE = 1 } dqn_env_kind;   /* device-resident vector envs */

/* dqn_config.flags (diagnostics; 0 = the fast paths). */
typedef enum {
    DQN_FLAG_NO_HANDOVER = 1,      /* no workgroup of a launch ever waits for another: the actor launch carries no sampler
                                    * workgroups (the update draws its own PER batch) and the row backward is its own launch.
                                    * The library takes this path by itself where the device (CU count) cannot hold the
                                    * co-resident grids; the flag forces it (tests). Same results, bit for bit. */
    DQN_FLAG_NO_ACTOR16 = 2,       /* never the 16-env small-net actor kernel */
    DQN_FLAG_BF16_F32_ACTOR = 4,   /* bf16 mode with the exact-f32 actor chain */
    DQN_FLAG_BIG_ROWS = 8,         /* take the 64-row large-batch kernels (dqn_net_big.hip; 2x256 nets) for every batch of
                                    * >= 64 rows instead of from 16 384 rows up (tests: same results at small sizes) */
    DQN_FLAG_PW_SEGMENTS = 16,     /* sorted priority write-back: always the leaf-segment kernel (k_per_write_seg, normally from
                                    * 1 024 batch positions) ... */
    DQN_FLAG_PW_CHUNKS = 32        /* ... / always the wave-per-64-positions kernel. Same tree, bit for bit. */
} dqn_flags;

typedef struct dqn_handle dqn_handle;

/* Mirrors the constants of Test/lunar_lander.py:23-48 plus the net dims of
 * LunarLander/dddqn.py:19-22 and the PER spec. */
typedef struct {
    int32_t obs_dim;          /* D  (9 in the reference: LunarLander/env.py:17)            */
    int32_t hidden1;          /* H1 (32: dddqn.py:19); multiple of 16                       */
    int32_t hidden2;          /* H2 (64: dddqn.py:20); multiple of 16                       */
    int32_t num_actions;      /* A  (4);  1 + A <= 16                                       */
    int64_t capacity;         /* replay ring size (BUFFER_SIZE, Test/lunar_lander.py:24)    */
    int32_t use_per;          /* 0: uniform sampling (reference); 1: proportional PER       */
    int32_t max_batch;        /* largest B any call will use                                */
    int32_t optimizer;        /* dqn_optimizer                                              */
    float   lr, b1, b2, eps, weight_decay;   /* optax defaults .9 .999 1e-8 (1e-4 adamw)    */
    float   gamma;            /* GAMMA, Test/lunar_lander.py:34                             */
    float   per_alpha;        /* 0.6                                                        */
    float   per_eps;          /* 1e-6                                                       */
    float   per_beta;         /* IS exponent used by dqn_update_fused (settable)            */
    int32_t precision;        /* dqn_precision                                              */
    uint64_t seed;            /* Philox key for the handle's own draws (fused path)         */
    int32_t world_size;       /* gradient is divided by this inside the optimizer           */
    int32_t n_step;           /* n-step returns of the device-resident vector actor (not in the
                               * reference): 0 / 1 = off; 2..8: every env keeps its last n_step (s, a, r, done) and each
                               * vector step adds the row (s_u, a_u, R, s_{t+1}, done_n) of the window u = t-n_step+1 .. t,
                               * R = r_u + gamma*(r_{u+1} + gamma*(...)) cut after the first done; updates bootstrap
                               * with gamma^n_step. Rows from dqn_replay_add are stored as given. */
    int32_t flags;            /* dqn_flags, or-ed */
} dqn_config;

const char *dqn_last_error(void);
int dqn_abi_version(void);
void dqn_default_config(dqn_config *cfg);

/* lifetime. Replaces the closure construction in q_agent.py:110-113 + ReplayBuffer.__init__
 * (General/Base/replay_buffer.py:20-34). */
int dqn_create(const dqn_config *cfg, dqn_handle **out);
int dqn_destroy(dqn_handle *h);

/* parameter / optimizer state I/O (flat f32, haiku leaf order: w1 b1 w2 b2 wv bv wa ba,
 * w is [in,out] row-major). which: DQN_BUF_PARAMS / _TARGET / _MU / _NU / _GRAD. */
int dqn_param_count(const dqn_handle *h, int64_t *n);
int dqn_set_params(dqn_handle *h, int which, const float *src, int src_is_host, void *stream);
int dqn_get_params(dqn_handle *h, int which, float *dst, int dst_is_host, void *stream);
int dqn_set_opt_count(dqn_handle *h, int32_t count, void *stream);   /* ScaleByAdamState.count */
int dqn_get_opt_count_host(dqn_handle *h, int32_t *count);           /* synchronises */
int dqn_buffer(dqn_handle *h, int which, void **ptr, int64_t *nbytes);
int dqn_set_schedule(dqn_handle *h, float per_beta, float lr, void *stream);
/* re-parameterisation between training runs (ParamAgent.inject, General/QLearning/hyperparameter_optimization.py:76-91):
 * gamma is baked into captured launches, so the handle's graphs are dropped and re-captured on next use. epsilon:
 * dqn_set_epsilon; batch size / train frequency: arguments of dqn_train_iters (one graph per combination). */
int dqn_set_gamma(dqn_handle *h, float gamma);

/* ReplayBuffer.add (replay_buffer.py:58-65), vectorised: n rows at consecutive slots
 * counter % capacity. With use_per the new leaves get the running max priority. */
int dqn_replay_add(dqn_handle *h, const float *s, const int32_t *a, const float *r,
                   const float *s2, const uint8_t *d, int32_t n, void *stream);
/* the handle as a positions-only prioritized index: n new positions at the running max priority (counter, size, leaves), no row data */
int dqn_per_index_advance(dqn_handle *h, int32_t n, void *stream);
/* the same in one launch with rows that are being overwritten in place: the zero_n positions from zero_first on (mod capacity) get
 * priority 0 first (out of the draw), then n new positions enter (n = 0: none) */
int dqn_per_index_step(dqn_handle *h, int32_t n, int64_t zero_first, int32_t zero_n, void *stream);
int dqn_replay_size_host(dqn_handle *h, int64_t *size, int64_t *counter);   /* synchronises */

/* sample_batch (replay_buffer.py:68-85). idx_in != NULL: gather exactly those rows
 * (reference-parity mode: numba's RNG is opaque, so indices are an explicit input);
 * idx_in == NULL: idx[k] = (Philox4x32-10(seed, ctr, k, stream 1).x * size) >> 32. */
int dqn_replay_sample_uniform(dqn_handle *h, int32_t B, uint64_t seed, uint64_t ctr,
                              const int32_t *idx_in, float *s, int32_t *a, float *r,
                              float *s2, uint8_t *d, int32_t *idx_out, void *stream);

/* proportional PER (not in the reference). Stratified descent of the
 * f32 sum-tree, gather, normalised IS weights. */
int dqn_per_sample(dqn_handle *h, int32_t B, float beta, uint64_t seed, uint64_t ctr,
                   float *s, int32_t *a, float *r, float *s2, uint8_t *d,
                   int32_t *idx, float *isw, void *stream);
/* p_i = (|td_i| + eps)^alpha; duplicates: highest batch position wins; parents
 * recomputed as left + right. */
int dqn_per_update(dqn_handle *h, const int32_t *idx, const float *td_abs, int32_t B, void *stream);
int dqn_per_set(dqn_handle *h, const int32_t *idx, const float *prio, int32_t B, void *stream);
/* same result as dqn_per_update for NON-DECREASING idx (what dqn_per_sample returns): spread over many
 * CUs, siblings and duplicates resolved between adjacent positions. Undefined for unsorted idx. */
int dqn_per_update_sorted(dqn_handle *h, const int32_t *idx, const float *td_abs, int32_t B, void *stream);
/* as dqn_per_set (raw priorities, e.g. 0 to take rows out of the draw), for non-decreasing idx: the many-CU write-back */
int dqn_per_set_sorted(dqn_handle *h, const int32_t *idx, const float *prio, int32_t B, void *stream);

/* Model.__call__ (LunarLander/dddqn.py:24-34): q[B,A]; feat (optional) = the H2
 * features of return_features=True (:32-33). */
int dqn_qnet_forward(dqn_handle *h, int which_net, const float *x, int32_t B,
                     float *q, float *feat, void *stream);

/* the per-sample arithmetic of compute_q_targets (q_learning_functions.py:55-60) +
 * Huber loss pieces (:36). d is f32 (after preprocessing :84). Any output may be NULL. */
int dqn_td_targets(dqn_handle *h, const float *q, const float *next_q, const float *next_q_tm,
                   const int32_t *a, const float *r, const float *d, const float *isw,
                   float gamma, int32_t B, float *targets, float *td, float *dq, float *loss,
                   void *stream);

/* compute_q_targets (q_learning_functions.py:42-64): three forwards + the above. */
int dqn_q_targets(dqn_handle *h, const float *s, const int32_t *a, const float *r,
                  const float *s2, const float *d, int32_t B, float *targets, void *stream);

/* compute_loss (q_learning_functions.py:31-39); isw NULL = unweighted. loss: 1 f32. */
int dqn_loss(dqn_handle *h, const float *s, const float *targets, const float *isw,
             int32_t B, float *loss, void *stream);

/* train_step (q_learning_functions.py:14-28) in two halves, so that a gradient
 * all-reduce can sit between them: grads -> handle's DQN_BUF_GRAD (+ loss), then
 * optimizer.update + apply_updates on the handle's params / moments. */
int dqn_grads(dqn_handle *h, const float *s, const float *targets, const float *isw,
              int32_t B, float *loss, void *stream);
int dqn_optimizer_step(dqn_handle *h, void *stream);
int dqn_train_step(dqn_handle *h, const float *s, const float *targets, int32_t B, void *stream);

/* Agent._step (q_agent.py:146-169) as one call on the handle's own replay:
 * sample -> q-targets -> backward -> optimizer -> PER write-back. Replayed from a
 * hipGraph. With world_size > 1 use the two halves around dqn_allreduce_grads (or a
 * torch.distributed all_reduce on DQN_BUF_GRAD). */
int dqn_update_fused(dqn_handle *h, int32_t B, void *stream);
int dqn_update_backward(dqn_handle *h, int32_t B, void *stream);   /* sample .. grads, + PER write-back */
int dqn_update_apply(dqn_handle *h, int32_t B, void *stream);      /* optimizer on the (all-reduced) grads */

/* compute_action (q_learning_functions.py:67-73) + Agent._policy (q_agent.py:137-141),
 * vectorised over n rows: greedy iff eps < U(0,1) else randint(0, A), Philox stream 2. */
int dqn_act(dqn_handle *h, const float *s, int32_t n, float epsilon, uint64_t seed,
            uint64_t ctr, int32_t *actions, void *stream);

/* One vector step of the env loop (q_agent.py:176-183) on n_envs synthetic environments that
 * live on the device: action = _policy(state) -> synthetic transition (obs' ~
 * N(0,1), r ~ N(0,1) or +-100 on terminals, done ~ Bernoulli(p_done); Philox stream 3) ->
 * replay.add -> state = observation. Graph-replayed; epsilon and the step counter live on
 * the device. dqn_env_reset uploads the initial observations [n_envs, D]. */
int dqn_set_epsilon(dqn_handle *h, float epsilon, void *stream);
/* kind of the device-resident envs: DQN_ENV_SYNTHETIC (default) or DQN_ENV_CARTPOLE (CartPole-v1 physics, obs 4,
 * actions 2, reward 1 per step, truncation at max_steps as q_agent.py:179-180, auto-reset). term_reward is the
 * reward of the step that TERMINATES an episode (gym: 1). NOTE: the reference's target rule (q_learning_functions.py
 * :58, kept verbatim) makes the terminal target q + r, so a positive terminal reward inflates Q at failures; pass a
 * negative term_reward (e.g. -1) to make CartPole learnable under that rule. dqn_env_stats_host returns the number
 * of finished episodes and their summed length (synchronises). */
int dqn_env_config(dqn_handle *h, int32_t kind, int32_t max_steps, float term_reward);
int dqn_env_stats_host(dqn_handle *h, int64_t *episodes, int64_t *episode_steps);
/* ObsWrapper (LunarLander/env.py:19-31) for the device-resident vector envs: with enable != 0 the LAST observation column
 * (obs_dim counts it: 8 + 1 = 9 as in the reference) is step / max_steps -- float32 of the float64 quotient, `step`
 * pre-incremented by every env step and zeroed when an episode ends -- and an episode also ends at max_steps
 * (q_agent.py:179-180; max_steps from dqn_env_config). Synthetic env, one-step returns; the vector steps between two
 * updates then run as separate launches with the feature pass behind each. dqn_env_reset's observations carry the
 * feature of step 0 (0.0) in that column. */
int dqn_env_time_feature(dqn_handle *h, int32_t enable);
int dqn_env_reset(dqn_handle *h, const float *obs, int32_t n_envs, float p_done, void *stream);
int dqn_actor_step(dqn_handle *h, int32_t n_envs, void *stream);
/* env_steps consecutive vector env steps (q_agent.py:176-183 x train_frequency, between two updates the parameters do
 * not change) as ONE launch: same transitions, ring slots and tree as env_steps calls of dqn_actor_step.
 * env_steps * n_envs <= capacity. */
int dqn_actor_steps(dqn_handle *h, int32_t env_steps, int32_t n_envs, void *stream);
/* the reference's inner loop (q_agent.py:174-187) for n_iters iterations as ONE hipGraph launch:
 * each iteration = env_steps vector env steps (train_frequency) followed by one Agent._step.
 * Single-GPU path (world_size == 1). */
int dqn_train_iters(dqn_handle *h, int32_t n_iters, int32_t env_steps, int32_t n_envs, int32_t B, void *stream);
/* data-parallel form of one iteration: env_steps vector env steps + dqn_update_backward as ONE graph launch;
 * follow with the gradient all-reduce and dqn_update_apply. */
int dqn_actor_backward(dqn_handle *h, int32_t env_steps, int32_t n_envs, int32_t B, void *stream);

/* Agent._update_target_model (q_agent.py:143-144) */
int dqn_sync_target(dqn_handle *h, void *stream);

/* per-kernel timing with HIP events on `stream` (bench.py's live roofline measurement): between
 * begin and end every hot-path launch is run eagerly with hipExtLaunchKernelGGL start/stop events (the
 * dispatch's own timestamps); end synchronises and returns up to max_entries (name, elapsed ms) pairs,
 * names as NUL-terminated strings of name_stride bytes each. A name covering two kernels reports the first. */
int dqn_profile_begin(dqn_handle *h, void *stream);
int dqn_profile_end(dqn_handle *h, void *stream, char *names_host, int32_t name_stride, float *ms_host,
                    int32_t max_entries, int32_t *count_host);

/* gradient all-reduce for independent per-GPU learners (no counterpart in the
 * reference). unique_id: the 128-byte ncclUniqueId from rank 0. */
int dqn_comm_unique_id(void *unique_id_128);
int dqn_comm_init(dqn_handle *h, const void *unique_id_128, int32_t rank, int32_t world);
int dqn_allreduce_grads(dqn_handle *h, void *stream);
int dqn_comm_count_host(dqn_handle *h, int32_t *ranks);   /* ncclCommCount of the handle's communicator (0: none) */

/* In-kernel hand-over waits are bounded (0.2 s): a launch whose partner workgroups never ran ends, poisons the loss with
 * NaN and counts here instead of hanging the GPU. Returns the count since dqn_create (synchronises); non-zero means the
 * results since then are not to be trusted. */
int dqn_device_errors_host(dqn_handle *h, int64_t *count);
/* resets the hand-over words and the error count after give-ups were reported (the timed-out launches' results stay) */
int dqn_clear_device_errors(dqn_handle *h);
/* diagnostic (tests): the fused forward's partner passes stop counting themselves in -> every hand-over wait of the next
 * updates runs into its bound (0.2 s), the error count goes up, the loss is NaN; on = 0 restores the normal path */
int dqn_debug_withhold_handover(dqn_handle *h, int32_t on);

/* ---- Nature-CNN dueling Q-network (BASELINE configs[4], PongNoFrameskip-v4 shape). Not in
 * the reference: the trunk (conv 32x8x8/4, 64x4x4/2, 64x3x3/1, fc 512, ReLU) ends in the reference's dueling head
 * (LunarLander/dddqn.py:29-31) and feeds the reference's TD rule (q_learning_functions.py:55-60). Frames: u8
 * [B][84][84][4] (NHWC, four stacked frames), scaled by 1/255 in the first layer. Parameters, flat f32:
 * conv1 w[8,8,4,32] b[32]  conv2 w[4,4,32,64] b[64]  conv3 w[3,3,64,64] b[64]  fc w[3136,512] b[512] (rows in [7][7][64]
 * order)  val w[512,1] b[1]  adv w[512,A] b[A].  precision: DQN_PREC_F32 = exact f32 MFMA (k-ascending fmaf chains),
 * DQN_PREC_BF16 = bf16 MFMA operands, f32 accumulate. */
typedef struct dqn_cnn_handle dqn_cnn_handle;
int dqn_cnn_create(int32_t num_actions, int32_t max_batch, int32_t precision, dqn_cnn_handle **out);
int dqn_cnn_destroy(dqn_cnn_handle *h);
/* diagnostics for the tests (results stay bit-identical): FC_WIDE_TILE = the bf16 mode's 128 x 128 fc tile (normally taken from
 * 8 192 rows) at every batch size; NO_SIDE_STREAM = everything in stream order on the caller's stream; LAYERWISE_CONV = the bf16
 * mode's three convolutions as one kernel per layer (r02's) instead of the fused trunk kernel (r03) */
enum dqn_cnn_flags { DQN_CNN_FLAG_FC_WIDE_TILE = 1, DQN_CNN_FLAG_NO_SIDE_STREAM = 2, DQN_CNN_FLAG_LAYERWISE_CONV = 4 };
int dqn_cnn_set_flags(dqn_cnn_handle *h, int32_t flags);
int dqn_cnn_param_count(const dqn_cnn_handle *h, int64_t *n);
int dqn_cnn_set_params(dqn_cnn_handle *h, int which_net, const float *src, int src_is_host, void *stream);
int dqn_cnn_forward(dqn_cnn_handle *h, int which_net, const uint8_t *frames, int32_t B, float *q, void *stream);
/* compute_q_targets (q_learning_functions.py:42-64) with the CNN as the model: d is f32 (preprocessing :84) */
int dqn_cnn_q_targets(dqn_cnn_handle *h, const uint8_t *s, const int32_t *a, const float *r, const uint8_t *s2,
                      const float *d, float gamma, int32_t B, float *targets, void *stream);
/* jax.grad(compute_loss) (q_learning_functions.py:23, :31-39: mean_i w_i sum_a huber(model(s_i)[a] - targets[i][a])) w.r.t.
 * every leaf, into the handle's gradient buffer; targets [B][A], isw [B] or NULL, loss_host optional (synchronises). */
int dqn_cnn_grads(dqn_cnn_handle *h, const uint8_t *frames, const float *targets, const float *isw, int32_t B,
                  float *loss_host, void *stream);
/* which_buf: DQN_BUF_PARAMS / _TARGET / _GRAD / _MU / _NU (dqn_cnn_param_count floats each) */
int dqn_cnn_get_buffer(dqn_cnn_handle *h, int which_buf, float *dst, int dst_is_host, void *stream);
/* the buffer itself (device pointer, bytes): per-GPU learners all-reduce DQN_BUF_GRAD in place between dqn_cnn_grads and
 * dqn_cnn_optimizer_step(grad_scale = 1 / world) */
int dqn_cnn_buffer(dqn_cnn_handle *h, int which_buf, void **ptr, int64_t *bytes);
/* optax.adam / adamw (Test/lunar_lander.py:48); resets moments and step count */
int dqn_cnn_set_optimizer(dqn_cnn_handle *h, int32_t adamw, float lr, float b1, float b2, float eps, float weight_decay,
                          void *stream);
/* optimizer.update + optax.apply_updates (q_learning_functions.py:24-25) on the gradient buffer (x grad_scale) */
int dqn_cnn_optimizer_step(dqn_cnn_handle *h, float grad_scale, void *stream);
/* train_step (q_learning_functions.py:14-28) = dqn_cnn_grads + dqn_cnn_optimizer_step */
int dqn_cnn_train_step(dqn_cnn_handle *h, const uint8_t *frames, const float *targets, const float *isw, int32_t B,
                       void *stream);
/* Agent._step (q_agent.py:146-169) on a given minibatch: compute_q_targets + train_step sharing the online pass over s */
int dqn_cnn_update(dqn_cnn_handle *h, const uint8_t *s, const int32_t *a, const float *r, const uint8_t *s2, const float *d,
                   const float *isw, float gamma, int32_t B, float *loss_host, void *stream);
/* Agent._update_target_model (q_agent.py:143-144) */
int dqn_cnn_sync_target(dqn_cnn_handle *h, void *stream);
/* Per-GPU learners for the CNN (BASELINE configs[4] in its 8-GPU form): the handle's own RCCL communicator (unique id from
 * dqn_comm_unique_id on rank 0). With it dqn_cnn_update / dqn_cnn_update_replay sum the gradient over the ranks -- the fc weight leaf
 * (6.4 MB of the 6.7 MB) on a stream of the communicator's own beside the rest of the backward -- and step with grad_scale = 1 / world.
 * dqn_cnn_allreduce_grads: the same reduction as a call of its own, between dqn_cnn_grads and dqn_cnn_optimizer_step. */
int dqn_cnn_comm_init(dqn_cnn_handle *h, const void *unique_id_128, int32_t rank, int32_t world);
int dqn_cnn_comm_count_host(dqn_cnn_handle *h, int32_t *ranks);
int dqn_cnn_allreduce_grads(dqn_cnn_handle *h, int32_t fc_leaf_done, void *stream);
/* Agent._policy (q_agent.py:137-141) with the CNN as the model: epsilon-greedy actions for n frame stacks (Philox as dqn_act) */
int dqn_cnn_act(dqn_cnn_handle *h, const uint8_t *frames, int32_t n, float epsilon, uint64_t seed, uint64_t ctr,
                int32_t *actions, void *stream);
/* ReplayBuffer (General/Base/replay_buffer.py:20-85) for u8 frame stacks inside the CNN handle: init (zeroed ring of s, s', a,
 * r, d), add n transitions at the head, gather given rows; the PER tree over the same positions is a dqn_handle of the same
 * capacity (its dqn_per_sample / dqn_per_update_sorted give and take the indices). d is f32 (preprocessing :84). */
int dqn_cnn_replay_init(dqn_cnn_handle *h, int64_t capacity);
int dqn_cnn_replay_add(dqn_cnn_handle *h, const uint8_t *s, const int32_t *a, const float *r, const uint8_t *s2,
                       const float *d, int32_t n, int64_t *first_index, void *stream);
/* The synthetic frame-stack vector env of BASELINE configs[4]'s shape on the device (no ALE here): reset n envs, then
 * one call per vector step = CNN act + synthetic transition (Philox frames / rewards / dones) + ReplayBuffer.add of the n rows. */
int dqn_cnn_env_reset_synth(dqn_cnn_handle *h, int32_t n, uint64_t seed, void *stream);
int dqn_cnn_env_step_synth(dqn_cnn_handle *h, float epsilon, float p_done, int64_t *first_index, void *stream);
int dqn_cnn_replay_size_host(const dqn_cnn_handle *h, int64_t *size, int64_t *counter);
/* n_step = 1: the stored rows. n_step 2..8 with the rows stored step-major (n_envs rows per env step, capacity a multiple of
 * n_envs): the n-step transition that STARTS at each row -- s and a of the row, R = r_0 + gamma (r_1 + ...) cut after the first done,
 * s' of the last step, that done flag; sample only rows whose n - 1 successors are already in the ring. */
int dqn_cnn_replay_gather(dqn_cnn_handle *h, const int32_t *idx, int32_t B, int32_t n_step, int32_t n_envs, float gamma,
                          uint8_t *s, int32_t *a, float *r, uint8_t *s2, float *d, void *stream);
/* Agent._step from the frame ring: gather idx (n-step transitions when n_step > 1, bootstrapped with gamma^n) + dqn_cnn_update;
 * td_abs_out (device, optional) = |delta| per sample */
int dqn_cnn_update_replay(dqn_cnn_handle *h, const int32_t *idx, const float *isw, float gamma, int32_t n_step, int32_t n_envs,
                          int32_t B, float *td_abs_out, float *loss_host, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DQN_HIP_H */
