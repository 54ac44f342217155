// csrc/dqn_api.hip -- the C ABI of libdqn_hip.so (include/dqn_hip.h): handle, device arena,
// kernel sequencing, hipGraph capture of the fused update, RCCL gradient all-reduce.
#include "../../include/dqn_hip.h"
#include "dqn_launch.h"
#include "dqn_per_device.h"

#include <dlfcn.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
thread_local hipEvent_t g_prof_ev0 = nullptr, g_prof_ev1 = nullptr;
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
int dqn_set_error(int code, const char *msg) { g_err = msg; return code; }    // for the other translation units (dqn_cnn.hip)
#define HIP_TRY(expr)                                                                          \
    do { hipError_t e_ = (expr);                                                               \
         if (e_ != hipSuccess) return fail(DQN_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define REQUIRE(cond, ...) do { if (!(cond)) return fail(DQN_ERR_INVALID, __VA_ARGS__); } while (0)

extern "C" const char *dqn_last_error(void) { return g_err.c_str(); }
extern "C" int dqn_abi_version(void) { return DQN_ABI_VERSION; }

extern "C" void dqn_default_config(dqn_config *c) {
    // Test/lunar_lander.py:23-48 + LunarLander/dddqn.py:19-22 + optax defaults
    memset(c, 0, sizeof(*c));
    c->obs_dim = 9; c->hidden1 = 32; c->hidden2 = 64; c->num_actions = 4;
    c->capacity = 100000; c->use_per = 0; c->max_batch = 64;
    c->optimizer = DQN_OPT_ADAMW; c->lr = 2e-4f; c->b1 = 0.9f; c->b2 = 0.999f; c->eps = 1e-8f;
    c->weight_decay = 1e-4f; c->gamma = 0.99f;
    c->per_alpha = 0.6f; c->per_eps = 1e-6f; c->per_beta = 0.4f;
    c->precision = DQN_PREC_F32; c->seed = 0; c->world_size = 1;
    c->n_step = 1; c->flags = 0;
}

// ------------------------------------------------------------------------------ handle
struct GraphSet { hipGraphExec_t fused = nullptr, bwd = nullptr, apply = nullptr, actor = nullptr; };

// RCCL entry points, resolved lazily so the library loads without librccl
struct NcclId { char b[128]; };   // ncclUniqueId (passed by value to ncclCommInitRank)
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, NcclId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

struct dqn_handle {
    dqn_config cfg;
    NetDims m;
    int L = 0;                 // tree levels (N_tree = 2^L >= capacity)
    long long Ntree = 0;
    int Bp = 0;                // max_batch rounded up to 16 (f32) / 32 (bf16)
    bool bf16 = false;         // DQN_PREC_BF16: packs and stashes hold bf16 data
    void *arena = nullptr;
    size_t arena_bytes = 0, pack_bytes = 0;
    DqnState *st = nullptr;
    float *params = nullptr, *target = nullptr, *mu = nullptr, *nu = nullptr, *grad = nullptr;
    float *pack = nullptr, *pack_t = nullptr;
    float *pack_act = nullptr;                                    // bf16 mode: f32 shadows for the actor kernel
    float *states = nullptr, *observations = nullptr, *rewards = nullptr;
    int32_t *actions = nullptr; uint8_t *dones = nullptr;
    float *tree = nullptr; unsigned long long *stamp = nullptr;
    float *bs = nullptr, *bs2 = nullptr, *br = nullptr, *bw_raw = nullptr, *bisw = nullptr, *btd = nullptr,
          *btd_abs = nullptr, *bdf = nullptr;
    int32_t *ba = nullptr, *bidx = nullptr; uint8_t *bd = nullptr;
    float *q = nullptr, *nq = nullptr, *nt = nullptr;
    float *px = nullptr, *ph1 = nullptr, *ph2 = nullptr, *pdz1 = nullptr, *pdz2 = nullptr, *pdz3 = nullptr;
    float *loss_part = nullptr, *loss_dev = nullptr, *scratch = nullptr;
    int *tile_cnt = nullptr;                                      // per-tile hand-over counters of the fused forward + row backward
    float *big_slab = nullptr, *big_colsum = nullptr;             // large-batch path (dqn_net_big.hip): split-K partial tiles, per-row-tile column sums
    float *pw_part = nullptr; int pw_force = 0;                   // per-segment priority maxima of k_per_write_seg; DQN_FLAG_PW_* (tests)
    unsigned int *wmax_tmp = nullptr;                             // batch max of the raw IS weights (API sampler -> normalise)
    int num_cus = 256;                                            // hipDeviceAttributeMultiprocessorCount of the handle's device
    float *env_obs = nullptr, *env_next = nullptr, *env_r = nullptr; int32_t *env_a = nullptr; uint8_t *env_d = nullptr;
    float *hist_s = nullptr, *hist_r = nullptr; int32_t *hist_a = nullptr, *hist_d = nullptr;   // n-step history
    int n_step = 1; float gamma_n = 0.0f;
    bool no_handover = false;                                     // DQN_FLAG_NO_HANDOVER: no in-launch waits (no sampler workgroups, k_bwd_rows on its own)
    bool no_actor16 = false, f32_actor = false;                   // DQN_FLAG_NO_ACTOR16 / DQN_FLAG_BF16_F32_ACTOR
    bool big_any = false;                                         // DQN_FLAG_BIG_ROWS
    int withhold = 0;                                             // dqn_debug_withhold_handover
    int tile_stride = 0;                                          // tile_cnt: [tile_stride] arrival counters + [tile_stride] consumed counts
    float p_done = 0.01f;
    int env_kind = 0, env_max_steps = 500; int32_t *env_t = nullptr; float env_term_reward = 1.0f;
    bool env_time_feature = false;                                // the last observation column is ObsWrapper's step / max_steps
    // per-kernel HIP-event timing (dqn_profile_*): events[i] .. events[i+1] brackets launch i
    bool profiling = false;
    std::vector<hipEvent_t> events; size_t ev_used = 0;
    std::vector<const char *> ev_names;
    std::map<int, GraphSet> graphs;
    std::map<std::vector<int>, hipGraphExec_t> loop_graphs;      // (iters, env_steps, n_envs, B) -> graph
    void *comm = nullptr; int rank = 0, world = 1;
    std::map<int, std::pair<void *, int64_t>> bufs;
};

static Rccl g_rccl;

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// precision dispatch
static void L_pack(dqn_handle *h, hipStream_t s, const float *params, float *pack) {
    if (h->bf16) {
        launch_pack_bf16(s, h->m, params, pack);
        if (pack == h->pack) { launch_pack(s, h->m, params, h->pack_act); launch_pack_w2k16(s, h->m, params, h->pack_act); }   // the actor kernel reads f32 shadows of the online net
    } else launch_pack(s, h->m, params, pack);
}
static bool big_from(const dqn_handle *h, int B) { return h->big_any || (h->bf16 && B >= DQN_BIG_MIN_BF16); }
static bool use_big(dqn_handle *h, int B) { return h->big_slab != nullptr && big_supported(h->m, B, big_from(h, B)); }
static void L_fwd(dqn_handle *h, hipStream_t s, const FwdPass *p, int n, int B, const SampleArgs *smp = nullptr,
                  const BwdArgs *fuse = nullptr) {
    if (use_big(h, B) && !smp && !fuse) {                          // large batches: 64-row tiles (dqn_net_big.hip), forward only
        bool plain = true;
        for (int i = 0; i < n; ++i) plain = plain && !p[i].act_out && !p[i].px && p[i].x;
        if (plain) { if (h->bf16) launch_big16_forward(s, h->m, p, n, B, h->num_cus); else launch_big_forward(s, h->m, p, n, B, h->num_cus); return; }
    }
    if (h->bf16) launch_qnet_fwd_bf16(s, h->m, p, n, B, smp, fuse, h->tile_cnt, h->st, h->tile_stride, h->withhold);
    else launch_qnet_fwd(s, h->m, p, n, B, smp, fuse, h->tile_cnt, h->st, h->tile_stride, h->withhold);
}
static void L_bwd(dqn_handle *h, hipStream_t s, const BwdArgs &g, int B) {
    if (h->bf16) launch_bwd_rows_bf16(s, h->m, g, B, h->st); else launch_bwd_rows(s, h->m, g, B, h->st);
}
static void L_dw(dqn_handle *h, hipStream_t s, int B, float *loss_out, int bump, const AdamArgs &ad, const PwArgs &pw) {
    if (h->bf16) launch_dw_bf16(s, h->m, h->px, h->ph1, h->ph2, h->pdz1, h->pdz2, h->pdz3, B, h->grad, h->loss_part, loss_out, h->st, bump, ad, pw);
    else launch_dw(s, h->m, h->px, h->ph1, h->ph2, h->pdz1, h->pdz2, h->pdz3, B, h->grad, h->loss_part, loss_out, h->st, bump, ad, pw);
}

// profiling mode: arm() before a launch hands the next (start, stop) event pair to DQN_LAUNCH, mark() names it
static void arm(dqn_handle *h) {
    if (!h->profiling) return;
    while (h->events.size() < h->ev_used + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        h->events.push_back(e);
    }
    g_prof_ev0 = h->events[h->ev_used]; g_prof_ev1 = h->events[h->ev_used + 1];
}
static void mark(dqn_handle *h, hipStream_t st, const char *name) {
    (void)st;
    if (!h->profiling) return;
    h->ev_used += 2;
    h->ev_names.push_back(name);
}

extern "C" int dqn_create(const dqn_config *cfg, dqn_handle **out) {
    REQUIRE(cfg && out, "dqn_create: null argument");
    const int D = cfg->obs_dim, H1 = cfg->hidden1, H2 = cfg->hidden2, A = cfg->num_actions;
    REQUIRE(D >= 1 && D <= 256, "obs_dim %d out of range [1,256]", D);
    REQUIRE(H1 >= 16 && H1 <= 256 && H1 % 16 == 0, "hidden1 %d must be a multiple of 16 in [16,256]", H1);
    REQUIRE(H2 >= 16 && H2 <= 256 && H2 % 16 == 0, "hidden2 %d must be a multiple of 16 in [16,256]", H2);
    REQUIRE(A >= 1 && A <= 15, "num_actions %d out of range [1,15]", A);
    REQUIRE(cfg->capacity >= 1 && cfg->capacity <= (1ll << 30), "capacity out of range");
    REQUIRE(cfg->max_batch >= 1 && cfg->max_batch <= (1 << 22), "max_batch out of range");
    REQUIRE(cfg->precision == DQN_PREC_F32 || cfg->precision == DQN_PREC_BF16, "unknown precision %d", cfg->precision);
    REQUIRE(cfg->optimizer == DQN_OPT_ADAM || cfg->optimizer == DQN_OPT_ADAMW, "unknown optimizer");
    REQUIRE(cfg->world_size >= 1, "world_size must be >= 1");
    REQUIRE(cfg->n_step >= 0 && cfg->n_step <= 8, "n_step %d out of range [0,8]", cfg->n_step);
    REQUIRE((cfg->flags & ~63) == 0, "unknown flags 0x%x", cfg->flags);
    REQUIRE(cfg->n_step <= 1 || cfg->capacity >= 64ll * cfg->max_batch, "n_step > 1 needs capacity >= 64 * max_batch "
            "(the n-step actor runs in k_actor only, whose steps of one launch must fit the ring)");

    dqn_handle *h = new (std::nothrow) dqn_handle();
    if (!h) return fail(DQN_ERR_NOMEM, "host allocation failed");
    h->cfg = *cfg;
    {   // machine size: every grid that must be resident as a whole is derived from it
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            h->num_cus = cus;
    }
    h->m = make_dims(D, H1, H2, A);
    h->world = cfg->world_size;
    h->bf16 = cfg->precision == DQN_PREC_BF16;
    h->no_handover = (cfg->flags & DQN_FLAG_NO_HANDOVER) != 0;
    h->no_actor16 = (cfg->flags & DQN_FLAG_NO_ACTOR16) != 0;
    h->f32_actor = (cfg->flags & DQN_FLAG_BF16_F32_ACTOR) != 0;
    h->big_any = (cfg->flags & DQN_FLAG_BIG_ROWS) != 0;
    h->pw_force = (cfg->flags & DQN_FLAG_PW_SEGMENTS) ? 2 : ((cfg->flags & DQN_FLAG_PW_CHUNKS) ? 1 : 0);
    h->n_step = cfg->n_step > 1 ? cfg->n_step : 1;
    h->gamma_n = cfg->gamma;
    for (int i = 1; i < h->n_step; ++i) h->gamma_n = h->gamma_n * cfg->gamma;       // f32 product, as the oracle's
    h->Bp = (cfg->max_batch + 63) / 64 * 64;                       // whole row tiles of every kernel family
    if (cfg->use_per) {
        int L = 0; while ((1ll << L) < cfg->capacity) ++L;
        if (L < 1) L = 1;
        h->L = L; h->Ntree = 1ll << L;
    }
    const long long N = cfg->capacity, P = h->m.P, Bp = h->Bp;
    const int K1 = h->m.KQ1 * 16;

    // carve one arena
    struct Item { void **p; size_t bytes; int id; };
    std::vector<Item> items;
    auto add = [&](void *pp, size_t bytes, int id = -1) { items.push_back({(void **)pp, bytes, id}); };
    add(&h->st, sizeof(DqnState));
    add(&h->params, P * 4, DQN_BUF_PARAMS); add(&h->target, P * 4, DQN_BUF_TARGET);
    add(&h->mu, P * 4, DQN_BUF_MU); add(&h->nu, P * 4, DQN_BUF_NU); add(&h->grad, P * 4, DQN_BUF_GRAD);
    const size_t esz = h->bf16 ? 2 : 4;                           // element size of packs and stashes
    const size_t pack_bytes = h->bf16 ? (size_t)bf16_pack_elems(h->m) * 2 : (size_t)h->m.pack_floats * 4;
    h->pack_bytes = pack_bytes;
    add(&h->pack, pack_bytes); add(&h->pack_t, pack_bytes);
    if (h->bf16) add(&h->pack_act, (size_t)h->m.pack_floats * 4 + (size_t)h->m.H1 * h->m.H2 * 2);   // f32 shadows + bf16 k-packed W2
    add(&h->states, N * D * 4, DQN_BUF_STATES); add(&h->observations, N * D * 4, DQN_BUF_OBSERVATIONS);
    add(&h->rewards, N * 4, DQN_BUF_REWARDS); add(&h->actions, N * 4, DQN_BUF_ACTIONS);
    add(&h->dones, N, DQN_BUF_DONES);
    if (cfg->use_per) { add(&h->tree, 2 * h->Ntree * 4, DQN_BUF_TREE); add(&h->stamp, h->Ntree * 8); add(&h->pw_part, 8192 * 4); }
    add(&h->bs, Bp * D * 4); add(&h->bs2, Bp * D * 4); add(&h->br, Bp * 4); add(&h->bw_raw, Bp * 4);
    add(&h->bisw, Bp * 4, DQN_BUF_BATCH_ISW); add(&h->btd, Bp * 4, DQN_BUF_BATCH_TD); add(&h->btd_abs, Bp * 4);
    add(&h->bdf, Bp * 4); add(&h->ba, Bp * 4); add(&h->bidx, Bp * 4, DQN_BUF_BATCH_IDX); add(&h->bd, Bp);
    add(&h->q, Bp * A * 4); add(&h->nq, Bp * A * 4); add(&h->nt, Bp * A * 4);
    add(&h->px, Bp * (h->bf16 ? 32 : K1) * esz);                  // (bf16: the 64-row kernels stash 32 columns)
    add(&h->ph1, Bp * H1 * esz); add(&h->ph2, Bp * H2 * esz);
    add(&h->pdz1, Bp * H1 * esz); add(&h->pdz2, Bp * H2 * esz); add(&h->pdz3, Bp * 16 * esz);
    add(&h->loss_part, (Bp / 16) * 4); add(&h->loss_dev, 4, DQN_BUF_LOSS); add(&h->scratch, Bp * 4);
    if (big_supported(h->m, cfg->max_batch, big_from(h, cfg->max_batch))) {
        add(&h->big_slab, big_slab_floats(cfg->max_batch, h->num_cus) * 4); add(&h->big_colsum, big_colsum_floats(cfg->max_batch) * 4);
    }
    h->tile_stride = (int)(Bp / 16) + 2;
    add(&h->tile_cnt, (size_t)h->tile_stride * 2 * 4); add(&h->wmax_tmp, 4);
    add(&h->env_obs, Bp * D * 4, DQN_BUF_ENV_OBS); add(&h->env_next, Bp * D * 4); add(&h->env_r, Bp * 4);
    add(&h->env_a, Bp * 4, DQN_BUF_ENV_ACTIONS); add(&h->env_d, Bp); add(&h->env_t, Bp * 4);
    if (h->n_step > 1) {
        const size_t ns = (size_t)h->n_step;
        add(&h->hist_s, ns * Bp * D * 4); add(&h->hist_r, ns * Bp * 4); add(&h->hist_a, ns * Bp * 4); add(&h->hist_d, ns * Bp * 4);
    }
    size_t total = 0;
    for (auto &it : items) total += align_up(it.bytes, 256);
    hipError_t e = hipMalloc(&h->arena, total);
    if (e != hipSuccess) { delete h; return fail(DQN_ERR_NOMEM, "hipMalloc(%zu bytes): %s", total, hipGetErrorString(e)); }
    h->arena_bytes = total;
    e = hipMemset(h->arena, 0, total);                   // zero ring (replay_buffer.py:28-32), tree, moments
    if (e != hipSuccess) { (void)hipFree(h->arena); delete h; return fail(DQN_ERR_HIP, "hipMemset: %s", hipGetErrorString(e)); }
    size_t off = 0;
    for (auto &it : items) {
        *it.p = (char *)h->arena + off;
        if (it.id >= 0) h->bufs[it.id] = {*it.p, (int64_t)it.bytes};
        off += align_up(it.bytes, 256);
    }
    DqnState s0{};
    s0.b1pow = 1.0; s0.b2pow = 1.0; s0.pmax = 1.0f; s0.beta = cfg->per_beta; s0.lr = cfg->lr;
    e = hipMemcpy(h->st, &s0, sizeof(s0), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(h->arena); delete h; return fail(DQN_ERR_HIP, "state init: %s", hipGetErrorString(e)); }
    *out = h;
    return DQN_OK;
}

static void destroy_graphs(dqn_handle *h) {
    for (auto &kv : h->graphs) {
        if (kv.second.fused) (void)hipGraphExecDestroy(kv.second.fused);
        if (kv.second.bwd) (void)hipGraphExecDestroy(kv.second.bwd);
        if (kv.second.apply) (void)hipGraphExecDestroy(kv.second.apply);
        if (kv.second.actor) (void)hipGraphExecDestroy(kv.second.actor);
    }
    h->graphs.clear();
    for (auto &kv : h->loop_graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
    h->loop_graphs.clear();
}

extern "C" int dqn_destroy(dqn_handle *h) {
    if (!h) return DQN_OK;
    (void)hipDeviceSynchronize();
    destroy_graphs(h);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    for (auto e : h->events) (void)hipEventDestroy(e);
    if (h->arena) (void)hipFree(h->arena);
    delete h;
    return DQN_OK;
}

extern "C" int dqn_param_count(const dqn_handle *h, int64_t *n) {
    REQUIRE(h && n, "null argument");
    *n = h->m.P;
    return DQN_OK;
}

static float *param_buf(dqn_handle *h, int which) {
    switch (which) {
    case DQN_BUF_PARAMS: return h->params; case DQN_BUF_TARGET: return h->target;
    case DQN_BUF_MU: return h->mu; case DQN_BUF_NU: return h->nu; case DQN_BUF_GRAD: return h->grad;
    default: return nullptr;
    }
}

extern "C" int dqn_set_params(dqn_handle *h, int which, const float *src, int src_is_host, void *stream) {
    REQUIRE(h && src, "null argument");
    float *dst = param_buf(h, which);
    REQUIRE(dst, "dqn_set_params: bad selector %d", which);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(dst, src, h->m.P * 4, src_is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    if (which == DQN_BUF_PARAMS) L_pack(h, s, h->params, h->pack);
    if (which == DQN_BUF_TARGET) L_pack(h, s, h->target, h->pack_t);
    HIP_TRY(hipGetLastError());
    if (src_is_host) HIP_TRY(hipStreamSynchronize(s));      // the host buffer may be pageable / reused
    return DQN_OK;
}

extern "C" int dqn_get_params(dqn_handle *h, int which, float *dst, int dst_is_host, void *stream) {
    REQUIRE(h && dst, "null argument");
    float *src = param_buf(h, which);
    REQUIRE(src, "dqn_get_params: bad selector %d", which);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(dst, src, h->m.P * 4, dst_is_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s));
    if (dst_is_host) HIP_TRY(hipStreamSynchronize(s));
    return DQN_OK;
}

extern "C" int dqn_set_opt_count(dqn_handle *h, int32_t count, void *stream) {
    REQUIRE(h && count >= 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const double b1pow = std::pow((double)h->cfg.b1, (double)count), b2pow = std::pow((double)h->cfg.b2, (double)count);
    HIP_TRY(hipMemcpyAsync(&h->st->b1pow, &b1pow, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(&h->st->b2pow, &b2pow, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(&h->st->adam_count, &count, 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DQN_OK;
}

extern "C" int dqn_get_opt_count_host(dqn_handle *h, int32_t *count) {
    REQUIRE(h && count, "null argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(count, &h->st->adam_count, 4, hipMemcpyDeviceToHost));
    return DQN_OK;
}

extern "C" int dqn_buffer(dqn_handle *h, int which, void **ptr, int64_t *nbytes) {
    REQUIRE(h && ptr && nbytes, "null argument");
    auto it = h->bufs.find(which);
    REQUIRE(it != h->bufs.end(), "dqn_buffer: selector %d not available on this handle", which);
    *ptr = it->second.first; *nbytes = it->second.second;
    return DQN_OK;
}

extern "C" int dqn_set_schedule(dqn_handle *h, float per_beta, float lr, void *stream) {
    REQUIRE(h, "null argument");
    hipStream_t s = (hipStream_t)stream;
    h->cfg.per_beta = per_beta; h->cfg.lr = lr;
    HIP_TRY(hipMemcpyAsync(&h->st->beta, &h->cfg.per_beta, 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(&h->st->lr, &h->cfg.lr, 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DQN_OK;
}

/* gamma is a by-value argument of the captured launches (and gamma^n_step is derived from it): changing it drops the graphs */
extern "C" int dqn_set_gamma(dqn_handle *h, float gamma) {
    REQUIRE(h && gamma >= 0.0f && gamma <= 1.0f, "dqn_set_gamma: gamma out of [0, 1]");
    if (gamma == h->cfg.gamma) return DQN_OK;
    h->cfg.gamma = gamma;
    h->gamma_n = gamma;
    for (int i = 1; i < h->n_step; ++i) h->gamma_n = h->gamma_n * gamma;
    destroy_graphs(h);
    return DQN_OK;
}

// ------------------------------------------------------------------------------ replay
extern "C" int dqn_replay_add(dqn_handle *h, const float *s, const int32_t *a, const float *r,
                              const float *s2, const uint8_t *d, int32_t n, void *stream) {
    REQUIRE(h && s && a && r && s2 && d, "null argument");
    REQUIRE(n >= 1 && n <= h->cfg.capacity, "dqn_replay_add: n=%d must be in [1, capacity]", n);
    hipStream_t st = (hipStream_t)stream;
    launch_replay_add(st, h->st, h->states, h->actions, h->rewards, h->observations, h->dones,
                      h->cfg.capacity, h->cfg.obs_dim, s, a, r, s2, d, n, nullptr, 0);
    if (h->cfg.use_per) launch_per_add(st, h->st, h->tree, h->Ntree, h->L, n, h->cfg.capacity);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

/* The handle as a POSITIONS-ONLY prioritized index (the CNN loop keeps its transitions in the frame ring of a dqn_cnn_handle): n new
 * positions enter at the running max priority exactly as dqn_replay_add's would -- counter, size, leaves -- without any row data. */
extern "C" int dqn_per_index_advance(dqn_handle *h, int32_t n, void *stream) {
    REQUIRE(h, "null argument");
    if (!h->cfg.use_per) return fail(DQN_ERR_STATE, "PER call on a handle created with use_per=0");
    REQUIRE(n >= 1 && n <= h->cfg.capacity, "dqn_per_index_advance: n=%d must be in [1, capacity]", n);
    hipStream_t st = (hipStream_t)stream;
    launch_per_add(st, h->st, h->tree, h->Ntree, h->L, n, h->cfg.capacity, 1);      // (one launch: counter, size and leaves)
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

/* One launch per vector env step for a positions-only index whose rows are overwritten in place (the frame ring of configs[4]): the
 * zero_n positions from zero_first on (mod capacity) get priority 0 -- out of the draw until their successors exist -- and then n new
 * positions enter at the running max priority as in dqn_per_index_advance (n = 0: none). Same tree as dqn_per_set_sorted(zeros) +
 * dqn_per_index_advance, which were three launches and a host-side index add. */
extern "C" int dqn_per_index_step(dqn_handle *h, int32_t n, int64_t zero_first, int32_t zero_n, void *stream) {
    REQUIRE(h, "null argument");
    if (!h->cfg.use_per) return fail(DQN_ERR_STATE, "PER call on a handle created with use_per=0");
    REQUIRE(n >= 0 && n <= h->cfg.capacity && zero_n >= 0 && zero_n <= h->cfg.capacity && zero_first >= 0 && (n > 0 || zero_n > 0),
            "dqn_per_index_step: n=%d, zero_n=%d must be in [0, capacity], not both 0", n, zero_n);
    launch_per_add((hipStream_t)stream, h->st, h->tree, h->Ntree, h->L, n, h->cfg.capacity, 1, zero_first, zero_n);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

extern "C" int dqn_replay_size_host(dqn_handle *h, int64_t *size, int64_t *counter) {
    REQUIRE(h, "null argument");
    DqnState s;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&s, h->st, sizeof(s), hipMemcpyDeviceToHost));
    if (size) *size = s.size;
    if (counter) *counter = (int64_t)s.ring_counter;
    return DQN_OK;
}

extern "C" int dqn_replay_sample_uniform(dqn_handle *h, int32_t B, uint64_t seed, uint64_t ctr,
                                         const int32_t *idx_in, float *s, int32_t *a, float *r,
                                         float *s2, uint8_t *d, int32_t *idx_out, void *stream) {
    REQUIRE(h && s && a && r && s2 && d, "null argument");
    REQUIRE(B >= 1, "B must be >= 1");
    launch_sample_uniform((hipStream_t)stream, h->st, h->states, h->actions, h->rewards, h->observations,
                          h->dones, h->cfg.obs_dim, B, seed, ctr, 0, idx_in, s, a, r, s2, d, idx_out);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

extern "C" int dqn_per_sample(dqn_handle *h, int32_t B, float beta, uint64_t seed, uint64_t ctr,
                              float *s, int32_t *a, float *r, float *s2, uint8_t *d,
                              int32_t *idx, float *isw, void *stream) {
    REQUIRE(h && s && a && r && s2 && d && idx && isw, "null argument");
    if (!h->cfg.use_per) return fail(DQN_ERR_STATE, "dqn_per_sample on a handle created with use_per=0");
    REQUIRE(B >= 1 && B <= h->cfg.max_batch, "B=%d exceeds max_batch=%d", B, h->cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(h->wmax_tmp, 0, 4, st));
    arm(h);
    launch_per_sample(st, h->st, h->tree, h->Ntree, h->L, h->states, h->actions, h->rewards, h->observations,
                      h->dones, h->cfg.obs_dim, B, beta, seed, ctr, 0, s, a, r, s2, d, idx, h->bw_raw, h->wmax_tmp, h->num_cus, true);
    mark(h, st, "per_sample");
    launch_isw_normalize(st, h->bw_raw, B, isw, h->st, h->wmax_tmp);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

static int per_write(dqn_handle *h, const int32_t *idx, const float *val, int32_t B, int mode, void *stream) {
    REQUIRE(h && idx && val, "null argument");
    if (!h->cfg.use_per) return fail(DQN_ERR_STATE, "PER call on a handle created with use_per=0");
    REQUIRE(B >= 1, "B must be >= 1");
    launch_per_write((hipStream_t)stream, h->st, h->tree, h->stamp, h->Ntree, h->L, idx, val, B, mode,
                     h->cfg.per_alpha, h->cfg.per_eps, h->cfg.capacity);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}
extern "C" int dqn_per_update(dqn_handle *h, const int32_t *idx, const float *td_abs, int32_t B, void *stream) {
    return per_write(h, idx, td_abs, B, 1, stream);
}
extern "C" int dqn_per_set(dqn_handle *h, const int32_t *idx, const float *prio, int32_t B, void *stream) {
    return per_write(h, idx, prio, B, 0, stream);
}
static int per_write_sorted(dqn_handle *h, const int32_t *idx, const float *val, int32_t B, int mode, void *stream) {
    REQUIRE(h && idx && val, "null argument");
    if (!h->cfg.use_per) return fail(DQN_ERR_STATE, "PER call on a handle created with use_per=0");
    REQUIRE(B >= 1, "B must be >= 1");
    launch_per_write_sorted((hipStream_t)stream, h->st, h->tree, h->Ntree, h->L, idx, val, B, mode,
                            h->cfg.per_alpha, h->cfg.per_eps, h->pw_part, h->pw_force);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}
extern "C" int dqn_per_update_sorted(dqn_handle *h, const int32_t *idx, const float *td_abs, int32_t B, void *stream) {
    return per_write_sorted(h, idx, td_abs, B, 1, stream);
}
extern "C" int dqn_per_set_sorted(dqn_handle *h, const int32_t *idx, const float *prio, int32_t B, void *stream) {
    return per_write_sorted(h, idx, prio, B, 0, stream);
}

// ------------------------------------------------------------------------------ network
static FwdPass make_pass(dqn_handle *h, int which_net, const float *x, float *q, float *feat, bool stash) {
    FwdPass p{};
    p.x = x;
    p.params = which_net == DQN_NET_TARGET ? h->target : h->params;
    p.pack = which_net == DQN_NET_TARGET ? h->pack_t : h->pack;
    p.q = q; p.feat = feat;
    if (stash) { p.px = h->px; p.ph1 = h->ph1; p.ph2 = h->ph2; }
    return p;
}

extern "C" int dqn_qnet_forward(dqn_handle *h, int which_net, const float *x, int32_t B,
                                float *q, float *feat, void *stream) {
    REQUIRE(h && x && q, "null argument");
    REQUIRE(which_net == DQN_NET_ONLINE || which_net == DQN_NET_TARGET, "bad net selector");
    REQUIRE(B >= 1, "B must be >= 1");
    FwdPass p = make_pass(h, which_net, x, q, feat, false);
    L_fwd(h, (hipStream_t)stream, &p, 1, B);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

extern "C" int dqn_td_targets(dqn_handle *h, const float *q, const float *next_q, const float *next_q_tm,
                              const int32_t *a, const float *r, const float *d, const float *isw,
                              float gamma, int32_t B, float *targets, float *td, float *dq, float *loss,
                              void *stream) {
    REQUIRE(h && q && next_q && next_q_tm && a && r && d, "null argument");
    REQUIRE(B >= 1 && B <= h->cfg.max_batch, "B=%d exceeds max_batch=%d", B, h->cfg.max_batch);
    launch_td((hipStream_t)stream, q, next_q, next_q_tm, a, r, d, isw, gamma, B, h->cfg.num_actions,
              targets, td, dq, loss, h->scratch);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

extern "C" int dqn_q_targets(dqn_handle *h, const float *s, const int32_t *a, const float *r,
                             const float *s2, const float *d, int32_t B, float *targets, void *stream) {
    REQUIRE(h && s && a && r && s2 && d && targets, "null argument");
    REQUIRE(B >= 1 && B <= h->cfg.max_batch, "B=%d exceeds max_batch=%d", B, h->cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    FwdPass p[3] = { make_pass(h, DQN_NET_ONLINE, s, h->q, nullptr, false),      // :52
                     make_pass(h, DQN_NET_ONLINE, s2, h->nq, nullptr, false),    // :53
                     make_pass(h, DQN_NET_TARGET, s2, h->nt, nullptr, false) };  // :54
    L_fwd(h, st, p, 3, B);
    launch_td(st, h->q, h->nq, h->nt, a, r, d, nullptr, h->cfg.gamma, B, h->cfg.num_actions, targets,
              nullptr, nullptr, nullptr, h->scratch);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

extern "C" int dqn_loss(dqn_handle *h, const float *s, const float *targets, const float *isw,
                        int32_t B, float *loss, void *stream) {
    REQUIRE(h && s && targets && loss, "null argument");
    REQUIRE(B >= 1 && B <= h->cfg.max_batch, "B=%d exceeds max_batch=%d", B, h->cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    FwdPass p = make_pass(h, DQN_NET_ONLINE, s, h->q, nullptr, false);
    L_fwd(h, st, &p, 1, B);
    launch_loss(st, h->q, targets, isw, B, h->cfg.num_actions, loss);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

extern "C" int dqn_grads(dqn_handle *h, const float *s, const float *targets, const float *isw,
                         int32_t B, float *loss, void *stream) {
    REQUIRE(h && s && targets, "null argument");
    REQUIRE(B >= 1 && B <= h->cfg.max_batch, "B=%d exceeds max_batch=%d", B, h->cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    FwdPass p = make_pass(h, DQN_NET_ONLINE, s, h->q, nullptr, true);
    BwdArgs g{};
    g.q = h->q; g.targets = targets; g.isw = isw; g.gamma = h->cfg.gamma;
    g.ph1 = h->ph1; g.ph2 = h->ph2; g.pack = h->pack;
    g.pdz1 = h->pdz1; g.pdz2 = h->pdz2; g.pdz3 = h->pdz3; g.loss_part = h->loss_part;
    if (use_big(h, B)) {                                           // forward + row backward of a row tile in one workgroup, split-K dW
        (h->bf16 ? launch_big16_rows_bwd : launch_big_rows_bwd)(st, h->m, &p, 1, B, g, h->px, h->ph1, h->ph2, h->big_colsum, h->st, h->num_cus);
        (h->bf16 ? launch_big16_dw : launch_big_dw)(st, h->m, h->px, h->ph1, h->ph2, h->pdz1, h->pdz2, h->pdz3, B, h->big_slab, h->big_colsum, h->grad,
                      h->loss_part, loss ? loss : h->loss_dev, h->st, 0, AdamArgs{}, h->num_cus);
        HIP_TRY(hipGetLastError());
        return DQN_OK;
    }
    L_fwd(h, st, &p, 1, B);
    L_bwd(h, st, g, B);
    L_dw(h, st, B, loss ? loss : h->loss_dev, 0, AdamArgs{}, PwArgs{});
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

static void enqueue_adam(dqn_handle *h, hipStream_t st) {
    if (h->bf16) launch_adam_bf16(st, h->m, h->st, h->params, h->grad, h->mu, h->nu, h->pack, h->cfg.optimizer == DQN_OPT_ADAMW,
                                  h->cfg.b1, h->cfg.b2, h->cfg.eps, h->cfg.weight_decay, 1.0f / (float)h->world, h->pack_act);
    else launch_adam(st, h->m, h->st, h->params, h->grad, h->mu, h->nu, h->pack, h->cfg.optimizer == DQN_OPT_ADAMW,
                     h->cfg.b1, h->cfg.b2, h->cfg.eps, h->cfg.weight_decay, 1.0f / (float)h->world);
}

extern "C" int dqn_optimizer_step(dqn_handle *h, void *stream) {
    REQUIRE(h, "null argument");
    enqueue_adam(h, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

extern "C" int dqn_train_step(dqn_handle *h, const float *s, const float *targets, int32_t B, void *stream) {
    int rc = dqn_grads(h, s, targets, nullptr, B, nullptr, stream);
    if (rc != DQN_OK) return rc;
    return dqn_optimizer_step(h, stream);
}

// ------------------------------------------------------------------------ fused update
static AdamArgs adam_args(dqn_handle *h) {
    return AdamArgs{h->params, h->mu, h->nu, h->pack, h->cfg.optimizer == DQN_OPT_ADAMW, h->cfg.b1, h->cfg.b2,
                    h->cfg.eps, h->cfg.weight_decay, 1.0f / (float)h->world, h->pack_act};
}

static void enqueue_per_writeback(dqn_handle *h, int B, hipStream_t st) {
    // the batch indices come from dqn_per_sample's stratified descent: non-decreasing
    arm(h);
    launch_per_write_sorted(st, h->st, h->tree, h->Ntree, h->L, h->bidx, h->btd_abs, B, 1,
                            h->cfg.per_alpha, h->cfg.per_eps, h->pw_part, h->pw_force);
    mark(h, st, "per_update");
}

// sample -> three forwards -> TD / row backward -> weight gradients.
// fuse_adam: optimizer applied in the dW epilogue (single GPU). fork: run the PER write-back on a
// parallel branch of the captured graph (it only needs idx and |delta|), joined by join_update().
static void enqueue_backward(dqn_handle *h, int B, hipStream_t st, bool fuse_adam = false, bool fuse_pw = false,
                             bool defer_top = false, bool presampled = false) {
    if (use_big(h, B) && !presampled) {
        // large batches (dqn_net_big.hip): the batch is drawn and gathered by the stand-alone sampler, then ONE launch takes
        // each 64-row tile through the three forwards, the TD rule and the row backward, then the split-K weight gradients
        // with the optimizer in the reduction, then the priority write-back
        if (h->cfg.use_per) {
            arm(h);
            launch_per_sample(st, h->st, h->tree, h->Ntree, h->L, h->states, h->actions, h->rewards, h->observations, h->dones,
                              h->cfg.obs_dim, B, 0.0f, h->cfg.seed, 0, 1, h->bs, h->ba, h->br, h->bs2, h->bd, h->bidx, h->bw_raw,
                              reinterpret_cast<unsigned int *>(&h->st->wmax), h->num_cus, false);
            mark(h, st, "per_sample");
        } else {
            launch_sample_uniform(st, h->st, h->states, h->actions, h->rewards, h->observations, h->dones, h->cfg.obs_dim, B,
                                  h->cfg.seed, 0, 1, nullptr, h->bs, h->ba, h->br, h->bs2, h->bd, h->bidx);
        }
        FwdPass p[3] = { make_pass(h, DQN_NET_ONLINE, h->bs2, h->nq, nullptr, false),     // :53
                         make_pass(h, DQN_NET_TARGET, h->bs2, h->nt, nullptr, false),     // :54
                         make_pass(h, DQN_NET_ONLINE, h->bs, h->q, nullptr, false) };     // :52 (= pred of :35)
        BwdArgs g{};
        g.a = h->ba; g.r = h->br; g.d_u8 = h->bd; g.w_raw = h->cfg.use_per ? h->bw_raw : nullptr;
        g.gamma = h->n_step > 1 ? h->gamma_n : h->cfg.gamma;
        g.pack = h->pack; g.pdz1 = h->pdz1; g.pdz2 = h->pdz2; g.pdz3 = h->pdz3;
        g.td = h->btd; g.td_abs = h->btd_abs; g.isw_out = h->bisw; g.loss_part = h->loss_part;
        arm(h);
        if (h->bf16) {
            // bf16: online(s'), target(s') by the forward kernel (two workgroups per CU), their Q rows through HBM (32 B per row)
            launch_big16_forward(st, h->m, p, 2, B, h->num_cus);
            mark(h, st, "big_fwd2");
            g.nq = h->nq; g.nt = h->nt;
            arm(h);
            launch_big16_rows_bwd(st, h->m, p + 2, 1, B, g, h->px, h->ph1, h->ph2, h->big_colsum, h->st, h->num_cus);
        } else launch_big_rows_bwd(st, h->m, p, 3, B, g, h->px, h->ph1, h->ph2, h->big_colsum, h->st, h->num_cus);
        mark(h, st, "big_rows_fwd3_bwd");
        arm(h);
        (h->bf16 ? launch_big16_dw : launch_big_dw)(st, h->m, h->px, h->ph1, h->ph2, h->pdz1, h->pdz2, h->pdz3, B, h->big_slab, h->big_colsum, h->grad,
                      h->loss_part, h->loss_dev, h->st, 1, fuse_adam ? adam_args(h) : AdamArgs{}, h->num_cus);
        mark(h, st, "big_dw");
        if (fuse_pw && h->cfg.use_per) enqueue_per_writeback(h, B, st);
        (void)defer_top;
        return;
    }
    // q_agent.py:147-153 sample_batch + :159-165 compute_q_targets' three forwards, ONE launch: every forward
    // workgroup draws its own 16 rows (stratified PER descent or uniform Philox index) and reads them from the ring
    SampleArgs sm{};
    sm.st = h->st; sm.tree = h->cfg.use_per ? h->tree : nullptr; sm.N = h->Ntree; sm.L = h->L;
    sm.states = h->states; sm.observations = h->observations; sm.rewards = h->rewards; sm.actions = h->actions;
    sm.dones = h->dones; sm.seed = h->cfg.seed;
    sm.idx = h->bidx; sm.a = h->ba; sm.r = h->br; sm.w_raw = h->bw_raw; sm.d = h->bd;
    sm.pre = presampled ? 1 : 0;         // idx / w_raw / wmax drawn by the actor launch that precedes (dqn_actor.hip)
    FwdPass p[3] = { make_pass(h, DQN_NET_ONLINE, nullptr, h->q, nullptr, true),
                     make_pass(h, DQN_NET_ONLINE, nullptr, h->nq, nullptr, false),
                     make_pass(h, DQN_NET_TARGET, nullptr, h->nt, nullptr, false) };
    p[0].src = 1; p[1].src = 2; p[2].src = 2;
    // targets + loss gradient + row backward (q_learning_functions.py:55-60, :35-36, :23)
    BwdArgs g{};
    g.q = h->q; g.nq = h->nq; g.nt = h->nt; g.a = h->ba; g.r = h->br; g.d_u8 = h->bd;
    g.w_raw = h->cfg.use_per ? h->bw_raw : nullptr;
    g.gamma = h->n_step > 1 ? h->gamma_n : h->cfg.gamma;       // rows of the n-step actor bootstrap with gamma^n
    g.ph1 = h->ph1; g.ph2 = h->ph2; g.pack = h->pack;
    g.pdz1 = h->pdz1; g.pdz2 = h->pdz2; g.pdz3 = h->pdz3;
    g.td = h->btd; g.td_abs = h->btd_abs; g.isw_out = h->bisw; g.loss_part = h->loss_part;
    // whole grid resident (3 * tiles <= 256 workgroups) and the batch weights final before the launch (drawn by the
    // actor launch, or uniform replay): the row backward rides in the forward launch (pass-0 workgroups go on with it)
    const int grid_tiles = h->bf16 ? 2 * ((B + 31) / 32) : (B + 15) / 16;
    const bool fuse_rows = 3 * grid_tiles <= h->num_cus && (presampled || !h->cfg.use_per) && !h->no_handover;
    if (fuse_rows) p[0].q = nullptr;        // the pass-0 workgroup keeps its Q rows in registers for the TD rule; nobody else reads them
    arm(h);
    L_fwd(h, st, p, 3, B, &sm, fuse_rows ? &g : nullptr);
    mark(h, st, fuse_rows ? "sample_fwd_x3_bwd" : "sample_fwd_x3");
    if (!fuse_rows) {
        arm(h);
        L_bwd(h, st, g, B);
        mark(h, st, "td_bwd_rows");
    }
    PwArgs pw{};
    if (fuse_pw && h->cfg.use_per)
        pw = PwArgs{h->tree, h->Ntree, h->L, h->bidx, h->btd_abs, B, h->cfg.per_alpha, h->cfg.per_eps};
    arm(h);
    L_dw(h, st, B, h->loss_dev, 1, fuse_adam ? adam_args(h) : AdamArgs{}, pw);
    mark(h, st, fuse_adam ? (pw.tree ? "dw_adam_perwrite" : "dw_adam") : (pw.tree ? "dw_perwrite" : "dw"));
    // the dense top of the tree: its own launch, unless the caller defers it into the next actor launch
    if (pw.tree && !defer_top) { arm(h); launch_per_top(st, h->st, h->tree, h->L); mark(h, st, "per_top"); }
}

// second half of an update when a gradient all-reduce sits in between: only the optimizer is left (the PER
// write-back does not depend on the all-reduce and already rode in the backward half's dW launch)
static void enqueue_apply(dqn_handle *h, int B, hipStream_t st) {
    (void)B;
    arm(h);
    enqueue_adam(h, st);                                           // q_learning_functions.py:24-25
    mark(h, st, "adam");
}

// the whole Agent._step. Single GPU: optimizer fused into the dW epilogue, and the PER write-back waves
// ride in the same launch as surplus workgroups (a forked graph branch measured slower: cross-queue
// dependencies cost more than they hide).
static void enqueue_update(dqn_handle *h, int B, hipStream_t st, bool capturing, bool defer_top = false,
                           bool presampled = false) {
    (void)capturing;
    if (h->world == 1) {
        enqueue_backward(h, B, st, true, true, defer_top, presampled);
    } else {
        enqueue_backward(h, B, st, false, true, false, presampled);
        enqueue_apply(h, B, st);
    }
}

static EnvArgs env_args(dqn_handle *h, int n_envs, bool rebuild_top) {
    EnvArgs e{};
    e.st = h->st; e.states = h->states; e.actions = h->actions; e.rewards = h->rewards; e.observations = h->observations;
    e.dones = h->dones; e.cap = h->cfg.capacity; e.tree = h->cfg.use_per ? h->tree : nullptr; e.Nt = h->Ntree; e.L = h->L;
    e.env_obs = h->env_obs; e.seed = h->cfg.seed; e.p_done = h->p_done; e.n = n_envs;
    e.kind = h->env_kind; e.max_steps = h->env_max_steps; e.env_t = h->env_t; e.term_reward = h->env_term_reward;
    e.time_feature = h->env_time_feature ? 1 : 0;
    e.rebuild_top = (rebuild_top && h->cfg.use_per) ? 1 : 0;
    e.n_step = h->n_step; e.hist_stride = h->Bp; e.gamma = h->cfg.gamma;
    e.hist_s = h->hist_s; e.hist_r = h->hist_r; e.hist_a = h->hist_a; e.hist_d = h->hist_d;
    return e;
}

// q_agent.py:176-183 x T in ONE launch (exact-f32 path, dqn_actor.hip): T dependent vector env steps with the weights
// resident in registers, the T*n new leaves inserted by a side workgroup, and -- presample_B > 0 -- the stratified PER
// draw of the update that follows done by further side workgroups once the leaves are in.
static bool actor_multi_ok(dqn_handle *h, int n_envs, int T) {
    return T >= 1 && (long long)T * n_envs <= h->cfg.capacity && actor_multi_supported(h->m, n_envs, T);
}
static bool enqueue_actor_multi(dqn_handle *h, int T, int n_envs, hipStream_t st, bool rebuild_top, int presample_B) {
    const EnvArgs e = env_args(h, n_envs, rebuild_top);
    SampleArgs sm{};
    sm.st = h->st; sm.tree = h->tree; sm.N = h->Ntree; sm.L = h->L; sm.seed = h->cfg.seed;
    sm.idx = h->bidx; sm.w_raw = h->bw_raw;
    arm(h);
    // bf16 mode: the same kernel on v_mfma_f32_4x4x4_16b_bf16 (weights / activations rounded to bf16 in registers from
    // the f32 shadows in pack_act: chains a quarter as long)
    if (presample_B > 0 && use_big(h, presample_B)) presample_B = 0;   // large batches draw their rows in the update (k_per_sample2)
    const bool pre = launch_actor_multi(st, h->m, e, T, h->params, h->bf16 ? h->pack_act : h->pack, h->env_a,
                                        (h->cfg.use_per && !h->no_handover ? presample_B : 0), &sm, h->bf16 && !h->f32_actor,
                                        h->num_cus, h->no_actor16);
    mark(h, st, "actor_steps");
    return pre;                                                    // the PER batch of the next update is drawn
}

// q_agent.py:176-183 for n_envs device-resident envs, one vector step: k_actor with T = 1 (forward + epsilon-greedy policy +
// env transition + ring insert per 4-env workgroup, leaves inserted by the tree workgroup)
static void enqueue_actor(dqn_handle *h, int n_envs, hipStream_t st, bool rebuild_top = false) {
    enqueue_actor_multi(h, 1, n_envs, st, rebuild_top, 0);      // (ObsWrapper's time-fraction column: kept inside the kernel, r03)
}

// capture `body` into an executable graph on the caller's stream (non-null streams only). body_err: set non-zero by a body
// whose capture must not be kept (a collective that failed to enqueue): the graph is dropped -- never cached, never
// launched -- so that no later call replays an update without its all-reduce.
template <class F>
static int run_captured(dqn_handle *h, hipGraphExec_t *slot, hipStream_t st, F body, const int *body_err = nullptr) {
    if (!st || h->profiling) { body(); HIP_TRY(hipGetLastError()); return DQN_OK; }   // default stream / profiling: eager
    if (!*slot) {
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        body();
        hipError_t e = hipStreamEndCapture(st, &graph);
        if (e != hipSuccess) return fail(DQN_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        if (body_err && *body_err) { (void)hipGraphDestroy(graph); return DQN_ERR_COMM; }
        e = hipGraphInstantiate(slot, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { *slot = nullptr; return fail(DQN_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(hipGraphLaunch(*slot, st));
    return DQN_OK;
}

static int check_B(dqn_handle *h, int32_t B) {
    REQUIRE(h, "null argument");
    REQUIRE(B >= 1 && B <= h->cfg.max_batch, "B=%d exceeds max_batch=%d", B, h->cfg.max_batch);
    return DQN_OK;
}

extern "C" int dqn_update_fused(dqn_handle *h, int32_t B, void *stream) {
    int rc = check_B(h, B); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    return run_captured(h, &h->graphs[B].fused, st, [&] { enqueue_update(h, B, st, st && !h->profiling); });
}
extern "C" int dqn_update_backward(dqn_handle *h, int32_t B, void *stream) {
    int rc = check_B(h, B); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    return run_captured(h, &h->graphs[B].bwd, st, [&] { enqueue_backward(h, B, st, false, true); });
}
extern "C" int dqn_update_apply(dqn_handle *h, int32_t B, void *stream) {
    int rc = check_B(h, B); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    return run_captured(h, &h->graphs[B].apply, st, [&] { enqueue_apply(h, B, st); });
}

extern "C" int dqn_act(dqn_handle *h, const float *s, int32_t n, float epsilon, uint64_t seed,
                       uint64_t ctr, int32_t *actions, void *stream) {
    REQUIRE(h && s && actions, "null argument");
    REQUIRE(n >= 1 && n <= h->cfg.max_batch, "n=%d exceeds max_batch=%d", n, h->cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    FwdPass p = make_pass(h, DQN_NET_ONLINE, s, nullptr, nullptr, false);
    p.act_out = actions; p.act_state = nullptr; p.act_eps = epsilon; p.act_seed = seed; p.act_ctr = ctr;
    L_fwd(h, st, &p, 1, n);
    HIP_TRY(hipGetLastError());
    return DQN_OK;
}

// ------------------------------------------------------------------- synthetic actor
extern "C" int dqn_set_epsilon(dqn_handle *h, float epsilon, void *stream) {
    REQUIRE(h, "null argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(&h->st->epsilon, &epsilon, 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DQN_OK;
}

extern "C" int dqn_env_reset(dqn_handle *h, const float *obs, int32_t n_envs, float p_done, void *stream) {
    REQUIRE(h && obs, "null argument");
    REQUIRE(n_envs >= 1 && n_envs <= h->cfg.max_batch, "n_envs=%d exceeds max_batch=%d", n_envs, h->cfg.max_batch);
    if (p_done != h->p_done) destroy_graphs(h);                   // p_done is a by-value argument of captured launches
    h->p_done = p_done;
    HIP_TRY(hipMemsetAsync(h->env_t, 0, (size_t)n_envs * 4, (hipStream_t)stream));
    HIP_TRY(hipMemsetAsync(&h->st->hist_steps, 0, sizeof(unsigned long long), (hipStream_t)stream));   // n-step history restarts
    HIP_TRY(hipMemcpyAsync(h->env_obs, obs, (size_t)n_envs * h->cfg.obs_dim * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DQN_OK;
}

extern "C" int dqn_env_config(dqn_handle *h, int32_t kind, int32_t max_steps, float term_reward) {
    REQUIRE(h, "null argument");
    REQUIRE(kind == DQN_ENV_SYNTHETIC || kind == DQN_ENV_CARTPOLE, "unknown env kind %d", kind);
    REQUIRE(kind != DQN_ENV_CARTPOLE || (h->cfg.obs_dim == 4 && h->cfg.num_actions == 2), "CartPole needs obs_dim 4, num_actions 2");
    REQUIRE(max_steps >= 1, "max_steps must be >= 1");
    REQUIRE(!(h->env_time_feature && kind != DQN_ENV_SYNTHETIC), "the time-fraction feature is built for the synthetic env only");
    h->env_kind = kind; h->env_max_steps = max_steps; h->env_term_reward = term_reward;
    destroy_graphs(h);                                           // env parameters are baked into captured launches
    return DQN_OK;
}

extern "C" int dqn_env_time_feature(dqn_handle *h, int32_t enable) {
    REQUIRE(h, "null argument");
    REQUIRE(!enable || (h->env_kind == DQN_ENV_SYNTHETIC && h->n_step <= 1 && h->cfg.obs_dim >= 2),
            "the time-fraction feature is built for the synthetic vector env with one-step returns (obs_dim counts the feature)");
    h->env_time_feature = enable != 0;
    destroy_graphs(h);
    return DQN_OK;
}

extern "C" int dqn_env_stats_host(dqn_handle *h, int64_t *episodes, int64_t *episode_steps) {
    REQUIRE(h, "null argument");
    DqnState s;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&s, h->st, sizeof(s), hipMemcpyDeviceToHost));
    if (episodes) *episodes = (int64_t)s.ep_count;
    if (episode_steps) *episode_steps = (int64_t)s.ep_steps;
    return DQN_OK;
}

extern "C" int dqn_actor_step(dqn_handle *h, int32_t n_envs, void *stream) {
    REQUIRE(h, "null argument");
    REQUIRE(n_envs >= 1 && n_envs <= h->cfg.max_batch && n_envs <= h->cfg.capacity,
            "n_envs=%d exceeds max_batch=%d or capacity", n_envs, h->cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    return run_captured(h, &h->graphs[-n_envs].actor, st, [&] { enqueue_actor(h, n_envs, st); });
}

extern "C" int dqn_actor_steps(dqn_handle *h, int32_t env_steps, int32_t n_envs, void *stream) {
    REQUIRE(h, "null argument");
    REQUIRE(env_steps >= 1 && env_steps <= 64, "env_steps out of range [1,64]");
    REQUIRE(n_envs >= 1 && n_envs <= h->cfg.max_batch && (long long)env_steps * n_envs <= h->cfg.capacity,
            "n_envs=%d exceeds max_batch=%d, or env_steps * n_envs exceeds the capacity", n_envs, h->cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    const std::vector<int> key{-2, env_steps, n_envs, 0};
    return run_captured(h, &h->loop_graphs[key], st, [&] {
        if (actor_multi_ok(h, n_envs, env_steps)) enqueue_actor_multi(h, env_steps, n_envs, st, false, 0);
        else for (int e = 0; e < env_steps; ++e) enqueue_actor(h, n_envs, st);
    });
}

/* env_steps vector env steps + the backward half of one update as ONE graph launch (data-parallel step:
 * this, then the gradient all-reduce, then dqn_update_apply) */
extern "C" int dqn_actor_backward(dqn_handle *h, int32_t env_steps, int32_t n_envs, int32_t B, void *stream) {
    int rc = check_B(h, B); if (rc) return rc;
    REQUIRE(env_steps >= 0 && env_steps <= 64, "env_steps out of range");
    REQUIRE(env_steps == 0 || (n_envs >= 1 && n_envs <= h->cfg.max_batch && n_envs <= h->cfg.capacity),
            "n_envs=%d exceeds max_batch=%d or capacity", n_envs, h->cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    const std::vector<int> key{-1, env_steps, n_envs, B};
    return run_captured(h, &h->loop_graphs[key], st, [&] {
        if (env_steps > 0 && actor_multi_ok(h, n_envs, env_steps)) {
            const bool pre = enqueue_actor_multi(h, env_steps, n_envs, st, false, B);
            enqueue_backward(h, B, st, false, true, false, pre);
            return;
        }
        for (int e = 0; e < env_steps; ++e) enqueue_actor(h, n_envs, st);
        enqueue_backward(h, B, st, false, true);
    });
}

/* n_iters x (env_steps vector env steps + one update) as ONE graph launch. With a communicator (dqn_comm_init) every
 * update is: backward half -> in-place RCCL sum all-reduce of the flat gradient, captured in the same graph -> optimizer */
extern "C" int dqn_train_iters(dqn_handle *h, int32_t n_iters, int32_t env_steps, int32_t n_envs, int32_t B,
                               void *stream) {
    int rc = check_B(h, B); if (rc) return rc;
    REQUIRE(n_iters >= 1 && n_iters <= 256 && env_steps >= 0 && env_steps <= 64, "n_iters / env_steps out of range");
    REQUIRE(env_steps == 0 || (n_envs >= 1 && n_envs <= h->cfg.max_batch && n_envs <= h->cfg.capacity),
            "n_envs=%d exceeds max_batch=%d or capacity", n_envs, h->cfg.max_batch);
    if (h->world != 1 && !h->comm)
        return fail(DQN_ERR_STATE, "dqn_train_iters with world_size > 1 needs dqn_comm_init (the all-reduce is captured in the "
                                   "graph); without a communicator use dqn_actor_backward / your all-reduce / dqn_update_apply");
    hipStream_t st = (hipStream_t)stream;
    const bool dp = h->comm != nullptr;                                   // also a 1-rank communicator (rehearsal)
    const std::vector<int> key{n_iters, env_steps, n_envs, B};
    int comm_err = 0;
    rc = run_captured(h, &h->loop_graphs[key], st, [&] {
        // one update: single learner = everything fused into the dW launch; data-parallel = backward half (with the
        // PER write-back), the gradient all-reduce, the optimizer
        auto update = [&](bool defer_top, bool presampled) {
            if (!dp) { enqueue_update(h, B, st, st && !h->profiling, defer_top, presampled); return; }
            enqueue_backward(h, B, st, false, true, defer_top, presampled);
            hipEvent_t ev0 = nullptr, ev1 = nullptr;                       // profiling: the collective is not one of our launches
            if (h->profiling) { arm(h); ev0 = g_prof_ev0; ev1 = g_prof_ev1; g_prof_ev0 = nullptr; g_prof_ev1 = nullptr; }
            if (ev0) (void)hipEventRecord(ev0, st);
            const int e = g_rccl.AllReduce(h->grad, h->grad, (size_t)h->m.P, /*ncclFloat32*/ 7, /*ncclSum*/ 0, h->comm, st);
            if (e) comm_err = e;
            if (ev1) (void)hipEventRecord(ev1, st);
            mark(h, st, "allreduce");
            enqueue_apply(h, B, st);
        };
        // Inside the loop the rebuild of the tree top after an update is deferred into the actor launch of the next
        // iteration (its tree workgroup is the next reader and has slack); the last update keeps k_per_top so that the
        // tree is consistent when the graph ends.
        bool can_defer = env_steps > 0 && h->cfg.use_per;
        if (env_steps > 0 && actor_multi_ok(h, n_envs, env_steps)) {
            // the env_steps actor steps of an iteration are ONE launch, which also inserts their leaves and draws the
            // update's PER batch on side workgroups
            for (int it = 0; it < n_iters; ++it) {
                const bool pre = enqueue_actor_multi(h, env_steps, n_envs, st, can_defer && it > 0, B);
                update(can_defer && it + 1 < n_iters, pre);
            }
            return;
        }
        can_defer = can_defer && !h->bf16;                            // (per-step bf16 actor launches: measured -1 %)
        for (int it = 0; it < n_iters; ++it) {
            for (int e = 0; e < env_steps; ++e) enqueue_actor(h, n_envs, st, can_defer && it > 0 && e == 0);
            update(can_defer && it + 1 < n_iters, false);
        }
    }, &comm_err);
    if (comm_err) return fail(DQN_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(comm_err) : "?");
    return rc;
}

extern "C" int dqn_sync_target(dqn_handle *h, void *stream) {
    REQUIRE(h, "null argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(h->target, h->params, h->m.P * 4, hipMemcpyDeviceToDevice, st));   // q_agent.py:144
    HIP_TRY(hipMemcpyAsync(h->pack_t, h->pack, h->pack_bytes, hipMemcpyDeviceToDevice, st));
    return DQN_OK;
}

// --------------------------------------------------------------------------- profiling
extern "C" int dqn_profile_begin(dqn_handle *h, void *stream) {
    REQUIRE(h, "null argument");
    (void)stream;
    h->profiling = true; h->ev_used = 0; h->ev_names.clear();
    return DQN_OK;
}

extern "C" int dqn_profile_end(dqn_handle *h, void *stream, char *names, int32_t name_stride, float *ms,
                               int32_t max_entries, int32_t *count) {
    REQUIRE(h && names && ms && count, "null argument");
    h->profiling = false;
    g_prof_ev0 = nullptr; g_prof_ev1 = nullptr;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    int n = (int)h->ev_names.size();
    if (n > max_entries) n = max_entries;
    for (int i = 0; i < n; ++i) {
        HIP_TRY(hipEventElapsedTime(&ms[i], h->events[2 * i], h->events[2 * i + 1]));
        snprintf(names + (size_t)i * name_stride, name_stride, "%s", h->ev_names[i]);
    }
    *count = n;
    return DQN_OK;
}

// -------------------------------------------------------------------------------- RCCL
static int load_rccl() {
    if (g_rccl.lib) return DQN_OK;
    void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(DQN_ERR_COMM, "cannot load librccl.so: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(lib, "ncclCommCount");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(DQN_ERR_COMM, "librccl.so lacks a required symbol");
    g_rccl.lib = lib;
    return DQN_OK;
}

// the same library for the CNN handle's communicator (dqn_cnn.hip): thin wrappers so that one dlopen serves both
int dqn_rccl_comm_init(void **comm, const void *unique_id_128, int rank, int world) {
    int rc = load_rccl(); if (rc) return rc;
    NcclId id;
    memcpy(&id, unique_id_128, 128);
    const int e = g_rccl.CommInitRank(comm, world, id, rank);
    if (e) return fail(DQN_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return DQN_OK;
}
int dqn_rccl_allreduce_sum_f32(void *comm, float *buf, size_t n, hipStream_t st) {
    const int e = g_rccl.AllReduce(buf, buf, n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, st);
    if (e) return fail(DQN_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return DQN_OK;
}
void dqn_rccl_comm_destroy(void *comm) { if (comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm); }
int dqn_rccl_comm_count(void *comm, int *n) {
    if (!g_rccl.CommCount) return fail(DQN_ERR_COMM, "librccl.so lacks ncclCommCount");
    const int e = g_rccl.CommCount(comm, n);
    if (e) return fail(DQN_ERR_COMM, "ncclCommCount: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return DQN_OK;
}

extern "C" int dqn_comm_unique_id(void *unique_id_128) {
    REQUIRE(unique_id_128, "null argument");
    int rc = load_rccl(); if (rc) return rc;
    const int e = g_rccl.GetUniqueId(unique_id_128);
    if (e) return fail(DQN_ERR_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return DQN_OK;
}

extern "C" int dqn_comm_init(dqn_handle *h, const void *unique_id_128, int32_t rank, int32_t world) {
    REQUIRE(h && unique_id_128, "null argument");
    REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank/world");
    int rc = load_rccl(); if (rc) return rc;
    NcclId id;
    memcpy(&id, unique_id_128, 128);
    const int e = g_rccl.CommInitRank(&h->comm, world, id, rank);
    if (e) return fail(DQN_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    h->rank = rank; h->world = world; h->cfg.world_size = world;
    destroy_graphs(h);                                        // grad scale is baked into captured launches
    return DQN_OK;
}

extern "C" int dqn_allreduce_grads(dqn_handle *h, void *stream) {
    REQUIRE(h, "null argument");
    if (!h->comm) return fail(DQN_ERR_STATE, "dqn_allreduce_grads before dqn_comm_init");
    // one in-place sum all-reduce of the flat f32 gradient (P elements); /world is in the optimizer
    const int e = g_rccl.AllReduce(h->grad, h->grad, (size_t)h->m.P, /*ncclFloat32*/ 7, /*ncclSum*/ 0, h->comm,
                                   (hipStream_t)stream);
    if (e) return fail(DQN_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return DQN_OK;
}

extern "C" int dqn_comm_count_host(dqn_handle *h, int32_t *ranks) {
    REQUIRE(h && ranks, "null argument");
    *ranks = 0;
    if (!h->comm) return DQN_OK;
    if (!g_rccl.CommCount) return fail(DQN_ERR_COMM, "librccl.so lacks ncclCommCount");
    int n = 0;
    const int e = g_rccl.CommCount(h->comm, &n);
    if (e) return fail(DQN_ERR_COMM, "ncclCommCount: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    *ranks = n;
    return DQN_OK;
}

/* Diagnostic (tests): with on != 0 the partner passes of the fused forward launch no longer count themselves in, so every
 * pass-0 workgroup sits out its bounded wait (DQN_WAIT_TICKS, 0.2 s), gives up, bumps the error count and poisons the loss --
 * the path a partitioned / shared device would take. Captured launches are dropped (the switch is a kernel argument). */
extern "C" int dqn_debug_withhold_handover(dqn_handle *h, int32_t on) {
    REQUIRE(h, "null handle");
    HIP_TRY(hipDeviceSynchronize());
    destroy_graphs(h);
    h->withhold = on ? 1 : 0;
    return DQN_OK;
}

/* After dqn_device_errors_host has reported give-ups: bring the hand-over words back to their initial state (per-tile
 * arrival / consumed counters, arrival ticket, fill count, error count) so that the handle can be used again. The results of
 * the launches that timed out stay what they are (NaN losses; parameters updated from incomplete rows): reload them. */
extern "C" int dqn_clear_device_errors(dqn_handle *h) {
    REQUIRE(h, "null handle");
    HIP_TRY(hipDeviceSynchronize());
    if (h->tile_cnt) HIP_TRY(hipMemset(h->tile_cnt, 0, (size_t)h->tile_stride * 2 * 4));
    const unsigned int z = 0;
    HIP_TRY(hipMemcpy(&h->st->err_count, &z, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&h->st->arrive, &z, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&h->st->fill_cnt, &z, 4, hipMemcpyHostToDevice));
    return DQN_OK;
}

extern "C" int dqn_device_errors_host(dqn_handle *h, int64_t *count) {
    REQUIRE(h && count, "null argument");
    unsigned int c = 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&c, &h->st->err_count, 4, hipMemcpyDeviceToHost));
    *count = (int64_t)c;
    return DQN_OK;
}
