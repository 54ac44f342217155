// csrc/dqn_per_device.h -- device code of the sorted priority write-back, shared by dqn_replay.hip
// (k_per_write_sorted) and dqn_net.hip (surplus workgroups of k_dw).
#pragma once
#include "dqn_device.h"

// The tree is cut at depth TOP = min(L, PW_TOP): subtrees below the cut hold 2^(L-TOP) leaves (64 at L = 20), so
// that the <= 64-items-per-wave register path almost always applies even when the newest leaves (inserted at the
// running max priority, contiguous) attract many samples; the dense top is rebuilt by k_per_top.
#define PW_TOP 14
#define PW_BOT 21            // bottom levels handled below the dense top image: L - PW_TOP <= 21 (L <= 31)

__device__ __forceinline__ int shfl_i(int v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ float shfl_f(float v, int src) { return __shfl(v, src, 64); }

// (-DDQN_STAMPS: the wave that owns chunk 0 -- in k_dw a surplus workgroup, not block 0)
#ifdef DQN_STAMPS
#define PWSTAMP(S)                                                                              \
    do { if (chunk == 0 && (threadIdx.x & 63) == 0) {                                            \
             g_stamps[4][S][0] = __builtin_amdgcn_s_memtime();                                   \
             g_stamps[4][S][1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define PWSTAMP(S) do { } while (0)
#endif
// one wave = one 64-position chunk of the batch (chunk index `chunk`); called by k_per_write_sorted and by
// the surplus workgroups of k_dw (the write-back then shares a launch with the weight gradients)
__device__ __forceinline__ void per_write_sorted_wave(DqnState *st, float *tree, long long N, int L,
                                                      const int32_t *__restrict__ idx, const float *__restrict__ val,
                                                      int B, int mode, float alpha, float eps, int chunk) {
    const int lane = threadIdx.x & 63, base = chunk * 64;
    const int TOP = L < PW_TOP ? L : PW_TOP, SH = L - TOP;       // depth-TOP subtree id = leaf >> SH
    PWSTAMP(0);
    // ---- which items does this wave own?
    const int i0 = base + lane;
    const int my = i0 < B ? idx[i0] : -1;
    const int prev = (i0 > 0 && i0 <= B) ? idx[i0 - 1] : -1;
    const bool starts = i0 < B && (i0 == 0 || (my >> SH) != (prev >> SH));
    const unsigned long long sm = __ballot(starts);
    if (sm == 0ull) return;                                       // every item here belongs to an earlier owner
    const int first = base + __ffsll((long long)sm) - 1;
    const int last_chunk = (base + 63 < B - 1) ? base + 63 : B - 1;
    const int s_last = shfl_i(my, last_chunk - base) >> SH;
    // extension past the chunk end: following items that still belong to subtree s_last
    int ext = 0;
    {
        int pos = base + 64;
        for (;;) {
            const int j = pos + lane;
            const bool same = j < B && (idx[j] >> SH) == s_last;
            const unsigned long long mm = __ballot(same);
            if (mm == ~0ull) { ext += 64; pos += 64; continue; }
            ext += __ffsll((long long)~mm) - 1;
            break;
        }
    }
    const int total = (last_chunk + 1 - first) + ext;
    float lmax = 0.0f;
    PWSTAMP(1);

    if (total <= 64) {
        // ---- fast path: one item per lane, registers only
        const int i = first + lane;
        bool live = lane < total;
        long long x = live ? N + (long long)idx[i] : 0;
        float v = 0.0f;
        if (live) { v = (mode == 0) ? val[i] : pow_det(val[i] + eps, alpha); lmax = v; }
        float sib[PW_BOT];
#pragma unroll
        for (int l = 0; l < PW_BOT; ++l) sib[l] = (live && l < SH) ? tree[(x >> l) ^ 1] : 0.0f;
        {   // equal leaves: the highest batch position (last of the run) wins
            const long long xn = __shfl_down(x, 1, 64);
            if (live && lane + 1 < total && xn == x) live = false;
        }
        if (live) tree[x] = v;
        PWSTAMP(2);
#pragma unroll
        for (int l = 0; l < PW_BOT; ++l) {
            if (l >= SH) break;
            const unsigned long long m = __ballot(live);
            const unsigned long long mr = (lane == 63) ? 0ull : (m & ~((2ull << lane) - 1ull));
            const unsigned long long ml = m & ((1ull << lane) - 1ull);
            const int r = mr ? __ffsll((long long)mr) - 1 : lane;
            const int lf = ml ? 63 - __clzll((long long)ml) : lane;
            const long long xr = __shfl(x, r, 64), xl = __shfl(x, lf, 64);
            const float vr = shfl_f(v, r), vl = shfl_f(v, lf);
            (void)vl;
            if (live) {
                if ((x & 1) == 0) {
                    const bool has = mr && xr == x + 1;
                    v = v + (has ? vr : sib[l]);
                    x >>= 1;
                    tree[x] = v;
                } else {
                    const bool has = ml && xl == x - 1;
                    if (has) live = false;                        // the left sibling carries the pair upward
                    else { v = sib[l] + v; x >>= 1; tree[x] = v; }
                }
            }
        }
    } else {
        // ---- slow path (a depth-TOP subtree holds > 64 sampled items): wave-serial, level-synchronous
        // through L2 (agent-scope accesses bypass this CU's L1 between levels)
        for (int j = first + lane; j < first + total; j += 64) {
            const float p = (mode == 0) ? val[j] : pow_det(val[j] + eps, alpha);
            lmax = fmaxf(lmax, p);
            const bool loser = (j + 1 < first + total) && idx[j + 1] == idx[j];
            if (!loser) __hip_atomic_store(&tree[N + idx[j]], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int lvl = 1; lvl <= SH; ++lvl) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int j = first + lane; j < first + total; j += 64) {
                const long long node = (N + (long long)idx[j]) >> lvl;
                const float a = __hip_atomic_load(&tree[2 * node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float b = __hip_atomic_load(&tree[2 * node + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&tree[node], a + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    PWSTAMP(3);
    // running max priority: order-independent (positive floats order like their bit patterns)
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
    if (lane == 0) atomicMax(reinterpret_cast<unsigned int *>(&st->pmax), __float_as_uint(lmax));
    PWSTAMP(4);
}


// ------------------------------------------------------------ synthetic env + leaf-range insert
// Shared by k_per_add (dqn_replay.hip) and k_actor (dqn_actor.hip), whose tree workgroup inserts the new leaves.
#define RANGE_MAX (1 << 30)       // the range insert takes any n (kept for its callers' piece loops)

struct EnvArgs {            // q_agent.py:177-183 on device-resident synthetic envs; st == NULL => not an actor launch
    DqnState *st;
    float *states; int32_t *actions; float *rewards; float *observations; uint8_t *dones;
    long long cap; float *tree; long long Nt; int L;
    float *env_obs; unsigned long long seed; float p_done; int n;
    int kind;               // 0: synthetic transitions; 1: CartPole-v1 physics
    int max_steps;          // episode truncation (q_agent.py:179-180)
    int32_t *env_t;         // per-env step counter
    int time_feature;       // ObsWrapper (LunarLander/env.py:19-24): the LAST observation column is step / max_steps, kept by the
                            // actor kernels themselves (synthetic env, one-step returns); the episode also ends at max_steps
    float term_reward;      // CartPole: reward of the step that terminates the episode (gym: 1; see dqn_env_config)
    int rebuild_top;        // the surplus tree workgroup first rebuilds the dense top of the tree (deferred k_per_top)
    // n-step returns (dqn_config.n_step > 1; k_actor only): per-env history of the last n_step steps, [n_step][hist_stride]
    int n_step, hist_stride; float gamma;
    float *hist_s, *hist_r; int32_t *hist_a, *hist_d;
};

// ---- CartPole-v1 (classic control; BASELINE.json configs[2]). Euler step of the published cart-pole equations in
// f32 with one rounding per written operation and polynomial sin / cos (|theta| stays < 0.25 rad), so the CPU
// restatement reproduces every state bit for bit. Auto-reset on termination / truncation: the stored transition
// keeps the terminal next-state, the env continues from a fresh U(-0.05, 0.05)^4 state (Philox stream 3).
__device__ __forceinline__ float cp_sin(float t) {            // t - t^3/6 + t^5/120 - t^7/5040
    const float t2 = t * t;
    float p = -1.98412701e-4f;
    p = p * t2; p = p + 8.33333377e-3f;
    p = p * t2; p = p + -0.166666672f;
    p = p * t2; p = p + 1.0f;
    return p * t;
}
__device__ __forceinline__ float cp_cos(float t) {            // 1 - t^2/2 + t^4/24 - t^6/720
    const float t2 = t * t;
    float p = -1.38888892e-3f;
    p = p * t2; p = p + 4.16666679e-2f;
    p = p * t2; p = p + -0.5f;
    p = p * t2; p = p + 1.0f;
    return p;
}
__device__ __forceinline__ bool cartpole_step(float (&s)[4], int action) {
    const float force = action == 1 ? 10.0f : -10.0f;
    const float ct = cp_cos(s[2]), sn = cp_sin(s[2]);
    const float temp = __fdiv_rn(force + (0.05f * (s[3] * s[3])) * sn, 1.1f);
    const float thetaacc = __fdiv_rn((9.8f * sn) - (ct * temp), 0.5f * (1.33333337f - __fdiv_rn(0.1f * (ct * ct), 1.1f)));
    const float xacc = temp - __fdiv_rn((0.05f * thetaacc) * ct, 1.1f);
    s[0] = s[0] + 0.02f * s[1];
    s[1] = s[1] + 0.02f * xacc;
    s[2] = s[2] + 0.02f * s[3];
    s[3] = s[3] + 0.02f * thetaacc;
    return s[0] < -2.4f || s[0] > 2.4f || s[2] < -0.20943951f || s[2] > 0.20943951f;
}

// a normal is the Irwin-Hall sum ((u0+u1)+(u2+u3) - 2) * sqrt(3) -- exactly reproducible on the CPU
__device__ __forceinline__ float ih_normal(const u32x4 o) {
    return (((u01(o.x) + u01(o.y)) + (u01(o.z) + u01(o.w))) - 2.0f) * 1.73205078f;
}

// level-synchronous fallback through global memory (ring wrap, or n > RANGE_MAX). Whole workgroup.
__device__ __forceinline__ void per_add_slow(float *tree, long long Nt, int L, unsigned long long c_base, int n,
                                             float pmax, long long cap) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < n; i += nt)
        tree[Nt + (long long)((c_base + (unsigned long long)i) % (unsigned long long)cap)] = pmax;
    __threadfence_block();
    __syncthreads();
    for (int lvl = 1; lvl <= L; ++lvl) {
        for (int i = tid; i < n; i += nt) {
            const long long node = (Nt + (long long)((c_base + (unsigned long long)i) % (unsigned long long)cap)) >> lvl;
            tree[node] = tree[2 * node] + tree[2 * node + 1];
        }
        __threadfence_block();
        __syncthreads();
    }
}

// Leaf-range insert: new transitions occupy CONSECUTIVE leaves [a, a+n), all at the running max priority pmax. At level l
// the touched nodes are the contiguous range [first>>l, last>>l]; every node strictly inside that range has its whole
// subtree inside [a, a+n), so its value is pmax * 2^l -- exactly what summing two equal children level by level gives
// (x + x is exact), no dependency on the level below. Only the two END nodes of each level are real sums: of an inner or
// end child and, possibly, one untouched sibling outside the range. So: the outside siblings of all levels are
// requested up front (L1-bypassing: an earlier piece of the same insert, or the tree-top rebuild, run by this workgroup,
// may have written them after this CU cached their lines); all inner nodes of all levels are plain parallel stores;
// one wave walks the <= 2 end nodes per level up to the root in registers. Any n, no LDS image of the range.
// Whole workgroup; lds: 64 floats; requires a + n <= ring capacity.
// Tree stores that another workgroup of the SAME launch reads (actor launch: tree workgroup / fillers -> samplers):
// device-coherent write-through stores (sc1). Once the issuing wave's vmcnt has drained they are visible to every XCD,
// so the hand-over needs no release fence -- a fence writes back the XCD's WHOLE dirty L2 (18 us beside 16 384 fresh
// ring rows). Plain kernels that use these helpers lose nothing: the data had to reach HBM anyway.
__device__ __forceinline__ void tree_st(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tree_st4(float4 *p, const float4 &v) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(q) : "memory");
}

// inner nodes of levels 0 .. lmax of the leaf range [a, a+n): share `w` of `nw` (whole workgroups; w = 0 also stores the
// few unaligned nodes at the ends of each level's run). 16-B stores for the aligned middle.
__device__ __forceinline__ void per_add_range_fill(float *tree, long long Nt, int lmax, long long a, int n, float pmax, int w, int nw, int nt = (int)blockDim.x) {
    const int tid = threadIdx.x;
    const long long first = Nt + a, last = Nt + a + n - 1;
    for (int l = 0; l <= lmax; ++l) {
        const long long lo = first >> l, hi = last >> l;
        if (hi - lo < 2) break;                                          // (runs only shrink going up)
        const float v = pmax * (float)(1u << l);
        const long long p0 = lo + 1, p1 = hi;                            // [p0, p1)
        const long long q0 = (p0 + 3) & ~3ll, q1 = p1 & ~3ll;
        if (q0 < q1) {
            if (w == 0) {
                for (long long p = p0 + tid; p < q0; p += nt) tree_st(tree + p, v);
                for (long long p = q1 + tid; p < p1; p += nt) tree_st(tree + p, v);
            }
            float4 *t4 = reinterpret_cast<float4 *>(tree);
            const float4 v4 = {v, v, v, v};
            for (long long q = (q0 >> 2) + (long long)w * nt + tid; q < (q1 >> 2); q += (long long)nw * nt) tree_st4(t4 + q, v4);
        } else if (w == 0) {
            for (long long p = p0 + tid; p < p1; p += nt) tree_st(tree + p, v);
        }
    }
}

// the two END nodes of levels 0 .. lmax (+ the parents they feed at level lmax + 1 when lmax < L - 1 is NOT wanted: the walk
// stops after writing level lmax). One thread; outside siblings requested up front by the whole workgroup. lds: 64 floats.
__device__ __forceinline__ void per_add_range_ends(float *tree, long long Nt, int L, int lmax, long long a, int n, float pmax, float *lds, int nt = (int)blockDim.x) {
    const int tid = threadIdx.x;
    float *bl = lds, *br = lds + 32;
    const long long first = Nt + a, last = Nt + a + n - 1;
    for (int l = tid; l < L; l += nt) {
        const long long lo = first >> l, hi = last >> l;
        bl[l] = (lo & 1) ? __hip_atomic_load(&tree[lo - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
        br[l] = (hi & 1) ? 0.0f : __hip_atomic_load(&tree[hi + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    LDS_BARRIER();                                                       // bl / br are in LDS
    if (tid == 0) {
        float vlo = pmax, vhi = pmax;                                    // values of the end nodes; inner neighbours in closed form
        long long lo = first, hi = last;
        tree_st(tree + lo, vlo); tree_st(tree + hi, vhi);
        for (int l = 0; l < lmax; ++l) {
            const float inner = pmax * (float)(1u << l);                 // value of a node strictly inside [lo, hi] at level l
            const long long plo = lo >> 1, phi = hi >> 1;
            float nlo;
            if (lo & 1) nlo = bl[l] + vlo;                               // left sibling is outside the range
            else nlo = vlo + (lo + 1 > hi ? br[l] : (lo + 1 == hi ? vhi : inner));
            float nhi = nlo;
            if (phi != plo) {
                if (hi & 1) nhi = (hi - 1 == lo ? vlo : inner) + vhi;     // hi - 1 >= lo here (different parents)
                else nhi = vhi + br[l];                                  // right sibling is outside the range
            }
            lo = plo; hi = phi; vlo = nlo; vhi = nhi;
            tree_st(tree + lo, vlo);
            if (hi != lo) tree_st(tree + hi, vhi);
        }
    }
    LDS_BARRIER();                                                       // bl / br may be reused by the next segment
}

// the whole insert by one workgroup (API path k_per_add; actor launches without sampler workgroups)
__device__ __forceinline__ void per_add_range_wg(float *tree, long long Nt, int L, long long a, int n, float pmax, float *lds, int nt = (int)blockDim.x) {
    per_add_range_fill(tree, Nt, L, a, n, pmax, 0, 1, nt);
    per_add_range_ends(tree, Nt, L, L, a, n, pmax, lds, nt);
}

// dense top of the tree, whole workgroup: depth TOP-1 from the depth-TOP pairs in HBM, the levels above out of an
// LDS image of 2^TOP floats -- 256 x 68 floats on the 256-thread fast path -- (same arithmetic as k_per_top). Ends with a barrier.
__device__ __forceinline__ void per_top_wg(float *tree, int L, float *top, int nt = (int)blockDim.x) {
    const int tid = threadIdx.x;
    const int TOP = L < PW_TOP ? L : PW_TOP;
    if (TOP == 0) return;
    if (TOP == 14 && nt == 256) {
        // 256 threads, full-size top: each thread owns 64 consecutive depth-14 nodes (16 requests of 16 B, all in
        // flight at once) and reduces its private 6-level subtree in registers -- depths 13..8 never touch LDS and need
        // no barrier; depth 8 (one node per thread) and above continue in the LDS image. Same pair sums as below.
        // (the 64 KB of depth 14 come in as fully coalesced 16-B loads and are handed to their owner threads through a
        // padded LDS image [256][68]: per-thread contiguous global loads touched 64 cache lines per instruction -- 5.8 us)
        const float4 *src = reinterpret_cast<const float4 *>(tree + (1 << 14));
        float4 c[16];
        STAMP(2, 0);
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u] = src[u * 256 + tid];
#pragma unroll
        for (int u = 0; u < 16; ++u)                                       // element 4*(256u + tid) + {0..3} -> row (256u + tid) / 16
            *reinterpret_cast<float4 *>(top + (16 * u + (tid >> 4)) * 68 + 4 * (tid & 15)) = c[u];
        LDS_BARRIER();
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u] = *reinterpret_cast<const float4 *>(top + tid * 68 + 4 * u);
        LDS_BARRIER();                                                     // the image is dead: top[] is reused below
        float v[32];
#pragma unroll
        for (int u = 0; u < 16; ++u) { v[2 * u] = c[u].x + c[u].y; v[2 * u + 1] = c[u].z + c[u].w; }
        STAMP(2, 1);
        // depths 13..8 go to the LDS image top[node] first and leave for HBM as coalesced 16-B stores (per-thread
        // contiguous global stores cost what the loads did)
#pragma unroll
        for (int d = 13, cnt = 32; d >= 8; --d, cnt >>= 1) {
            float *dst = top + (1 << d) + cnt * tid;                     // this thread's cnt nodes of depth d
            if (cnt >= 4) {
#pragma unroll
                for (int u = 0; u < 32; u += 4) if (u < cnt) *reinterpret_cast<float4 *>(dst + u) = float4{v[u], v[u + 1], v[u + 2], v[u + 3]};
            } else if (cnt == 2) {
                *reinterpret_cast<float2 *>(dst) = float2{v[0], v[1]};
            } else {
                dst[0] = v[0];
            }
            if (d > 8) {
#pragma unroll
                for (int u = 0; u < 16; ++u) if (2 * u + 1 < cnt) v[u] = v[2 * u] + v[2 * u + 1];
            }
        }
        STAMP(2, 2);
        LDS_BARRIER();
        {
            const float4 *img = reinterpret_cast<const float4 *>(top + 256);
            float4 *out = reinterpret_cast<float4 *>(tree + 256);
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int q = u * 256 + tid; if (q < (16384 - 256) / 4) tree_st4(out + q, img[q]); }
        }
        STAMP(2, 3);
        for (int d = 7; d >= 0; --d) {
            const int cnt = 1 << d;
            if (tid < cnt) {
                const int p = cnt + tid;
                const float s = top[2 * p] + top[2 * p + 1];
                top[p] = s;
                tree_st(tree + p, s);
            }
            LDS_BARRIER();
        }
        STAMP(2, 4);
        __syncthreads();                                         // drain this workgroup's tree stores before it re-reads them
        STAMP(2, 5);
        return;
    }
    const int h = 1 << (TOP - 1);
    for (int j = tid; j < h; j += nt) {
        const float2 c = *reinterpret_cast<const float2 *>(tree + 2 * (h + j));
        const float v = c.x + c.y;
        top[h + j] = v;
        tree_st(tree + h + j, v);
    }
    __syncthreads();
    for (int d = TOP - 2; d >= 0; --d) {
        const int cnt = 1 << d;
        for (int j = tid; j < cnt; j += nt) {
            const int p = cnt + j;
            const float v = top[2 * p] + top[2 * p + 1];
            top[p] = v;
            tree_st(tree + p, v);
        }
        LDS_BARRIER();
    }
    __syncthreads();                                         // drain this workgroup's tree stores before it re-reads them
}

// ------------------------------------------------------------ sampling fused into the forward launch
// q_agent.py:147-153 (sample_batch) for the fused update: every forward workgroup draws the 16 batch rows it is about
// to push through the network (same Philox draw / stratified descent / clamp / raw IS weight as k_per_sample, so
// the indices are identical), reads those rows straight from the ring, and -- pass 0 only -- publishes idx, a, r, d,
// w_raw and the batch max weight for the backward kernel. The dependent 20-load descent overlaps the weight stream.
struct SampleArgs {
    DqnState *st;                        // NULL => rows come from FwdPass.x
    const float *tree; long long N; int L;   // tree == NULL => uniform sampling (Philox stream 1)
    const float *states, *observations, *rewards; const int32_t *actions; const uint8_t *dones;
    unsigned long long seed;
    int32_t *idx, *a; float *r, *w_raw; uint8_t *d;
    int pre;                             // idx / w_raw / wmax were already drawn by the actor launch (dqn_actor.hip): gather only
};

__device__ __forceinline__ long long sample_leaf(const SampleArgs &s, int k, int B, float *w_out) {
    const unsigned long long ctr = s.st->sample_ctr;
    const long long size = s.st->size;
    if (!s.tree) {                                                           // replay_buffer.py:77
        const u32x4 o = philox_draw(s.seed, ctr, (uint32_t)k, DQN_STREAM_UNIFORM);
        *w_out = 1.0f;
        return (long long)(((unsigned long long)o.x * (unsigned long long)size) >> 32);
    }
    const float beta = s.st->beta;
    const float total = s.tree[1];
    const float seg = __fdiv_rn(total, (float)B);
    const u32x4 o = philox_draw(s.seed, ctr, (uint32_t)k, DQN_STREAM_PER);
    float u = ((float)k + u01(o.x)) * seg;
    long long node = 1;
    for (int lvl = 0; lvl < s.L; ++lvl) {
        const float l = s.tree[2 * node];
        if (u < l) { node = 2 * node; }
        else { u = u - l; node = 2 * node + 1; }
    }
    long long leaf = node - s.N;
    if (leaf >= size) leaf = size - 1;
    *w_out = pow_det(__fdiv_rn((float)size * s.tree[s.N + leaf], total), -beta);
    return leaf;
}

// lanes 0..15 of wave 0: draw row row0+lane, leave the leaf in lidx[lane]; pass 0 publishes the batch fields
__device__ __forceinline__ void sample_tile(const SampleArgs &s, int row0, int B, int lane16, bool publish, int *lidx) {
    const int k = row0 + lane16, kk = k < B ? k : B - 1;
    float w;
    const long long leaf = sample_leaf(s, kk, B, &w);
    lidx[lane16] = (int)leaf;
    if (publish) {
        if (k < B) {
            s.idx[k] = (int32_t)leaf; s.w_raw[k] = w;
            s.a[k] = s.actions[leaf]; s.r[k] = s.rewards[leaf]; s.d[k] = s.dones[leaf];
        }
        float mx = k < B ? w : 0.0f;
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane16 == 0 && s.tree) atomicMax(reinterpret_cast<unsigned int *>(&s.st->wmax), __float_as_uint(mx));
    }
}

// Cooperative form of sample_tile for a 256-thread workgroup: 16 lanes per batch row. Per round trip the group
// fetches the whole 4-level subtree below its current node (2+4+8+16 = 30 nodes, contiguous per level) into LDS and
// walks it there -- 5 dependent memory latencies for L = 20 instead of 20. Same compares / subtractions in the same
// order as the scalar descent, hence the same leaf. lsub: 16*32 floats, lw: 16 floats.
__device__ __forceinline__ void sample_tile_coop(const SampleArgs &s, int row0, int B, int tid, bool publish,
                                                 int *lidx, float *lsub, float *lw) {
    const int g = tid >> 4, j = tid & 15;
    const int k = row0 + g, kk = k < B ? k : B - 1;
    const unsigned long long ctr = s.st->sample_ctr;
    const long long size = s.st->size;
    const float beta = s.st->beta;
    const float total = s.tree[1];
    const float seg = __fdiv_rn(total, (float)B);
    const u32x4 o = philox_draw(s.seed, ctr, (uint32_t)kk, DQN_STREAM_PER);
    float u = ((float)kk + u01(o.x)) * seg;
    long long cur = 1;
    float *sub = lsub + g * 32;
    for (int done = 0; done < s.L; done += 4) {
        const int nl = s.L - done < 4 ? s.L - done : 4;
        const int cnt = (2 << nl) - 2;
        for (int f = j; f < cnt; f += 16) {
            const int t = 31 - __clz(f + 2), i = f + 2 - (1 << t);        // flattened index -> (relative depth t, position i)
            sub[f] = s.tree[(cur << t) + i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the group is inside one wave: LDS program order
        int p = 0;
        for (int t = 1; t <= nl; ++t) {
            const float l = sub[(1 << t) - 2 + 2 * p];
            if (u < l) { p = 2 * p; }
            else { u = u - l; p = 2 * p + 1; }
        }
        cur = (cur << nl) + p;
    }
    long long leaf = cur - s.N;
    if (leaf >= size) leaf = size - 1;
    if (j == 0) {
        const float w = pow_det(__fdiv_rn((float)size * s.tree[s.N + leaf], total), -beta);
        lidx[g] = (int)leaf;
        lw[g] = k < B ? w : 0.0f;
        if (publish && k < B) {
            s.idx[k] = (int32_t)leaf; s.w_raw[k] = w;
            s.a[k] = s.actions[leaf]; s.r[k] = s.rewards[leaf]; s.d[k] = s.dones[leaf];
        }
    }
    LDS_BARRIER();
    if (publish && tid == 0) {
        float mx = 0.0f;
        for (int q = 0; q < 16; ++q) mx = fmaxf(mx, lw[q]);
        atomicMax(reinterpret_cast<unsigned int *>(&s.st->wmax), __float_as_uint(mx));
    }
}
