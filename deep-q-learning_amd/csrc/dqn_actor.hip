// csrc/dqn_actor.hip -- the vector actor of the inner loop (q_agent.py:176-183) for T consecutive env steps in
// ONE launch, exact-f32 path.
//
// Between two updates the online parameters do not change and the envs do not interact, so the T = train_frequency
// actor steps of an iteration (q_agent.py:186) are T dependent forwards of the SAME rows through the SAME weights.
// One workgroup owns 4 envs ("tile") for the whole launch:
//   * its slab of W1 / W2 (column 64*wave + lane, every k) is fetched ONCE into registers straight from the
//     canonical [in,out] row-major parameters (a k-row of 64 columns = one coalesced 256-B wave load; no packing);
//   * every layer is a chain of v_mfma_f32_4x4x1_16b_f32: 16 blocks of 4x4, K = 1, i.e. 4 rows x 64 columns per
//     instruction -- the narrowest row tile the matrix core offers at the full f32 rate, so a 256x256 layer costs
//     256 issues per wave instead of the 1024 of a 16-row tile. One fused multiply-add per (row, column, k) in
//     ascending k: bit-for-bit the fmaf chain of the CPU restatement (and of the 16x16x4 kernels of dqn_net.hip);
//   * the two skinny heads (4 rows x (1+A) columns, K = H2) run as VALU fmaf chains out of LDS: four partial chains per output
//     (k groups of four, chain j = (k/4) % 4, one per wave), combined (c0+c1)+(c2+c3) -- the oracle's heads_row;
//   * the env state never leaves LDS between steps; ring rows go straight to their (deterministic) slots
//     counter + t*n + i (replay_buffer.py:58-65).
// Synthetic env: no physics. Per env i and vector step c: obs' ~ N(0,1)^D, r ~ N(0,1) (+-100 on terminals),
// d ~ Bernoulli(p_done), all from Philox stream 3 and exactly reproducible on the CPU (a normal is the Irwin-Hall sum
// ((u0+u1)+(u2+u3) - 2) * sqrt(3)). CartPole-v1: dqn_per_device.h.
// Riding in the same launch, off the actors' critical path:
//   * workgroup 0 rebuilds the tree top left stale by the previous update's priority write-back and inserts the
//     T*n new leaves (all at the running max priority, one contiguous range), then releases a flag;
//   * ceil(B/16) sampler workgroups wait for that flag and draw the NEXT update's stratified PER batch (indices, raw
//     IS weights, batch max) -- the tree is final once the leaves are in, whatever the actors are still doing; the
//     sampled rows themselves are gathered by the forward launch that follows (SampleArgs.pre).
#include <type_traits>
#include "dqn_device.h"
#include "dqn_launch.h"
#include "dqn_per_device.h"
#include "dqn_net_common.h"
#include <cstdlib>

#define MFMA1(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define MFMA4B(a, b, c) __builtin_amdgcn_mfma_f32_4x4x4bf16_1k((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ short bf16_bits(float f) { return __builtin_bit_cast(short, (__bf16)f); }   // round to nearest even
__device__ __forceinline__ s16x4 cvt4(float a, float b, float c, float d) { return s16x4{bf16_bits(a), bf16_bits(b), bf16_bits(c), bf16_bits(d)}; }

#ifdef DQN_STAMPS
extern __device__ unsigned long long g_stamps[8][64][2];
#define ASTAMP(S)                                                                             \
    do { if (threadIdx.x == 0 && wg == 0) {                                                    \
             g_stamps[7][S][0] = __builtin_amdgcn_s_memtime();                                 \
             g_stamps[7][S][1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define WSTAMP(S)                                                                             \
    do { if (threadIdx.x == 64 && wg == 0) {                                                   \
             g_stamps[7][S][0] = __builtin_amdgcn_s_memtime();                                 \
             g_stamps[7][S][1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define BSTAMP(S)                                                                             \
    do { if (threadIdx.x == 0) {                                                               \
             g_stamps[3][S][0] = __builtin_amdgcn_s_memtime();                                 \
             g_stamps[3][S][1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define ASTAMP(S) do { } while (0)
#define BSTAMP(S) do { } while (0)
#define WSTAMP(S) do { } while (0)
#endif

struct ActorArgs {
    EnvArgs env;                 // envs, ring, tree, device state
    int T;                       // vector env steps taken by this launch
    const float *params;         // canonical online parameters (LunarLander/dddqn.py:19-22 leaf order)
    const float *pack;           // their packed shadows (p_w2k: k-packed W2, p_wht: heads)
    int32_t *act_out;            // [n] actions of the last step
    int tiles, G;                // 4-env tiles, actor workgroups (tile = wg, wg + G, ...)
    int TC;                      // steps whose Philox draws are made at once (what the LDS draw buffer holds)
    int n_tree, n_smp, NR;                 // NR: batch rows per sampler lane group (1, 2, 4)           // role split of the grid: [tree][actors x G][samplers x n_smp]
    int B; SampleArgs smp;       // presampling of the next update's batch (n_smp > 0)
};

// ---- one sampler workgroup: 16 batch rows, 16 lanes per row, 4 tree levels per memory round trip (the descent of
// sample_tile_coop in dqn_per_device.h: same compares / subtractions in the same order, hence the same leaves)
// ring position of a counter: capacities are powers of two whenever a sum-tree is in use (a mask instead of the ~60-instruction
// software 64-bit modulo in every workgroup's prologue); any other capacity takes the modulo
__device__ __forceinline__ unsigned long long mod_cap(unsigned long long c, long long cap) {
    const unsigned long long uc = (unsigned long long)cap;
    return (uc & (uc - 1)) == 0 ? (c & (uc - 1)) : c % uc;
}

// the stratified uniforms of rows row0 + 16 r + g, before scaling by the tree total: k + u01 (they depend on nothing
// the launch changes, so a sampler workgroup draws them while it waits for the tree)
template <int NR>
__device__ __forceinline__ void presample_draw(const SampleArgs &s, int row0, int B, int tid, float (&uu)[NR]) {
    const int g = tid >> 4;
    const unsigned long long ctr = s.st->sample_ctr;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int k = row0 + 16 * r + g, kk = k < B ? k : B - 1;
        const u32x4 o = philox_draw(s.seed, ctr, (uint32_t)kk, DQN_STREAM_PER);
        uu[r] = (float)kk + u01(o.x);
    }
}

template <int NR>
__device__ __forceinline__ void presample_tile(const SampleArgs &s, int row0, int B, int tid, long long size,
                                               float *lsub, float *lw, const float (&uu)[NR], int sb = -1) {
#define PSTAMP(i) do { if (sb >= 0) BSTAMP(sb + (i)); } while (0)
    // 16 lanes per descent, 4 tree levels per round trip; NR descents per lane group in flight together (rows
    // row0 + 16 r + g) when the batch has more 16-row tiles than the launch has sampler workgroups
    const int g = tid >> 4, j = tid & 15;
    const float beta = s.st->beta;
    const float total = s.tree[1];
    const float seg = __fdiv_rn(total, (float)B);
    int k[NR];
    float u[NR];
    long long cur[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        k[r] = row0 + 16 * r + g;
        u[r] = uu[r] * seg;
        cur[r] = 1;
    }
    PSTAMP(0);
    float *sub = lsub + g * 32;
    for (int done = 0; done < s.L; done += 4) {
        const int nl = s.L - done < 4 ? s.L - done : 4;
        const int cnt = (2 << nl) - 2;
#pragma unroll
        for (int r = 0; r < NR; ++r)
            for (int f = j; f < cnt; f += 16) {
                const int t = 31 - __clz(f + 2), i = f + 2 - (1 << t);
                sub[r * 512 + f] = s.tree[(cur[r] << t) + i];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // a row group lives inside one wave
        int p[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) p[r] = 0;
        for (int t = 1; t <= nl; ++t) {                                   // the NR walks side by side: their LDS reads overlap
            float l[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) l[r] = sub[r * 512 + (1 << t) - 2 + 2 * p[r]];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (u[r] < l[r]) { p[r] = 2 * p[r]; }
                else { u[r] = u[r] - l[r]; p[r] = 2 * p[r] + 1; }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) cur[r] = (cur[r] << nl) + p[r];
        PSTAMP(1 + (done >> 2));
    }
    if (j == 0) {
        long long leaf[NR];
        float pr[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            leaf[r] = cur[r] - s.N;
            if (leaf[r] >= size) leaf[r] = size - 1;
            pr[r] = s.tree[s.N + leaf[r]];
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float w = pow_det(__fdiv_rn((float)size * pr[r], total), -beta);
            lw[16 * r + g] = k[r] < B ? w : 0.0f;
            if (k[r] < B) { s.idx[k[r]] = (int32_t)leaf[r]; s.w_raw[k[r]] = w; }
        }
    }
    PSTAMP(6);
    LDS_BARRIER();
    PSTAMP(7);
    if (tid == 0) {
        float mx = 0.0f;
        for (int q = 0; q < 16 * NR; ++q) mx = fmaxf(mx, lw[q]);
        atomicMax(reinterpret_cast<unsigned int *>(&s.st->wmax), __float_as_uint(mx));
    }
}

// ---- n-step returns (dqn_config.n_step > 1). History rows are re-read by the thread that wrote them, n_step - 1 steps
// later in the same launch: L1-bypassing loads, because this CU's L1 may still hold the line from the previous read.
template <class T> __device__ __forceinline__ T ld_sc1(const T *p) {
    return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// files (action, reward, done) of env i at history slot hpos and, when the window is complete, replaces them by the row of
// the window that starts at slot hold: its first action, R = r_0 + gamma*(r_1 + gamma*(...)) cut after the first done, and
// that done flag (Horner form, one rounding per operation: the CPU restatement's arithmetic)
__device__ __forceinline__ void nstep_row(const EnvArgs &e, int ns, int hpos, int hold, int i, bool emit, int &a, float &r, int &d) {
    const long long hs = e.hist_stride;
    e.hist_a[hpos * hs + i] = a; e.hist_r[hpos * hs + i] = r; e.hist_d[hpos * hs + i] = d;
    if (!emit) return;
    float rr[8]; int dd[8];
#pragma unroll
    for (int k = 0; k < 7; ++k) {                                        // the ns-1 older steps; the newest is in registers
        const int p = hold + k < ns ? hold + k : hold + k - ns;
        rr[k] = k < ns - 1 ? ld_sc1(e.hist_r + p * hs + i) : 0.0f;
        dd[k] = k < ns - 1 ? ld_sc1(e.hist_d + p * hs + i) : 0;
    }
    const int a0 = ld_sc1(e.hist_a + hold * hs + i);
    int last = ns - 1;
#pragma unroll
    for (int k = 6; k >= 0; --k) if (k < ns - 1 && dd[k]) last = k;      // first done in the window
    float acc = last == ns - 1 ? r : 0.0f;
    int dn = last == ns - 1 ? d : 1;
#pragma unroll
    for (int k = 6; k >= 0; --k) {
        if (k < ns - 1) {
            if (k == last) acc = rr[k];
            else if (k < last) acc = rr[k] + e.gamma * acc;
        }
    }
    a = a0; r = acc; d = dn;
}

// ---- the side workgroups of an actor launch (shared by k_actor and k_actor16). role 0: the tree workgroup -- deferred top
// rebuild, then the leaves of all steps of the launch (q_agent.py:182 x T; pmax only moves in a priority write-back, so the
// T inserts of the sequential loop are one range insert), then the flag; role 2: a sampler workgroup -- q_agent.py:147-153
// for the update that follows this launch, once the flag is up.
__device__ __forceinline__ void actor_side_role(int role, int wg, const ActorArgs &g, unsigned long long c0, unsigned long long nT,
                                                unsigned long long c1, unsigned long long ticket_val, float *lds) {
    const EnvArgs &e = g.env;
    const int tid = threadIdx.x;
    // the new slots are one contiguous leaf range, or two when the ring wraps (any order of inserting gives the same tree:
    // a parent is always the sum of its two current children)
    const long long a = (long long)mod_cap(c0, e.cap);
    const long long seg_a[2] = {a, 0};
    const long long seg_n[2] = {a + (long long)nT < e.cap ? (long long)nT : e.cap - a, a + (long long)nT < e.cap ? 0 : a + (long long)nT - e.cap};
    // A single CU stores at ~7 B/clk: a long insert by the tree workgroup alone would be the launch's critical path
    // (16 384 leaves: 37 us). With sampler workgroups in the launch, THEY store the inner nodes of the new range (closed
    // form pmax * 2^level, no dependencies) before they wait for the flag; the tree workgroup only walks the end nodes.
    // When the tree top is rebuilt in this launch, both stop at the top's base depth: the rebuild then derives everything
    // above from final values.
    const int TOPD = e.L < PW_TOP ? e.L : PW_TOP;
    const int lmax = e.rebuild_top ? e.L - TOPD : e.L;                   // levels (leaf = 0) written by fill / end-node walk
    // enough fillers to spread the stores (one per 512 leaves, <= 32), few enough that their counter bumps -- device-scope
    // atomics on one line, ~0.1 us each -- stay short
    int nfill = (int)(nT >> 9);
    nfill = nfill < 1 ? 1 : (nfill > 32 ? 32 : nfill);
    if (nfill > g.n_smp) nfill = g.n_smp;
    if (role == 0) {
        BSTAMP(0);
        const float pmax = e.st->pmax;
        if (g.n_smp > 0) {
            for (int seg = 0; seg < 2; ++seg)
                if (seg_n[seg] > 0) per_add_range_ends(e.tree, e.Nt, e.L, lmax, seg_a[seg], (int)seg_n[seg], pmax, lds, 256);
            BSTAMP(1);
            // every sampler workgroup has stored (and released) its share of the inner nodes?
            if (tid == 0 && !wait_word_eq(&e.st->fill_cnt, (unsigned)nfill, 4)) flag_wait_timeout(e.st);   // bounded: dqn_device.h
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (e.rebuild_top) per_top_wg(e.tree, e.L, lds, 256);
            BSTAMP(2);
            __syncthreads();                                                 // every wave's tree stores (sc1) have been acknowledged
            if (tid == 0) __hip_atomic_store(&e.st->tree_ready, ticket_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (e.rebuild_top) per_top_wg(e.tree, e.L, lds, 256);
            BSTAMP(1);
            for (int seg = 0; seg < 2; ++seg)
                if (seg_n[seg] > 0) { per_add_range_wg(e.tree, e.Nt, e.L, seg_a[seg], (int)seg_n[seg], pmax, lds, 256); __syncthreads(); }
            BSTAMP(2);
        }
        BSTAMP(3);
    } else if (role == 2) {
        // ---- sampler workgroup: its share of the new leaves' inner nodes, then q_agent.py:147-153 for the update that follows
        if (wg == 0) BSTAMP(4);
        if (wg < nfill) {
            const float pmax = e.st->pmax;
            for (int seg = 0; seg < 2; ++seg)
                if (seg_n[seg] > 0) per_add_range_fill(e.tree, e.Nt, lmax, seg_a[seg], (int)seg_n[seg], pmax, wg, nfill, 256);
            __syncthreads();                                                 // the stores (sc1) have been acknowledged
            if (tid == 0) atomicAdd(&e.st->fill_cnt, 1u);
        }
        const long long size = (long long)(c1 < (unsigned long long)e.cap ? c1 : (unsigned long long)e.cap);
        auto run = [&](auto nr_tag) {
            constexpr int NR = decltype(nr_tag)::value;
            const int rows = 16 * NR, ntile = (g.B + rows - 1) / rows;
            float uu[NR];
            int tile = wg;
            if (tile < ntile) presample_draw<NR>(g.smp, tile * rows, g.B, tid, uu);
            if (tid == 0 && !wait_word_eq(&e.st->tree_ready, ticket_val, 8)) flag_wait_timeout(e.st);   // bounded; (hundreds of pollers of one line: poll sparsely; acquire below)
            if (wg == 0) BSTAMP(5);
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (wg == 0) BSTAMP(6);
            for (; tile < ntile; tile += g.n_smp) {
                presample_tile<NR>(g.smp, tile * rows, g.B, tid, size, lds, lds + 2048, uu, wg == 0 && tile == 0 ? 8 : -1);
                LDS_BARRIER();                                               // lw is rewritten by the next tile
                if (tile + g.n_smp < ntile) presample_draw<NR>(g.smp, (tile + g.n_smp) * rows, g.B, tid, uu);
            }
        };
        if (g.NR == 1) run(std::integral_constant<int, 1>{});
        else if (g.NR == 2) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 4>{});
        if (wg == 0) BSTAMP(7);
    }
}

// `s_waitcnt vmcnt(VM)` whose operands are CNT registers of the slab (AG: accumulation VGPRs): nothing that reads them can be
// scheduled ahead of the wait. Vector-memory results return in request order, so "at most VM outstanding" = "everything
// but the youngest VM requests has landed" (younger requests of any kind only make the wait more conservative).
template <int CNT, bool AG, int VM>
__device__ __forceinline__ void wait_tied(f32x4 *w) {
    static_assert(CNT == 16 || CNT == 8 || CNT == 4, "segment sizes of the W2 slab");
    if constexpr (CNT == 16) {
        if constexpr (AG) asm volatile("s_waitcnt vmcnt(%[n])" : "+a"(w[0]), "+a"(w[1]), "+a"(w[2]), "+a"(w[3]), "+a"(w[4]), "+a"(w[5]), "+a"(w[6]), "+a"(w[7]), "+a"(w[8]), "+a"(w[9]), "+a"(w[10]), "+a"(w[11]), "+a"(w[12]), "+a"(w[13]), "+a"(w[14]), "+a"(w[15]) : [n] "n"(VM) : "memory");
        else asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]) : [n] "n"(VM) : "memory");
    } else if constexpr (CNT == 8) {
        if constexpr (AG) asm volatile("s_waitcnt vmcnt(%[n])" : "+a"(w[0]), "+a"(w[1]), "+a"(w[2]), "+a"(w[3]), "+a"(w[4]), "+a"(w[5]), "+a"(w[6]), "+a"(w[7]) : [n] "n"(VM) : "memory");
        else asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : [n] "n"(VM) : "memory");
    } else {
        if constexpr (AG) asm volatile("s_waitcnt vmcnt(%[n])" : "+a"(w[0]), "+a"(w[1]), "+a"(w[2]), "+a"(w[3]) : [n] "n"(VM) : "memory");
        else asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : [n] "n"(VM) : "memory");
    }
}

// KB = 16-row k-blocks of the register-resident W2 slab: hidden1 <= 16*KB (columns / rows past hidden1 are zeros, and
// x*0 + acc leaves every chain unchanged), so the layer-2 chain is straight-line code for each size class
// KB2: the same for hidden2 (the heads' chains run over 16*KB2 zero-padded k).
// NSTEP: dqn_config.n_step > 1 (compiled apart: the n-step bookkeeping costs scalar registers in the step loop)
// BF: bf16 mode (DQN_PREC_BF16) -- the same workgroup / step structure on v_mfma_f32_4x4x4_16b_bf16: weights and
// activations rounded to bf16 (as the update's bf16 kernels do), f32 accumulation, FOUR k per instruction, so every
// K-long chain is a quarter as long; the slabs are converted once per launch from the f32 shadows
template <int KB, int KB2, bool NSTEP, bool BF>
__global__ void __launch_bounds__(256)
k_actor(NetDims m, ActorArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const EnvArgs &e = g.env;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long c0 = e.st->ring_counter, ec = e.st->env_ctr;
    // n-step returns: a vector step adds rows only once n_step steps are on file, i.e. not during the first `warm` steps
    // after dqn_env_reset (n_step == 1: warm = 0, every step adds its n rows)
    const unsigned long long hs0 = NSTEP ? e.st->hist_steps : 0ull;
    unsigned int ticket = 0u;                    // this workgroup's arrival ticket (thread 0; see the commit at the end)
    const int warm = (NSTEP && hs0 + 1ull < (unsigned long long)e.n_step) ? (int)((unsigned long long)e.n_step - 1ull - hs0) : 0;
    const int n_emit = g.T > warm ? g.T - warm : 0;
    const unsigned long long nT = (unsigned long long)n_emit * (unsigned long long)e.n, c1 = c0 + nT;
    const unsigned long long ticket_val = ec + (unsigned long long)g.T;   // flag value of THIS launch (the env step counter only grows)
    const unsigned total_wgs = (unsigned)(g.n_tree + g.G + g.n_smp);   // (= gridDim.x, without reading the dispatch packet)
    int role = 1, wg = (int)blockIdx.x - g.n_tree;
    if ((int)blockIdx.x < g.n_tree) role = 0;
    else if (wg >= g.G) { role = 2; wg -= g.G; }

    if (role != 1) {
        actor_side_role(role, wg, g, c0, nT, c1, ticket_val, lds);
    } else {
        // ---- actor workgroup
        const int D = m.D, H1 = m.H1, H2 = m.H2, A = m.A;
        const int DP = (D + 3) & ~3;
        const int sx = DP + 4, s1 = 16 * KB + 4, s2 = 16 * KB2 + 4;
        float *lx = lds, *l1 = lx + 4 * sx, *l2 = l1 + 4 * s1, *lwh = l2 + 4 * s2, *lq = lwh + (A + 1) * s2;
        int *lt = reinterpret_cast<int *>(lq + 64);                          // step counters of the tile's envs, [2][4]: CartPole uses [0];
                                                                             // the time feature reads [t & 1] and writes [(t + 1) & 1]
        // Philox draws of TC steps at a time, [TC][4 envs][DW]: the synthetic env's next observation (D floats), then the
        // policy's u, its random action, done, reward. None depends on the forward pass: waves 1..3 make them for a whole
        // chunk of steps up front (behind the W2 stream for the first tile) instead of inside every step.
        const int DO = (D + 3) & ~3, DW = DO + 4;                            // the four scalars sit 16-B aligned behind the observation
        float *lpart = lq + 64 + 8;                                          // heads: the four waves' partial sums [4][64]
        float *ldraw = lpart + 256;
        // bf16 mode: bf16 images of h1 / h2 / the heads' weights (rounded ONCE, by the lanes that produce them: a chain link
        // is then one 8-B LDS read + one MFMA, no conversion in its shadow)
        const int s1h = 16 * KB + 8, s2h = 16 * KB2 + 8;
        __bf16 *l1h = reinterpret_cast<__bf16 *>(ldraw + g.TC * 4 * DW), *l2h = l1h + 4 * s1h, *lwh16 = l2h + 4 * s2h;
        const float *P = g.params;
        const int r4 = lane & 3;
        ASTAMP(0);

        // Every parameter request of the prologue is an inline-asm buffer load (SGPR descriptor + 32-bit lane offset;
        // rows past the end of a matrix are out of range and come back as 0.0 from the bounds check). The compiler does
        // not count them, so the waits are written by hand. Order of issue = order of arrival: the first half of the W2
        // slab leads (the compiler guards the first reuse of a few registers with a full wait: harmless while nothing is
        // in flight), the small operands (first tile's rows, W1 slab, biases, heads) follow, and the second half of the
        // slab goes out as soon as those have been staged; it lands behind layer 1 of the first step.
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        auto mkrs = [](const void *ptr, int bytes) -> i32x4 {
            const unsigned long long b = reinterpret_cast<unsigned long long>(ptr);
            return i32x4{(int)(unsigned)(b & 0xffffffffull), (int)(unsigned)(b >> 32), bytes, 0x00020000};
        };
        const i32x4 rsP = mkrs(P, (int)(m.P * 4)), rsW1 = mkrs(P + m.o_w1, D * H1 * 4), rsX = mkrs(e.env_obs, e.n * D * 4),
                    rsH = mkrs(g.pack + m.p_wht, 16 * H2 * 4),
                    rsW2 = BF ? mkrs(g.pack + m.pack_floats, H1 * H2 * 2)                  // bf16 mode: eight k per 16 bytes, ready-made
                              : mkrs(g.pack + m.p_w2k, H1 * H2 * 4);
#define BLD1(dst, voff, rs, soff) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff))
#define BLD4(cls, dst, voff, rs, soff) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : cls(dst) : "v"(voff), "s"(rs), "s"(soff))
        const unsigned col = 64u * wave + lane;
        const unsigned cc1 = col < (unsigned)H1 ? col : (unsigned)H1 - 1u, cc2 = col < (unsigned)H2 ? col : (unsigned)H2 - 1u;
        const int hr = lane & 3, hc = lane >> 2;                             // head chain of this lane (wave 0)
        const bool hlane = wave == 0 && hc <= A;
        const bool hany = hc <= A;                                           // (every wave runs one of the four partial chains)
        const int nq4 = 4 * H2;
        // The W2 slab goes straight to its final registers as 16-B loads of the k-packed shadow (four consecutive k of
        // the lane's column; measured 69 B/clk/CU against 36 for dword rows): k < 64 to architectural VGPRs, the rest to
        // accumulation VGPRs (the ISA addresses 256 of each; the matrix core reads its B operand from either file; the
        // compiler would stage every row in an architectural VGPR first: spills, serialised copies).
        // bf16 mode: the slab is the bf16 shadow the optimizer keeps (eight consecutive k of the column per 16-B load): half
        // the requests, nothing to convert
        f32x4 w2q[BF ? 1 : 4 * KB], w2q8[BF ? 2 * KB : 1];
        constexpr int N1 = BF ? 0 : (4 * KB < 32 ? 4 * KB : 32);   // first part of the slab; the rest follows the small operands
        const int vw2 = (int)(cc2 * 16u);
        // An actor workgroup has no memory operation in flight here, but the compiler's bookkeeping merges in the side roles' stores
        // (same registers) and put a vmcnt(0) after the THIRD slab request below: the rest of the slab waited for a round trip.
        // A wait it can see costs nothing now and clears that bookkeeping (vmcnt 0, expcnt 7, lgkmcnt 15).
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if constexpr (BF) {
#pragma unroll
            for (int u = 0; u < 2 * KB; ++u) {           // (as below: the later half straight to accumulation VGPRs -- an "=v" result the
                if (u < 16) BLD4("=v", w2q8[u], vw2, rsW2, u * H2 * 16);   //  allocator parks there is copied before its data has arrived)
                else BLD4("=a", w2q8[u], vw2, rsW2, u * H2 * 16);
            }
        } else {
#pragma unroll
            for (int kq = 0; kq < N1; ++kq) {
                if (kq < 16) BLD4("=v", w2q[kq], vw2, rsW2, kq * H2 * 16); else BLD4("=a", w2q[kq], vw2, rsW2, kq * H2 * 16);
            }
        }
        float x0, w1r[16], b1r, b2r, bh, bh16r = 0.0f;
        f32x4 whv[4];
        int vx, v1, v2, vh, vh16 = 0, vq[4];
        {
            // element tid of the first tile's lx image (out of range => 0.0: padding columns, rows past n)
            const int il = tid / sx, el = tid - il * sx, i = 4 * wg + il;
            vx = (tid < 4 * sx && i < e.n && el < D) ? (i * D + el) * 4 : 0x7ffffff0;
            BLD1(x0, vx, rsX, 0);
            v1 = (int)(cc1 * 4u); v2 = (int)(cc2 * 4u);
#pragma unroll
            for (int k = 0; k < 16; ++k) BLD1(w1r[k], v1, rsW1, k * H1 * 4);
            BLD1(b1r, v1, rsP, (int)m.o_b1 * 4);
            BLD1(b2r, v2, rsP, (int)m.o_b2 * 4);
            vh = ((hc == 0 || hc > A) ? (int)m.o_bv : (int)m.o_ba + hc - 1) * 4;
            BLD1(bh, vh, rsP, 0);
            if constexpr (BF) {                                               // bf16 mode: head bias of column `lane`
                vh16 = ((lane == 0 || lane > A) ? (int)m.o_bv : (int)m.o_ba + lane - 1) * 4;
                BLD1(bh16r, vh16, rsP, 0);
            }
            // heads, from the fragment-ordered transposed shadow (dqn_net.hip: packed(WH^T), K = 1+A padded to 16, C = H2):
            // float4 number ct*64 + ln holds WH^T[4j + (ln>>4)][16ct + (ln&15)], j = 0..3
#pragma unroll
            for (int u = 0; u < 4; ++u) { vq[u] = (tid + 256 * u) * 16; BLD4("=v", whv[u], vq[u], rsH, 0); }
        }
        // (the address registers of the requests above stay allocated until here: were one of them reused as a destination
        // of the slab requests, the compiler would put a full wait in front of that request)
        asm volatile("" :: "v"(vx), "v"(v1), "v"(v2), "v"(vh), "v"(vh16), "v"(vq[0]), "v"(vq[1]), "v"(vq[2]), "v"(vq[3]));
        float eps = e.st->epsilon;
        // ---- Philox draws of steps t0 .. t0+TC-1 of a tile (waves 1..3): for the first tile right here, while the parameter
        // requests above are in flight
        auto make_draws = [&](int i0, int cnt, int t0) {
            if (wave == 0) return;
            const int nobs = e.kind == 0 ? cnt * D : 0, per = nobs + 2 * cnt;
            const int nst = g.T - t0 < g.TC ? g.T - t0 : g.TC;
            for (int u = tid - 64; u < nst * per; u += 192) {
                const int ts = u / per, v = u - ts * per;
                const unsigned long long ecs = ec + (unsigned long long)(t0 + ts);
                float *dw = ldraw + (ts * 4) * DW;
                if (v < nobs) {
                    const int il = v / D, el = v - il * D, i = i0 + il;
                    dw[il * DW + el] = ih_normal(philox_draw(e.seed, ecs, (uint32_t)(i * (D + 1) + el), DQN_STREAM_ENV));
                } else if (v < nobs + cnt) {
                    const int il = v - nobs, i = i0 + il;
                    const u32x4 o = philox_draw(e.seed, ecs, (uint32_t)i, DQN_STREAM_POLICY);   // as policy_row()
                    dw[il * DW + DO] = u01(o.x);
                    dw[il * DW + DO + 1] = __int_as_float((int)(((unsigned long long)o.y * (unsigned long long)A) >> 32));
                } else if (e.kind == 0) {
                    const int il = v - nobs - cnt, i = i0 + il;
                    const u32x4 o = philox_draw(e.seed, ecs, (uint32_t)(i * (D + 1) + D), DQN_STREAM_ENV);
                    const bool done = u01(o.x) < e.p_done;
                    float rew = (((u01(o.y) + u01(o.z)) + (u01(o.w) + u01(o.x))) - 2.0f) * 1.73205078f;
                    if (done) rew = (o.y & 1u) ? 100.0f : -100.0f;
                    dw[il * DW + DO + 2] = done ? 1.0f : 0.0f;
                    dw[il * DW + DO + 3] = rew;
                }
            }
        };
        if (wg < g.tiles) make_draws(4 * wg, e.n - 4 * wg < 4 ? e.n - 4 * wg : 4, 0);
        ASTAMP(20);
        // zero images: h1 / h2 columns past hidden1 / hidden2 and the heads' padding stay zero for the whole launch
        // (16-B stores: every stride is a multiple of four floats / eight bf16, the images start 16-B aligned)
        for (int t = tid; t < (4 * s1 + 4 * s2 + (A + 1) * s2) / 4; t += 256) reinterpret_cast<float4 *>(l1)[t] = float4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) for (int t = tid; t < (4 * s1h + 4 * s2h + (A + 1) * s2h) / 8; t += 256) reinterpret_cast<float4 *>(l1h)[t] = float4{0.f, 0.f, 0.f, 0.f};
        LDS_BARRIER();
        ASTAMP(21);
        // the small operands are the youngest requests: everything issued so far has landed after this wait; the registers
        // are operands of the wait (or of a pin right behind it) so that no use of them can be scheduled ahead of it
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(b1r), "+v"(b2r), "+v"(bh), "+v"(bh16r), "+v"(whv[0]), "+v"(whv[1]), "+v"(whv[2]), "+v"(whv[3]) :: "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(w1r[k]));
        // The commit at the end of the launch needs "every workgroup has READ the counters", not "has finished": an actor
        // workgroup takes its ticket here -- every wave is past the barrier above, i.e. has c0 / ec (/ hs0) in registers --
        // and looks at it at the end, so the returning atomic's round trip is not the tail of the launch.
        asm volatile("" : "+v"(eps));            // (compiler-tracked load: waited for HERE, not by the pin inside the step loop,
                                                 // where a full vmcnt wait would also wait for the previous step's ring stores --
                                                 // and BEFORE the ticket below, whose round trip the same wait would sit out)
        if (tid == 0) {
            unsigned int one = 1u;
            asm volatile("" : "+s"(one) : "s"(c0), "s"(ec), "s"(hs0));
            ticket = ticket_take_early(&e.st->arrive, one);
        }
        // heads: lwh[c][k], c = 0: value column (dddqn.py:29), c = 1..A: advantage columns (:30)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = tid + 256 * u, ct = q >> 6, ln = q & 63, gq = ln >> 4, k = 16 * ct + (ln & 15);
            if (q < nq4) {
                if (gq <= A) lwh[gq * s2 + k] = whv[u][0];
                if (4 + gq <= A) lwh[(4 + gq) * s2 + k] = whv[u][1];
                if (8 + gq <= A) lwh[(8 + gq) * s2 + k] = whv[u][2];
                if (12 + gq <= A) lwh[(12 + gq) * s2 + k] = whv[u][3];
                if constexpr (BF) {
                    if (gq <= A) lwh16[gq * s2h + k] = (__bf16)whv[u][0];
                    if (4 + gq <= A) lwh16[(4 + gq) * s2h + k] = (__bf16)whv[u][1];
                    if (8 + gq <= A) lwh16[(8 + gq) * s2h + k] = (__bf16)whv[u][2];
                    if (12 + gq <= A) lwh16[(12 + gq) * s2h + k] = (__bf16)whv[u][3];
                }
            }
        }
        if (tid < 4 * sx) lx[tid] = x0;                                      // first tile's rows
        ASTAMP(22);
#pragma unroll
        for (int kq = N1; kq < (BF ? 0 : 4 * KB); ++kq) BLD4("=a", w2q[kq], vw2, rsW2, kq * H2 * 16);
        ASTAMP(23);
        // ring slot of env i at step t = (c0 + t*n + i) mod capacity; T*n <= capacity, so one conditional subtraction
        const long long a0 = (long long)mod_cap(c0, e.cap);
        bool w2_landed = false;
        s16x4 w1p[4], w2p[BF ? 4 * KB : 1];                               // bf16 mode: the slabs as four-k bf16 groups
        float bh16 = 0.0f;                                               // bf16 mode: head bias of column `lane` (wave 0)
        if constexpr (BF) {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) w1p[kq] = cvt4(w1r[4 * kq], w1r[4 * kq + 1], w1r[4 * kq + 2], w1r[4 * kq + 3]);
            bh16 = bh16r;                                                // (requested with the small operands of the prologue)
            asm volatile("" : "+v"(bh16));
        }
        const int hpos0 = NSTEP ? (int)(hs0 % (unsigned long long)e.n_step) : 0;   // one 64-bit modulo per launch, not per step

        for (int tile = wg; tile < g.tiles; tile += g.G) {
            const int i0 = 4 * tile;
            const int cnt = e.n - i0 < 4 ? e.n - i0 : 4;
            LDS_BARRIER();                                                   // previous tile's LDS is dead
            for (int t = tid + (tile == wg ? 256 : 0); t < 4 * sx; t += 256) {   // (first tile: elements 0..255 are staged)
                const int il = t / sx, el = t - il * sx;
                lx[t] = (il < cnt && el < D) ? e.env_obs[(long long)(i0 + il) * D + el] : 0.0f;
            }
            if ((e.kind == 1 || e.time_feature) && tid < 4) lt[tid] = tid < cnt ? e.env_t[i0 + tid] : 0;
            LDS_BARRIER();
            ASTAMP(1);

            for (int t = 0, tc = 0; t < g.T; ++t, tc = tc + 1 == g.TC ? 0 : tc + 1) {   // tc = t % TC without the division
                if (tc == 0 && !(tile == wg && t == 0)) make_draws(i0, cnt, t);   // (first chunk of the first tile: made in the prologue)
                const float *dstep = ldraw + (tc * 4) * DW;
                const bool last = t == g.T - 1;
                const unsigned long long ect = ec + (unsigned long long)t;
                const bool emit = !NSTEP || t >= warm;                          // (n-step warm-up: the step is only filed)
                const long long at = a0 + (long long)(emit ? t - warm : 0) * e.n + i0;   // slot of the tile's first env, before the wrap
                const int ns = NSTEP ? e.n_step : 1;
                int hpos = 0, hold = 0;                                         // history slot of this step / of the oldest step of the window
                if constexpr (NSTEP) { hpos = hpos0 + t; hpos -= hpos >= ns ? ns : 0; hpos -= hpos >= ns ? ns : 0;   // (hpos0 < ns, t < 64: loop below)
                              while (hpos >= ns) hpos -= ns;
                              hold = hpos + 1 == ns ? 0 : hpos + 1; }
                // The slabs live in registers for the whole launch: the empty asm makes their values opaque here, so
                // the compiler can neither re-request them from memory inside the step loop nor forget them.
                if constexpr (BF) {
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) asm volatile("" : "+v"(w1p[kq]));
                    asm volatile("" : "+v"(bh16));
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(w1r[k]));
                }
                asm volatile("" : "+v"(b1r), "+v"(b2r), "+v"(bh), "+v"(eps));
                if (t == 1) ASTAMP(30);
                // layer 1: h1 = relu(x @ w1 + b1)                              dddqn.py:25-26
                {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const float *ar = lx + r4 * sx;
                    if constexpr (BF) {
                        if (D <= 16) {
#pragma unroll
                            for (int kq = 0; kq < 4; ++kq) {
                                if (4 * kq < DP) {
                                    const float4 a4 = *reinterpret_cast<const float4 *>(ar + 4 * kq);
                                    acc = MFMA4B(cvt4(a4.x, a4.y, a4.z, a4.w), w1p[kq], acc);
                                }
                            }
                        } else {
                            for (int k0 = 0; k0 < D; k0 += 4) {                 // (padding columns of lx are zero)
                                const float4 a4 = *reinterpret_cast<const float4 *>(ar + k0);
                                float wv[4];
#pragma unroll
                                for (int u = 0; u < 4; ++u) wv[u] = k0 + u < D ? P[m.o_w1 + (long long)(k0 + u) * H1 + cc1] : 0.0f;
                                acc = MFMA4B(cvt4(a4.x, a4.y, a4.z, a4.w), cvt4(wv[0], wv[1], wv[2], wv[3]), acc);
                            }
                        }
                    } else if (D <= 16) {
                        float4 ab[4];
#pragma unroll
                        for (int kq = 0; kq < 4; ++kq) ab[kq] = *reinterpret_cast<const float4 *>(ar + (4 * kq < DP ? 4 * kq : 0));
#pragma unroll
                        for (int kq = 0; kq < 4; ++kq) {
                            if (4 * kq < DP) {
                                acc = MFMA1(ab[kq].x, w1r[4 * kq + 0], acc);
                                acc = MFMA1(ab[kq].y, w1r[4 * kq + 1], acc);
                                acc = MFMA1(ab[kq].z, w1r[4 * kq + 2], acc);
                                acc = MFMA1(ab[kq].w, w1r[4 * kq + 3], acc);
                            }
                        }
                    } else {
                        for (int k = 0; k < D; ++k) acc = MFMA1(ar[k], P[m.o_w1 + (long long)k * H1 + cc1], acc);
                    }
                    if (t == 1) ASTAMP(31);
                    if (col < (unsigned)H1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[r] + b1r;
                            if constexpr (BF) l1h[r * s1h + col] = (__bf16)(v > 0.0f ? v : 0.0f);
                            else l1[r * s1 + col] = v > 0.0f ? v : 0.0f;
                        }
                    }
                }
                if (t == 1) ASTAMP(32);
                if (!w2_landed) {
                    // bf16 mode converts the whole slab once: wait for all of it. Exact mode: the layer-2 chain below waits
                    // for the slab quarter by quarter, so the first step's chain runs while the rest is still arriving.
                    if constexpr (BF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    w2_landed = true; ASTAMP(24);
                    if constexpr (BF) {
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
                        for (int u = 0; u < 2 * KB; ++u) {
                            if (u < 16) asm volatile("" : "+v"(w2q8[u])); else asm volatile("" : "+a"(w2q8[u]));
                            const s16x8 h8 = __builtin_bit_cast(s16x8, w2q8[u]);
                            w2p[2 * u] = s16x4{h8[0], h8[1], h8[2], h8[3]};
                            w2p[2 * u + 1] = s16x4{h8[4], h8[5], h8[6], h8[7]};
                        }
                    }
                }
                LDS_BARRIER();
                ASTAMP(2 + 4 * t);
                if constexpr (BF) {
#pragma unroll
                    for (int kq = 0; kq < 4 * KB; ++kq) asm volatile("" : "+v"(w2p[kq]));
                } else {
#pragma unroll
                    for (int kq = 0; kq < 4 * KB; ++kq) {
                        if (kq < 16) asm volatile("" : "+v"(w2q[kq]));
                        else asm volatile("" : "+a"(w2q[kq]));
                    }
                }
                // layer 2: h2 = relu(h1 @ w2 + b2)                             dddqn.py:27-28
                // A operand (4 consecutive k of this lane's row) read from LDS four groups ahead of its MFMAs
                if constexpr (BF) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const s16x4 *ar = reinterpret_cast<const s16x4 *>(l1h + r4 * s1h);      // four consecutive k of this lane's row
                    s16x4 ab[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) ab[q] = ar[q];
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const s16x4 a4 = ab[u];
                            if (kb + 1 < KB) ab[u] = ar[4 * (kb + 1) + u];
                            acc = MFMA4B(a4, w2p[4 * kb + u], acc);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (col < (unsigned)H2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float v = acc[r] + b2r; l2h[r * s2h + col] = (__bf16)(v > 0.0f ? v : 0.0f); }
                    }
                } else {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const float *ar = l1 + r4 * s1;
                    float4 ab[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) ab[q] = *reinterpret_cast<const float4 *>(ar + 4 * q);
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        // (after the first step these waits find nothing of the slab outstanding)
                        if constexpr (KB >= 4) {
                            constexpr int SEG = KB;                             // 4 KB / 4 float4 registers per quarter
                            if (kb % (KB / 4) == 0) {
                                const int sg = kb / (KB / 4);
                                if (sg == 0) wait_tied<SEG, false, 3 * SEG>(w2q);
                                else if (sg == 1) wait_tied<SEG, (SEG >= 16), 2 * SEG>(w2q + SEG);
                                else if (sg == 2) wait_tied<SEG, (2 * SEG >= 16), SEG>(w2q + 2 * SEG);
                                else wait_tied<SEG, (3 * SEG >= 16), 0>(w2q + 3 * SEG);
                            }
                        } else if (kb == 0) {
                            if constexpr (KB == 2) wait_tied<8, false, 0>(w2q); else wait_tied<4, false, 0>(w2q);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float4 a4 = ab[u];
                            if (kb + 1 < KB) ab[u] = *reinterpret_cast<const float4 *>(ar + 16 * (kb + 1) + 4 * u);
                            acc = MFMA1(a4.x, w2q[4 * kb + u][0], acc);
                            acc = MFMA1(a4.y, w2q[4 * kb + u][1], acc);
                            acc = MFMA1(a4.z, w2q[4 * kb + u][2], acc);
                            acc = MFMA1(a4.w, w2q[4 * kb + u][3], acc);
                            __builtin_amdgcn_sched_barrier(0);               // keep the LDS reads four groups ahead
                        }
                    }
                    if (col < (unsigned)H2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float v = acc[r] + b2r; l2[r * s2 + col] = v > 0.0f ? v : 0.0f; }
                    }
                }
                LDS_BARRIER();
                ASTAMP(3 + 4 * t);
                if constexpr (!BF) {
                    // heads (dddqn.py:29-30): 4 rows x (1+A) columns, each the sum of four partial fmaf chains -- chain j over
                    // the k with (k / 4) % 4 == j (oracle: heads_row) -- one per WAVE: a quarter of the dependent chain each
                    if (hany) {
                        const float *ar = l2 + hr * s2 + 4 * wave, *wr = lwh + hc * s2 + 4 * wave;
                        float acc = 0.0f;
                        float4 ab[4], wb[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            ab[q] = *reinterpret_cast<const float4 *>(ar + 16 * (q < KB2 ? q : 0));
                            wb[q] = *reinterpret_cast<const float4 *>(wr + 16 * (q < KB2 ? q : 0));
                        }
#pragma unroll
                        for (int kb = 0; kb < KB2; ++kb) {
                            const float4 a4 = ab[kb & 3], w4 = wb[kb & 3];
                            if (kb + 4 < KB2) {
                                ab[kb & 3] = *reinterpret_cast<const float4 *>(ar + 16 * (kb + 4));
                                wb[kb & 3] = *reinterpret_cast<const float4 *>(wr + 16 * (kb + 4));
                            }
                            acc = fmaf(a4.x, w4.x, acc); acc = fmaf(a4.y, w4.y, acc);
                            acc = fmaf(a4.z, w4.z, acc); acc = fmaf(a4.w, w4.w, acc);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        lpart[64 * wave + hr * 16 + hc] = acc;
                    }
                    LDS_BARRIER();
                } else {
                    // bf16 mode: the same split on v_mfma_f32_4x4x4_16b_bf16 -- wave w takes the w-th four-k group of every
                    // 16-deep k-block; lane l supplies column l of [wv|wa] (lanes past 1+A: a duplicate of column A, result
                    // unused) and row l&3 of h2; acc[r] of lane l = partial head l of row r
                    const int hcl = lane <= A ? lane : A;
                    const s16x4 *ar = reinterpret_cast<const s16x4 *>(l2h + (lane & 3) * s2h) + wave, *wr = reinterpret_cast<const s16x4 *>(lwh16 + hcl * s2h) + wave;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    s16x4 ab[4], wb[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ab[q] = ar[4 * (q < KB2 ? q : 0)]; wb[q] = wr[4 * (q < KB2 ? q : 0)]; }
#pragma unroll
                    for (int kb = 0; kb < KB2; ++kb) {
                        const s16x4 a4 = ab[kb & 3], w4 = wb[kb & 3];
                        if (kb + 4 < KB2) { ab[kb & 3] = ar[4 * (kb + 4)]; wb[kb & 3] = wr[4 * (kb + 4)]; }
                        acc = MFMA4B(a4, w4, acc);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (lane <= A) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) lpart[64 * wave + r * 16 + lane] = acc[r];
                    }
                    LDS_BARRIER();
                }
                if (wave == 0) {
                    // everything of the policy lanes that does not depend on the forward pass, ahead of the chain (it then
                    // issues in the chain's dependency stalls instead of after it)
                    const int pil = lane < cnt ? lane : 0, pi = i0 + pil;
                    long long pk = at + pil; if (pk >= e.cap) pk -= e.cap;
                    int32_t *p_act = e.actions + pk; float *p_rew = e.rewards + pk; uint8_t *p_done = e.dones + pk;
                    // heads (dddqn.py:29-30): 4 rows x (1+A) columns = 4*(1+A) fmaf chains over hidden2, one per lane
                    if constexpr (BF) {
                        if (lane <= A) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                lq[r * 16 + lane] = ((lpart[r * 16 + lane] + lpart[64 + r * 16 + lane]) + (lpart[128 + r * 16 + lane] + lpart[192 + r * 16 + lane])) + bh16;
                        }
                    } else if (hlane) {
                        const float4 pp = {lpart[hr * 16 + hc], lpart[64 + hr * 16 + hc], lpart[128 + hr * 16 + hc], lpart[192 + hr * 16 + hc]};
                        lq[hr * 16 + hc] = ((pp.x + pp.y) + (pp.z + pp.w)) + bh;
                    }
                    if (t == 1) ASTAMP(25);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // same wave: LDS program order (lq)
                    if (t == 1) ASTAMP(26);
                    if (lane < cnt) {
                        // dueling combine (dddqn.py:31) + epsilon-greedy (q_agent.py:137-141, compute_action :70)
                        const int il = pil, i = pi;
                        // (the 16 head outputs and the draws in registers after ONE LDS latency; unrolled, no indexed arrays)
                        const float4 *hrow = reinterpret_cast<const float4 *>(lq + il * 16);
                        const float4 h0 = hrow[0], h1 = hrow[1], h2 = hrow[2], h3 = hrow[3];
                        const float4 dr = *reinterpret_cast<const float4 *>(dstep + il * DW + DO);   // u, random action, done, reward
                        const float h[16] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z, h2.w, h3.x, h3.y, h3.z, h3.w};
                        float sum = 0.0f;
                        int act = 0;
                        if (A <= 4) {                  // (uniform branch: the usual case without the general one's 15 select-guarded links)
#pragma unroll
                            for (int a = 0; a < 4; ++a) if (a < A) sum = sum + h[1 + a];
                            const float mean = __fdiv_rn(sum, (float)A);
                            float best = (h[0] + h[1]) - mean;
#pragma unroll
                            for (int a = 1; a < 4; ++a) {
                                const float qa = (h[0] + h[1 + a]) - mean;
                                if (a < A && qa > best) { best = qa; act = a; }      // first max wins (compute_action :70)
                            }
                        } else {
#pragma unroll
                            for (int a = 0; a < 15; ++a) if (a < A) sum = sum + h[1 + a];
                            const float mean = __fdiv_rn(sum, (float)A);
                            float best = (h[0] + h[1]) - mean;
#pragma unroll
                            for (int a = 1; a < 15; ++a) {
                                const float qa = (h[0] + h[1 + a]) - mean;
                                if (a < A && qa > best) { best = qa; act = a; }      // first max wins (compute_action :70)
                            }
                        }
                        if (!(eps < dr.x)) act = __float_as_int(dr.y);
                        if (last) g.act_out[i] = act;
                        // the action-dependent part of the transition (q_agent.py:177-183)
                        const long long k = pk;
                        if (e.kind == 1) {
                            float sv[4], s0v[4];
                            for (int j = 0; j < 4; ++j) { sv[j] = lx[il * sx + j]; s0v[j] = sv[j]; }
                            const bool term = cartpole_step(sv, act);
                            const int tt = lt[il] + 1;
                            const bool done = term || tt >= e.max_steps;             // q_agent.py:179-180
                            float rew = term ? e.term_reward : 1.0f;
                            int a_row = act, d_row = done ? 1 : 0;
                            if constexpr (NSTEP) {
                                for (int j = 0; j < 4; ++j) {
                                    e.hist_s[((long long)hpos * e.hist_stride + i) * 4 + j] = s0v[j];
                                    if (emit) s0v[j] = ld_sc1(e.hist_s + ((long long)hold * e.hist_stride + i) * 4 + j);
                                }
                                nstep_row(e, ns, hpos, hold, i, emit, a_row, rew, d_row);
                            }
                            if (emit) {
                                for (int j = 0; j < 4; ++j) { e.states[k * 4 + j] = s0v[j]; e.observations[k * 4 + j] = sv[j]; }
                                e.actions[k] = a_row; e.rewards[k] = rew; e.dones[k] = d_row ? 1 : 0;
                            }
                            if (done) {
                                atomicAdd(&e.st->ep_count, 1ull);
                                atomicAdd(&e.st->ep_steps, (unsigned long long)tt);
                                const u32x4 o = philox_draw(e.seed, ect, (uint32_t)i, DQN_STREAM_ENV);
                                sv[0] = (u01(o.x) * 0.1f) - 0.05f; sv[1] = (u01(o.y) * 0.1f) - 0.05f;
                                sv[2] = (u01(o.z) * 0.1f) - 0.05f; sv[3] = (u01(o.w) * 0.1f) - 0.05f;
                            }
                            lt[il] = done ? 0 : tt;
                            for (int j = 0; j < 4; ++j) lx[il * sx + j] = sv[j];
                            if (last) {
                                e.env_t[i] = done ? 0 : tt;
                                for (int j = 0; j < 4; ++j) e.env_obs[(long long)i * 4 + j] = sv[j];
                            }
                        } else {
                            int a_row = act, d_row = dr.z != 0.0f ? 1 : 0;
                            if constexpr (!NSTEP) { if (e.time_feature && lt[4 * (t & 1) + il] + 1 >= e.max_steps) d_row = 1; }   // q_agent.py:179-180
                            float rew = dr.w;
                            if constexpr (NSTEP) nstep_row(e, ns, hpos, hold, i, emit, a_row, rew, d_row);
                            if (emit) {
                                *p_act = a_row;                                      // replay_buffer.py:60
                                *p_rew = rew;                                        // :61
                                *p_done = d_row ? 1 : 0;                             // :63
                            }
                        }
                    }
                } else {
                    // waves 1..3, meanwhile: the synthetic env's transition of this step out of the draw buffer (ring rows,
                    // LDS state); nothing to do for CartPole, whose transition depends on the action (wave 0 above)
                    const int nobs = e.kind == 0 ? cnt * D : 0;
                    for (int u = tid - 64; u < nobs; u += 192) {
                        const int il = u / D, el = u - il * D, i = i0 + il;
                        long long k = at + il; if (k >= e.cap) k -= e.cap;
                        float nx = dstep[il * DW + el], nx_env = nx;
                        if constexpr (!NSTEP) {
                            if (e.time_feature && el == D - 1) {
                                // LunarLander/env.py:19-24: step pre-incremented, feature = float32(step / max_steps) (python float
                                // division = f64), reset() -> 0; the thread of this column keeps the env's counter
                                const int tt = lt[4 * (t & 1) + il] + 1;
                                const bool done = dstep[il * DW + DO + 2] != 0.0f || tt >= e.max_steps;
                                nx = (float)((double)tt / (double)e.max_steps);
                                nx_env = done ? 0.0f : nx;
                                lt[4 * ((t + 1) & 1) + il] = done ? 0 : tt;
                                if (last) e.env_t[i] = done ? 0 : tt;
                            }
                        }
                        float s_row = lx[il * sx + el];
                        if constexpr (NSTEP) {                                            // n-step: the row starts at the oldest step on file
                            e.hist_s[((long long)hpos * e.hist_stride + i) * D + el] = s_row;
                            if (emit) s_row = ld_sc1(e.hist_s + ((long long)hold * e.hist_stride + i) * D + el);
                        }
                        if (emit) {
                            e.states[k * D + el] = s_row;                        // replay_buffer.py:59
                            e.observations[k * D + el] = nx;                     // :62
                        }
                        lx[il * sx + el] = nx_env;                               // q_agent.py:183 (read by this thread only)
                        if (last) e.env_obs[(long long)i * D + el] = nx_env;
                    }
                }
                LDS_BARRIER();
                ASTAMP(4 + 4 * t);
                ASTAMP(5 + 4 * t);
            }
        }
    }

    // the last workgroup of the launch to arrive commits the counters (every thread's stores depend on c0 / ec, so a
    // workgroup that reaches this barrier has finished reading them)
    LDS_BARRIER();
    if (tid == 0) {
        if (role != 1) ticket = atomicAdd(&e.st->arrive, 1u);                                     // (actor workgroups: taken in the prologue,
                                                                                                  //  ticket_take_early, dqn_device.h)
        if (ticket == total_wgs - 1u) {
            e.st->ring_counter = c1;                                                              // replay_buffer.py:64
            e.st->size = (long long)(c1 < (unsigned long long)e.cap ? c1 : (unsigned long long)e.cap);   // :65
            e.st->env_ctr = ec + (unsigned long long)g.T;
            if constexpr (NSTEP) e.st->hist_steps = hs0 + (unsigned long long)g.T;
            e.st->fill_cnt = 0;
            e.st->arrive = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_actor16 -- the same launch (T vector env steps, side workgroups, chunked draws) for SMALL nets (hidden sizes <= 128,
// obs_dim <= 16, exact f32, one-step returns): 16 envs per workgroup on v_mfma_f32_16x16x4_f32. With so little weight data
// the per-wave share of all three layers (fragment-packed shadows of dqn_net.hip: <= 2 column tiles x <= 8 k-blocks) lives in
// registers for the whole launch, a layer is 16-128 MFMAs, and what a step costs is its fixed latency (three barriers, the
// policy / physics lanes) -- paid once per 16 envs here instead of once per 4: BASELINE configs[2] (4 096 CartPole envs,
// 2x64 net) has 1 024 four-env tiles but only 256 sixteen-env ones, one per CU. Same arithmetic as k_qnet_fwd (k-ordered
// fmaf chains), hence the same actions and rows as k_actor and the CPU restatement.
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ int perm16a(int c) { return ((c & 3) << 2) | (c >> 2); }   // A-operand position of column c in its 16-block

template <int KQ>                     // 16-deep k-blocks held per layer: hidden1, hidden2 <= 16*KQ; KQ/4 column tiles per wave
__global__ void __launch_bounds__(256)
k_actor16(NetDims m, ActorArgs g) {
    constexpr int TN = KQ / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const EnvArgs &e = g.env;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long c0 = e.st->ring_counter, ec = e.st->env_ctr;
    const unsigned long long nT = (unsigned long long)g.T * (unsigned long long)e.n, c1 = c0 + nT;
    const unsigned long long ticket_val = ec + (unsigned long long)g.T;
    unsigned int ticket = 0u;                    // arrival ticket (thread 0), taken early by actor workgroups as in k_actor
    const unsigned total_wgs = (unsigned)(g.n_tree + g.G + g.n_smp);   // (= gridDim.x, without reading the dispatch packet)
    int role = 1, wg = (int)blockIdx.x - g.n_tree;
    if ((int)blockIdx.x < g.n_tree) role = 0;
    else if (wg >= g.G) { role = 2; wg -= g.G; }

    if (role != 1) {
        actor_side_role(role, wg, g, c0, nT, c1, ticket_val, lds);
    } else {
        const int D = m.D, H1 = m.H1, H2 = m.H2, A = m.A;
        const int sx = 16 + 4, sh = 16 * KQ + 4;
        const int DO = (D + 3) & ~3, DW = DO + 4;
        float *lx = lds, *l1 = lx + 16 * sx, *l2 = l1 + 16 * sh, *lh = l2 + 16 * sh;
        int *lt = reinterpret_cast<int *>(lh + 256);                         // [2][16] (see k_actor)
        float *ldraw = lh + 256 + 32;
        const float *P = g.params;
        const int c15 = lane & 15, g4 = lane >> 4;
        const int KQ2 = H1 / 16, KQH = H2 / 16, CT1 = H1 / 16, CT2 = H2 / 16;

        // this wave's share of the three layers, register-resident (clamped requests + selects: no branch around a load)
        const f32x4 *p1 = reinterpret_cast<const f32x4 *>(g.pack + m.p_w1), *p2 = reinterpret_cast<const f32x4 *>(g.pack + m.p_w2),
                    *ph = reinterpret_cast<const f32x4 *>(g.pack + m.p_wh);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 w1f[TN], w2f[KQ][TN], whf[KQ];
        float b1[TN], b2[TN], bh;
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int ct = wave + 4 * t;
            const f32x4 v = p1[(long long)(ct < CT1 ? ct : 0) * 64 + lane];          // KQ1 = 1 (obs_dim <= 16)
            w1f[t] = ct < CT1 ? v : z4;
            const float bv1 = P[m.o_b1 + 16 * (ct < CT1 ? ct : 0) + c15], bv2 = P[m.o_b2 + 16 * (ct < CT2 ? ct : 0) + c15];
            b1[t] = ct < CT1 ? bv1 : 0.0f; b2[t] = ct < CT2 ? bv2 : 0.0f;
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                const bool ok = ct < CT2 && kq < KQ2;
                const f32x4 w = p2[((long long)(ok ? ct : 0) * KQ2 + (ok ? kq : 0)) * 64 + lane];
                w2f[kq][t] = ok ? w : z4;
            }
        }
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) { const f32x4 w = ph[(long long)(kq < KQH ? kq : 0) * 64 + lane]; whf[kq] = kq < KQH ? w : z4; }
        { const float bvv = P[c15 == 0 || c15 > A ? m.o_bv : m.o_ba + c15 - 1]; bh = c15 <= A ? bvv : 0.0f; }
        float eps = e.st->epsilon;
        const long long a0 = (long long)mod_cap(c0, e.cap);
        ASTAMP(0);

        auto make_draws = [&](int i0, int cnt, int t0) {                     // Philox draws of steps t0 .. t0+TC-1 (waves 1..3)
            if (wave == 0) return;
            const int nobs = e.kind == 0 ? cnt * D : 0, per = nobs + 2 * cnt;
            const int nst = g.T - t0 < g.TC ? g.T - t0 : g.TC;
            for (int u = tid - 64; u < nst * per; u += 192) {
                const int ts = u / per, v = u - ts * per;
                const unsigned long long ecs = ec + (unsigned long long)(t0 + ts);
                float *dw = ldraw + (ts * 16) * DW;
                if (v < nobs) {
                    const int il = v / D, el = v - il * D, i = i0 + il;
                    dw[il * DW + el] = ih_normal(philox_draw(e.seed, ecs, (uint32_t)(i * (D + 1) + el), DQN_STREAM_ENV));
                } else if (v < nobs + cnt) {
                    const int il = v - nobs, i = i0 + il;
                    const u32x4 o = philox_draw(e.seed, ecs, (uint32_t)i, DQN_STREAM_POLICY);   // as policy_row()
                    dw[il * DW + DO] = u01(o.x);
                    dw[il * DW + DO + 1] = __int_as_float((int)(((unsigned long long)o.y * (unsigned long long)A) >> 32));
                } else if (e.kind == 0) {
                    const int il = v - nobs - cnt, i = i0 + il;
                    const u32x4 o = philox_draw(e.seed, ecs, (uint32_t)(i * (D + 1) + D), DQN_STREAM_ENV);
                    const bool done = u01(o.x) < e.p_done;
                    float rew = (((u01(o.y) + u01(o.z)) + (u01(o.w) + u01(o.x))) - 2.0f) * 1.73205078f;
                    if (done) rew = (o.y & 1u) ? 100.0f : -100.0f;
                    dw[il * DW + DO + 2] = done ? 1.0f : 0.0f;
                    dw[il * DW + DO + 3] = rew;
                }
            }
        };
        for (int t = tid; t < 16 * sx + 2 * 16 * sh; t += 256) lx[t] = 0.0f;  // padding columns stay zero for the whole launch
        // every load above is waited for here, outside the step loop
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            asm volatile("" : "+v"(w1f[t]), "+v"(b1[t]), "+v"(b2[t]));
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) asm volatile("" : "+v"(w2f[kq][t]));
        }
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) asm volatile("" : "+v"(whf[kq]));
        asm volatile("" : "+v"(bh), "+v"(eps));

        for (int tile = wg; tile < g.tiles; tile += g.G) {
            const int i0 = 16 * tile;
            const int cnt = e.n - i0 < 16 ? e.n - i0 : 16;
            LDS_BARRIER();                                                   // previous tile's LDS is dead; zero-fill done
            {
                const int rl = tid >> 4, c = tid & 15;
                lx[rl * sx + perm16a(c)] = (rl < cnt && c < D) ? e.env_obs[(long long)(i0 + rl) * D + c] : 0.0f;
            }
            if ((e.kind == 1 || e.time_feature) && tid < 16) lt[tid] = tid < cnt ? e.env_t[i0 + tid] : 0;
            LDS_BARRIER();
            if (tile == wg) ASTAMP(1);
            if (tile == wg && tid == 0) {                                    // arrival ticket, early (see k_actor): every wave has c0 / ec
                unsigned int one = 1u;
                asm volatile("" : "+s"(one) : "s"(c0), "s"(ec));
                ticket = ticket_take_early(&e.st->arrive, one);
            }

            for (int t = 0, tc = 0; t < g.T; ++t, tc = tc + 1 == g.TC ? 0 : tc + 1) {   // tc = t % TC without the division
                if (tc == 0) make_draws(i0, cnt, t);
                const float *dstep = ldraw + (tc * 16) * DW;
                const bool last = t == g.T - 1;
                const unsigned long long ect = ec + (unsigned long long)t;
                const long long at = a0 + (long long)t * e.n + i0;
#pragma unroll
                for (int tt = 0; tt < TN; ++tt) {
                    asm volatile("" : "+v"(w1f[tt]), "+v"(b1[tt]), "+v"(b2[tt]));
#pragma unroll
                    for (int kq = 0; kq < KQ; ++kq) asm volatile("" : "+v"(w2f[kq][tt]));
                }
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq) asm volatile("" : "+v"(whf[kq]));
                asm volatile("" : "+v"(bh), "+v"(eps));
                // layer 1: h1 = relu(x @ w1 + b1)                              dddqn.py:25-26
                {
                    const float4 a4 = *reinterpret_cast<const float4 *>(lx + c15 * sx + 4 * g4);
#pragma unroll
                    for (int tt = 0; tt < TN; ++tt) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        acc = MFMA4(a4.x, w1f[tt][0], acc); acc = MFMA4(a4.y, w1f[tt][1], acc);
                        acc = MFMA4(a4.z, w1f[tt][2], acc); acc = MFMA4(a4.w, w1f[tt][3], acc);
                        const int ct = wave + 4 * tt;
                        if (ct < CT1) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { const float v = acc[r] + b1[tt]; l1[(4 * g4 + r) * sh + 16 * ct + perm16a(c15)] = v > 0.0f ? v : 0.0f; }
                        }
                    }
                }
                LDS_BARRIER();
                if (tile == wg) ASTAMP(2 + 4 * t);
                // layer 2: h2 = relu(h1 @ w2 + b2)                             dddqn.py:27-28
                {
                    f32x4 acc[TN];
#pragma unroll
                    for (int tt = 0; tt < TN; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const float *ar = l1 + c15 * sh + 4 * g4;
#pragma unroll
                    for (int kq = 0; kq < KQ; ++kq) {
                        const float4 a4 = *reinterpret_cast<const float4 *>(ar + 16 * kq);
#pragma unroll
                        for (int tt = 0; tt < TN; ++tt) {
                            acc[tt] = MFMA4(a4.x, w2f[kq][tt][0], acc[tt]); acc[tt] = MFMA4(a4.y, w2f[kq][tt][1], acc[tt]);
                            acc[tt] = MFMA4(a4.z, w2f[kq][tt][2], acc[tt]); acc[tt] = MFMA4(a4.w, w2f[kq][tt][3], acc[tt]);
                        }
                    }
#pragma unroll
                    for (int tt = 0; tt < TN; ++tt) {
                        const int ct = wave + 4 * tt;
                        if (ct < CT2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { const float v = acc[tt][r] + b2[tt]; l2[(4 * g4 + r) * sh + 16 * ct + perm16a(c15)] = v > 0.0f ? v : 0.0f; }
                        }
                    }
                }
                LDS_BARRIER();
                if (tile == wg) ASTAMP(3 + 4 * t);
                if (wave == 0) {
                    // heads: column 0 = val (dddqn.py:29), columns 1..A = adv (:30); then dueling combine + policy per env
                    // (four chains, the j-th MFMA of every k-block into chain j, combined (c0 + c1) + (c2 + c3): as k_qnet_fwd / heads_row)
                    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
                    const float *ar = l2 + c15 * sh + 4 * g4;
#pragma unroll
                    for (int kq = 0; kq < KQ; ++kq) {
                        const float4 a4 = *reinterpret_cast<const float4 *>(ar + 16 * kq);
                        c0 = MFMA4(a4.x, whf[kq][0], c0); c1 = MFMA4(a4.y, whf[kq][1], c1);
                        c2 = MFMA4(a4.z, whf[kq][2], c2); c3 = MFMA4(a4.w, whf[kq][3], c3);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) lh[(4 * g4 + r) * 16 + c15] = ((c0[r] + c1[r]) + (c2[r] + c3[r])) + bh;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // same wave: LDS program order
                    if (lane < cnt) {
                        const int il = lane, i = i0 + il;
                        const float4 *hrow = reinterpret_cast<const float4 *>(lh + il * 16);
                        const float4 h0 = hrow[0], h1 = hrow[1], h2 = hrow[2], h3 = hrow[3];
                        const float4 dr = *reinterpret_cast<const float4 *>(dstep + il * DW + DO);   // u, random action, done, reward
                        const float h[16] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z, h2.w, h3.x, h3.y, h3.z, h3.w};
                        float sum = 0.0f;
                        int act = 0;
                        if (A <= 4) {                  // (uniform branch: the usual case without the general one's 15 select-guarded links)
#pragma unroll
                            for (int a = 0; a < 4; ++a) if (a < A) sum = sum + h[1 + a];
                            const float mean = __fdiv_rn(sum, (float)A);             // dddqn.py:31
                            float best = (h[0] + h[1]) - mean;
#pragma unroll
                            for (int a = 1; a < 4; ++a) {
                                const float qa = (h[0] + h[1 + a]) - mean;
                                if (a < A && qa > best) { best = qa; act = a; }      // first max wins (compute_action :70)
                            }
                        } else {
#pragma unroll
                            for (int a = 0; a < 15; ++a) if (a < A) sum = sum + h[1 + a];
                            const float mean = __fdiv_rn(sum, (float)A);
                            float best = (h[0] + h[1]) - mean;
#pragma unroll
                            for (int a = 1; a < 15; ++a) {
                                const float qa = (h[0] + h[1 + a]) - mean;
                                if (a < A && qa > best) { best = qa; act = a; }
                            }
                        }
                        if (!(eps < dr.x)) act = __float_as_int(dr.y);               // q_agent.py:137-141
                        if (last) g.act_out[i] = act;
                        long long k = at + il; if (k >= e.cap) k -= e.cap;
                        if (e.kind == 1) {
                            float sv[4];
                            for (int j = 0; j < 4; ++j) { sv[j] = lx[il * sx + perm16a(j)]; e.states[k * 4 + j] = sv[j]; }
                            const bool term = cartpole_step(sv, act);
                            const int tt = lt[il] + 1;
                            const bool done = term || tt >= e.max_steps;             // q_agent.py:179-180
                            for (int j = 0; j < 4; ++j) e.observations[k * 4 + j] = sv[j];
                            e.actions[k] = act; e.rewards[k] = term ? e.term_reward : 1.0f; e.dones[k] = done ? 1 : 0;
                            if (done) {
                                atomicAdd(&e.st->ep_count, 1ull);
                                atomicAdd(&e.st->ep_steps, (unsigned long long)tt);
                                const u32x4 o = philox_draw(e.seed, ect, (uint32_t)i, DQN_STREAM_ENV);
                                sv[0] = (u01(o.x) * 0.1f) - 0.05f; sv[1] = (u01(o.y) * 0.1f) - 0.05f;
                                sv[2] = (u01(o.z) * 0.1f) - 0.05f; sv[3] = (u01(o.w) * 0.1f) - 0.05f;
                            }
                            lt[il] = done ? 0 : tt;
                            for (int j = 0; j < 4; ++j) lx[il * sx + perm16a(j)] = sv[j];
                            if (last) {
                                e.env_t[i] = done ? 0 : tt;
                                for (int j = 0; j < 4; ++j) e.env_obs[(long long)i * 4 + j] = sv[j];
                            }
                        } else {
                            e.actions[k] = act;                                      // replay_buffer.py:60
                            e.rewards[k] = dr.w;                                     // :61
                            e.dones[k] = (dr.z != 0.0f || (e.time_feature && lt[16 * (t & 1) + il] + 1 >= e.max_steps)) ? 1 : 0;   // :63, q_agent.py:179-180
                        }
                    }
                } else if (e.kind == 0) {
                    // waves 1..3, meanwhile: the synthetic transition of the step out of the draw buffer
                    for (int u = tid - 64; u < cnt * D; u += 192) {
                        const int il = u / D, el = u - il * D, i = i0 + il;
                        long long k = at + il; if (k >= e.cap) k -= e.cap;
                        float nx = dstep[il * DW + el], nx_env = nx;
                        if (e.time_feature && el == D - 1) {                         // LunarLander/env.py:19-24 (see k_actor)
                            const int tt = lt[16 * (t & 1) + il] + 1;
                            const bool done = dstep[il * DW + DO + 2] != 0.0f || tt >= e.max_steps;
                            nx = (float)((double)tt / (double)e.max_steps);
                            nx_env = done ? 0.0f : nx;
                            lt[16 * ((t + 1) & 1) + il] = done ? 0 : tt;
                            if (last) e.env_t[i] = done ? 0 : tt;
                        }
                        e.states[k * D + el] = lx[il * sx + perm16a(el)];            // replay_buffer.py:59
                        e.observations[k * D + el] = nx;                             // :62
                        lx[il * sx + perm16a(el)] = nx_env;                          // q_agent.py:183 (read by this thread only)
                        if (last) e.env_obs[(long long)i * D + el] = nx_env;
                    }
                }
                LDS_BARRIER();
                if (tile == wg) { ASTAMP(4 + 4 * t); ASTAMP(5 + 4 * t); }
            }
        }
    }
    LDS_BARRIER();
    if (tid == 0) {
        if (role != 1) ticket = atomicAdd(&e.st->arrive, 1u);                                     // (actor workgroups: taken in the prologue,
                                                                                                  //  ticket_take_early, dqn_device.h)
        if (ticket == total_wgs - 1u) {
            e.st->ring_counter = c1;                                                              // replay_buffer.py:64
            e.st->size = (long long)(c1 < (unsigned long long)e.cap ? c1 : (unsigned long long)e.cap);   // :65
            e.st->env_ctr = ec + (unsigned long long)g.T;
            e.st->fill_cnt = 0;
            e.st->arrive = 0;
        }
    }
}

// host side -------------------------------------------------------------------------------------------------
bool actor_multi_supported(const NetDims &m, int n_envs, int T) {
    (void)n_envs;
    return T >= 1 && m.H1 <= 256 && m.H2 <= 256 && m.D <= 256;
}

// sampler workgroups of an actor launch: one per 16 NR batch rows, at most `cap`; NR (rows per lane group, in flight
// together) doubles up to 4 before a workgroup has to loop over tiles
static void set_samplers(ActorArgs &g, int B, int cap) {
    g.NR = 1; g.n_smp = 0;
    if (B <= 0 || cap < 1) return;
    while (g.NR < 4 && (B + 16 * g.NR - 1) / (16 * g.NR) > cap) g.NR *= 2;
    g.n_smp = (B + 16 * g.NR - 1) / (16 * g.NR);
    if (g.n_smp > cap) g.n_smp = cap;
}

bool launch_actor_multi(hipStream_t s, const NetDims &m, const EnvArgs &env, int T, const float *params, const float *pack,
                        int32_t *act_out, int B, const SampleArgs *smp, bool bf16, int num_cus, bool no_wide) {
    ActorArgs g{};
    g.env = env; g.T = T; g.params = params; g.pack = pack; g.act_out = act_out;
    // The tree workgroup, the samplers and (through the arrival tickets) the actors of one launch wait for each other: the
    // whole grid has to be resident. Residency comes from the grid size alone (cdna guide, Workgroups / residency):
    // res = workgroups of this kernel a CU holds x the device's CUs, minus one CU of margin.
    const bool want_smp = smp && env.tree && B > 0;
    // small nets (exact f32, one-step returns): 16 envs per workgroup on the 16x16x4 MFMA, all weights in registers (k_actor16)
    const bool wide = !bf16 && env.n_step <= 1 && m.D <= 16 && m.H1 <= 128 && m.H2 <= 128 && !no_wide;
    if (wide) {
        // (about 110-180 registers per thread and <= 70 KB of LDS: two workgroups of this kernel share a CU -- room for one
        // sampler workgroup per 16 batch rows beside the actors)
        const int res = 2 * (num_cus - 1);
        g.n_tree = env.tree ? 1 : 0;
        g.tiles = (env.n + 15) / 16;
        const int half = res / 2;                                        // at least half the machine stays with the actors' tiles
        set_samplers(g, want_smp ? B : 0, (g.tiles <= half ? res - 1 - g.tiles : half - 1));
        const int room = res - g.n_tree - g.n_smp;
        g.G = g.tiles < room ? g.tiles : room;
        if (g.G < 1) { g.n_smp = 0; g.G = 1; }                           // (a device of one CU: actors loop, the tree workgroup queues behind them without anyone waiting on it)
        g.B = B;
        if (g.n_smp) g.smp = *smp;
    const int KQ = (m.H1 <= 64 && m.H2 <= 64) ? 4 : 8;
        const int DWh = ((m.D + 3) & ~3) + 4;
        g.TC = 6144 / (16 * DWh);
        if (g.TC > T) g.TC = T;
        if (g.TC < 1) g.TC = 1;
        size_t lds = sizeof(float) * (16 * 20 + 2 * 16 * (size_t)(16 * KQ + 4) + 256 + 32 + (size_t)g.TC * 16 * DWh);
        if (g.n_tree) {
            size_t need = sizeof(float) * 64;
            if (lds < need) lds = need;
            if (env.rebuild_top) { need = sizeof(float) * (env.L >= PW_TOP ? (size_t)256 * 68 : (size_t)1 << env.L); if (lds < need) lds = need; }
        }
        if (g.n_smp && lds < sizeof(float) * (2048 + 64)) lds = sizeof(float) * (2048 + 64);
        const dim3 grid(g.n_tree + g.G + g.n_smp), block(256);
        if (KQ == 4) DQN_LAUNCH((k_actor16<4>), grid, block, lds, s, m, g);
        else DQN_LAUNCH((k_actor16<8>), grid, block, lds, s, m, g);
        return g.n_smp > 0;
    }
    // the register-resident weight slab limits a CU to ONE workgroup of this kernel, whatever its role: keep the
    // grid within the device's CUs so that tree, sampler and actor workgroups all run side by side
    const int res = num_cus - 1;
    g.n_tree = env.tree ? 1 : 0;
    set_samplers(g, want_smp ? B : 0, (res + 1) / 4 < 64 ? (res + 1) / 4 : 64);
    g.tiles = (env.n + 3) / 4;
    const int room = res - g.n_tree - g.n_smp;
    g.G = g.tiles < room ? g.tiles : room;
    if (g.G < 1) { g.n_smp = 0; g.G = 1; }
    g.B = B;
    if (g.n_smp) g.smp = *smp;
    const int DP = (m.D + 3) & ~3;
    const int KB = m.H1 <= 16 ? 1 : (m.H1 <= 32 ? 2 : (m.H1 <= 64 ? 4 : (m.H1 <= 128 ? 8 : 16)));
    const int KB2 = m.H2 <= 64 ? 4 : 16;
    size_t lds = sizeof(float) * (4 * (size_t)(DP + 4) + 4 * (size_t)(16 * KB + 4) + 4 * (size_t)(16 * KB2 + 4) +
                                  (size_t)(m.A + 1) * (16 * KB2 + 4) + 64 + 8 + 256);
    const int DWh = ((m.D + 3) & ~3) + 4;
    g.TC = 6144 / (4 * DWh);                                              // draw buffer: <= 24 KB
    if (g.TC > T) g.TC = T;
    if (g.TC < 1) g.TC = 1;
    lds += sizeof(float) * (size_t)g.TC * 4 * DWh;
    if (bf16) lds += 2 * (4 * (size_t)(16 * KB + 8) + (size_t)(m.A + 5) * (16 * KB2 + 8));
    if (g.n_tree) {
        size_t need = sizeof(float) * 64;
        if (lds < need) lds = need;
        if (env.rebuild_top) { need = sizeof(float) * (env.L >= PW_TOP ? (size_t)256 * 68 : (size_t)1 << env.L); if (lds < need) lds = need; }
    }
    if (g.n_smp && lds < sizeof(float) * (2048 + 64)) lds = sizeof(float) * (2048 + 64);
    const dim3 grid(g.n_tree + g.G + g.n_smp), block(256);
#define ACTOR_CASE(K1, K2) if (KB == K1 && KB2 == K2) {                                                     \
        if (bf16) { if (env.n_step > 1) DQN_LAUNCH((k_actor<K1, K2, true, true>), grid, block, lds, s, m, g);    \
                    else DQN_LAUNCH((k_actor<K1, K2, false, true>), grid, block, lds, s, m, g); }                \
        else { if (env.n_step > 1) DQN_LAUNCH((k_actor<K1, K2, true, false>), grid, block, lds, s, m, g);        \
               else DQN_LAUNCH((k_actor<K1, K2, false, false>), grid, block, lds, s, m, g); }                    \
        return g.n_smp > 0; }
    ACTOR_CASE(1, 4) ACTOR_CASE(2, 4) ACTOR_CASE(4, 4) ACTOR_CASE(8, 4) ACTOR_CASE(16, 4)
    ACTOR_CASE(1, 16) ACTOR_CASE(2, 16) ACTOR_CASE(4, 16) ACTOR_CASE(8, 16) ACTOR_CASE(16, 16)
#undef ACTOR_CASE
    return false;
}
