// csrc/dqn_device.h -- shared device-side definitions (gfx950 only).
// Compiled with -ffp-contract=off: every f32 op written here is one IEEE rounding, so the
// integer/indexing paths (Philox, sum-tree, deterministic pow, Adam) are bit-reproducible
// against the CPU restatement. FMA is used only where written explicitly (fmaf / MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, NOT for
// its outstanding global loads / stores (a __syncthreads() would drain prefetches and stores)
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Mutable scalars that must live on the device so that a hipGraph replay sees them change.
struct DqnState {
    unsigned long long ring_counter;   // ReplayBuffer._counter      (replay_buffer.py:33)
    long long          size;           // ReplayBuffer._num_samples  (replay_buffer.py:34)
    unsigned long long sample_ctr;     // Philox counter of the fused path (one per update)
    unsigned long long epoch;          // write-back epoch for duplicate resolution
    unsigned long long env_ctr;        // vector env steps taken (Philox counter of policy / synthetic env)
    double             b1pow, b2pow;   // running b1^t, b2^t
    int                adam_count;     // ScaleByAdamState.count
    float              pmax;           // running max priority
    float              beta;           // IS exponent for the fused path
    float              lr;
    float              loss;           // last loss
    float              wmax;           // max raw IS weight of the batch in flight (atomicMax by the sampler,
                                       // reset by k_dw once consumed)
    float              epsilon;        // exploration rate of the fused actor step
    float              pad0;
    unsigned int       arrive;         // last-block tickets
    unsigned int       pad;
    unsigned long long ep_count;       // finished episodes of the device-resident envs (CartPole)
    unsigned long long ep_steps;       // env steps (= return, reward 1 per step) summed over finished episodes
    unsigned long long hist_steps;     // vector env steps filed in the n-step history since dqn_env_reset
    unsigned int       err_count;      // in-kernel hand-over waits that gave up (bounded spins; dqn_device_errors_host). While it is
                                       // non-zero every loss written is NaN: a timed-out launch has computed on incomplete data
    unsigned int       pad1;
    // cross-workgroup hand-overs of an actor launch, each on its own 128-B line (hundreds of workgroups poll / bump them):
    alignas(128) unsigned long long tree_ready;   // env step counter up to which the leaves are in the tree: released by the
                                       // tree workgroup, awaited by the sampler workgroups (dqn_actor.hip)
    alignas(128) unsigned int fill_cnt;           // sampler workgroups that have stored their share of the new leaves' inner
                                       // nodes (reset by the launch's commit)
};

// Bounded wait of the in-launch hand-overs (one lane polls one word): gives up after DQN_WAIT_TICKS of the 100 MHz
// constant clock (0.2 s; the waits themselves last microseconds) so that a launch whose partner workgroups are not
// resident -- a partitioned or shared device -- ends with an error count instead of hanging the GPU.
#define DQN_WAIT_TICKS 20000000ull
template <class T>
__device__ __forceinline__ bool wait_word_eq(const T *word, T want, int sleep) {
    if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        for (int i = 0; i < 64; ++i) {
            if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) return true;
            if (sleep >= 8) __builtin_amdgcn_s_sleep(8); else if (sleep >= 4) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(2);
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > DQN_WAIT_TICKS) return false;
    }
}
__device__ __forceinline__ void flag_wait_timeout(DqnState *st) { atomicAdd(&st->err_count, 1u); }

enum { DQN_STREAM_PER = 0, DQN_STREAM_UNIFORM = 1, DQN_STREAM_POLICY = 2, DQN_STREAM_ENV = 3 };

// ---------------------------------------------------------------- Philox4x32-10
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ u32x4 philox_draw(unsigned long long seed, unsigned long long ctr,
                                             uint32_t k, uint32_t stream) {
    return philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), k, stream,
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 0x1.0p-24f; }

// ------------------------------------------------- deterministic f32 pow (no FMA)
__device__ __forceinline__ float log2_det(float x) {
    const uint32_t u = __float_as_uint(x);
    int e = (int)((u >> 23) & 0xFF) - 127;
    float m = __uint_as_float((u & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float s = __fdiv_rn(m - 1.0f, m + 1.0f);
    const float z = s * s;
    float p = 0.111111112f;
    p = p * z; p = p + 0.142857149f;
    p = p * z; p = p + 0.2f;
    p = p * z; p = p + 0.333333343f;
    p = p * z; p = p + 1.0f;
    const float ln_m = (2.0f * s) * p;
    const float r = ln_m * 1.44269502f;
    return (float)e + r;
}

__device__ __forceinline__ float exp2_det(float y) {
    const float fi = floorf(y + 0.5f);
    int i = (int)fi;
    const float f = y - fi;
    const float t = f * 0.693147182f;
    float p = 1.98412701e-4f;
    p = p * t; p = p + 1.38888892e-3f;
    p = p * t; p = p + 8.33333377e-3f;
    p = p * t; p = p + 4.16666679e-2f;
    p = p * t; p = p + 0.166666672f;
    p = p * t; p = p + 0.5f;
    p = p * t; p = p + 1.0f;
    p = p * t; p = p + 1.0f;
    i = i < -126 ? -126 : (i > 127 ? 127 : i);
    return p * __uint_as_float((uint32_t)(i + 127) << 23);
}

__device__ __forceinline__ float pow_det(float x, float a) { return exp2_det(a * log2_det(x)); }

// Arrival ticket of the "last workgroup commits" blocks, taken early and looked at at the end of the kernel. `atomicAdd` on
// a uniform address goes through the compiler's atomic optimiser (one lane adds, s_waitcnt vmcnt(0), v_readfirstlane
// broadcast): the caller stalls for the whole round trip where the ticket is TAKEN (seen in the ISA, r02). Here the address
// is made opaque (a "v" asm output is a divergent value for the optimiser), so the one calling lane issues a plain returning
// `global_atomic_add_u32 ... sc0` whose result is an ordinary compiler-tracked VGPR: the compiler places the vmcnt wait in
// front of the first instruction that reads it. (r02's inline-asm atomic handed the compiler a result register that was not
// valid yet; under register pressure it was copied to an AGPR at once -- a stale ticket -- and the VGPR reused while the
// atomic was in flight: ADVICE r02, tools/isa_scan.py now rejects any such pattern.)
__device__ __forceinline__ unsigned int ticket_take_early(unsigned int *p, unsigned int v) {
    // (as a GLOBAL pointer: an opaque generic pointer becomes a flat atomic, which also counts in lgkmcnt -- the next LDS
    // barrier would sit out its round trip)
    __attribute__((address_space(1))) unsigned int *g = (__attribute__((address_space(1))) unsigned int *)p;
    asm volatile("" : "+v"(g));
    return __hip_atomic_fetch_add(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ diagnostic stamps
// -DDQN_STAMPS builds (tools only, never shipped): thread 0 of block (0,0) records
// (s_memtime shader-clock ticks, s_memrealtime 100 MHz ticks) at named points of a kernel.
#ifdef DQN_STAMPS
extern __device__ unsigned long long g_stamps[8][64][2];   // [kernel][slot][clock kind]
#define STAMP(K, S)                                                                         \
    do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {                       \
             g_stamps[K][S][0] = __builtin_amdgcn_s_memtime();                                \
             g_stamps[K][S][1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define STAMP(K, S) do { } while (0)

#endif
