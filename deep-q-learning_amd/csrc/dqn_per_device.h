// csrc/dqn_per_device.h -- device code of the sorted priority write-back, shared by dqn_replay.hip
// (k_per_write_sorted) and dqn_net.hip (surplus workgroups of k_dw).
#pragma once
#include "dqn_device.h"

#define PW_TOP 10
#define PW_BOT 21            // bottom levels handled below the dense top image: L - PW_TOP <= 21 (L <= 31)

__device__ __forceinline__ int shfl_i(int v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ float shfl_f(float v, int src) { return __shfl(v, src, 64); }

// one wave = one 64-position chunk of the batch (chunk index `chunk`); called by k_per_write_sorted and by
// the surplus workgroups of k_dw (the write-back then shares a launch with the weight gradients)
__device__ __forceinline__ void per_write_sorted_wave(DqnState *st, float *tree, long long N, int L,
                                                      const int32_t *__restrict__ idx, const float *__restrict__ val,
                                                      int B, int mode, float alpha, float eps, int chunk) {
    const int lane = threadIdx.x & 63, base = chunk * 64;
    const int TOP = L < PW_TOP ? L : PW_TOP, SH = L - TOP;       // depth-TOP subtree id = leaf >> SH
    STAMP(4, 0);
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
    STAMP(4, 1);

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
        STAMP(4, 2);
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
    STAMP(4, 3);
    // running max priority: order-independent (positive floats order like their bit patterns)
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
    if (lane == 0) atomicMax(reinterpret_cast<unsigned int *>(&st->pmax), __float_as_uint(lmax));
    STAMP(4, 4);
}

