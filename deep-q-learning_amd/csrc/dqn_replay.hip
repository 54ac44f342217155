// csrc/dqn_replay.hip -- replay ring and proportional-PER sum-tree kernels (gfx950).
//
// Integer / indexing work, HBM- (in practice Infinity-Cache-) bound. Bit-exact against the
// CPU restatement: Philox draws, the f32 tree descent (compares and subtractions only), the
// deterministic pow, and parents recomputed as left + right (never delta-added, no atomics
// on tree values).
//
// Reference: General/Base/replay_buffer.py:20-85 (ring + uniform gather). PER is not in the
// reference; SURVEY.md 8(c2) is its specification.
#include "dqn_device.h"
#include "dqn_launch.h"
#include "dqn_per_device.h"


// ------------------------------------------------------------------ ring insert
// ReplayBuffer.add (replay_buffer.py:58-65) for n rows at consecutive slots. Every block
// reads the same ring_counter; the last block to finish commits counter and size.
__global__ void __launch_bounds__(256)
k_replay_add(DqnState *st, float *states, int32_t *actions, float *rewards, float *observations,
             uint8_t *dones, long long N, int D, const float *s,
             const int32_t *__restrict__ a, const float *__restrict__ r,
             const float *s2, const uint8_t *__restrict__ d, int n, float *s_advance,
             int bump_env) {
    const unsigned long long c0 = st->ring_counter;
    const int total = n * D;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int j = t / D, e = t - j * D;
        const long long k = (long long)((c0 + (unsigned long long)j) % (unsigned long long)N);
        const float sv = s[t], ov = s2[t];
        states[k * D + e] = sv;                                         // :59
        observations[k * D + e] = ov;                                   // :62
        if (s_advance) s_advance[t] = ov;                               // q_agent.py:183 state = observation
        if (e == 0) {
            actions[k] = a[j];                                          // :60
            rewards[k] = r[j];                                          // :61
            dones[k] = d[j] ? 1 : 0;                                    // :63
        }
    }
    // every thread's stores above depend on c0, so past this barrier the block has read it
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int ticket = atomicAdd(&st->arrive, 1u);
        if (ticket == gridDim.x - 1) {
            const unsigned long long c1 = c0 + (unsigned long long)n;
            st->ring_counter = c1;                                                     // :64
            st->size = (long long)(c1 < (unsigned long long)N ? c1 : (unsigned long long)N);  // :65
            if (bump_env) st->env_ctr += 1ull;
            st->arrive = 0;
        }
    }
}

// ------------------------------------------------------------- uniform sampling
// sample_batch (replay_buffer.py:68-85): indices (given, or Philox stream 1) + 5 gathers.
// One thread per sampled row; a row's D floats are contiguous.
__device__ __forceinline__ void gather_row(const float *__restrict__ states, const int32_t *__restrict__ actions,
                                           const float *__restrict__ rewards, const float *__restrict__ observations,
                                           const uint8_t *__restrict__ dones, int D, long long i, int k,
                                           float *s, int32_t *a, float *r, float *s2, uint8_t *d) {
    const float *ps = states + i * D, *po = observations + i * D;
    float *qs = s + (long long)k * D, *qo = s2 + (long long)k * D;
    if ((D & 3) == 0) {
        for (int e = 0; e < D; e += 4) {
            *reinterpret_cast<float4 *>(qs + e) = *reinterpret_cast<const float4 *>(ps + e);
            *reinterpret_cast<float4 *>(qo + e) = *reinterpret_cast<const float4 *>(po + e);
        }
    } else {
        for (int e = 0; e < D; ++e) { qs[e] = ps[e]; qo[e] = po[e]; }
    }
    a[k] = actions[i];
    r[k] = rewards[i];
    d[k] = dones[i];
}

__global__ void __launch_bounds__(256)
k_sample_uniform(const DqnState *st, const float *states, const int32_t *actions, const float *rewards,
                 const float *observations, const uint8_t *dones, int D, int B,
                 unsigned long long seed, unsigned long long ctr_arg, int from_state, const int32_t *idx_in,
                 float *s, int32_t *a, float *r, float *s2, uint8_t *d, int32_t *idx_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B) return;
    const unsigned long long ctr = from_state ? st->sample_ctr : ctr_arg;
    long long i;
    if (idx_in) {
        i = idx_in[k];
    } else {
        const u32x4 o = philox_draw(seed, ctr, (uint32_t)k, DQN_STREAM_UNIFORM);
        i = (long long)(((unsigned long long)o.x * (unsigned long long)st->size) >> 32);  // :77
    }
    if (idx_out) idx_out[k] = (int32_t)i;
    gather_row(states, actions, rewards, observations, dones, D, i, k, s, a, r, s2, d);   // :78-84
}

// ------------------------------------------------------------------ PER sampling
// Stratified proportional sampling: u_k = (k + U_k) * total / B, binary descent
//   k=1; while k<N: l=tree[2k]; if u<l: k=2k else: u-=l; k=2k+1
// clamp to < size, raw IS weight (size*p/total)^-beta via pow_det, then the five gathers of sample_batch
// (General/Base/replay_buffer.py:78-84). Same compares / subtractions in the same order as the CPU restatement, so
// indices, weights and rows are bit-identical; only WHERE the tree nodes are read from differs:
//
//   * u_k is non-decreasing in k and the descent keeps order, so the 64 samples of a wave visit, at every level, a
//     CONTIGUOUS run of nodes [node(lane 0), node(lane 63)]. A wave therefore never chases pointers through HBM:
//   * levels 0 .. TL-1 come from an LDS image of the tree top (depths 0 .. TL, 2^(TL+1) floats, staged once per
//     workgroup by coalesced 16-B loads and shared by its waves over all their chunks);
//   * below, the wave fetches its band -- the children of its run for the next t levels, one coalesced 8-B-per-lane
//     load per level -- into a private LDS slice and walks t levels there (t = as many levels as fit PS_BAND floats:
//     with B ~ N the whole rest of the tree in ONE round trip);
//   * when the run gets wider than PS_WIDE nodes (B << N: every sample is alone in its subtree and a band would be
//     mostly unused bytes) the lanes finish with one dependent 4-B load per level, as the scalar descent does;
//   * rows are gathered lane-cooperatively: a row of D floats is D/4 16-byte pieces, consecutive lanes take consecutive
//     pieces, so both the ring reads (sorted leaves: neighbouring rows) and the batch writes are full-width.
// One wave = one 64-sample chunk at a time; a workgroup walks a contiguous range of chunks (persistent grid).
// ctr_from_state: take the Philox counter / beta from the device state (graph replay) and publish the batch max there.
#ifndef PS_BAND
#define PS_BAND 512          // floats of LDS per wave for a band
#endif
#define PS_WIDE 256          // run width beyond which a band costs more bytes than per-lane loads
#ifndef PS_TOPL
#define PS_TOPL 12           // deepest level walked in the shared LDS image (32 KiB)
#endif
#ifndef PS_WGS_PER_CU
#define PS_WGS_PER_CU 2      // 16-wave workgroups per CU (LDS: 4 * 2^(PS_TOPL+1) + 64 * PS_BAND bytes each)
// r02 choice of the three: B = 2^20 runs in one of two placement modes. One workgroup per CU with a 64 KiB image
// and 4 KiB bands (13 / 1024 / 1) is 34 us in the cache-resident mode and 54 us in the other; two workgroups per CU with half
// the LDS each (12 / 512 / 2: 32 waves per CU in flight) is 39 us and 46 us -- above 60 % of the HBM peak in algorithmic bytes
// in BOTH modes, which the first is not.
#endif

struct PerSampleArgs {
    const DqnState *st; const float *tree; long long N; int L;
    const float *states; const int32_t *actions; const float *rewards; const float *observations; const uint8_t *dones;
    int D, B; float beta; unsigned long long seed, ctr; int from_state;
    float *s; int32_t *a; float *r; float *s2; uint8_t *d; int32_t *idx; float *w_raw;
    unsigned int *wmax_bits;     // batch max of the raw weights (positive floats order like their bit patterns)
    int TL;                      // levels walked in the LDS image (0: none)
    int chunks_per_wg;
};

// one chunk's descent, cut where it waits for memory: start() walks the LDS image and REQUESTS the first band (eight 8-byte
// pieces per lane, in registers); finish() stores them to the wave's LDS slice, walks on (further bands / per-lane tail)
// and ends with the leaf and its priority. Between the two the wave gathers the PREVIOUS chunk's rows: a wave always has
// one chunk's band and one chunk's rows in flight.
struct PsChunk {
    int k0, kk; bool in_range;
    float u; unsigned node; int lvl;
    unsigned nf; int w, t;                   // first band: run [nf, nf + w), t levels (0: none requested)
    float2 pc[PS_BAND / 128];
};

// NT: the batch is written with non-temporal stores (the caller consumes it in a later launch at the earliest: keeping 85 MB of
// output out of the L2 / Infinity Cache leaves them to the ring -- 34.5 instead of 37.3 us at B = 2^20); the large-batch update,
// whose forward reads the rows right away, takes the plain form
template <int WAVES, bool NT>
__global__ void __launch_bounds__(WAVES * 64)
k_per_sample2(const PerSampleArgs p) {
    extern __shared__ __attribute__((aligned(16))) float ps_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TL = p.TL, L = p.L, B = p.B, D = p.D;
    float *top = ps_lds;                                             // nodes [0, 2^(TL+1)) (index 0 unused)
    float *band = ps_lds + (TL > 0 ? (2 << TL) : 0) + wave * PS_BAND;
    int *lidx = reinterpret_cast<int *>(band);                       // the wave's 64 leaves (after the descent)
    const float *tree = p.tree;
    const float2 *tree2 = reinterpret_cast<const float2 *>(tree);
    float2 *band2 = reinterpret_cast<float2 *>(band);
    STAMP(3, 0);
    if (TL > 0) {                                                    // stage the top image
        const float4 *src = reinterpret_cast<const float4 *>(tree);
        float4 *dst = reinterpret_cast<float4 *>(top);
        for (int q = tid; q < (2 << TL) / 4; q += WAVES * 64) dst[q] = src[q];
    }
    const unsigned long long ctr = p.from_state ? p.st->sample_ctr : p.ctr;
    const float beta = p.from_state ? p.st->beta : p.beta;
    const long long size = p.st->size;
    const float total = tree[1];
    const float seg = __fdiv_rn(total, (float)B);
    if (TL > 0) __syncthreads();
    STAMP(3, 1);
    const int nchunks = (B + 63) >> 6;
    const int c_begin = blockIdx.x * p.chunks_per_wg;
    int c_end = c_begin + p.chunks_per_wg; if (c_end > nchunks) c_end = nchunks;

    // levels this round for a run of w nodes: the most with w * (2^(t+1) - 2) <= PS_BAND (0: run too wide for a band)
    auto band_levels = [&](int w, int left) -> int {
        if (w > PS_WIDE) return 0;
        int t = 0;
        while (t < left && (long long)w * ((4ll << t) - 2) <= PS_BAND) ++t;
        return t;
    };
    // request the band of t levels below the run [nf, nf + w): level lvl+j = nodes [nf << j, (nf + w) << j), stored from
    // float offset w * (2^j - 2). As 8-byte pieces the t levels are one flat run of w * (2^t - 1) <= PS_BAND / 2 pieces:
    // piece q lies in level j = 1 + floor(log2(q / w + 1)) and is tree2[(nf << (j-1)) + q - w * (2^(j-1) - 1)]. Eight pieces
    // per lane, a clamped index instead of a branch around the load.
    auto band_request = [&](unsigned nf, int w, int t, float2 (&pc)[PS_BAND / 128]) {
        const int total2 = w * ((1 << t) - 1);
        const float rw = __frcp_rn((float)w);                        // (q + 0.5) / w is never within 1/(2w) of an integer
#pragma unroll
        for (int i = 0; i < PS_BAND / 128; ++i) {
            const int q0 = lane + 64 * i, q = q0 < total2 ? q0 : total2 - 1;
            const int x = (int)(((float)q + 0.5f) * rw);
            const int jm1 = 31 - __clz(x + 1);                        // j - 1
            pc[i] = tree2[((unsigned long long)nf << jm1) + (unsigned)(q - w * ((1 << jm1) - 1))];
        }
    };
    auto band_store = [&](int w, int t, const float2 (&pc)[PS_BAND / 128]) {
        const int total2 = w * ((1 << t) - 1);
#pragma unroll
        for (int i = 0; i < PS_BAND / 128; ++i) {
            const int q0 = lane + 64 * i;
            if (q0 < total2) band2[q0] = pc[i];
        }
    };
    // walk t levels in the stored band (one wave: LDS program order makes the stores visible to the reads)
    auto band_walk = [&](unsigned nf, int w, int t, bool to_leaf, float &u, unsigned &node, float &pleaf) {
        for (int j = 1; j <= t; ++j) {
            const int at = w * ((1 << j) - 2) + (int)(2u * node - (nf << j));   // slot of the left child
            const float l = band[at];
            if (u < l) { node = 2 * node; if (to_leaf && j == t) pleaf = l; }
            else { u = u - l; node = 2 * node + 1; if (to_leaf && j == t) pleaf = band[at + 1]; }
        }
    };
    auto start = [&](int c, PsChunk &d) {
        d.k0 = c << 6;
        const int k = d.k0 + lane;
        d.in_range = k < B;
        d.kk = d.in_range ? k : B - 1;                               // surplus lanes redo the last sample (no stores)
        const u32x4 o = philox_draw(p.seed, ctr, (uint32_t)d.kk, DQN_STREAM_PER);
        float u = ((float)d.kk + u01(o.x)) * seg;
        unsigned node = 1;                                           // node ids < 2^31 (capacity <= 2^30)
        for (int lvl = 0; lvl < TL; ++lvl) {
            const float l = top[2 * node];
            if (u < l) { node = 2 * node; }
            else { u = u - l; node = 2 * node + 1; }
        }
        d.u = u; d.node = node; d.lvl = TL; d.t = 0; d.w = 1; d.nf = node;
        if (TL < L) {
            d.nf = __builtin_amdgcn_readfirstlane(node);
            d.w = (int)(__builtin_amdgcn_readlane(node, 63) - d.nf) + 1;
            d.t = band_levels(d.w, L - TL);
            if (d.t > 0) band_request(d.nf, d.w, d.t, d.pc);
        }
    };
    // -> leaf (clamped) and its priority
    auto finish = [&](PsChunk &d, long long &leaf, float &pleaf) {
        float u = d.u; unsigned node = d.node; int lvl = d.lvl;
        bool have_p = false; pleaf = 0.0f;
        if (TL > 0 && lvl == L) { pleaf = top[node]; have_p = true; }
        if (d.t > 0) {
            band_store(d.w, d.t, d.pc);
            band_walk(d.nf, d.w, d.t, lvl + d.t == L, u, node, pleaf);
            lvl += d.t; have_p = lvl == L;
            while (lvl < L) {                                        // run wider than one band holds: further rounds
                const unsigned nf = __builtin_amdgcn_readfirstlane(node);
                const int w = (int)(__builtin_amdgcn_readlane(node, 63) - nf) + 1;
                const int t = band_levels(w, L - lvl);
                if (t == 0) break;
                float2 pc[PS_BAND / 128];
                band_request(nf, w, t, pc);
                band_store(w, t, pc);
                band_walk(nf, w, t, lvl + t == L, u, node, pleaf);
                lvl += t; have_p = lvl == L;
            }
        }
        for (; lvl < L; ++lvl) {                                     // sparse regime: per-lane dependent loads
            const float l = tree[2ull * node];
            if (u < l) { node = 2 * node; }
            else { u = u - l; node = 2 * node + 1; }
        }
        leaf = (long long)node - p.N;
        if (leaf >= size) { leaf = size - 1; have_p = false; }
        if (!have_p) pleaf = tree[p.N + leaf];
    };

    float wmx = 0.0f;
    PsChunk d;
    int c = c_begin + wave;
    if (c < c_end) start(c, d);
    while (c < c_end) {
        long long leaf; float pleaf;
        finish(d, leaf, pleaf);
        const int k0 = d.k0, k = k0 + lane; const bool in_range = d.in_range;
        const float w_is = pow_det(__fdiv_rn((float)size * pleaf, total), -beta);
        wmx = fmaxf(wmx, w_is);
        lidx[lane] = (int)leaf;
        // ---- the five gathers (replay_buffer.py:78-84): every load first; the next chunk's image walk and band request;
        // then the stores
        const int nrow = (B - k0) < 64 ? (B - k0) : 64;
        const int32_t av = p.actions[leaf]; const float rv = p.rewards[leaf]; const uint8_t dv = p.dones[leaf];
        const int cn = c + WAVES;
        if (D == 8) {
            // a row = two 16-byte pieces; lane q handles pieces q and q + 64 of the chunk's 128 (all 64 leaves are valid rows)
            const float4 *S4 = reinterpret_cast<const float4 *>(p.states), *O4 = reinterpret_cast<const float4 *>(p.observations);
            float4 *s4 = reinterpret_cast<float4 *>(p.s) + (long long)k0 * 2, *o4 = reinterpret_cast<float4 *>(p.s2) + (long long)k0 * 2;
            const long long src0 = (long long)lidx[lane >> 1] * 2 + (lane & 1), src1 = (long long)lidx[32 + (lane >> 1)] * 2 + (lane & 1);
            const float4 a0 = S4[src0], b0 = O4[src0], a1 = S4[src1], b1 = O4[src1];
            if (cn < c_end) start(cn, d);
            if constexpr (NT) {
                typedef float nt4 __attribute__((ext_vector_type(4)));
                auto nts = [](const float4 &v, float4 *q) { const nt4 x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, reinterpret_cast<nt4 *>(q)); };
                if (lane < 2 * nrow) { nts(a0, s4 + lane); nts(b0, o4 + lane); }
                if (lane + 64 < 2 * nrow) { nts(a1, s4 + lane + 64); nts(b1, o4 + lane + 64); }
            } else {
                if (lane < 2 * nrow) { s4[lane] = a0; o4[lane] = b0; }
                if (lane + 64 < 2 * nrow) { s4[lane + 64] = a1; o4[lane + 64] = b1; }
            }
        } else {
            if ((D & 3) == 0) {
                const int C = D >> 2, tot = nrow * C;                // 16-byte pieces of this chunk's rows
                const float4 *S4 = reinterpret_cast<const float4 *>(p.states), *O4 = reinterpret_cast<const float4 *>(p.observations);
                float4 *s4 = reinterpret_cast<float4 *>(p.s) + (long long)k0 * C, *o4 = reinterpret_cast<float4 *>(p.s2) + (long long)k0 * C;
                for (int q = lane; q < tot; q += 64) {
                    const int smp = q / C, part = q - smp * C;
                    const long long src = (long long)lidx[smp] * C + part;
                    const float4 a4 = S4[src], b4 = O4[src];
                    s4[q] = a4; o4[q] = b4;
                }
            } else {
                const int tot = nrow * D;
                float *s1 = p.s + (long long)k0 * D, *o1 = p.s2 + (long long)k0 * D;
                for (int q = lane; q < tot; q += 64) {
                    const int smp = q / D, e = q - smp * D;
                    const long long src = (long long)lidx[smp] * D + e;
                    const float x0 = p.states[src], x1 = p.observations[src];
                    s1[q] = x0; o1[q] = x1;
                }
            }
            if (cn < c_end) start(cn, d);
        }
        if (in_range) {
            if constexpr (NT) {
                __builtin_nontemporal_store((int32_t)leaf, p.idx + k); __builtin_nontemporal_store(w_is, p.w_raw + k); __builtin_nontemporal_store(av, p.a + k);
                __builtin_nontemporal_store(rv, p.r + k); __builtin_nontemporal_store(dv, p.d + k);
            } else { p.idx[k] = (int32_t)leaf; p.w_raw[k] = w_is; p.a[k] = av; p.r[k] = rv; p.d[k] = dv; }
        }
        c = cn;
    }
    STAMP(3, 2);
    // batch max of the raw weights: one atomic per workgroup
    for (int o = 32; o > 0; o >>= 1) wmx = fmaxf(wmx, __shfl_xor(wmx, o, 64));
    if (WAVES > 1) {
        __syncthreads();
        if (lane == 0) ps_lds[wave] = wmx;
        __syncthreads();
        if (tid == 0) for (int q = 1; q < WAVES; ++q) wmx = fmaxf(wmx, ps_lds[q]);
    }
    if (tid == 0 && wmx > 0.0f) atomicMax(p.wmax_bits, __float_as_uint(wmx));
}

// isw[k] = w_raw[k] / max_j w_raw[j] (the max was gathered by the sampler's workgroups); st->wmax = that max
__global__ void __launch_bounds__(256)
k_isw_normalize(const float *__restrict__ w_raw, int B, float *isw, DqnState *st, const unsigned int *wmax_bits) {
    const float wmax = __uint_as_float(*wmax_bits);
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < B) isw[k] = __fdiv_rn(w_raw[k], wmax);
    if (k == 0) st->wmax = wmax;
}

// ------------------------------------------------------------- priority write-back
// One workgroup (level-synchronous; all cross-level dependencies are inside it).
//   mode 0: prio given            (dqn_per_set)
//   mode 1: prio = (|td|+eps)^alpha (dqn_per_update)
//   mode 2: leaves = last n ring slots, prio = pmax (ring insert with PER)
// Duplicates: highest batch position wins, resolved with a 64-bit (epoch, position) stamp
// per leaf. Parents are recomputed bottom-up as left + right; threads sharing a parent
// store the same value.
__global__ void __launch_bounds__(1024)
k_per_write(DqnState *st, float *tree, unsigned long long *stamp, long long N, int L,
            const int32_t *__restrict__ idx, const float *__restrict__ val, int B, int mode,
            float alpha, float eps, long long ring_capacity) {
    __shared__ float red[1024];
    const int tid = threadIdx.x, nt = blockDim.x;
    const unsigned long long epoch = st->epoch + 1ull;
    const float pmax_old = st->pmax;
    unsigned long long c_base = 0;
    if (mode == 2) c_base = st->ring_counter - (unsigned long long)B;   // slots just written
    float lmax = 0.0f;

    if (mode != 2) {
        // warm the path: the siblings along every leaf-to-root path are independent loads, so
        // fetch them all at once instead of paying one cold miss per level below
        float warm = 0.0f;
        for (int i = tid; i < B; i += nt) {
            const long long leafnode = N + idx[i];
            for (int lvl = 0; lvl < L; ++lvl) warm += tree[(leafnode >> lvl) ^ 1];
        }
        if (warm == -1.2345e30f) st->pad = 1u;            // never true; keeps the loads alive
        for (int i = tid; i < B; i += nt)
            atomicMax(&stamp[idx[i]], (epoch << 32) | (unsigned long long)(unsigned)i);
        __threadfence_block();
        __syncthreads();
    }
    for (int i = tid; i < B; i += nt) {
        long long leaf;
        float p;
        bool mine = true;
        if (mode == 2) {
            leaf = (long long)((c_base + (unsigned long long)i) % (unsigned long long)ring_capacity);
            p = pmax_old;
        } else {
            leaf = idx[i];
            p = (mode == 0) ? val[i] : pow_det(val[i] + eps, alpha);
            mine = stamp[leaf] == ((epoch << 32) | (unsigned long long)(unsigned)i);
            lmax = fmaxf(lmax, p);
        }
        if (mine) tree[N + leaf] = p;
    }
    __threadfence_block();
    __syncthreads();
    for (int lvl = 1; lvl <= L; ++lvl) {
        for (int i = tid; i < B; i += nt) {
            const long long leaf = (mode == 2)
                ? (long long)((c_base + (unsigned long long)i) % (unsigned long long)ring_capacity)
                : (long long)idx[i];
            const long long node = (N + leaf) >> lvl;
            tree[node] = tree[2 * node] + tree[2 * node + 1];
        }
        __threadfence_block();
        __syncthreads();
    }
    if (mode != 2) {
        red[tid] = lmax;
        __syncthreads();
        for (int sft = nt >> 1; sft > 0; sft >>= 1) {
            if (tid < sft) red[tid] = fmaxf(red[tid], red[tid + sft]);
            __syncthreads();
        }
        if (tid == 0) { st->pmax = fmaxf(pmax_old, red[0]); st->epoch = epoch; }
    }
}

// ------------------------------------------------ priority write-back, LDS-resident (B <= 4096)
// Same contract as k_per_write modes 0/1 (duplicates: highest batch position wins; every touched
// parent = left + right), but no value ever travels between threads through global memory:
//   * winners per leaf are resolved in an LDS hash (key = leaf, atomicMax of the batch position);
//   * the old values of the siblings along every path are fetched up front, all levels at once;
//   * bottom levels (depth L .. TOP+1): per level, each live thread publishes (node -> new value)
//     in an LDS hash, looks its sibling up (found => both touched, the left child carries on; not
//     found => use the prefetched old value) and stores the parent to HBM without waiting;
//   * top TOP = min(L,10) levels: a dense LDS image of nodes [1, 2^(TOP+1)) is patched with the new
//     depth-TOP values and re-reduced level by level (recomputing an untouched parent from unchanged
//     children reproduces its value bit for bit).
// Only LDS barriers inside the loops. One workgroup, IPT items per thread held in registers.
__device__ __forceinline__ unsigned pw_hash(unsigned key, unsigned mask) { return (key * 2654435761u) >> 7 & mask; }

#define PWL_TOP 10           // dense-top depth of this single-workgroup variant (2^11 floats of LDS)
template <int IPT>
__global__ void __launch_bounds__(1024)
k_per_write_lds(DqnState *st, float *tree, long long N, int L, const int32_t *__restrict__ idx,
                const float *__restrict__ val, int B, int mode, float alpha, float eps, int TS) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float red[1024];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int TOP = L < PWL_TOP ? L : PWL_TOP;
    const unsigned mask = (unsigned)TS - 1u;
    int *keys0 = reinterpret_cast<int *>(smem);             // two hash tables, alternating per level
    float *vals0 = reinterpret_cast<float *>(keys0 + TS);
    int *keys1 = reinterpret_cast<int *>(vals0 + TS);
    float *vals1 = reinterpret_cast<float *>(keys1 + TS);
    float *top = vals1 + TS;                                 // dense image of nodes [0, 2^(TOP+1))
    const int topn = 2 << TOP;

    STAMP(2, 0);
    const float pmax_old = st->pmax;
    long long node[IPT]; float nv[IPT]; bool live[IPT]; float sib[IPT][PW_BOT];
    float lmax = 0.0f;

    // ---- prologue: independent loads first (dense top image, sibling old values), tables cleared
    for (int j = tid; j < topn; j += nt) top[j] = j ? tree[j] : 0.0f;
    for (int j = tid; j < TS; j += nt) { keys0[j] = 0; keys1[j] = 0; reinterpret_cast<int *>(vals0)[j] = 0; }
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        const int i = tid + q * nt;
        live[q] = i < B;
        node[q] = live[q] ? N + (long long)idx[i] : 1;
        nv[q] = 0.0f;
        if (live[q]) {
            const float p = (mode == 0) ? val[i] : pow_det(val[i] + eps, alpha);
            nv[q] = p;
            lmax = fmaxf(lmax, p);
        }
#pragma unroll
        for (int l = 0; l < PW_BOT; ++l)
            sib[q][l] = (live[q] && l < L - TOP) ? tree[(node[q] >> l) ^ 1] : 0.0f;
    }
    __syncthreads();
    STAMP(2, 1);

    // ---- winners: highest batch position per leaf (positions kept in vals0 as ints)
    int slot[IPT];
    int *pos0 = reinterpret_cast<int *>(vals0);
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        if (live[q]) {
            const int key = (int)node[q];                    // node ids < 2^31
            unsigned s = pw_hash((unsigned)key, mask);
            for (;;) {
                const int prev = atomicCAS(&keys0[s], 0, key);
                if (prev == 0 || prev == key) break;
                s = (s + 1) & mask;
            }
            slot[q] = (int)s;
            atomicMax(&pos0[s], tid + q * nt + 1);           // positions are stored +1; 0 = none yet
        }
    }
    LDS_BARRIER();
#pragma unroll
    for (int q = 0; q < IPT; ++q)
        if (live[q]) {
            live[q] = pos0[slot[q]] == tid + q * nt + 1;
            if (live[q]) tree[node[q]] = nv[q];              // the leaf itself
        }
    LDS_BARRIER();
#pragma unroll
    for (int q = 0; q < IPT; ++q)                            // table 0 now carries values: winners publish
        if (live[q]) vals0[slot[q]] = nv[q];
    // (losers' slots are their leaf's slot too, already holding the winner's key)
    LDS_BARRIER();

    STAMP(2, 2);
    // ---- bottom levels: depth L down to TOP+1 (fully unrolled so that sib[][l] stays in registers)
#pragma unroll
    for (int l = 0; l < PW_BOT; ++l) {
        if (l >= L - TOP) break;
        int *kc = (l & 1) ? keys1 : keys0; float *vc = (l & 1) ? vals1 : vals0;
        int *kn = (l & 1) ? keys0 : keys1;
        if (l > 0) {                                         // publish this level's nodes (level 0 done above)
#pragma unroll
            for (int q = 0; q < IPT; ++q)
                if (live[q]) {
                    const int key = (int)node[q];
                    unsigned s = pw_hash((unsigned)key, mask);
                    for (;;) {
                        const int prev = atomicCAS(&kc[s], 0, key);
                        if (prev == 0) break;
                        s = (s + 1) & mask;
                    }
                    vc[s] = nv[q];
                }
            LDS_BARRIER();
        }
#pragma unroll
        for (int q = 0; q < IPT; ++q)
            if (live[q]) {
                const int skey = (int)(node[q] ^ 1);
                unsigned s = pw_hash((unsigned)skey, mask);
                bool found = false; float sv = 0.0f;
                for (;;) {
                    const int k = kc[s];
                    if (k == skey) { found = true; sv = vc[s]; break; }
                    if (k == 0) break;
                    s = (s + 1) & mask;
                }
                const bool right = node[q] & 1;
                if (found && right) { live[q] = false; }     // the left sibling carries the pair upward
                else {
                    const float other = found ? sv : sib[q][l];
                    const float pv = right ? other + nv[q] : nv[q] + other;
                    node[q] >>= 1; nv[q] = pv;
                    tree[node[q]] = pv;
                }
            }
        for (int j = tid; j < TS; j += nt) kn[j] = 0;        // next level's table (its readers are done)
        LDS_BARRIER();
    }
    // the hash for level 0 used positions in vals0; if L == TOP the loop above did not run

    STAMP(2, 3);
    // ---- top levels: patch the dense image at depth TOP, then reduce it
#pragma unroll
    for (int q = 0; q < IPT; ++q)
        if (live[q]) top[node[q]] = nv[q];
    LDS_BARRIER();
    for (int d = TOP - 1; d >= 0; --d) {
        const int cnt = 1 << d;
        for (int j = tid; j < cnt; j += nt) {
            const int p = cnt + j;
            const float v = top[2 * p] + top[2 * p + 1];
            top[p] = v;
            tree[p] = v;
        }
        LDS_BARRIER();
    }

    STAMP(2, 4);
    // ---- running max priority
    red[tid] = lmax;
    LDS_BARRIER();
    for (int sft = nt >> 1; sft > 0; sft >>= 1) {
        if (tid < sft) red[tid] = fmaxf(red[tid], red[tid + sft]);
        LDS_BARRIER();
    }
    if (tid == 0) { st->pmax = fmaxf(pmax_old, red[0]); st->epoch += 1ull; }
    STAMP(2, 5);
}

// --------------------------------------- priority write-back for SORTED indices (many CUs)
// dqn_per_sample's indices are non-decreasing (stratified u_k increase and the descent is monotone),
// so duplicates and sibling pairs sit in ADJACENT positions of the batch. The tree is cut at depth
// TOP = min(L,10): below it the depth-TOP subtrees are independent, above it lies one dense
// 2^TOP-node image.
//   k_per_write_sorted  one wave per 64-position chunk of the batch; a wave owns every depth-TOP subtree
//                       whose first item lies in its chunk, and all of that subtree's items (it reads on
//                       past its chunk end). With <= 64 owned items everything stays in registers: the
//                       last of a run of equal leaves wins, sibling detection is a shuffle from the nearest
//                       live lane, the untouched siblings' old values are prefetched for all levels at
//                       once, parents go to HBM unawaited. Longer runs take a wave-serial slow path.
//   k_per_top           one workgroup reloads depth TOP (contiguous) and re-reduces the top image in LDS.
// Scattered traffic is thereby spread over B/64 CUs instead of one. Same arithmetic as k_per_write.
__global__ void __launch_bounds__(64)
k_per_write_sorted(DqnState *st, float *tree, long long N, int L, const int32_t *__restrict__ idx,
                   const float *__restrict__ val, int B, int mode, float alpha, float eps) {
    per_write_sorted_wave(st, tree, N, L, idx, val, B, mode, alpha, eps, blockIdx.x);
}

__global__ void __launch_bounds__(1024)
k_per_top(DqnState *st, float *tree, int L) {
    // depth TOP holds n = 2^TOP nodes (already final in HBM); depth TOP-1 is summed straight from them
    // (coalesced pair loads), the levels above are reduced out of an LDS image of 2^TOP floats.
    extern __shared__ __attribute__((aligned(16))) float top[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int TOP = L < PW_TOP ? L : PW_TOP;
    if (TOP == 0) { if (tid == 0) st->epoch += 1ull; return; }
    const int h = 1 << (TOP - 1);                            // nodes at depth TOP-1: indices [h, 2h)
    for (int j = tid; j < h; j += nt) {
        const float2 c = *reinterpret_cast<const float2 *>(tree + 2 * (h + j));
        const float v = c.x + c.y;
        top[h + j] = v;
        tree[h + j] = v;
    }
    __syncthreads();
    for (int d = TOP - 2; d >= 0; --d) {
        const int cnt = 1 << d;
        for (int j = tid; j < cnt; j += nt) {
            const int p = cnt + j;
            const float v = top[2 * p] + top[2 * p + 1];
            top[p] = v;
            tree[p] = v;
        }
        LDS_BARRIER();
    }
    if (tid == 0) st->epoch += 1ull;
}


// ---------------------------------------------------------------------------------------------------------------------
// r03: the sorted write-back rebuilt for BANDWIDTH (large batches; the wave-per-64-positions kernel above stays for small
// ones and inside k_dw). Instead of walking up from every touched leaf (per item: L scattered sibling reads + L scattered
// parent writes), the LEAF SPACE is cut into segments of 2^PWS_LOG leaves, one workgroup each:
//   * the workgroup finds its slice [lo, hi) of the sorted batch by two interleaved 64-ary searches (4 dependent probes at
//     B = 2^20 instead of 20), leaves at once if the slice is empty (nothing of that subtree is read or written);
//   * otherwise it loads the segment's leaves (coalesced 16-B loads) into an LDS heap, applies its batch items there (the last
//     of a run of equal leaves wins -- the highest batch position, as everywhere), and recomputes EVERY inner node of the
//     subtree as left + right: three levels in registers (a thread owns 8 consecutive leaves), six by wave shuffles, the
//     rest through LDS. Untouched nodes are re-derived from untouched children: the tree's invariant (a parent is always
//     fl(left + right), never delta-added) makes that the value they already have -- the result is bit-identical to the
//     touched-path update of the restatement;
//   * leaves and inner nodes go back as dense, coalesced stores (one contiguous run per level).
// k_per_top_seg then rebuilds the levels above the segment roots from the N / 2^PWS_LOG roots (one workgroup, LDS) and folds
// the per-segment priority maxima into the running max. No atomics anywhere: hundreds of workgroups bumping one word
// serialise at ~0.1 us each.
// Traffic at L = 20 with every segment touched: 4 MB leaves in, 4 + 4 MB out, 8 B per item -- independent of the tree depth.
#define PWS_LOG 11
#define PWS_MIN_B 1024              // below: the wave-per-64-positions kernel (touched paths only) + k_per_top (measured: 10.3 vs 11.0 us per call at B = 1 024, 10.6 vs 16.8 at 8 192, 14.3 vs 517 at 2^20)
#define PWS_MAX_ROOT_LOG 12          // at most 2^12 segment roots in the top kernel's LDS heap (32 KB; L <= 23; deeper trees: the chunk kernels)

__global__ void __launch_bounds__(256)
k_per_write_seg(float *__restrict__ tree, int L, const int32_t *__restrict__ idx, const float *__restrict__ val, int B,
                int mode, float alpha, float eps, float *__restrict__ pmax_part) {
    __shared__ __attribute__((aligned(16))) float heap[2 << PWS_LOG];      // heap[1] = segment root, leaf i at heap[S + i]
    __shared__ int bounds[2];
    __shared__ float wmaxs[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int SL = L < PWS_LOG ? L : PWS_LOG, S = 1 << SL;
    const int seg = blockIdx.x;
    const long long N = 1ll << L;
    if (wave == 0) {
        // first batch position whose leaf is >= key, for key = seg * S (lo) and (seg + 1) * S (hi): 64-ary searches over the
        // non-decreasing idx[], both in flight together
        const int key0 = seg << SL, key1 = (seg + 1) << SL;
        int a0 = 0, b0 = B, a1 = 0, b1 = B;
        while (b0 > a0 || b1 > a1) {
            const int st0 = (b0 - a0 + 63) >> 6, st1 = (b1 - a1 + 63) >> 6;
            const int p0 = a0 + lane * st0, p1 = a1 + lane * st1;
            const bool in0 = b0 > a0 && p0 < b0, in1 = b1 > a1 && p1 < b1;
            const int v0 = in0 ? idx[p0] : 0x7fffffff, v1 = in1 ? idx[p1] : 0x7fffffff;
            const int c0 = __popcll(__ballot(in0 && v0 < key0)), c1 = __popcll(__ballot(in1 && v1 < key1));
            if (b0 > a0) { if (c0 == 0) b0 = a0; else { const int na = a0 + (c0 - 1) * st0 + 1, nb = a0 + c0 * st0; a0 = na; b0 = nb < b0 ? nb : b0; } }
            if (b1 > a1) { if (c1 == 0) b1 = a1; else { const int na = a1 + (c1 - 1) * st1 + 1, nb = a1 + c1 * st1; a1 = na; b1 = nb < b1 ? nb : b1; } }
        }
        if (lane == 0) { bounds[0] = a0; bounds[1] = a1; }
    }
    __syncthreads();
    const int lo = bounds[0], hi = bounds[1];
    if (lo >= hi) { if (tid == 0) pmax_part[seg] = 0.0f; return; }
    float *leaves = tree + N + ((long long)seg << SL);
    const int per = S >= 2048 ? 8 : (S >= 256 ? S / 256 : 1);               // leaves per thread (S < 256: threads >= S idle)
    const bool act = tid * per < S;
    float lf[8];
    if (per == 8) {
        const float4 x0 = reinterpret_cast<const float4 *>(leaves)[2 * tid], x1 = reinterpret_cast<const float4 *>(leaves)[2 * tid + 1];
        lf[0] = x0.x; lf[1] = x0.y; lf[2] = x0.z; lf[3] = x0.w; lf[4] = x1.x; lf[5] = x1.y; lf[6] = x1.z; lf[7] = x1.w;
        reinterpret_cast<float4 *>(heap + S)[2 * tid] = x0; reinterpret_cast<float4 *>(heap + S)[2 * tid + 1] = x1;
    } else if (act) {
        for (int k = 0; k < per; ++k) { lf[k] = leaves[tid * per + k]; heap[S + tid * per + k] = lf[k]; }
    }
    __syncthreads();
    float lmax = 0.0f;
    for (int j = lo + tid; j < hi; j += 256) {
        const int leaf = idx[j];
        const float p = (mode == 0) ? val[j] : pow_det(val[j] + eps, alpha);
        lmax = fmaxf(lmax, p);
        const bool loser = j + 1 < B && idx[j + 1] == leaf;                 // a later batch position holds the same leaf
        if (!loser) heap[S + (leaf - (seg << SL))] = p;
    }
    __syncthreads();
    // ---- all inner nodes, bottom-up, each as left + right
    if (per == 8) {
        const float4 x0 = reinterpret_cast<const float4 *>(heap + S)[2 * tid], x1 = reinterpret_cast<const float4 *>(heap + S)[2 * tid + 1];
        reinterpret_cast<float4 *>(leaves)[2 * tid] = x0; reinterpret_cast<float4 *>(leaves)[2 * tid + 1] = x1;
        const float s0 = x0.x + x0.y, s1 = x0.z + x0.w, s2 = x1.x + x1.y, s3 = x1.z + x1.w;     // depth SL-1: nodes S/2 + 4 tid ..
        const float t0 = s0 + s1, t1 = s2 + s3;                                                  // depth SL-2: S/4 + 2 tid ..
        float v = t0 + t1;                                                                       // depth SL-3: S/8 + tid
        reinterpret_cast<float4 *>(heap + (S >> 1))[tid] = float4{s0, s1, s2, s3};
        reinterpret_cast<float2 *>(heap + (S >> 2))[tid] = float2{t0, t1};
        heap[(S >> 3) + tid] = v;
        // six levels inside the wave: at step s the lanes that are multiples of 2^s hold the node of 2^s threads
#pragma unroll
        for (int s = 1; s <= 6; ++s) {
            const float r = __shfl_down(v, 1 << (s - 1), 64);
            v = v + r;                                                       // left (this lane) + right
            if ((lane & ((1 << s) - 1)) == 0) heap[(S >> (3 + s)) + (tid >> s)] = v;
        }
        // wave w's lane 0 wrote node (S >> 9) + w = 4 + w (S = 2048): two more levels by one thread
        __syncthreads();
        if (tid == 0) { heap[2] = heap[4] + heap[5]; heap[3] = heap[6] + heap[7]; heap[1] = heap[2] + heap[3]; }
        __syncthreads();
    } else {
        if (act) for (int k = 0; k < per; ++k) leaves[tid * per + k] = heap[S + tid * per + k];
        for (int d = SL - 1; d >= 0; --d) {
            const int cnt = 1 << d;
            for (int t = tid; t < cnt; t += 256) heap[cnt + t] = heap[2 * (cnt + t)] + heap[2 * (cnt + t) + 1];
            __syncthreads();
        }
    }
    // ---- inner nodes out: heap index k at local depth d = floor(log2 k) is global node 2^(L-SL+d) + seg * 2^d + (k - 2^d)
    for (int k = tid; k < S; k += 256) {
        if (k == 0) continue;
        const int d = 31 - __clz(k);
        tree[(1ll << (L - SL + d)) + ((long long)seg << d) + (k - (1 << d))] = heap[k];
    }
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
    if (lane == 0) wmaxs[wave] = lmax;
    __syncthreads();
    if (tid == 0) pmax_part[seg] = fmaxf(fmaxf(wmaxs[0], wmaxs[1]), fmaxf(wmaxs[2], wmaxs[3]));
}

// levels above the segment roots (depth L - SL, R = 2^(L-SL) nodes, final in HBM) + running max + epoch
__global__ void __launch_bounds__(1024)
k_per_top_seg(DqnState *st, float *tree, int L, const float *__restrict__ pmax_part) {
    extern __shared__ __attribute__((aligned(16))) float top[];             // heap of 2 R floats
    __shared__ float wm[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int SL = L < PWS_LOG ? L : PWS_LOG, RL = L - SL, R = 1 << RL;
    float m = 0.0f;
    for (int j = tid; j < R; j += nt) { top[R + j] = tree[R + j]; m = fmaxf(m, pmax_part[j]); }
    __syncthreads();
    for (int d = RL - 1; d >= 0; --d) {
        const int cnt = 1 << d;
        for (int j = tid; j < cnt; j += nt) {
            const int p = cnt + j;
            const float v = top[2 * p] + top[2 * p + 1];
            top[p] = v;
            tree[p] = v;
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0) wm[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < (nt >> 6); ++w) m = fmaxf(m, wm[w]);
        if (m > st->pmax) st->pmax = m;                                      // running max priority (max is order-independent)
        st->epoch += 1ull;
    }
}

// ------------------------------------------------- leaf-range insert (ring add with PER)
// device code: per_add_range_wg / per_add_slow in dqn_per_device.h
__global__ void __launch_bounds__(1024)
k_per_add(DqnState *st, float *tree, long long Nt, int L, int n, long long cap, int advance, long long zero_first, int zero_n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // zero_n > 0 (dqn_per_index_step): the zero_n positions from zero_first on leave the draw first (priority 0: their rows are being
    // overwritten) -- a consecutive leaf range like the insert's, so the same range code with the value 0
    if (zero_n > 0) {
        const long long z = zero_first % cap;
        if (zero_n <= RANGE_MAX && z + zero_n <= cap) per_add_range_wg(tree, Nt, L, z, zero_n, 0.0f, lds);
        else per_add_slow(tree, Nt, L, (unsigned long long)z, zero_n, 0.0f, cap);
        __threadfence_block();
        __syncthreads();
        if (n <= 0) return;
    }
    // advance = 0: the n slots have just been written (ring_counter is past them); 1 (dqn_per_index_advance): this launch IS the
    // add -- the slots start at ring_counter, committed by thread 0 once every thread has read it
    const unsigned long long c_base = advance ? st->ring_counter : st->ring_counter - (unsigned long long)n;
    const long long a = (long long)(c_base % (unsigned long long)cap);
    const float pmax = st->pmax;
    if (n <= RANGE_MAX && a + n <= cap) per_add_range_wg(tree, Nt, L, a, n, pmax, lds);
    else per_add_slow(tree, Nt, L, c_base, n, pmax, cap);
    if (advance) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long c1 = c_base + (unsigned long long)n;                  // replay_buffer.py:64-65
            st->ring_counter = c1;
            st->size = (long long)(c1 < (unsigned long long)cap ? c1 : (unsigned long long)cap);
        }
    }
}


// --------------------------------------------------------------------- launchers
void launch_replay_add(hipStream_t st_, DqnState *st, float *states, int32_t *actions, float *rewards,
                       float *observations, uint8_t *dones, long long N, int D, const float *s,
                       const int32_t *a, const float *r, const float *s2, const uint8_t *d, int n,
                       float *s_advance, int bump_env) {
    const int total = n * D;
    int blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_replay_add, dim3(blocks), dim3(256), 0, st_, st, states, actions, rewards,
                       observations, dones, N, D, s, a, r, s2, d, n, s_advance, bump_env);
}


void launch_sample_uniform(hipStream_t st_, const DqnState *st, const float *states, const int32_t *actions,
                           const float *rewards, const float *observations, const uint8_t *dones, int D,
                           int B, unsigned long long seed, unsigned long long ctr, int from_state,
                           const int32_t *idx_in, float *s, int32_t *a, float *r, float *s2, uint8_t *d,
                           int32_t *idx_out) {
    hipLaunchKernelGGL(k_sample_uniform, dim3((B + 255) / 256), dim3(256), 0, st_, st, states, actions,
                       rewards, observations, dones, D, B, seed, ctr, from_state, idx_in, s, a, r, s2, d, idx_out);
}

// grid / LDS choice of k_per_sample2: many chunks -> 16-wave workgroups (one per CU, persistent over their chunk range) that
// share a 2^(TL+1)-float image of the tree top; few chunks -> one-wave workgroups spread over the CUs, no image (the band
// rounds start at the root: a lone wave would spend longer staging 64 KiB than walking)
void launch_per_sample(hipStream_t st_, const DqnState *st, const float *tree, long long N, int L,
                       const float *states, const int32_t *actions, const float *rewards,
                       const float *observations, const uint8_t *dones, int D, int B, float beta,
                       unsigned long long seed, unsigned long long ctr, int from_state,
                       float *s, int32_t *a, float *r, float *s2, uint8_t *d, int32_t *idx, float *w_raw,
                       unsigned int *wmax_bits, int num_cus, bool nt_out) {
    PerSampleArgs p{st, tree, N, L, states, actions, rewards, observations, dones, D, B, beta, seed, ctr, from_state,
                    s, a, r, s2, d, idx, w_raw, wmax_bits, 0, 1};
    const int nchunks = (B + 63) / 64;
    if (num_cus < 1) num_cus = 256;
    if (nchunks >= 4 * num_cus) {
        p.TL = L < PS_TOPL ? L : PS_TOPL;
        int wgs = (nchunks + 15) / 16; if (wgs > PS_WGS_PER_CU * num_cus) wgs = PS_WGS_PER_CU * num_cus;
        p.chunks_per_wg = (nchunks + wgs - 1) / wgs;
        wgs = (nchunks + p.chunks_per_wg - 1) / p.chunks_per_wg;
        const size_t lds = sizeof(float) * ((size_t)(2 << p.TL) + 16 * PS_BAND);
        if (nt_out) DQN_LAUNCH((k_per_sample2<16, true>), dim3(wgs), dim3(1024), lds, st_, p);
        else DQN_LAUNCH((k_per_sample2<16, false>), dim3(wgs), dim3(1024), lds, st_, p);
    } else {
        p.TL = 0; p.chunks_per_wg = 1;
        DQN_LAUNCH((k_per_sample2<1, false>), dim3(nchunks), dim3(64), sizeof(float) * PS_BAND, st_, p);
    }
}

void launch_isw_normalize(hipStream_t st_, const float *w_raw, int B, float *isw, DqnState *st, const unsigned int *wmax_bits) {
    hipLaunchKernelGGL(k_isw_normalize, dim3((B + 255) / 256), dim3(256), 0, st_, w_raw, B, isw, st, wmax_bits);
}

void launch_per_write(hipStream_t st_, DqnState *st, float *tree, unsigned long long *stamp, long long N,
                      int L, const int32_t *idx, const float *val, int B, int mode, float alpha, float eps,
                      long long ring_capacity) {
    int threads = 64;
    while (threads < B && threads < 1024) threads <<= 1;
    if (mode != 2 && B <= 4096 && L <= 31) {
        int TS = 128;
        while (TS < 2 * B) TS <<= 1;                          // load factor <= 0.5
        const int TOP = L < PWL_TOP ? L : PWL_TOP;
        const size_t lds = (size_t)TS * 16 + sizeof(float) * (size_t)(2 << TOP);
        const int ipt = (B + threads - 1) / threads;
        if (ipt <= 1)      hipLaunchKernelGGL((k_per_write_lds<1>), dim3(1), dim3(threads), lds, st_, st, tree, N, L, idx, val, B, mode, alpha, eps, TS);
        else if (ipt <= 2) hipLaunchKernelGGL((k_per_write_lds<2>), dim3(1), dim3(threads), lds, st_, st, tree, N, L, idx, val, B, mode, alpha, eps, TS);
        else               hipLaunchKernelGGL((k_per_write_lds<4>), dim3(1), dim3(threads), lds, st_, st, tree, N, L, idx, val, B, mode, alpha, eps, TS);
        return;
    }
    hipLaunchKernelGGL(k_per_write, dim3(1), dim3(threads), 0, st_, st, tree, stamp, N, L, idx, val, B, mode,
                       alpha, eps, ring_capacity);
}

static inline int pow2_threads(int n, int lo, int hi) { int t = lo; while (t < n && t < hi) t <<= 1; return t; }

void launch_per_add(hipStream_t st_, const DqnState *st, float *tree, long long Nt, int L, int n, long long cap, int advance, long long zero_first, int zero_n) {
    const size_t lds = sizeof(float) * 64;
    hipLaunchKernelGGL(k_per_add, dim3(1), dim3(pow2_threads(n > zero_n ? n : zero_n, 64, 1024)), lds, st_, const_cast<DqnState *>(st), tree, Nt, L, n, cap, advance, zero_first, zero_n);
}


void launch_per_top(hipStream_t st_, DqnState *st, float *tree, int L) {
    const int TOP = L < PW_TOP ? L : PW_TOP;
    const int n = 1 << TOP;                                   // LDS image: indices [0, n) = depths 0 .. TOP-1
    DQN_LAUNCH(k_per_top, dim3(1), dim3(n / 2 < 1024 ? (n / 2 < 64 ? 64 : n / 2) : 1024), sizeof(float) * n, st_, st, tree, L);
}

// pw_part: scratch of 2^PWS_MAX_ROOT_LOG floats (per-segment priority maxima), or NULL
bool per_write_seg_applies(int L, int B) { return B >= PWS_MIN_B && L - (L < PWS_LOG ? L : PWS_LOG) <= PWS_MAX_ROOT_LOG; }

void launch_per_write_sorted(hipStream_t st_, DqnState *st, float *tree, long long N, int L, const int32_t *idx,
                             const float *val, int B, int mode, float alpha, float eps, float *pw_part, int force) {
    if (pw_part && force != 1 && (force == 2 || per_write_seg_applies(L, B)) && L - (L < PWS_LOG ? L : PWS_LOG) <= PWS_MAX_ROOT_LOG) {
        const int SL = L < PWS_LOG ? L : PWS_LOG, R = 1 << (L - SL);
        DQN_LAUNCH(k_per_write_seg, dim3(R), dim3(256), 0, st_, tree, L, idx, val, B, mode, alpha, eps, pw_part);
        DQN_LAUNCH(k_per_top_seg, dim3(1), dim3(R >= 1024 ? 1024 : (R < 64 ? 64 : R)), sizeof(float) * 2 * R, st_, st, tree, L, pw_part);
        return;
    }
    DQN_LAUNCH(k_per_write_sorted, dim3((B + 63) / 64), dim3(64), 0, st_, st, tree, N, L, idx, val, B, mode,
                       alpha, eps);
    launch_per_top(st_, st, tree, L);
}
