// csrc/dqn_cnn_trunk.h -- r03: the three convolutions of the Nature-CNN trunk (dqn_cnn.hip's header: conv1 32 x 8x8 / 4, conv2
// 64 x 4x4 / 2, conv3 64 x 3x3 / 1, ReLU after each) as ONE persistent kernel in the bf16 mode: a workgroup takes a PAIR of frame
// stacks from the u8 frames to conv3's output without leaving the CU.
//
//   frames   28 224 B of u8 per image arrive in LDS by LDS-DMA (global_load_lds_dwordx4, no registers), one image at a time: the
//            pair's second image travels behind conv1 of the first, the next pair's first image behind conv1 of the second,
//            conv2 and conv3; every pixel is converted to bf16 ONCE into an LDS image (integers 0..255 are exact in bf16; 1/255
//            is in the weight shadow)
//   conv1    per image M = 400 rows = 13 tiles of 32, K = 256: the patch operand of k-step (kh, half) is ONE 16-byte LDS read
//            (two pixels x 4 channels) at base(row) + constant
//   conv2    M = 162 rows = 6 tiles x 2 column tiles, K = 512: the patch operand of k-step (kh, kw, channel half) is ONE 16-byte
//            LDS read at base(row) + constant from conv1's map, which is kept in LDS with its columns split by parity
//            ([iy][ix & 1][ix >> 1]: the stride-2 walk of conv2 becomes a stride-1 walk)
//   conv3    M = 98 rows = 4 tiles x 2, K = 576, the same from conv2's map
//   weights  ALL THREE LAYERS' MFMA FRAGMENTS LIVE IN REGISTERS for the whole kernel (one wave per SIMD = the whole 512-register
//            file: conv1 64 + conv2 128 + conv3 144 registers per lane; wave (wm, wn) owns column tile wn of conv2 / conv3):
//            no weight traffic per image at all
//   outputs  transposed accumulators (weights as the A operand): a lane holds 4 consecutive channels of one position = one
//            8-byte store into the NHWC maps -- into LDS for the next layer and, when the caller wants the map (the backward
//            does), to HBM in the per-layer kernels' layout
// Same MFMA (v_mfma_f32_32x32x16_bf16), same k order per output element, same epilogue arithmetic as k_cnn_layer: the maps are
// BIT-IDENTICAL to the per-layer kernels' (test_cnn_trunk_bitexact). Included by dqn_cnn.hip.
#pragma once

typedef __bf16 bf16x4t __attribute__((ext_vector_type(4)));

// (diagnostic builds: stamps of workgroup 0, or with -DTRUNK_STAMP_LAST of the last one -- the first gets its weights first)
#if defined(DQN_STAMPS) && defined(TRUNK_STAMP_LAST)
#define TSTAMP(S) do { if (threadIdx.x == 0 && blockIdx.x == 0 && (S == 0 || S == 10)) { g_stamps[7][20 + S][0] = __builtin_amdgcn_s_memtime(); g_stamps[7][20 + S][1] = __builtin_amdgcn_s_memrealtime(); } \
                       if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) { g_stamps[7][S][0] = __builtin_amdgcn_s_memtime(); g_stamps[7][S][1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define TSTAMP(S) STAMP(7, S)
#endif

struct TrunkJob {
    const uint8_t *f1, *f2; int images1, B;            // images 0 .. images1-1 from f1, the rest from f2 (the paired pass)
    const __bf16 *w0, *w1, *w2;                        // fragment-packed shadows (CnnShadows::wp; conv1's holds W / 255)
    const float *b0, *b1, *b2;
    __bf16 *a0, *a1, *a2;                              // conv1 / conv2 maps: nullptr = not wanted; conv3's is always written
    // idx != nullptr: f1 / f2 are rings of `cap` frame stacks and image b is the row (idx[b] + off1) mod cap of f1, image
    // images1 + b the row (idx[b] + off2) mod cap of f2 -- the sampled transitions are read where they lie (the s / s' gather of
    // Agent._step from the frame ring is this kernel's DMA source address: no 2 x 14.4 MB staging copy in front of the pass)
    const int32_t *idx; long long cap; int off1, off2;
};
// One launch runs up to two passes (Agent._step: the online net over s | s' and the target net over s'): the pairs of pass 0,
// then those of pass 1, form ONE list that the workgroups walk with stride gridDim -- every workgroup gets the same number of
// pairs (768 pairs on 256 CUs = 3 each), and fetches the other net's weight fragments when its next pair belongs to the other
// pass. (Two launches cannot share a CU -- each workgroup owns the whole LDS and register file -- so the target pass on a side
// stream ran BEHIND the online pass, not beside it.)
struct TrunkArgs { TrunkJob j[2]; int pairs0, total; };

constexpr int TR_FR = 84 * 84 * 4;                     // one frame stack
constexpr int TR_U8 = TR_FR + 1024;                    // DMA target (1 KB slack: the last piece is a whole 1-KB wave instruction)
// bf16 image of ONE frame stack: row = two planes of 21 pixel PAIRS (16 B each; plane = pair & 1) -- conv1's operand of k-step
// (kh, half) for output (oy, ox) is the pair 2 ox + 2 half + h, so the lanes of a tile walk a plane 16 B at a time; the row
// stride makes the 4-row step to the next oy land 4 slots further (conflict-free 16-lane groups, see tools/diag/trunk_banks.py)
constexpr int TR_BF_PL = 21 * 16, TR_BF_RS = 720, TR_BF = 84 * TR_BF_RS;
constexpr int TR_PIX0 = 64 + 16, TR_IMG0 = 400 * TR_PIX0;      // conv1 map in LDS: 32 channels + 16 B (bank spread)
constexpr int TR_PIX1 = 128 + 16, TR_ROW1 = 1520, TR_IMG1 = 13712;   // conv2 map (lives where the bf16 image was: dead after conv1); strides
                                                       // searched for conflict-free conv3 reads (tools/diag/trunk_banks.py)
constexpr int TR_OFF_BF = TR_U8;
constexpr int TR_OFF_A1 = TR_OFF_BF;
constexpr int TR_OFF_A0 = TR_OFF_BF + TR_BF;
constexpr int TR_OFF_B = TR_OFF_A0 + 2 * TR_IMG0;
constexpr int TR_LDS = TR_OFF_B + 2 * 160 * 4;
constexpr int TR_PF = 6;                               // operand reads in flight ahead of the MFMAs (register ring)
static_assert(TR_OFF_A0 % 16 == 0 && TR_OFF_BF % 16 == 0 && 2 * TR_IMG1 <= TR_BF && TR_LDS <= 160 * 1024, "LDS carve");

// one accumulator tile: D[channel][position] += W-fragment (registers) x patch fragment (one 16-byte LDS read at ab + off(s)), the
// reads TR_PF steps ahead of the MFMAs in a register ring (written step by step the compiler waits for every read right before
// its MFMA; sched_barrier keeps the issue order). The ring is filled by the caller -- for the NEXT tile right after this tile's
// last MFMA, so that the fill's latency passes behind the epilogue.
template <typename OFF>
__device__ __forceinline__ void trunk_fill(bf16x8c (&ring)[TR_PF], const unsigned char *ab, OFF off) {
#pragma unroll
    for (int s = 0; s < TR_PF; ++s) ring[s] = *reinterpret_cast<const bf16x8c *>(ab + off(s));
}
template <int NS, typename OFF>
__device__ __forceinline__ void trunk_tile(const bf16x8c (&w)[NS], bf16x8c (&ring)[TR_PF], const unsigned char *ab, OFF off, f32x16c &acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bf16x8c cur = ring[s % TR_PF];
        if (s + TR_PF < NS) ring[s % TR_PF] = *reinterpret_cast<const bf16x8c *>(ab + off(s + TR_PF));
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], cur, acc, 0, 0, 0);
    }
}
// bias + ReLU + bf16 of one accumulator tile (k_cnn_layer's epilogue arithmetic): element 4 gq + u = channel ch0 + 8 gq + u
__device__ __forceinline__ void trunk_out(const f32x16c &acc, const float (&bias)[16], bf16x4t (&o)[4]) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float v = acc[4 * gq + u] + bias[4 * gq + u]; o[gq][u] = (__bf16)(v > 0.0f ? v : 0.0f); }
}

__global__ void __launch_bounds__(256)
k_cnn_trunk16(TrunkArgs ga) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[TR_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31, wm = wave & 1, wn = wave >> 1;
    float *lb = reinterpret_cast<float *>(lds + TR_OFF_B);
#pragma unroll
    for (int n = 0; n < 2; ++n)                        // (pass 1 absent: its pointers repeat pass 0's)
        if (tid < 160) lb[160 * n + tid] = tid < 32 ? ga.j[n].b0[tid] : (tid < 96 ? ga.j[n].b1[tid - 32] : ga.j[n].b2[tid - 96]);
    int net = (int)blockIdx.x >= ga.pairs0 ? 1 : 0;      // the pass of the current pair (set by the pass loop below)
    // biases of this lane's 16 accumulator elements (channel ch0 + 8 gq + u): read once per phase, before its first tile
    auto biases = [&](float (&bias)[16], int ch0) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 bv = *reinterpret_cast<const float4 *>(lb + 160 * net + ch0 + 8 * gq);
            bias[4 * gq] = bv.x; bias[4 * gq + 1] = bv.y; bias[4 * gq + 2] = bv.z; bias[4 * gq + 3] = bv.w;
        }
    };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    // one image's frames: 28 pieces of 1 KB (wave instruction = 64 lanes x 16 B, LDS destination lane-linear), 7 per wave.
    // (inline asm, not __builtin_amdgcn_global_load_lds: with the builtin the compiler puts s_waitcnt vmcnt(0) in front of EVERY
    //  later LDS read -- the DMA may alias it -- and vmcnt(0) also waits for the map stores just issued: the stamps had 600 cycles
    //  per epilogue block, 4/5 of the kernel. The DMA's completion is waited for by hand: frames_landed.)
    auto dma = [&](const TrunkJob &g, int image) {
        image = image < g.B ? image : g.B - 1;
        const bool second = image >= g.images1;
        long long row = second ? image - g.images1 : image;
        if (g.idx) {
            row = g.idx[row];
            row = row < 0 ? 0 : (row >= g.cap ? g.cap - 1 : row);
            row = (row + (second ? g.off2 : g.off1)) % g.cap;
        }
        const uint8_t *p = (second ? g.f2 : g.f1) + row * TR_FR;
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int piece = wave + 4 * u, o = 1024 * piece + 16 * lane;
            const uint8_t *src = o < TR_FR ? p + o : p;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + 1024u * (unsigned)piece);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
    auto frames_landed = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share (and every store before it)
        __syncthreads();                                         // everyone's
    };
    TSTAMP(0);
    if ((int)blockIdx.x < ga.total) dma(ga.j[net], 2 * ((int)blockIdx.x - (net ? ga.pairs0 : 0)));
    // the weights: fragment of channel c (+ 32 wn) for k-step s = 8 consecutive k at 16 s + 8 h, packed 1 KB per (column tile, step)
    bf16x8c w0r[16], w1r[32], w2r[36];
    auto fetch_weights = [&](const TrunkJob &g) {
        int lo = lane * 8;                             // opaque per call: the 84 addresses are formed here, next to their loads (hoisted
        asm volatile("" : "+v"(lo));                   // out of the pass loop they were 28 spilled offsets, each reload a vmcnt(0))
        const __bf16 *p0 = g.w0 + lo, *p1 = g.w1 + wn * (32 * 512) + lo, *p2 = g.w2 + wn * (36 * 512) + lo;
#pragma unroll
        for (int s = 0; s < 16; ++s) w0r[s] = *reinterpret_cast<const bf16x8c *>(p0 + s * 512);
#pragma unroll
        for (int s = 0; s < 32; ++s) w1r[s] = *reinterpret_cast<const bf16x8c *>(p1 + s * 512);
#pragma unroll
        for (int s = 0; s < 36; ++s) w2r[s] = *reinterpret_cast<const bf16x8c *>(p2 + s * 512);
        // conv1's and conv2's fragments are PINNED to accumulation registers (the MFMA reads its A operand from either file):
        // left to itself the allocator parks them there as spills and copies every one (4 x v_accvgpr_mov) in front of its MFMA
#pragma unroll
        for (int s = 0; s < 16; ++s) asm volatile("" : "+a"(w0r[s]));
#pragma unroll
        for (int s = 0; s < 32; ++s) asm volatile("" : "+a"(w1r[s]));
    };

    // u8 frames (DMA target) -> the bf16 image: every pixel converted ONCE (integers 0..255 are exact in bf16; 1/255 is in the
    // weight shadow). Converting inside conv1's k-loop instead (each pixel is in 4 patches) made conv1 VALU-bound: 16 VALU
    // instructions per MFMA, 148 cycles per step by the stamps.
    const int cv_y = tid / 21, cv_j = tid - 21 * cv_y;        // thread (row within a group of 12, 16-byte piece of the row); tid < 252
    auto convert = [&]() {
        if (tid < 252) {
#pragma unroll
            for (int u = 0; u < 7; ++u) {                          // rows cv_y + 12 u; piece = 4 pixels = pairs 2 j, 2 j + 1
                const int y = cv_y + 12 * u;
                const uint4 raw = *reinterpret_cast<const uint4 *>(lds + (y * 21 + cv_j) * 16);
                const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
                bf16x8c p0, p1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    p0[i] = (__bf16)(float)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
                    p1[i] = (__bf16)(float)((w[2 + (i >> 2)] >> (8 * (i & 3))) & 0xffu);
                }
                unsigned char *d = lds + TR_OFF_BF + y * TR_BF_RS + 16 * cv_j;
                *reinterpret_cast<bf16x8c *>(d) = p0;
                *reinterpret_cast<bf16x8c *>(d + TR_BF_PL) = p1;
            }
        }
    };
    // conv1 of one image: 13 row tiles of 32 (the last one half empty), tiles wave, wave + 4, ...
    // k = (kh, kw, ch): step s = row kh = s >> 1, pixels 4 (s & 1) + 2 h, + 1 = pair 2 ox + 2 (s & 1) + h
    auto off0 = [](int s) { return (s >> 1) * TR_BF_RS + 16 * (s & 1); };
    auto conv1 = [&](const TrunkJob &g, int img, int imgG) {
        bf16x8c ring[TR_PF];
        int pos = 32 * wave + c, oy = pos / 20, ox = pos - 20 * oy;
        const unsigned char *ab = lds + TR_OFF_BF + (4 * oy) * TR_BF_RS + h * TR_BF_PL + ox * 16;
        trunk_fill(ring, ab, off0);
        float bias0[16];
        biases(bias0, 4 * h);
        for (int t = wave; t < 13; t += 4) {
            f32x16c acc;
            trunk_tile<16>(w0r, ring, ab, off0, acc);
            const int cpos = pos, coy = oy, cox = ox;
            const bool valid = 32 * t + c < 400;
            if (t + 4 < 13) {
                pos = 32 * (t + 4) + c; pos = pos < 400 ? pos : 399;
                oy = pos / 20; ox = pos - 20 * oy;
                ab = lds + TR_OFF_BF + (4 * oy) * TR_BF_RS + h * TR_BF_PL + ox * 16;
                trunk_fill(ring, ab, off0);
            }
            bf16x4t o[4];
            trunk_out(acc, bias0, o);
            if (valid) {
                unsigned char *dst = lds + TR_OFF_A0 + img * TR_IMG0 + ((coy * 2 + (cox & 1)) * 10 + (cox >> 1)) * TR_PIX0 + 8 * h;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<bf16x4t *>(dst + 16 * gq) = o[gq];
                if (g.a0 && imgG < g.B) {
                    __bf16 *gd = g.a0 + ((long long)imgG * 400 + cpos) * 32 + 4 * h;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<bf16x4t *>(gd + 8 * gq) = o[gq];
                }
            }
        }
    };

    int job = blockIdx.x;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {             // (a loop, not a conditional reload inside the pair loop: the fragment registers
    const int end = pass == 0 ? ga.pairs0 : ga.total;  //  are rewritten unconditionally at its top -- the conditional form spilled)
    if (job >= end) continue;
    net = pass;
    const TrunkJob g = ga.j[pass];                     // BY VALUE, in scalar registers: through a reference every use of a field was an
                                                       // s_load + lgkmcnt(0), which also drained the operand reads in flight (once per tile)
    fetch_weights(g);
    for (; job < end; job += gridDim.x) {
        const int pair = job - (pass ? ga.pairs0 : 0);
        TSTAMP(1);
        frames_landed();                               // image 0 in the DMA target; conv3 of the previous pair is done with its input
        TSTAMP(2);
        convert();
        LDS_BARRIER();
        dma(g, 2 * pair + 1);                          // image 1 travels behind conv1 of image 0
        TSTAMP(3);
        conv1(g, 0, 2 * pair);
        TSTAMP(4);
        frames_landed();
        convert();
        LDS_BARRIER();
        {                                              // the next pair's first image: behind everything below
            const int nj = job + (int)gridDim.x;
            if (nj < ga.total) { const int nn = nj >= ga.pairs0 ? 1 : 0; dma(ga.j[nn], 2 * (nj - (nn ? ga.pairs0 : 0))); }
        }
        TSTAMP(5);
        conv1(g, 1, 2 * pair + 1);
        TSTAMP(6);
        LDS_BARRIER();                                 // conv1's maps complete; the bf16 image is dead (conv2's map replaces it)
        TSTAMP(7);
        // ---- conv2: column tile wn, row tiles wm, wm + 2, wm + 4 of 6
        {
            auto off1 = [](int s) {                    // step s = tap (kh, kw) = s >> 1, channels 16 (s & 1) + 8 h ..
                const int tap = s >> 1, kh = tap >> 2, kw = tap & 3;
                return ((kh * 2 + (kw & 1)) * 10 + (kw >> 1)) * TR_PIX0 + 32 * (s & 1); };
            bf16x8c ring[TR_PF];
            int m = 32 * wm + c, img = m >= 81 ? 1 : 0, pos = m - 81 * img, oy = pos / 9, ox = pos - 9 * oy;
            const unsigned char *ab = lds + TR_OFF_A0 + img * TR_IMG0 + ((4 * oy) * 10 + ox) * TR_PIX0 + 16 * h;
            trunk_fill(ring, ab, off1);
            float bias1[16];
            biases(bias1, 32 + 32 * wn + 4 * h);
            for (int t = wm; t < 6; t += 2) {
                f32x16c acc;
                trunk_tile<32>(w1r, ring, ab, off1, acc);
                TSTAMP(11 + 2 * (t >> 1));                 // (diagnostic builds: conv2's tiles -- MFMA loop | epilogue)
                const int cimg = img, cpos = pos, coy = oy, cox = ox;
                const bool valid = 32 * t + c < 162;
                if (t + 2 < 6) {
                    m = 32 * (t + 2) + c; m = m < 162 ? m : 161;
                    img = m >= 81 ? 1 : 0; pos = m - 81 * img; oy = pos / 9; ox = pos - 9 * oy;
                    ab = lds + TR_OFF_A0 + img * TR_IMG0 + ((4 * oy) * 10 + ox) * TR_PIX0 + 16 * h;
                    trunk_fill(ring, ab, off1);
                }
                bf16x4t o[4];
                trunk_out(acc, bias1, o);
                if (valid) {
                    const int ch = 32 * wn + 4 * h, imgG = 2 * pair + cimg;
                    unsigned char *dst = lds + TR_OFF_A1 + cimg * TR_IMG1 + coy * TR_ROW1 + cox * TR_PIX1 + ch * 2;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<bf16x4t *>(dst + 16 * gq) = o[gq];
                    if (g.a1 && imgG < g.B) {
                        __bf16 *gd = g.a1 + ((long long)imgG * 81 + cpos) * 64 + ch;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<bf16x4t *>(gd + 8 * gq) = o[gq];
                    }
                }
                TSTAMP(12 + 2 * (t >> 1));
            }
        }
        TSTAMP(8);
        LDS_BARRIER();
        TSTAMP(9);
        // ---- conv3: column tile wn, row tiles wm, wm + 2 of 4
        {
            auto off2 = [](int s) {                    // step s = tap s >> 2 = (kh, kw), channels 16 (s & 3) + 8 h ..
                const int tap = s >> 2, kh = tap / 3, kw = tap - 3 * kh;
                return kh * TR_ROW1 + kw * TR_PIX1 + 32 * (s & 3); };
            bf16x8c ring[TR_PF];
            int m = 32 * wm + c, img = m >= 49 ? 1 : 0, pos = m - 49 * img, oy = pos / 7, ox = pos - 7 * oy;
            const unsigned char *ab = lds + TR_OFF_A1 + img * TR_IMG1 + oy * TR_ROW1 + ox * TR_PIX1 + 16 * h;
            trunk_fill(ring, ab, off2);
            float bias2[16];
            biases(bias2, 96 + 32 * wn + 4 * h);
            for (int t = wm; t < 4; t += 2) {
                f32x16c acc;
                trunk_tile<36>(w2r, ring, ab, off2, acc);
                const int cimg = img, cpos = pos;
                const bool valid = 32 * t + c < 98;
                if (t + 2 < 4) {
                    m = 32 * (t + 2) + c; m = m < 98 ? m : 97;
                    img = m >= 49 ? 1 : 0; pos = m - 49 * img; oy = pos / 7; ox = pos - 7 * oy;
                    ab = lds + TR_OFF_A1 + img * TR_IMG1 + oy * TR_ROW1 + ox * TR_PIX1 + 16 * h;
                    trunk_fill(ring, ab, off2);
                }
                bf16x4t o[4];
                trunk_out(acc, bias2, o);
                const int imgG = 2 * pair + cimg;
                if (valid && imgG < g.B) {
                    __bf16 *gd = g.a2 + ((long long)imgG * 49 + cpos) * 64 + 32 * wn + 4 * h;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<bf16x4t *>(gd + 8 * gq) = o[gq];
                }
            }
        }
        TSTAMP(10);
    }
    }
}
