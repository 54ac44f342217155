// csrc/dqn_cnn_btrunk.h -- r03: the backward-data steps of conv3 and conv2 (k_cnn_bwd_data<TC, 2> and <TC, 1> of dqn_cnn.hip: the
// gradient at a layer's INPUT pre-activations in gather form, masked by the ReLU gate of the layer below) as ONE persistent kernel in
// the bf16 mode, the backward twin of k_cnn_trunk16: a workgroup takes a PAIR of images from conv3's dZ to conv1's dZ without
// leaving the CU.
//
//   dZ3 in   [B][7][7][64] (gradient at conv3's pre-activations, from the fc layer's backward-data) -> LDS, zero-padded to 11 x 11
//   conv3^T  rows = the 2 x 81 input pixels (ih, iw), K = 9 taps x 64 oc, N = 64 ic: tap (kh, kw) reads dZ3[ih - kh][iw - kw] -- one
//            16-byte LDS read at base(row) + constant, the zero border stands in for "outside the map"; gate a1 > 0 -> dZ2: to LDS
//            (zero-padded 11 x 11 again) and to HBM (conv3's dW reads it)
//   conv2^T  stride 2: an input pixel is reached only by the taps of its parity, so WAVE w OWNS PARITY CLASS w (rows = its 2 x 100
//            pixels (i2, j2), K = 4 taps x 64 oc, N = 32 ic, its own 32 x 256 weight matrix): tap (th, tw) reads dZ2[i2 - th][j2 - tw];
//            gate a0 > 0 -> dZ1 to HBM (conv1's dW reads it)
//   weights  all fragments in registers for the whole kernel (conv3^T 144 + one class of conv2^T 64 per lane), from fragment-packed
//            shadows (CnnShadows::wpb)
// Same MFMA, same k order per output element, same gate arithmetic as k_cnn_bwd_data: BIT-IDENTICAL gradients
// (test_cnn_trunk_bitexact compares every leaf with DQN_CNN_FLAG_LAYERWISE_CONV). Included by dqn_cnn.hip after dqn_cnn_trunk.h.
#pragma once

struct BTrunkArgs {
    const __bf16 *dz3;                                 // [B][49][64]
    const __bf16 *a1, *a0;                             // forward maps (gates): [B][81][64], [B][400][32]
    __bf16 *dz2, *dz1;                                 // out: [B][81][64], [B][400][32]
    const __bf16 *w3t, *w2t;                           // packed: [2 column tiles][36 steps][64 lanes][8], [4 classes][16 steps][64][8]
    int B;
};

constexpr int BT_PIX = 128 + 16, BT_IMG = 121 * BT_PIX;        // an 11 x 11 map of 64 channels (+ 16 B: bank spread)
constexpr int BT_OFF_Z3 = 0, BT_OFF_Z2 = 2 * BT_IMG, BT_LDS = 4 * BT_IMG;

// gate + bf16 of one accumulator tile: element 4 gq + u = channel ch0 + 8 gq + u; gate = the forward map's value (> 0: open)
__device__ __forceinline__ void btrunk_out(const f32x16c &acc, const bf16x4t (&gate)[4], bf16x4t (&o)[4]) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int u = 0; u < 4; ++u) o[gq][u] = (__bf16)((float)gate[gq][u] > 0.0f ? acc[4 * gq + u] : 0.0f);
}

__global__ void __launch_bounds__(256)
k_cnn_btrunk16(BTrunkArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[BT_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, wm = wave & 1, wn = wave >> 1;
    const int npairs = (g.B + 1) >> 1;
    // zero the maps once: the borders are never written again
    for (int q = tid; q < BT_LDS / 16; q += 256) reinterpret_cast<uint4 *>(lds)[q] = uint4{0u, 0u, 0u, 0u};
    bf16x8c w3r[36], w2r[16];
    {
        const __bf16 *p3 = g.w3t + wn * (36 * 512) + lane * 8, *p2 = g.w2t + wave * (16 * 512) + lane * 8;
#pragma unroll
        for (int s = 0; s < 36; ++s) w3r[s] = *reinterpret_cast<const bf16x8c *>(p3 + s * 512);
#pragma unroll
        for (int s = 0; s < 16; ++s) w2r[s] = *reinterpret_cast<const bf16x8c *>(p2 + s * 512);
#pragma unroll
        for (int s = 0; s < 16; ++s) asm volatile("" : "+a"(w2r[s]));
    }
    // dZ3 of a pair: 2 x 392 pieces of 16 B into the padded maps
    auto load_z3 = [&](int pair) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = tid + 256 * u;
            if (q < 784) {
                const int img = q >= 392 ? 1 : 0, r = q - 392 * img, p = r >> 3, j = r & 7, y = p / 7, x = p - 7 * y;
                int imgG = 2 * pair + img; imgG = imgG < g.B ? imgG : g.B - 1;
                const uint4 v = *reinterpret_cast<const uint4 *>(g.dz3 + ((long long)imgG * 49 + p) * 64 + 8 * j);
                *reinterpret_cast<uint4 *>(lds + BT_OFF_Z3 + img * BT_IMG + ((y + 2) * 11 + x + 2) * BT_PIX + 16 * j) = v;
            }
        }
    };
    __syncthreads();                                   // (the zero fill, before the first interior writes)
    for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
        int c = lane & 31;                             // opaque per pair: tile positions are formed next to their use (see dqn_cnn_trunk.h)
        asm volatile("" : "+v"(c));
        load_z3(pair);
        LDS_BARRIER();
        // ---- conv3^T: column tile wn (32 ic), row tiles wm, wm + 2, wm + 4 of 6 (162 rows)
        {
            auto off3 = [](int s) {                    // step s: tap s >> 2 = (kh, kw), oc 16 (s & 3) + 8 h ..
                const int tap = s >> 2, kh = tap / 3, kw = tap - 3 * kh;
                return -(kh * 11 + kw) * BT_PIX + 32 * (s & 3); };
            bf16x8c ring[TR_PF];
            int m = 32 * wm + c, img = m >= 81 ? 1 : 0, pos = m - 81 * img, iy = pos / 9, ix = pos - 9 * iy;
            const unsigned char *ab = lds + BT_OFF_Z3 + img * BT_IMG + ((iy + 2) * 11 + ix + 2) * BT_PIX + 16 * h;
            trunk_fill(ring, ab, off3);
            // the ReLU gates of a tile are asked for ONE TILE AHEAD (a global round trip is longer than a tile's MFMAs)
            auto gates3 = [&](int gimg, int gpos, bf16x4t (&gt)[4]) {
                int ig = 2 * pair + gimg; ig = ig < g.B ? ig : g.B - 1;
                const __bf16 *gp = g.a1 + ((long long)ig * 81 + gpos) * 64 + 32 * wn + 4 * h;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) gt[gq] = *reinterpret_cast<const bf16x4t *>(gp + 8 * gq);
            };
            bf16x4t gate[4], gate_next[4];
            gates3(img, pos, gate_next);
            for (int t = wm; t < 6; t += 2) {
                const int cimg = img, cpos = pos, ciy = iy, cix = ix, imgG = 2 * pair + cimg;
                const bool valid = 32 * t + c < 162 && imgG < g.B;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) gate[gq] = gate_next[gq];
                if (t + 2 < 6) {
                    m = 32 * (t + 2) + c; m = m < 162 ? m : 161;
                    img = m >= 81 ? 1 : 0; pos = m - 81 * img; iy = pos / 9; ix = pos - 9 * iy;
                    gates3(img, pos, gate_next);
                }
                f32x16c acc;
                trunk_tile<36>(w3r, ring, ab, off3, acc);
                if (t + 2 < 6) {
                    ab = lds + BT_OFF_Z3 + img * BT_IMG + ((iy + 2) * 11 + ix + 2) * BT_PIX + 16 * h;
                    trunk_fill(ring, ab, off3);
                }
                bf16x4t o[4];
                btrunk_out(acc, gate, o);
                if (32 * t + c < 162) {                // (an absent second image: finite garbage into its map, never stored to HBM)
                    unsigned char *dst = lds + BT_OFF_Z2 + cimg * BT_IMG + ((ciy + 1) * 11 + cix + 1) * BT_PIX + (32 * wn + 4 * h) * 2;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<bf16x4t *>(dst + 16 * gq) = o[gq];
                }
                if (valid) {
                    __bf16 *gd = g.dz2 + ((long long)imgG * 81 + cpos) * 64 + 32 * wn + 4 * h;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<bf16x4t *>(gd + 8 * gq) = o[gq];
                }
            }
        }
        LDS_BARRIER();
        // ---- conv2^T: this wave's parity class (ph, pw) = (wave >> 1, wave & 1): 200 rows = 7 tiles, 32 ic
        {
            const int ph = wave >> 1, pw = wave & 1;
            auto off2 = [](int s) {                    // step s: tap s >> 2 = (th, tw), oc 16 (s & 3) + 8 h ..
                const int tt = s >> 2, th = tt >> 1, tw = tt & 1;
                return -(th * 11 + tw) * BT_PIX + 32 * (s & 3); };
            bf16x8c ring[TR_PF];
            int m = c, img = 0, pos = m, i2 = pos / 10, j2 = pos - 10 * i2;
            const unsigned char *ab = lds + BT_OFF_Z2 + img * BT_IMG + ((i2 + 1) * 11 + j2 + 1) * BT_PIX + 16 * h;
            trunk_fill(ring, ab, off2);
            auto opix_of = [&](int gimg, int gi2, int gj2) {
                int ig = 2 * pair + gimg; ig = ig < g.B ? ig : g.B - 1;
                return ((long long)ig * 20 + 2 * gi2 + ph) * 20 + 2 * gj2 + pw; };
            auto gates2 = [&](long long px, bf16x4t (&gt)[4]) {
                const __bf16 *gp = g.a0 + px * 32 + 4 * h;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) gt[gq] = *reinterpret_cast<const bf16x4t *>(gp + 8 * gq);
            };
            bf16x4t gate[4], gate_next[4];
            long long opix_next = opix_of(img, i2, j2);
            gates2(opix_next, gate_next);
            for (int t = 0; t < 7; ++t) {
                const int cimg = img, imgG = 2 * pair + cimg;
                const bool valid = 32 * t + c < 200 && imgG < g.B;
                const long long opix = opix_next;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) gate[gq] = gate_next[gq];
                if (t + 1 < 7) {
                    m = 32 * (t + 1) + c; m = m < 200 ? m : 199;
                    img = m >= 100 ? 1 : 0; pos = m - 100 * img; i2 = pos / 10; j2 = pos - 10 * i2;
                    opix_next = opix_of(img, i2, j2);
                    gates2(opix_next, gate_next);
                }
                f32x16c acc;
                trunk_tile<16>(w2r, ring, ab, off2, acc);
                if (t + 1 < 7) {
                    ab = lds + BT_OFF_Z2 + img * BT_IMG + ((i2 + 1) * 11 + j2 + 1) * BT_PIX + 16 * h;
                    trunk_fill(ring, ab, off2);
                }
                bf16x4t o[4];
                btrunk_out(acc, gate, o);
                if (valid) {
                    __bf16 *gd = g.dz1 + opix * 32 + 4 * h;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<bf16x4t *>(gd + 8 * gq) = o[gq];
                }
            }
        }
        LDS_BARRIER();                                 // (the next pair's dZ3 / dZ2 writes: everyone is done reading)
    }
}
