// Host packer: reference state_dict tensors -> per-lane MFMA A fragments.  Layout: nr_layout.h.
#include "nr_pack.h"

#include <cstring>

namespace nr {

static bool g_pack_unscaled = false;     // pack_pass_index_map: pack without the scaled-ELU factors
static bool g_pack_fp32_quads = false;   // the index maps are built by packing POSITIONS: fp32 quad slots in every build

#ifdef NR_BF16_QUADS
static unsigned short to_bf16(float f) {          // round to nearest even, as v_cvt_pk_bf16_f32
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
#endif

void pack_layer(float* dst, int layer, const float* W, int ldw, const float* bias, const LayerMaps& maps) {
    const LayerShape s = kShape[layer];
    float* q = dst + quads_offset(layer);
    float* s1 = dst + single_offset(layer);
    float* b = dst + bias_offset(layer);
    // scaled-ELU bookkeeping (nr_layout.h): rows x L when this layer's activation is the scaled ELU, columns / L when
    // its input is one; both -> exactly the original weight
    const double so = g_pack_unscaled ? 1.0 : (kOutScaled[layer] ? kLog2e : 1.0);
    const double sw = g_pack_unscaled ? 1.0 : ((kOutScaled[layer] == kInScaled[layer]) ? 1.0 : (kOutScaled[layer] ? kLog2e : 1.0 / kLog2e));
    for (int mo = 0; mo < s.mt_out; ++mo) {
        for (int kq = 0; kq < s.kq; ++kq)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int o = maps.out_map[mo * 16 + (lane & 15)];
                    const int i = maps.in_map[(4 * kq + j) * 4 + (lane >> 4)];
                    const float v = (o >= 0 && i >= 0) ? (float)(W[o * ldw + i] * sw) : 0.0f;
                    float* slot = q + ((mo * s.kq + kq) * 64 + lane) * 4;
#ifdef NR_BF16_QUADS      // the four weights of the quad as bf16 in the slot's first two dwords (the rest stays zero)
                    if (g_pack_fp32_quads) { slot[j] = v; continue; }
                    reinterpret_cast<unsigned short*>(slot)[j] = to_bf16(v);
#ifdef NR_BF16_SPLIT      // ... and what bf16 dropped, again as bf16, in the last two dwords: v = hi + lo to 2^-16
                    {
                        const unsigned hb = (unsigned)reinterpret_cast<unsigned short*>(slot)[j] << 16;
                        float hf;
                        std::memcpy(&hf, &hb, 4);
                        reinterpret_cast<unsigned short*>(slot)[4 + j] = to_bf16(v - hf);
                    }
#endif
#else
                    slot[j] = v;
#endif
                }
        for (int k1 = 0; k1 < s.k1; ++k1)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = maps.out_map[mo * 16 + (lane & 15)];
                const int i = maps.in1_map[k1 * 4 + (lane >> 4)];
                s1[(mo * s.k1 + k1) * 64 + lane] = (o >= 0 && i >= 0) ? (float)(W[o * ldw + i] * sw) : 0.0f;
            }
        for (int m = 0; m < 16; ++m) {
            const int o = maps.out_map[mo * 16 + m];
            b[mo * 16 + m] = (o >= 0 && bias) ? (float)(bias[o] * so) : 0.0f;
        }
    }
}

// vector rows (nr_layout.h kVec): row j of the layer = weight row rows[j], input feature f = column col0 + f of W
void pack_vec(float* dst, int layer, const float* W, int ldw, const float* bias, const int* rows, int col0, int nfeat) {
    const VecShape v = kVec[layer];
    const double sw = g_pack_unscaled ? 1.0 : (kInScaled[layer] ? 1.0 / kLog2e : 1.0);      // vector rows are never scaled-ELU outputs
    float* w = dst + vec_offset(layer);
    float* b = dst + vec_bias_offset(layer);
    for (int j = 0; j < v.n; ++j) {
        for (int t = 0; t < v.tiles; ++t)
            for (int g = 0; g < 4; ++g)
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * t + 4 * g + r;
                    w[(j * v.tiles + t) * 16 + 4 * g + r] = f < nfeat ? (float)(W[rows[j] * ldw + col0 + f] * sw) : 0.0f;
                }
        for (int g = 0; g < 4; ++g) b[4 * g + j] = bias ? bias[rows[j]] : 0.0f;
    }
}

namespace {

std::vector<int> out_natural(int mt_out, int n_real) {
    std::vector<int> m(mt_out * 16);
    for (int i = 0; i < mt_out * 16; ++i) m[i] = i < n_real ? i : -1;
    return m;
}
// 32 gathered channels: k-step s (0..7), lane group g <-> channel 8g + s
void in_gathered32(std::vector<int>& in, int col0) {
    for (int s = 0; s < 8; ++s)
        for (int g = 0; g < 4; ++g) in.push_back(col0 + 8 * g + s);
}
// D layout: k-step 4t + r, lane group g <-> feature 16t + 4g + r
void in_dlayout(std::vector<int>& in, int col0, int nfeat, int ntiles) {
    for (int t = 0; t < ntiles; ++t)
        for (int r = 0; r < 4; ++r)
            for (int g = 0; g < 4; ++g) {
                const int f = 16 * t + 4 * g + r;
                in.push_back(f < nfeat ? col0 + f : -1);
            }
}

void pack_mlp32(float* dst, int l1, int l2, const float* w0, const float* b0, const float* w2, const float* b2) {
    LayerMaps m1; m1.out_map = out_natural(2, 32); in_gathered32(m1.in_map, 0);
    pack_layer(dst, l1, w0, 32, b0, m1);
    LayerMaps m2; m2.out_map = out_natural(2, 32); in_dlayout(m2.in_map, 0, 32, 2);
    pack_layer(dst, l2, w2, 32, b2, m2);
}

}  // namespace

int pack_pass_weights(const float* const* t_in, float* dst, bool fold) {
    for (int i = 0; i < T_COUNT; ++i) {
        const bool optional = (i >= T_VIS0_W && i <= T_VIS4_B);
        if (!t_in[i] && !optional) return 1 + i;
    }
    const bool has_vis = t_in[T_VIS0_W] != nullptr;
    if (has_vis)
        for (int i = T_VIS0_W; i <= T_VIS4_B; ++i) if (!t_in[i]) return 1 + i;
    std::memset(dst, 0, sizeof(float) * kPackedPassFloats);
    const float* t[T_COUNT];
    for (int i = 0; i < T_COUNT; ++i) t[i] = t_in[i];
    // ---- fold: e = W2 h + b2 (prob_embed.2, no activation) feeds only neuray_fc.0 and base_fc.0's columns 175..206:
    //   Wn e + bn = (Wn W2) h + (Wn b2 + bn),   Wb[:, 175:207] e + bb = (Wb[:, 175:207] W2) h + (Wb[:, 175:207] b2 + bb)
    // products in double, rounded once to fp32
    std::vector<float> nf_w, nf_b, base_w, base_b;
    if (fold) {
        const float* W2 = t_in[T_PE2_W];
        const float* b2 = t_in[T_PE2_B];
        nf_w.resize(8 * 32); nf_b.resize(8);
        for (int o = 0; o < 8; ++o) {
            double bb = t_in[T_NF0_B][o];
            for (int k = 0; k < 32; ++k) bb += (double)t_in[T_NF0_W][o * 32 + k] * b2[k];
            nf_b[o] = (float)bb;
            for (int i = 0; i < 32; ++i) {
                double a = 0.0;
                for (int k = 0; k < 32; ++k) a += (double)t_in[T_NF0_W][o * 32 + k] * W2[k * 32 + i];
                nf_w[o * 32 + i] = (float)a;
            }
        }
        base_w.assign(t_in[T_BASE0_W], t_in[T_BASE0_W] + 64 * 207);
        base_b.resize(64);
        for (int o = 0; o < 64; ++o) {
            const float* row = t_in[T_BASE0_W] + o * 207 + 175;
            double bb = t_in[T_BASE0_B][o];
            for (int k = 0; k < 32; ++k) bb += (double)row[k] * b2[k];
            base_b[o] = (float)bb;
            for (int i = 0; i < 32; ++i) {
                double a = 0.0;
                for (int k = 0; k < 32; ++k) a += (double)row[k] * W2[k * 32 + i];
                base_w[o * 207 + 175 + i] = (float)a;
            }
        }
        t[T_NF0_W] = nf_w.data(); t[T_NF0_B] = nf_b.data();
        t[T_BASE0_W] = base_w.data(); t[T_BASE0_B] = base_b.data();
    }

    // ---- dist decoder -----------------------------------------------------------------------
    pack_mlp32(dst, L_DM1, L_DM2, t[T_MEAN0_W], t[T_MEAN0_B], t[T_MEAN2_W], t[T_MEAN2_B]);
    pack_mlp32(dst, L_DV1, L_DV2, t[T_VAR0_W], t[T_VAR0_B], t[T_VAR2_W], t[T_VAR2_B]);
    pack_mlp32(dst, L_DA1, L_DA2, t[T_AW0_W], t[T_AW0_B], t[T_AW2_W], t[T_AW2_B]);
    if (has_vis) pack_mlp32(dst, L_DS1, L_DS2, t[T_VIS0_W], t[T_VIS0_B], t[T_VIS2_W], t[T_VIS2_B]);
    {   // output rows of the heads (vector rows on the D-layout h2 of each head)
        const int r01[2] = {0, 1}, r0[1] = {0};
        pack_vec(dst, L_DFIN_M, t[T_MEAN4_W], 32, t[T_MEAN4_B], r01, 0, 32);
        pack_vec(dst, L_DFIN_V, t[T_VAR4_W], 32, t[T_VAR4_B], r01, 0, 32);
        pack_vec(dst, L_DFIN_A, t[T_AW4_W], 32, t[T_AW4_B], r0, 0, 32);
        if (has_vis) pack_vec(dst, L_DFIN_S, t[T_VIS4_W], 32, t[T_VIS4_B], r0, 0, 32);
    }
    // ---- prob_embed: input [ray_feats(32), hit, vis] ---------------------------------------------
    {
        LayerMaps m; m.out_map = out_natural(2, 32); in_gathered32(m.in_map, 0);
        for (int g = 0; g < 4; ++g) m.in1_map.push_back(g < 2 ? 32 + g : -1);
        pack_layer(dst, L_PE1, t[T_PE0_W], 34, t[T_PE0_B], m);
        LayerMaps m2; m2.out_map = out_natural(2, 32); in_dlayout(m2.in_map, 0, 32, 2);
        if (!fold) pack_layer(dst, L_PE2, t[T_PE2_W], 32, t[T_PE2_B], m2);     // (folded: the slot stays zero, the kernel skips the layer)
    }
    // ---- ray_dir_fc: 4 -> 16 -> 35; output rows re-ordered to [img channels in gathered order | rgb]
    {
        LayerMaps m; m.out_map = out_natural(1, 16);
        for (int g = 0; g < 4; ++g) m.in1_map.push_back(g);
        pack_layer(dst, L_RD1, t[T_RD0_W], 4, t[T_RD0_B], m);
        LayerMaps m2; m2.out_map.assign(32, -1);
        for (int tt = 0; tt < 2; ++tt)
            for (int g = 0; g < 4; ++g)
                for (int r = 0; r < 4; ++r) m2.out_map[tt * 16 + 4 * g + r] = 3 + (8 * g + 4 * tt + r);
        in_dlayout(m2.in_map, 0, 16, 1);
        pack_layer(dst, L_RD2, t[T_RD2_W], 16, t[T_RD2_B], m2);
        const int rgb_rows[3] = {0, 1, 2};
        pack_vec(dst, L_RD2, t[T_RD2_W], 16, t[T_RD2_B], rgb_rows, 0, 16);
    }
    // ---- neuray_fc 32 -> 8 -> 1 ---------------------------------------------------------------------
    {
        LayerMaps m; m.out_map = out_natural(1, 8); in_dlayout(m.in_map, 0, 32, 2);
        pack_layer(dst, L_NF1, t[T_NF0_W], 32, t[T_NF0_B], m);
        const int r0[1] = {0};
        pack_vec(dst, L_NF2, t[T_NF2_W], 8, t[T_NF2_B], r0, 0, 8);
    }
    // ---- base_fc.0 (64 x 207), columns [mean0 var0 mean1 var1 | rgb_feat neuray_feat]   ibrnet.py:340-342
    {
        LayerMaps m; m.out_map = out_natural(4, 64);
        for (int j = 0; j < 4; ++j) in_gathered32(m.in_map, 35 * j + 3);
        for (int j = 0; j < 4; ++j)                       // single k-step j = rgb part of statistic j
            for (int g = 0; g < 4; ++g) m.in1_map.push_back(g < 3 ? 35 * j + g : -1);
        pack_layer(dst, L_BG, t[T_BASE0_W], 207, t[T_BASE0_B], m);
        for (int half = 0; half < 2; ++half) {            // per-view columns, output rows 32*half .. 32*half+31
            LayerMaps v; v.out_map.resize(32);
            for (int i = 0; i < 32; ++i) v.out_map[i] = 32 * half + i;
            in_gathered32(v.in_map, 140 + 3);
            in_dlayout(v.in_map, 175, 32, 2);
            for (int g = 0; g < 4; ++g) v.in1_map.push_back(g < 3 ? 140 + g : -1);
            pack_layer(dst, half == 0 ? L_BV0 : L_BV1, t[T_BASE0_W], 207, nullptr, v);
        }
        LayerMaps m2; m2.out_map = out_natural(2, 32); in_dlayout(m2.in_map, 0, 64, 4);
        pack_layer(dst, L_B2, t[T_BASE2_W], 64, t[T_BASE2_B], m2);
    }
    // ---- vis_fc 32 -> 32 -> 33 (row 32 = visibility logit: vector row) -----------------------------
    {
        LayerMaps m; m.out_map = out_natural(2, 32); in_dlayout(m.in_map, 0, 32, 2);
        pack_layer(dst, L_VF1, t[T_VF0_W], 32, t[T_VF0_B], m);
        LayerMaps m2; m2.out_map = out_natural(2, 32); in_dlayout(m2.in_map, 0, 32, 2);
        pack_layer(dst, L_VF2, t[T_VF2_W], 32, t[T_VF2_B], m2);
        const int r32[1] = {32};
        pack_vec(dst, L_VF2, t[T_VF2_W], 32, t[T_VF2_B], r32, 0, 32);
    }
    // ---- vis_fc2 32 -> 32 -> 1 -----------------------------------------------------------------------
    {
        LayerMaps m; m.out_map = out_natural(2, 32); in_dlayout(m.in_map, 0, 32, 2);
        pack_layer(dst, L_V21, t[T_V20_W], 32, t[T_V20_B], m);
        const int r0[1] = {0};
        pack_vec(dst, L_V22, t[T_V22_W], 32, t[T_V22_B], r0, 0, 32);
    }
    // ---- rgb_fc [x(32), vis(1), ray_diff(4)] -> 16 -> 8 -> 1 -----------------------------------------
    {
        LayerMaps m; m.out_map = out_natural(1, 16); in_dlayout(m.in_map, 0, 32, 2);
        for (int g = 0; g < 4; ++g) m.in1_map.push_back(32 + g);          // vis, d0, d1, d2
        for (int g = 0; g < 4; ++g) m.in1_map.push_back(g == 0 ? 36 : -1);  // d3 (dot product)
        pack_layer(dst, L_RF1, t[T_RF0_W], 37, t[T_RF0_B], m);
        LayerMaps m2; m2.out_map = out_natural(1, 8); in_dlayout(m2.in_map, 0, 16, 1);
        pack_layer(dst, L_RF2, t[T_RF2_W], 16, t[T_RF2_B], m2);
        const int r0[1] = {0};
        pack_vec(dst, L_RF3, t[T_RF4_W], 8, t[T_RF4_B], r0, 0, 8);
    }
    // ---- geometry_fc [mean(32), var(32), mean weight(1)] -> 64 -> 16 --------------------------------
    {
        LayerMaps m; m.out_map = out_natural(4, 64);
        in_dlayout(m.in_map, 0, 32, 2); in_dlayout(m.in_map, 32, 32, 2);
        for (int g = 0; g < 4; ++g) m.in1_map.push_back(g == 0 ? 64 : -1);
        pack_layer(dst, L_GF1, t[T_GF0_W], 65, t[T_GF0_B], m);
        LayerMaps m2; m2.out_map = out_natural(1, 16); in_dlayout(m2.in_map, 0, 64, 4);
        pack_layer(dst, L_GF2, t[T_GF2_W], 64, t[T_GF2_B], m2);
    }
    // ---- ray kernel weights (row major copies) ---------------------------------------------------------
    float* r = dst + kPackedPointFloats;
    std::memcpy(r + RW_WQ, t[T_WQ], 256 * 4); std::memcpy(r + RW_WK, t[T_WK], 256 * 4);
    std::memcpy(r + RW_WV, t[T_WV], 256 * 4); std::memcpy(r + RW_FC, t[T_FC], 256 * 4);
    std::memcpy(r + RW_LNW, t[T_LN_W], 16 * 4); std::memcpy(r + RW_LNB, t[T_LN_B], 16 * 4);
    std::memcpy(r + RW_OG0W, t[T_OG0_W], 256 * 4); std::memcpy(r + RW_OG0B, t[T_OG0_B], 16 * 4);
    std::memcpy(r + RW_OG2W, t[T_OG2_W], 16 * 4); std::memcpy(r + RW_OG2B, t[T_OG2_B], 4);
    return 0;
}

// ---- AR_X3 (nr_layout.h): the folded fp32 pack re-written with every quad weight split three ways --------------------------
namespace {
unsigned short bf16_rn(float f) {                 // round to nearest even (finite weights)
    unsigned u; std::memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
float bf16_value(unsigned short b) { const unsigned u = (unsigned)b << 16; float f; std::memcpy(&f, &u, 4); return f; }
}  // namespace

void split3_bf16(float w, unsigned short (&part)[3]) {
    part[0] = bf16_rn(w);
    const float r1 = w - bf16_value(part[0]);     // exact: both are multiples of w's last bit, the difference needs <= 24 bits
    part[1] = bf16_rn(r1);
    const float r2 = r1 - bf16_value(part[1]);    // exact, and representable in 8 significant bits
    part[2] = bf16_rn(r2);
}

int pack_pass_weights_x3(const float* const* tensors, float* dst) {
    std::vector<float> f32(kPackedPassFloats, 0.0f);
    const int rc = pack_pass_weights(tensors, f32.data(), true);
    if (rc) return rc;
    std::memset(dst, 0, sizeof(float) * (size_t)kPackedPointFloatsX3);
    unsigned* words = reinterpret_cast<unsigned*>(dst);
    for (int l = 0; l < L_FWD_COUNT; ++l) {
        if (ar_omits(l, AR_X3)) continue;
        if (!ar_splits(l, AR_X3)) {              // per-point layers: the fp32 layer as it is
            std::memcpy(dst + layer_offset(l, AR_X3), f32.data() + layer_offset(l), sizeof(float) * (size_t)layer_floats(l));
            continue;
        }
        const int MT = kShape[l].mt_out, KQ = kShape[l].kq;
        const float* q = f32.data() + quads_offset(l);
        for (int mo = 0; mo < MT; ++mo)
            for (int kq = 0; kq < KQ; ++kq)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        unsigned short part[3];
                        split3_bf16(q[((mo * KQ + kq) * 64 + lane) * 4 + j], part);
                        const bool half = (KQ & 1) && kq == KQ - 1;        // the last quad of an odd count stands alone (K = 16)
                        const int unit = quads_offset(l, AR_X3) + mo * tile_quads_floats(l, AR_X3) + (kq / 2) * 768;
                        for (int pt = 0; pt < 3; ++pt) {
                            // pair: [part][lane] 4 words, value i = 4 (kq & 1) + j in word i / 2; single quad: [part][lane] 2 words
                            const int i = half ? j : 4 * (kq & 1) + j;
                            const int word = half ? unit + pt * 128 + lane * 2 + i / 2 : unit + pt * 256 + lane * 4 + i / 2;
                            words[word] |= (unsigned)part[pt] << (16 * (i & 1));
                        }
                    }
        // singles, biases, vector rows: as in the fp32 pack
        std::memcpy(dst + single_offset(l, AR_X3), f32.data() + single_offset(l), sizeof(float) * (size_t)(layer_floats(l) - quads_floats(l)));
    }
    return 0;
}

// ---- transposed layers (nr_layout.h LT_*): dX = W^T dY for the backward pass, second packed buffer --------------------
namespace {

// Wt[c * rows + r] = W[r * ldw + c]
std::vector<float> transposed(const float* W, int rows, int ldw, int cols) {
    std::vector<float> t((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = W[(size_t)r * ldw + c];
    return t;
}
// output rows in the gathered register order: tile mo, lane group g, register r <-> channel 8g + 4mo + r (+ col0)
std::vector<int> out_gathered32(int col0) {
    std::vector<int> m(32);
    for (int mo = 0; mo < 2; ++mo)
        for (int g = 0; g < 4; ++g)
            for (int r = 0; r < 4; ++r) m[mo * 16 + 4 * g + r] = col0 + 8 * g + 4 * mo + r;
    return m;
}
std::vector<int> out_natural_at(int col0, int n, int tiles) {
    std::vector<int> m(tiles * 16);
    for (int i = 0; i < tiles * 16; ++i) m[i] = i < n ? col0 + i : -1;
    return m;
}
void append(std::vector<int>& a, const std::vector<int>& b) { a.insert(a.end(), b.begin(), b.end()); }

// 32 -> 32 head layers: first (input = gathered f_ray) and second (input = D layout)
void pack_mlp32_t(float* dst, int lt1, int lt2, const float* w0, const float* w2) {
    const std::vector<float> t0 = transposed(w0, 32, 32, 32), t2 = transposed(w2, 32, 32, 32);
    LayerMaps m1; m1.out_map = out_gathered32(0); in_dlayout(m1.in_map, 0, 32, 2);
    pack_layer(dst, lt1, t0.data(), 32, nullptr, m1);
    LayerMaps m2; m2.out_map = out_natural(2, 32); in_dlayout(m2.in_map, 0, 32, 2);
    pack_layer(dst, lt2, t2.data(), 32, nullptr, m2);
}

}  // namespace

int pack_pass_t_weights(const float* const* t, float* dst) {
    for (int i = 0; i < T_COUNT; ++i) {
        const bool optional = (i >= T_VIS0_W && i <= T_VIS4_B);
        if (!t[i] && !optional) return 1 + i;
    }
    const bool has_vis = t[T_VIS0_W] != nullptr;
    std::memset(dst, 0, sizeof(float) * kPackedTFloats);
    pack_mlp32_t(dst, LT_DM1, LT_DM2, t[T_MEAN0_W], t[T_MEAN2_W]);
    pack_mlp32_t(dst, LT_DV1, LT_DV2, t[T_VAR0_W], t[T_VAR2_W]);
    pack_mlp32_t(dst, LT_DA1, LT_DA2, t[T_AW0_W], t[T_AW2_W]);
    if (has_vis) pack_mlp32_t(dst, LT_DS1, LT_DS2, t[T_VIS0_W], t[T_VIS2_W]);
    {   // prob_embed.0 (32 x 34): d h -> d f_ray (gathered) + vector rows d hit, d vis;  prob_embed.2 (32 x 32)
        const std::vector<float> w = transposed(t[T_PE0_W], 32, 34, 34);
        LayerMaps m; m.out_map = out_gathered32(0); in_dlayout(m.in_map, 0, 32, 2);
        pack_layer(dst, LT_PE1, w.data(), 32, nullptr, m);
        const int rows[2] = {32, 33};
        pack_vec(dst, LT_PE1, w.data(), 32, nullptr, rows, 0, 32);
        const std::vector<float> w2 = transposed(t[T_PE2_W], 32, 32, 32);
        LayerMaps m2; m2.out_map = out_natural(2, 32); in_dlayout(m2.in_map, 0, 32, 2);
        pack_layer(dst, LT_PE2, w2.data(), 32, nullptr, m2);
    }
    {   // ray_dir_fc.2 (35 x 16): input rows [3 + gathered img channel | rgb rows 0..2 as one single K-step] -> d h16
        const std::vector<float> w = transposed(t[T_RD2_W], 35, 16, 16);
        LayerMaps m; m.out_map = out_natural(1, 16); in_gathered32(m.in_map, 3);
        for (int g = 0; g < 4; ++g) m.in1_map.push_back(g < 3 ? g : -1);
        pack_layer(dst, LT_RD2, w.data(), 35, nullptr, m);
    }
    {   // neuray_fc.0 (8 x 32): d h8 -> d e
        const std::vector<float> w = transposed(t[T_NF0_W], 8, 32, 32);
        LayerMaps m; m.out_map = out_natural(2, 32); in_dlayout(m.in_map, 0, 8, 1);
        pack_layer(dst, LT_NF1, w.data(), 8, nullptr, m);
    }
    {   // base_fc.0 (64 x 207): per-view columns [140..142 rgb | 143..174 img (gathered) | 175..206 e], per-point 0..139
        const std::vector<float> w = transposed(t[T_BASE0_W], 64, 207, 207);
        LayerMaps m; m.out_map = out_gathered32(143); append(m.out_map, out_natural_at(175, 32, 2)); in_dlayout(m.in_map, 0, 64, 4);
        pack_layer(dst, LT_BV, w.data(), 64, nullptr, m);
        const int rgb[3] = {140, 141, 142};
        pack_vec(dst, LT_BV, w.data(), 64, nullptr, rgb, 0, 64);
        LayerMaps g; in_dlayout(g.in_map, 0, 64, 4);
        for (int j = 0; j < 4; ++j) append(g.out_map, out_gathered32(35 * j + 3));
        pack_layer(dst, LT_BG, w.data(), 64, nullptr, g);
        for (int j = 0; j < 4; ++j) {
            const int rows[3] = {35 * j, 35 * j + 1, 35 * j + 2};
            pack_vec(dst, LT_BG_R0 + j, w.data(), 64, nullptr, rows, 0, 64);
        }
        const std::vector<float> w2 = transposed(t[T_BASE2_W], 32, 64, 64);
        LayerMaps m2; m2.out_map = out_natural(4, 64); in_dlayout(m2.in_map, 0, 32, 2);
        pack_layer(dst, LT_B2, w2.data(), 32, nullptr, m2);
    }
    {   // vis_fc (32 x 32, 33 x 32: row 32 as a single K-step), vis_fc2.0 (32 x 32)
        const std::vector<float> w0 = transposed(t[T_VF0_W], 32, 32, 32), w2 = transposed(t[T_VF2_W], 33, 32, 32),
                                 v0 = transposed(t[T_V20_W], 32, 32, 32);
        LayerMaps m; m.out_map = out_natural(2, 32); in_dlayout(m.in_map, 0, 32, 2);
        pack_layer(dst, LT_VF1, w0.data(), 32, nullptr, m);
        LayerMaps m2; m2.out_map = out_natural(2, 32); in_dlayout(m2.in_map, 0, 32, 2);
        for (int g = 0; g < 4; ++g) m2.in1_map.push_back(g == 0 ? 32 : -1);
        pack_layer(dst, LT_VF2, w2.data(), 33, nullptr, m2);
        LayerMaps m3; m3.out_map = out_natural(2, 32); in_dlayout(m3.in_map, 0, 32, 2);
        pack_layer(dst, LT_V21, v0.data(), 32, nullptr, m3);
    }
    {   // rgb_fc.0 (16 x 37): d h16 -> d x2 (32) + vector row d vis (column 32);  rgb_fc.2 (8 x 16): d h8 -> d h16
        const std::vector<float> w = transposed(t[T_RF0_W], 16, 37, 37), w2 = transposed(t[T_RF2_W], 8, 16, 16);
        LayerMaps m; m.out_map = out_natural(2, 32); in_dlayout(m.in_map, 0, 16, 1);
        pack_layer(dst, LT_RF1, w.data(), 16, nullptr, m);
        const int row[1] = {32};
        pack_vec(dst, LT_RF1, w.data(), 16, nullptr, row, 0, 16);
        LayerMaps m2; m2.out_map = out_natural(1, 16); in_dlayout(m2.in_map, 0, 8, 1);
        pack_layer(dst, LT_RF2, w2.data(), 8, nullptr, m2);
    }
    {   // geometry_fc.0 (64 x 65): d h64 -> d [mean 32 | var 32] + vector row d mean weight;  geometry_fc.2 (16 x 64)
        const std::vector<float> w = transposed(t[T_GF0_W], 64, 65, 65), w2 = transposed(t[T_GF2_W], 16, 64, 64);
        LayerMaps m; m.out_map = out_natural(4, 64); in_dlayout(m.in_map, 0, 64, 4);
        pack_layer(dst, LT_GF1, w.data(), 64, nullptr, m);
        const int row[1] = {64};
        pack_vec(dst, LT_GF1, w.data(), 64, nullptr, row, 0, 64);
        LayerMaps m2; m2.out_map = out_natural(4, 64); in_dlayout(m2.in_map, 0, 16, 1);
        pack_layer(dst, LT_GF2, w2.data(), 16, nullptr, m2);
    }
    return 0;
}

// packed_t[i] = flat[index[i]] (index -1: padding, 0): the transposed layers carry the true weights, no factors
int pack_pass_t_index_map(bool has_vis, int* index) {
    std::vector<float> pos(kFlatPassFloats), tmp(kPackedTFloats);
    for (int i = 0; i < kFlatPassFloats; ++i) pos[i] = (float)(i + 1);        // < 2^24: exact
    const float* tp[T_COUNT];
    for (int t = 0; t < T_COUNT; ++t) {
        const bool vis_slot = t >= T_VIS0_W && t <= T_VIS4_B;
        tp[t] = (vis_slot && !has_vis) ? nullptr : pos.data() + tensor_offset(t);
    }
    g_pack_fp32_quads = true;
    const int rc = pack_pass_t_weights(tp, tmp.data());
    g_pack_fp32_quads = false;
    if (rc) return rc;
    for (int i = 0; i < kPackedTFloats; ++i) index[i] = (int)tmp[i] - 1;
    return 0;
}

// The packing is a gather with a per-element factor: packed[i] = flat[index[i]] * scale[i] (index -1: padding, 0).
// Obtained from the packer itself: once on tensors whose elements are their own flat position + 1 (scaling off), once
// on all-ones tensors (scaling on).
int pack_pass_index_map(bool has_vis, int* index, float* scale) {
    std::vector<float> pos(kFlatPassFloats), ones(kFlatPassFloats, 1.0f), tmp(kPackedPassFloats);
    for (int i = 0; i < kFlatPassFloats; ++i) pos[i] = (float)(i + 1);        // < 2^24: exact
    const float* tp[T_COUNT];
    const float* to[T_COUNT];
    for (int t = 0; t < T_COUNT; ++t) {
        const bool vis_slot = t >= T_VIS0_W && t <= T_VIS4_B;
        tp[t] = (vis_slot && !has_vis) ? nullptr : pos.data() + tensor_offset(t);
        to[t] = (vis_slot && !has_vis) ? nullptr : ones.data() + tensor_offset(t);
    }
    g_pack_unscaled = true;
    g_pack_fp32_quads = true;
    int rc = pack_pass_weights(tp, tmp.data());
    g_pack_unscaled = false;
    if (rc) { g_pack_fp32_quads = false; return rc; }
    for (int i = 0; i < kPackedPassFloats; ++i) index[i] = (int)tmp[i] - 1;
    rc = pack_pass_weights(to, tmp.data());
    g_pack_fp32_quads = false;
    if (rc) return rc;
    for (int i = 0; i < kPackedPassFloats; ++i) scale[i] = index[i] >= 0 ? tmp[i] : 0.0f;
    return 0;
}

// ---- np.random.shuffle of the reference's ray sampler, off the interpreter ------------------------------------------------------
// utils/base_utils.py:585-603 draws a training step's rays with two np.random.shuffle calls over the full pixel lists of the query
// image (640 000 entries at 800 x 800): 8 ms per step inside numpy, holding the interpreter lock.  The same permutation from the same
// generator state: numpy's legacy RandomState is MT19937 (key[624], pos), its shuffle is Fisher-Yates from the top - for i = n - 1 ...
// 1: j = random_interval(i), swap(x[i], x[j]) - and random_interval(max) masks 32-bit draws (64-bit above 2^32 - 1) to the smallest
// 2^k - 1 >= max and rejects values > max.  The caller hands over np.random.get_state()'s key / pos and puts them back afterwards.
namespace {
struct Mt19937 {
    unsigned int* key;
    int pos;
    void refill() {
        constexpr int N = 624, M = 397;
        constexpr unsigned int A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
        int k = 0;
        for (; k < N - M; ++k) { const unsigned int y = (key[k] & UP) | (key[k + 1] & LO); key[k] = key[k + M] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
        for (; k < N - 1; ++k) { const unsigned int y = (key[k] & UP) | (key[k + 1] & LO); key[k] = key[k + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
        const unsigned int y = (key[N - 1] & UP) | (key[0] & LO);
        key[N - 1] = key[M - 1] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        pos = 0;
    }
    unsigned int next32() {
        if (pos >= 624) refill();
        unsigned int y = key[pos++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    unsigned long long next64() { const unsigned long long hi = next32(); return (hi << 32) | next32(); }
    unsigned long long interval(unsigned long long max) {
        if (max == 0) return 0;
        unsigned long long mask = max, v;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        if (max <= 0xffffffffull) { while ((v = (next32() & mask)) > max) {} }
        else { while ((v = (next64() & mask)) > max) {} }
        return v;
    }
};
}  // namespace

int mt19937_shuffle(unsigned int* key, int* pos, void* data, long long n, int itemsize) {
    if (!key || !pos || *pos < 0 || *pos > 624 || (n > 0 && !data) || (itemsize != 4 && itemsize != 8)) return 1;
    Mt19937 g{key, *pos};
    if (itemsize == 4) {
        unsigned int* x = static_cast<unsigned int*>(data);
        for (long long i = n - 1; i >= 1; --i) { const long long j = (long long)g.interval((unsigned long long)i); const unsigned int t = x[j]; x[j] = x[i]; x[i] = t; }
    } else {
        unsigned long long* x = static_cast<unsigned long long*>(data);
        for (long long i = n - 1; i >= 1; --i) { const long long j = (long long)g.interval((unsigned long long)i); const unsigned long long t = x[j]; x[j] = x[i]; x[i] = t; }
    }
    *pos = g.pos;
    return 0;
}

}  // namespace nr
