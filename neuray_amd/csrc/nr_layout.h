// Packed-weight layout of one render pass, shared by the host packer (nr_pack.cpp) and the kernels.
//
// Every MLP layer of the per-ray path (reference: network/dist_decoder.py:64-97,
// network/aggregate_net.py:27-31, network/ibrnet.py:249-293) is stored as MFMA "A" fragments for
// v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand (M = output features) and the sample points
// as the N dimension, so that a layer's D registers are directly the next layer's B operands:
//
//   lane l = 16*g + c   (c = point column 0..15, g = lane group 0..3)
//   an activation vector with T tiles of 16 features lives in registers x[4*t + r]:
//        "D layout":      x[4t+r] (lane c,g)  = feature 16t + 4g + r of point c
//   a layer consumes a list of K-steps; K-step s takes one register per tile as the B operand and the
//   lane group g contributes input feature in_map(s, g).  K-steps are stored in quads ("KQ", one
//   float4 per lane = 4 K-steps) plus optional single K-steps ("K1", one float per lane) used for the
//   few scalar inputs (hit/vis, direction difference, rgb part).
//
//   w_quads [(mo*KQ + kq)*64 + lane]  float4 : component j = W[out_map(mo, c)][in_map(4kq+j, g)]
//   w_single[(mo*K1 + k1)*64 + lane]  float  :               W[out_map(mo, c)][in1_map(k1, g)]
//   bias    [mo*4 + g]                float4 : component r = bias[out_map(mo, 4g + r)]
//
// Rows/columns mapped to -1 are stored as 0, so padded features stay exactly 0 through ELU/ReLU.
#pragma once

namespace nr {

enum LayerId {
    // dist decoder heads (mean, var, aw, vis): 32 -> 32 -> 32 -> out                  dist_decoder.py:64-97
    // (order = LDS staging phases, see kPhase below)
    L_DM1, L_DM2,
    L_DFIN_M,    // vector rows: [mu0 mu1] pre-activations from h2(mean)
    L_DV1, L_DV2,
    L_DFIN_V,    // vector rows: [s0 s1] from h2(var)
    L_DA1, L_DA2,
    L_DFIN_A,    // vector row: [aw] from h2(aw)
    L_DS1, L_DS2,
    L_DFIN_S,    // vector row: [vis] from h2(vis)          (decoder with vis head only)
    L_PE1, L_PE2,            // prob_embed 34 -> 32 -> 32                           aggregate_net.py:27-31
    L_RD1, L_RD2,            // ray_dir_fc 4 -> 16 -> 35 (32 image rows on MFMA + 3 rgb vector rows)  ibrnet.py:249-252
    L_NF1, L_NF2,            // neuray_fc 32 -> 8 -> 1 (vector row)                 ibrnet.py:287-291
    L_BV0, L_BV1,            // base_fc.0 columns [rgb_feat neuray_feat] (per view), output rows 0..31 / 32..63  ibrnet.py:254
    L_B2,                    // base_fc.2 64 -> 32
    L_VF1, L_VF2,            // vis_fc 32 -> 32 -> 33 (32 MFMA rows + the visibility logit as a vector row)  ibrnet.py:259-263
    L_V21, L_V22,            // vis_fc2 32 -> 32 -> 1 (vector row)                  ibrnet.py:265-269
    L_RF1, L_RF2, L_RF3,     // rgb_fc 37 -> 16 -> 8 -> 1 (vector row)              ibrnet.py:281-285
    L_BG,                    // base_fc.0 columns [mean0 var0 mean1 var1] (per point; read from global by the owner waves)
    L_GF1, L_GF2,            // geometry_fc 65 -> 64 -> 16 (per point)              ibrnet.py:271-274
    L_FWD_COUNT,
    // ---- transposed layers of the backward pass (points_backward2_kernel): dX = W^T dY as one more layer in the same
    // fragment format, in a SECOND packed buffer (offsets restart at 0).  Input = the gradient w.r.t. the forward layer's
    // output rows in the D layout, output = the gradient w.r.t. its input columns in the register order the forward
    // pass holds them (natural D layout, or the gathered order "lane group g holds channels 8g..8g+7").  Unscaled (true
    // weights); extra scalar inputs / outputs of the forward layer appear as single K-steps / vector rows.
    LT_DM1 = L_FWD_COUNT, LT_DM2, LT_DV1, LT_DV2, LT_DA1, LT_DA2, LT_DS1, LT_DS2,   // dist heads: 32 -> 32 (x2)
    LT_PE1, LT_PE2,          // prob_embed: d h -> d [f_ray | hit, vis (vector rows)], d e -> d h
    LT_RD2,                  // ray_dir_fc.2: d [img rows (gathered) | rgb rows (single K-step)] -> d h16
    LT_NF1,                  // neuray_fc.0: d h8 -> d e
    LT_BV,                   // base_fc.0 per-view columns: d h64 -> d [gi (gathered) | e] (+ d rgb: vector rows)
    LT_B2,                   // base_fc.2: d x -> d h64
    LT_BG,                   // base_fc.0 per-point columns: sum_v d h64 -> d [mean0 var0 mean1 var1] image part (8 tiles)
    LT_BG_R0, LT_BG_R1, LT_BG_R2, LT_BG_R3,   // ... rgb part of statistic j (vector rows only)
    LT_VF1, LT_VF2,          // vis_fc (row 32 of vis_fc.2 enters as a single K-step)
    LT_V21,                  // vis_fc2.0
    LT_RF1, LT_RF2,          // rgb_fc.0: d h16 -> d x2 (+ d vis: vector row), rgb_fc.2: d h8 -> d h16
    LT_GF1, LT_GF2,          // geometry_fc.0: d h64 -> d [mean var] (+ d mean weight: vector row), geometry_fc.2: d G -> d h64
    L_COUNT
};

// ---- scaled ELU ------------------------------------------------------------------------------
// A hidden ELU layer whose output only feeds another linear layer computes L*ELU(y) with L = log2(e): its weights
// and bias are packed pre-multiplied by L, so the MFMA result is y' = L*y and
//     L*ELU(y) = median(y', fma(exp2(y'), L, -L), 0)
// is 3 VALU instructions (v_exp, v_fma, v_med3) instead of 4; the consuming layer's weights are packed divided by L
// (ELU -> ELU chains keep their weights untouched: L/L).  Layers whose ELU output is used by non-MFMA arithmetic
// (ray_dir_fc.2, base_fc.2, vis_fc.2, geometry_fc.2) keep the plain form.
constexpr double kLog2e = 1.4426950408889634074;
//                                     DM1 DM2 FM  DV1 DV2 FV  DA1 DA2 FA  DS1 DS2 FS  PE1 PE2 RD1 RD2 NF1 NF2 BV0 BV1 B2  VF1 VF2 V21 V22 RF1 RF2 RF3 BG  GF1 GF2
constexpr bool kOutScaledFwd[31] = {    1,  1,  0,  1,  1,  0,  1,  1,  0,  1,  1,  0,  0,  0,  1,  0,  1,  0,  1,  1,  0,  1,  0,  1,  0,  1,  1,  0,  1,  1,  0};
constexpr bool kInScaledFwd[31] = {     0,  1,  1,  0,  1,  1,  0,  1,  1,  0,  1,  1,  0,  0,  0,  1,  0,  1,  0,  0,  1,  0,  1,  0,  1,  0,  1,  1,  0,  0,  1};
static_assert(L_FWD_COUNT == 31, "kOutScaledFwd / kInScaledFwd follow the LayerId order");
// (the transposed layers carry true weights: never scaled)
struct ScaledFlags {
    bool v[L_COUNT];
    constexpr bool operator[](int l) const { return v[l]; }
};
constexpr ScaledFlags make_scaled(const bool (&f)[31]) {
    ScaledFlags s{};
    for (int i = 0; i < L_COUNT; ++i) s.v[i] = i < L_FWD_COUNT ? f[i] : false;
    return s;
}
constexpr ScaledFlags kOutScaled = make_scaled(kOutScaledFwd), kInScaled = make_scaled(kInScaledFwd);

struct LayerShape { int mt_out, kq, k1; };

constexpr LayerShape kShape[L_COUNT] = {
    {2, 2, 0}, {2, 2, 0}, {0, 0, 0},
    {2, 2, 0}, {2, 2, 0}, {0, 0, 0},
    {2, 2, 0}, {2, 2, 0}, {0, 0, 0},
    {2, 2, 0}, {2, 2, 0}, {0, 0, 0},
    {2, 2, 1}, {2, 2, 0},
    {1, 0, 1}, {2, 1, 0},
    {1, 2, 0}, {0, 0, 0},
    {2, 4, 1}, {2, 4, 1}, {2, 4, 0},
    {2, 2, 0}, {2, 2, 0},
    {2, 2, 0}, {0, 0, 0},
    {1, 2, 2}, {1, 1, 0}, {0, 0, 0},
    {4, 8, 4},
    {4, 4, 1}, {1, 4, 0},
    // transposed layers
    {2, 2, 0}, {2, 2, 0}, {2, 2, 0}, {2, 2, 0}, {2, 2, 0}, {2, 2, 0}, {2, 2, 0}, {2, 2, 0},
    {2, 2, 0}, {2, 2, 0},
    {1, 2, 1},
    {2, 1, 0},
    {4, 4, 0},
    {4, 2, 0},
    {8, 4, 0},
    {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0},
    {2, 2, 0}, {2, 2, 1},
    {2, 2, 0},
    {2, 1, 0}, {1, 1, 0},
    {4, 4, 0}, {4, 1, 0},
};

// ---- vector rows ---------------------------------------------------------------------------------
// Output rows that would waste most of a 16-row MFMA tile (the 1..4-wide output layers and the odd rows 32..34 of the
// 35- and 33-wide layers) are evaluated on the VALU instead: the input is in the D layout (lane (c, g) holds features
// 16t + 4g + r of point c), so each lane multiplies its 4*tiles features with per-lane-group weights and the four lane
// groups are summed with two permlane swaps; every lane group ends up with the full dot product.
//   n = output rows (<= 4), tiles = 16-feature input tiles.  Storage: [n][tiles][g][r] weights, then the biases
//   replicated per lane group, [g][4] (so that every load of a vector layer uses the same per-group address).
struct VecShape { int n, tiles; };
constexpr VecShape kVec[L_COUNT] = {
    {0, 0}, {0, 0}, {2, 2},
    {0, 0}, {0, 0}, {2, 2},
    {0, 0}, {0, 0}, {1, 2},
    {0, 0}, {0, 0}, {1, 2},
    {0, 0}, {0, 0},
    {0, 0}, {3, 1},
    {0, 0}, {1, 1},
    {0, 0}, {0, 0}, {0, 0},
    {0, 0}, {1, 2},
    {0, 0}, {1, 2},
    {0, 0}, {0, 0}, {1, 1},
    {0, 0},
    {0, 0}, {0, 0},
    // transposed layers
    {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0},
    {2, 2}, {0, 0},          // LT_PE1: d hit, d vis
    {0, 0},
    {0, 0},
    {3, 4},                  // LT_BV: d rgb_feat[0..2]
    {0, 0},
    {0, 0},
    {3, 4}, {3, 4}, {3, 4}, {3, 4},     // LT_BG_Rj: d (rgb part of statistic j)
    {0, 0}, {0, 0},
    {0, 0},
    {1, 1}, {0, 0},          // LT_RF1: d vis
    {1, 4}, {0, 0},          // LT_GF1: d mean weight
};

// ---- arithmetic of the MFMA layers ("AR") -----------------------------------------------------------------
// AR_F32: v_mfma_f32_16x16x4_f32 on fp32 operands; the layout described at the top of this file.
// AR_X3 (inference point kernel only): v_mfma_f32_16x16x32_bf16 on operands split THREE ways,
//   x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest: exact for an fp32 x), six products
//   hh, hm, mh, hl, lh, mm per K = 32 with fp32 accumulation; the dropped m l + l m + l l is below 2^-23 of the product.
//   A lane's 8 K-values of a K = 32 step are its 4 + 4 B registers of TWO consecutive quads (a "pair"), so the K order and the
//   D -> B register chaining are those of the fp32 layout.  A layer's quad fragments become, per output tile mo,
//       [pair kp = 0 .. KQ/2 - 1][part 0..2][lane] 16 bytes : 8 bf16 = part of W[out(mo, c)][in(8 kp + i, g)], i = 0..7
//                                                           (i < 4: component i of quad 2 kp, else component i - 4 of quad 2 kp + 1)
//       then, for an odd KQ, the last quad alone (v_mfma_f32_16x16x16_bf16): [part 0..2][lane] 8 bytes : 4 bf16
//   = KQ * 384 floats per tile instead of KQ * 256.  Singles (fp32 MFMA), biases and vector rows are stored as in the fp32 layout.
//   Always the folded network (prob_embed.2 inside its consumers): L_PE2 takes no space.
//   The per-point layers (L_BG, L_GF1, L_GF2: run by the owner waves between the cross-view all-reduces, 69 fp32 MFMAs per tile, bound by
//   the latency of their L2-resident fragments, not by the matrix pipe) keep the fp32 MFMA and the fp32 fragment format inside the X3
//   pack: splitting their operands would add VALU work and 1.5 x the fragment bytes to the tile's serial frame for no MFMA time won.
constexpr int AR_F32 = 0, AR_X3 = 1;
constexpr bool ar_omits(int l, int ar) { return ar == AR_X3 && l == L_PE2; }
constexpr bool ar_splits(int l, int ar) { return ar == AR_X3 && l != L_BG && l != L_GF1 && l != L_GF2; }

// sizes in floats
constexpr int quads_floats(int l, int ar = AR_F32) { return ar_omits(l, ar) ? 0 : kShape[l].mt_out * kShape[l].kq * 64 * (ar_splits(l, ar) ? 6 : 4); }
constexpr int single_floats(int l, int ar = AR_F32) { return ar_omits(l, ar) ? 0 : kShape[l].mt_out * kShape[l].k1 * 64; }
constexpr int bias_floats(int l, int ar = AR_F32) { return ar_omits(l, ar) ? 0 : kShape[l].mt_out * 16; }
constexpr int vec_floats(int l, int ar = AR_F32) { return (kVec[l].n > 0 && !ar_omits(l, ar)) ? kVec[l].n * kVec[l].tiles * 16 + 16 : 0; }
constexpr int layer_floats(int l, int ar = AR_F32) { return quads_floats(l, ar) + single_floats(l, ar) + bias_floats(l, ar) + vec_floats(l, ar); }
// floats of one output tile's quad fragments (the run-time tile index of the owner waves strides by this)
constexpr int tile_quads_floats(int l, int ar = AR_F32) { return kShape[l].kq * 64 * (ar_splits(l, ar) ? 6 : 4); }

// float offset of layer l inside its packed buffer: the forward layers in the pass buffer, the transposed layers in
// the second ("T") buffer, where the offsets restart at 0
constexpr int layer_offset(int l, int ar = AR_F32) {
    int off = 0;
    for (int i = (l >= L_FWD_COUNT ? (int)L_FWD_COUNT : 0); i < l; ++i) off += layer_floats(i, ar);
    return off;
}
constexpr int quads_offset(int l, int ar = AR_F32) { return layer_offset(l, ar); }
constexpr int single_offset(int l, int ar = AR_F32) { return layer_offset(l, ar) + quads_floats(l, ar); }
constexpr int bias_offset(int l, int ar = AR_F32) { return layer_offset(l, ar) + quads_floats(l, ar) + single_floats(l, ar); }
constexpr int vec_offset(int l, int ar = AR_F32) { return bias_offset(l, ar) + bias_floats(l, ar); }               // [n][tiles][16]
constexpr int vec_bias_offset(int l, int ar = AR_F32) { return vec_offset(l, ar) + kVec[l].n * kVec[l].tiles * 16; }

constexpr int packed_range_floats(int first, int last, int ar = AR_F32) {
    int off = 0;
    for (int i = first; i < last; ++i) off += layer_floats(i, ar);
    return off;
}
constexpr int kPackedPointFloats = packed_range_floats(0, L_FWD_COUNT);
constexpr int kPackedPointFloatsX3 = packed_range_floats(0, L_FWD_COUNT, AR_X3);
constexpr int kPackedTFloats = packed_range_floats(L_FWD_COUNT, L_COUNT);      // the transposed layers (backward pass)

// ---- LDS staging phases of the point kernel ------------------------------------------------------------
// Contiguous layer ranges [first, last] that are copied into LDS before they are used (the per-point layers L_BG,
// L_GF1, L_GF2 are read from global by their owner waves).  The weight stage is two regions of kStageRegionBytes:
// while the waves compute phase k out of one region, the LDS-DMA copy of phase k+1 (buffer_load ... lds, no VGPRs)
// lands in the other, so a phase costs one barrier and no exposed copy latency.  PH_DIST_S exists only for a decoder
// with a vis head.
enum PhaseId { PH_DIST_M, PH_DIST_VA, PH_DIST_S, PH_EMBED, PH_NF_BV0, PH_BV1, PH_B2_VF1, PH_TAIL, PH_COUNT };
struct PhaseRange { int first, last; };
constexpr PhaseRange kPhase[PH_COUNT] = {
    {L_DM1, L_DV1},          // mean head + its output rows, var.0
    {L_DV2, L_DFIN_A},       // var.2 + output rows, aw head
    {L_DS1, L_DFIN_S},       // vis head (HAS_VIS only)
    {L_PE1, L_RD2},          // prob_embed, ray_dir_fc
    {L_NF1, L_BV0},          // neuray_fc, base_fc.0 per-view rows 0..31
    {L_BV1, L_BV1},          // base_fc.0 per-view rows 32..63
    {L_B2, L_VF1},           // base_fc.2, vis_fc.0
    {L_VF2, L_RF3},          // vis_fc.2, vis_fc2, rgb_fc
};
constexpr int phase_begin(int ph, int ar = AR_F32) { return layer_offset(kPhase[ph].first, ar); }
constexpr int phase_floats(int ph, int ar = AR_F32) { return layer_offset(kPhase[ph].last + 1, ar) - layer_offset(kPhase[ph].first, ar); }
constexpr int max_phase_floats(int ar = AR_F32) {
    int m = 0;
    for (int i = 0; i < PH_COUNT; ++i) m = phase_floats(i, ar) > m ? phase_floats(i, ar) : m;
    return m;
}
constexpr int stage_region_bytes(int ar = AR_F32) { return (max_phase_floats(ar) * 4 + 1023) / 1024 * 1024; }   // DMA pieces are 1 KiB per wave
constexpr int weight_lds_floats(int ar = AR_F32) { return 2 * stage_region_bytes(ar) / 4; }
constexpr int kStageRegionBytes = stage_region_bytes();
constexpr int kWeightLdsFloats = weight_lds_floats();
// position of a phase in the sequence a kernel variant runs (PH_DIST_S is skipped without a vis head), its successor
constexpr int phase_count(bool vis) { return vis ? PH_COUNT : PH_COUNT - 1; }
constexpr int phase_seq(int ph, bool vis) { return (vis || ph < PH_DIST_S) ? ph : ph - 1; }
constexpr int phase_next(int ph, bool vis) {
    return ph == PH_COUNT - 1 ? 0 : ((!vis && ph + 1 == PH_DIST_S) ? ph + 2 : ph + 1);
}

// ---- weights of the ray kernel (attention + sigma head), plain row-major ------------------------
//   reference: network/ibrnet.py:52-102 (MultiHeadAttention 4 heads x d_k=4, no bias), :276-279
constexpr int RW_WQ = 0, RW_WK = 256, RW_WV = 512, RW_FC = 768, RW_LNW = 1024, RW_LNB = 1040,
              RW_OG0W = 1056, RW_OG0B = 1312, RW_OG2W = 1328, RW_OG2B = 1344, kPackedRayFloats = 1348;

constexpr int kPackedPassFloats = kPackedPointFloats + kPackedRayFloats;

// ---- per-point record written by the point kernel and read by the ray kernel -------------------
//   [0..15] geometry feature (geometry_fc output, before the positional encoding), [16..18] blended rgb,
//   [19] number of valid views
constexpr int kPointRec = 20;

// ---- per-view constants produced by the view-setup kernel ---------------------------------------
//   [0..11] H = K [R|t] (row major 3x4), [12..14] camera centre -R^T t, [15] -1/near, [16] -1/far
constexpr int kViewConst = 20;
constexpr int kMaxViews = 16;          // NEURAY_MAX_VIEWS of include/neuray_hip.h
// query constants: [0..8] K^-1, [9..20] pose 3x4, [21..23] centre, [24] -1/near, [25] -1/far
constexpr int kQueryConst = 28;

// tensor order of the `tensors` array of neuray_pack_pass_weights (host pointers, fp32, contiguous)
enum PassTensor {
    T_MEAN0_W, T_MEAN0_B, T_MEAN2_W, T_MEAN2_B, T_MEAN4_W, T_MEAN4_B,
    T_VAR0_W, T_VAR0_B, T_VAR2_W, T_VAR2_B, T_VAR4_W, T_VAR4_B,
    T_AW0_W, T_AW0_B, T_AW2_W, T_AW2_B, T_AW4_W, T_AW4_B,
    T_VIS0_W, T_VIS0_B, T_VIS2_W, T_VIS2_B, T_VIS4_W, T_VIS4_B,     // may be NULL (no vis head)
    T_PE0_W, T_PE0_B, T_PE2_W, T_PE2_B,
    T_RD0_W, T_RD0_B, T_RD2_W, T_RD2_B,
    T_BASE0_W, T_BASE0_B, T_BASE2_W, T_BASE2_B,
    T_VF0_W, T_VF0_B, T_VF2_W, T_VF2_B,
    T_V20_W, T_V20_B, T_V22_W, T_V22_B,
    T_GF0_W, T_GF0_B, T_GF2_W, T_GF2_B,
    T_WQ, T_WK, T_WV, T_FC, T_LN_W, T_LN_B,
    T_OG0_W, T_OG0_B, T_OG2_W, T_OG2_B,
    T_RF0_W, T_RF0_B, T_RF2_W, T_RF2_B, T_RF4_W, T_RF4_B,
    T_NF0_W, T_NF0_B, T_NF2_W, T_NF2_B,
    T_COUNT
};

// ---- flat natural layout (backward kernels): the T_COUNT tensors of a pass, row-major as in the state_dict, one
// after the other in PassTensor order (the vis-decoder slots are always present; zero without a vis head).  The
// gradient buffer of the backward kernels has the same layout.
constexpr int kTensorRows[T_COUNT] = {
    32, 32, 32, 32, 2, 2,   32, 32, 32, 32, 2, 2,   32, 32, 32, 32, 1, 1,   32, 32, 32, 32, 1, 1,
    32, 32, 32, 32,   16, 16, 35, 35,   64, 64, 32, 32,   32, 32, 33, 33,   32, 32, 1, 1,   64, 64, 16, 16,
    16, 16, 16, 16, 16, 16,   16, 16, 1, 1,   16, 16, 8, 8, 1, 1,   8, 8, 1, 1,
};
constexpr int kTensorCols[T_COUNT] = {      // 1 = bias / vector
    32, 1, 32, 1, 32, 1,   32, 1, 32, 1, 32, 1,   32, 1, 32, 1, 32, 1,   32, 1, 32, 1, 32, 1,
    34, 1, 32, 1,   4, 1, 16, 1,   207, 1, 64, 1,   32, 1, 32, 1,   32, 1, 32, 1,   65, 1, 64, 1,
    16, 16, 16, 16, 1, 1,   16, 1, 16, 1,   37, 1, 16, 1, 8, 1,   32, 1, 8, 1,
};
constexpr int tensor_floats(int t) { return kTensorRows[t] * kTensorCols[t]; }
constexpr int tensor_offset(int t) {
    int off = 0;
    for (int i = 0; i < t; ++i) off += tensor_floats(i);
    return off;
}
constexpr int kFlatPassFloats = tensor_offset(T_COUNT);

}  // namespace nr
