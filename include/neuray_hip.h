/* neuray_hip.h - C ABI of libneuray_hip.so: the MI355X-native (gfx950) NeuRay per-ray render path.
 *
 * The reference (liuyuan-pal/NeuRay) has no FFI layer: its "operator API" for this path is the Python
 * module surface network/render_ops.py + network/renderer.py.  Each entry point below
 * names the reference function(s) it replaces; neuray_amd/network/*.py binds them with ctypes and keeps
 * the reference's call surface on top (INTEGRATION.md shows the binding a maintainer would add).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; neuray_last_error() gives the message
 *   - "dev" pointers are device (HIP) pointers to contiguous fp32 unless stated; "host" pointers are CPU
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and never synchronise the host
 *   - all arrays are borrowed for the duration of the enqueued work; outputs are caller-allocated
 *   - a single query view per call (qn = 1), as in NeuralRayBaseRenderer.render (network/renderer.py:228-254)
 */
#ifndef NEURAY_HIP_H
#define NEURAY_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEURAY_ABI_VERSION 11
#define NEURAY_POINT_REC 20      /* floats per sample-point record (see neuray_render_points) */
#define NEURAY_VIEW_CONST 20     /* floats per reference-view constant block */
#define NEURAY_QUERY_CONST 28    /* floats of the query constant block */
#define NEURAY_PASS_TENSORS 68   /* length of the tensor array of neuray_pack_pass_weights */
#define NEURAY_DBG_FIELDS 16
#define NEURAY_MAX_VIEWS 16
#define NEURAY_MAX_SAMPLES 128

int neuray_abi_version(void);
const char* neuray_last_error(void);
/* 1 if the library was built for the GPU (hipcc, gfx950); 0 for the CPU test emulator build */
int neuray_is_device_build(void);
/* 32: the library computes with fp32 MFMA operands (the product, libneuray_hip.so).  16: the separately built and separately
 * reported bf16-operand variant (libneuray_hip_bf16.so: the quad K-steps of every MFMA layer take bf16 weights and
 * bf16-rounded activations, fp32 accumulation; inference only). */
int neuray_operand_precision(void);

/* ---- weights --------------------------------------------------------------------------------------
 * Packs the state_dict tensors of ONE pass (dist_decoder + agg_net, or fine_dist_decoder + fine_agg_net;
 * names and shapes: SURVEY.md Appendix B) into the per-lane MFMA fragment layout the kernels read.
 * `tensors` holds NEURAY_PASS_TENSORS host pointers in the order of enum nr::PassTensor (nr_layout.h):
 *   mean_decoder.{0,2,4}.{weight,bias}, var_decoder.*, aw_decoder.*, vis_decoder.* (6 NULLs when the decoder
 *   has no vis head), prob_embed.{0,2}.*, ray_dir_fc.{0,2}.*, base_fc.{0,2}.*, vis_fc.{0,2}.*, vis_fc2.{0,2}.*,
 *   geometry_fc.{0,2}.*, ray_attention.{w_qs,w_ks,w_vs,fc}.weight, ray_attention.layer_norm.{weight,bias},
 *   out_geometry_fc.{0,2}.*, rgb_fc.{0,2,4}.*, neuray_fc.{0,2}.*
 * Replaces: nn.Module parameter access of network/dist_decoder.py:64-97, network/aggregate_net.py:27-31,
 * network/ibrnet.py:249-293. */
size_t neuray_packed_pass_floats(void);
int neuray_pack_pass_weights(const float* const* tensors_host, float* packed_host);
/* The same pass with prob_embed.2 FOLDED into its consumers (inference packs).  aggregate_net.py:27-31: prob_embed is
 * Linear(34,32) -> ReLU -> Linear(32,32) with no activation behind the second Linear, and its output feeds only Linear layers:
 * neuray_fc.0 (ibrnet.py:287,337) and the last 32 columns of base_fc.0 (ibrnet.py:254,342-343).  With e = W2 h + b2:
 *   neuray_fc.0(e) = (Wn W2) h + (Wn b2 + bn),   base_fc.0(..., e) = ... + (Wb[:, 175:207] W2) h + (Wb[:, 175:207] b2 + bb)
 * so the 32 x 32 layer disappears (1,024 of the 19,504 MAC per (point, view)); the products are formed in double precision and
 * rounded once.  Same size and layout as neuray_pack_pass_weights (the prob_embed.2 slot is zero); pass NeurayPointsArgs.folded = 1
 * with it.  The backward kernels and the training forward (saved_dev) take the unfolded pack only. */
int neuray_pack_pass_weights_folded(const float* const* tensors_host, float* packed_host);
/* The same folded network for NEURAY_ARITH_X3: every weight of the MFMA fragments as three bf16 parts w = h + m + l (exact), laid out for
 * the K = 32 bf16 MFMA (csrc/nr_layout.h AR_X3).  packed_host: neuray_packed_points_floats_x3() floats; it holds the point kernel's layers
 * only - neuray_render_rays and every other entry point keep the neuray_pack_pass_weights[_folded] buffer. */
size_t neuray_packed_points_floats_x3(void);
int neuray_pack_pass_weights_x3(const float* const* tensors_host, float* packed_host);
/* The same packing as a gather, for callers that keep the weights on the device (training: they change every step):
 * packed[i] = flat[index[i]] * scale[i], index -1 = padding (0); flat = the flat natural layout described at
 * neuray_render_points_backward.  index_host / scale_host: neuray_packed_pass_floats() entries each. */
int neuray_pack_pass_index_map(int has_vis_head, int* index_host, float* scale_host);

/* ---- camera constants -------------------------------------------------------------------------------
 * view_const[v] = { H = K[R|t] (12), centre -R^T t (3), -1/near, -1/far, pad }   render_ops.py:94,110
 * query_const   = { K^-1 (9), pose (12), centre (3), -1/near, -1/far, near, far } render_ops.py:15-21
 * (K^-1 is supplied by the caller: the reference uses torch.inverse, render_ops.py:20). */
int neuray_setup_views(const float* poses_dev /*[n][3][4]*/, const float* Ks_dev /*[n][3][3]*/,
                       const float* depth_range_dev /*[n][2]*/, int n, float* view_const_dev, void* stream);
int neuray_setup_query(const float* pose_dev /*[3][4]*/, const float* Kinv_dev /*[3][3]*/,
                       const float* depth_range_dev /*[2]*/, float* query_const_dev, void* stream);

/* ---- map relayout: NCHW -> channels-last with c_pad >= c channels (zero filled) -------------------------
 * Feature maps are gathered with one contiguous 128-byte line per bilinear tap (32 ch) and images as
 * RGBA texels (c = 3, c_pad = 4).  Replaces the implicit NCHW layout of F.grid_sample, network/ops.py:32. */
int neuray_relayout_nhwc(const float* src_dev, float* dst_dev, int n, int c, int h, int w, int c_pad, void* stream);

/* ---- a1: sample_depth (network/render_ops.py:146-170, random_sample=False) -> depth [rn][dn] ------------ */
int neuray_sample_coarse_depth(const float* que_depth_range_dev /*[2]*/, int rn, int dn, float* depth_dev, void* stream);
/* the random_sample=True branch (render_ops.py:160-161): uniforms_dev [rn][dn-2] in [0,1), drawn by the caller with
 * torch.rand as the reference does; interior tick i becomes i + (u - 0.5) * 0.999.  NULL = neuray_sample_coarse_depth. */
int neuray_sample_coarse_depth_jittered(const float* depth_range_dev, const float* uniforms_dev, int rn, int dn, float* depth_dev,
                                        void* stream);

/* ---- a2-a14: one render pass over the sample points of a ray batch ------------------------------------------
 * Replaces, fused: depth2inv_dists, depth2points, project_points_dict (render_ops.py:27-52,82-144),
 * predict_proj_ray_prob, get_img_feats (renderer.py:67-83,127-135), DefaultAggregationNet.forward
 * (aggregate_net.py:34-68) and IBRNetWithNeuRay.forward up to geometry_fc and the colour blend
 * (ibrnet.py:315-355,362-367).
 * point_out[rn*dn][NEURAY_POINT_REC] = { geometry feature (16), blended rgb (3), number of valid views (1) }.
 * dbg (optional, may be NULL): [rn*dn][rfn][NEURAY_DBG_FIELDS] =
 *   { mask, u, v, z, hit_prob, vis, mu0, mu1, s0, s1, aw, vis_dec, sigmoid(neuray_fc), vis', vis'', rgb logit }. */
typedef struct NeurayPointsArgs {
    const float* query_const_dev;
    const float* view_const_dev;   /* [rfn][NEURAY_VIEW_CONST] */
    const float* coords_dev;       /* [rn][2] pixel (x,y) */
    const float* depth_dev;        /* [rn][dn] ascending sample depths */
    const float* ray_feats_nhwc_dev;   /* [rfn][fh][fw][32] */
    const float* img_feats_nhwc_dev;   /* [rfn][fh][fw][32] */
    const float* rgba_dev;             /* [rfn][h][w][4] */
    const float* packed_weights_dev;   /* neuray_pack_pass_weights output, uploaded */
    float* point_out_dev;
    float* dbg_dev;
    int rfn, rn, dn, h, w, fh, fw;
    int has_vis_head;    /* this pass's decoder has a vis_decoder */
    int use_vis;         /* the COARSE decoder's cfg['use_vis'] (renderer.py:75 uses it for both passes) */
    float var_bias;      /* dist_decoder cfg['bias_val'] (0.05) */
    int views_per_wave;  /* reference views processed by one wavefront: 0 = default (2), 1 or 2 */
    float* saved_dev;    /* NULL, or (training forward, rfn <= 8) [neuray_points_saved_floats(rn * dn)]: the cross-view quantities of
                          * every 16-point tile - base_fc.0's per-point part, the four weighted statistics, the mask sum, the dist
                          * decoder outputs (csrc/nr_kernels.h kSaved*) - which neuray_render_points_backward reads instead of
                          * recomputing them */
    int folded;          /* packed_weights_dev comes from neuray_pack_pass_weights_folded (not with saved_dev) */
    unsigned long long* slot_stats_dev;   /* NULL, or [2] counters the launch ADDS to: { (16-point tile, view) slots whose per-view layers
                          * ran, slots in total }.  The inference kernels skip a slot whose 16 (point, view) columns are all outside the
                          * view (mask = 0, render_ops.py:100-104,127-128): every quantity of such a column is multiplied by the mask
                          * downstream (ibrnet.py:333-349,365), so the result is the same; the counters give the executed share. */
    int arith;           /* arithmetic of the MLP contractions (the nn.Linear layers of dist_decoder.py:64-97, aggregate_net.py:27-31,
                          * ibrnet.py:249-293): NEURAY_ARITH_F32 = v_mfma_f32_16x16x4_f32 on fp32 operands; NEURAY_ARITH_X3 = every operand
                          * split exactly into three bf16 parts, six v_mfma_f32_16x16x32_bf16 products per K = 32, fp32 accumulation
                          * (each product within 2^-23 of exact: the fp32 grade, 2.5 x less matrix-pipe time).  X3: inference only,
                          * packed_weights_dev = the neuray_pack_pass_weights_x3 buffer, no saved_dev. */
} NeurayPointsArgs;
#define NEURAY_ARITH_F32 0
#define NEURAY_ARITH_X3 1
size_t neuray_points_saved_floats(int npts);
int neuray_render_points(const NeurayPointsArgs* args, void* stream);

/* ---- a14-a16: per-ray attention + sigma head + compositing ---------------------------------------------------
 * Replaces: `+ pos_encoding`, MultiHeadAttention, out_geometry_fc (ibrnet.py:356-360), network_rendering
 * (renderer.py:157-166), alpha_values2hit_prob (render_ops.py:72-80), ray_mask / render_depth
 * (renderer.py:195-202).  pos_enc [dn][16] is the sinusoid table of ibrnet.py:305-313. */
typedef struct NeurayRaysArgs {
    const float* point_rec_dev;     /* [rn][dn][NEURAY_POINT_REC] */
    const float* depth_dev;         /* [rn][dn] */
    const float* pos_enc_dev;       /* [dn][16] */
    const float* packed_weights_dev;
    float* hit_prob_dev;            /* [rn][dn] */
    float* pixel_dev;               /* [rn][3] */
    float* render_depth_dev;        /* [rn] or NULL */
    unsigned char* ray_mask_dev;    /* [rn] or NULL */
    float* density_dev;             /* [rn][dn] or NULL */
    int rn, dn, ray_mask_view_num, ray_mask_point_num;
    float* att_save_dev;            /* NULL, or (training forward) [rn][dn][NEURAY_RAY_ATT_SAVE]: softmax shift (4 heads), 1 / denominator
                                     * (4), attention output (16) of every sample, for neuray_render_rays_backward */
} NeurayRaysArgs;
int neuray_render_rays(const NeurayRaysArgs* args, void* stream);

/* ---- backward of neuray_render_rays (autograd of ibrnet.py:52-102,356-360, renderer.py:157-166, render_ops.py:72-80):
 * gradients of a scalar loss with respect to the per-point records and the ray-part weights, given the gradients of
 * pixel [rn][3], hit_prob [rn][dn] (NULL = none) and render_depth [rn] (NULL = none).  The forward is recomputed.
 * d_point_rec [rn][dn][NEURAY_POINT_REC]: [0..15] d geometry feature, [16..18] d blended colour, [19] = 0.
 * d_ray_weights [NEURAY_PACKED_RAY_FLOATS] is ACCUMULATED into (zero it first); layout = the ray part of the packed
 * pass, row-major copies at the NEURAY_RW_* offsets: ray_attention.w_qs / w_ks / w_vs / fc .weight (16x16 each),
 * layer_norm.weight / .bias (16), out_geometry_fc.0.weight (16x16) / .bias (16), out_geometry_fc.2.weight (16) / .bias.
 * dn <= NEURAY_MAX_SAMPLES: one wave per ray, one sample per lane up to 64 samples, two per lane above. */
#define NEURAY_PACKED_RAY_FLOATS 1348
#define NEURAY_RAY_ATT_SAVE 24
#define NEURAY_RW_WQ 0
#define NEURAY_RW_WK 256
#define NEURAY_RW_WV 512
#define NEURAY_RW_FC 768
#define NEURAY_RW_LNW 1024
#define NEURAY_RW_LNB 1040
#define NEURAY_RW_OG0W 1056
#define NEURAY_RW_OG0B 1312
#define NEURAY_RW_OG2W 1328
#define NEURAY_RW_OG2B 1344
typedef struct NeurayRaysBwdArgs {
    const float* point_rec_dev;       /* [rn][dn][NEURAY_POINT_REC] (forward input) */
    const float* depth_dev;           /* [rn][dn] */
    const float* pos_enc_dev;         /* [dn][16] */
    const float* packed_weights_dev;
    const float* d_pixel_dev;         /* [rn][3] */
    const float* d_hit_prob_dev;      /* [rn][dn] or NULL */
    const float* d_render_depth_dev;  /* [rn] or NULL */
    float* d_point_rec_dev;           /* [rn][dn][NEURAY_POINT_REC] */
    float* d_ray_weights_dev;         /* [NEURAY_PACKED_RAY_FLOATS], accumulated */
    int rn, dn;
    const float* att_saved_dev;       /* NULL (the attention forward is recomputed), or NeurayRaysArgs.att_save_dev of the same inputs */
} NeurayRaysBwdArgs;
int neuray_render_rays_backward(const NeurayRaysBwdArgs* args, void* stream);

/* ---- backward of neuray_render_points (autograd of dist_decoder.py:53-140, renderer.py:67-83,127-135,
 * aggregate_net.py:34-68, ibrnet.py:315-354,361-367): gradients of a scalar loss with respect to every weight of the pass
 * and to the ray_feats / img_feats maps, given d_point_rec (the output of neuray_render_rays_backward).
 * Weights and their gradients use the FLAT NATURAL layout: the NEURAY_PASS_TENSORS tensors, row-major as in the
 * state_dict, concatenated in the order of neuray_pack_pass_weights (vis-decoder slots always present; zeros without a
 * vis head); neuray_flat_pass_floats() floats, neuray_flat_tensor_offset(i) = start of tensor i.
 * d_flat, d_ray_feats_nhwc and d_img_feats_nhwc are ACCUMULATED into (zero them first).
 * The kernel (csrc/nr_kernels_bwd2.h; rfn <= 8 - no shipped configuration trains with more, dataset/train_dataset.py:73-74 - any dn):
 * 8 waves per 16-point tile, wave = reference view, activations and gradients chained in registers on the fp32 MFMA, weight gradients
 * accumulated in registers over the whole launch, run as TWO launches (the network's tail, then its front; each half keeps only its own
 * weight-gradient accumulators and chain state in registers, the tail hands 20 floats per (point, view) lane over through handover_dev).
 * Needs packed_weights_dev (the forward's packed weights, neuray_pack_pass_weights layout), packed_t_weights_dev (the transposed
 * layers: packed_t[i] = flat[index[i]] with the index map of neuray_pack_pass_t_index_map, neuray_packed_t_floats() floats), saved_dev
 * and handover_dev.  (Rounds 1-5 also carried a first-version kernel for rfn 9..16 and a one-launch form of this one; round 6 removed
 * both: ABI 10.) */
size_t neuray_flat_pass_floats(void);
size_t neuray_flat_tensor_offset(int tensor);
typedef struct NeurayPointsBwdArgs {
    const float* query_const_dev;
    const float* view_const_dev;
    const float* coords_dev;          /* [rn][2] */
    const float* depth_dev;           /* [rn][dn] */
    const float* ray_feats_nhwc_dev;  /* [rfn][fh][fw][32] */
    const float* img_feats_nhwc_dev;  /* [rfn][fh][fw][32] */
    const float* rgba_dev;            /* [rfn][h][w][4] */
    const float* flat_weights_dev;    /* [neuray_flat_pass_floats()] */
    const float* d_point_rec_dev;     /* [rn*dn][NEURAY_POINT_REC] */
    float* d_flat_weights_dev;        /* accumulated */
    float* d_ray_feats_nhwc_dev;      /* accumulated */
    float* d_img_feats_nhwc_dev;      /* accumulated */
    int rfn, rn, dn, h, w, fh, fw, has_vis_head, use_vis;
    float var_bias;
    const float* packed_weights_dev;   /* [neuray_packed_pass_floats()] */
    const float* packed_t_weights_dev; /* [neuray_packed_t_floats()] */
    const float* saved_dev;            /* what neuray_render_points left in NeurayPointsArgs.saved_dev for the same inputs */
    float* handover_dev;               /* neuray_points_backward_handover_floats(rn * dn) floats of scratch between the two launches */
} NeurayPointsBwdArgs;
size_t neuray_points_backward_handover_floats(int npoints);
size_t neuray_packed_t_floats(void);
/* index[neuray_packed_t_floats()] (host, int32): packed_t[i] = index[i] >= 0 ? flat[index[i]] : 0 */
int neuray_pack_pass_t_index_map(int has_vis_head, int* index_host);
/* float ranges [begin, end) of the quad fragments inside the packed pass buffer (transposed = 0) or the transposed pack (= 1), as
 * (begin, end) int pairs; returns MINUS the number of pairs (a positive value is an error).  The split library
 * (neuray_operand_precision() == 48) stores a quad's four weights as (hi, hi | lo, lo) bf16 pairs in the same 16 bytes: a device-side
 * packer gathers fp32 values with the index maps above and converts exactly these ranges. */
int neuray_packed_quad_ranges(int transposed, int* ranges_host, int max_pairs);
int neuray_render_points_backward(const NeurayPointsBwdArgs* args, void* stream);

/* ---- backward of the a19 path (renderer.py:137-155): hit_prob_self [rn][dn] as a function of the gathered query-view
 * features feats [rn][32] (neuray_interpolate_feats of que ray_feats) and the dist decoder weights.
 * -> d_feats [rn][32]; d_flat (flat natural layout, only the dist decoder tensors are touched) is ACCUMULATED into.
 * One wave per 16 rays, the decoder in registers on the packed and transposed packs (the scheme of neuray_render_points_backward):
 * packed_weights_dev [neuray_packed_pass_floats()], packed_t_weights_dev [neuray_packed_t_floats()].  Not in the bf16-operand build. */
int neuray_self_hit_prob_backward(const float* query_const_dev, const float* depth_dev, const float* feats_dev,
                                           const float* packed_weights_dev, const float* packed_t_weights_dev, int has_vis_head,
                                           int use_vis, float var_bias, const float* d_hit_dev, int rn, int dn, float* d_feats_dev,
                                           float* d_flat_weights_dev, void* stream);
/* ---- backward of neuray_dist_decoder_rows: gradients w.r.t. the decoder outputs (any of d_mean [n][2], d_var [n][2],
 * d_aw [n], d_vis [n] may be NULL: that head is skipped - predict_mean, renderer.py:280-316, needs the mean head only) -> d_feats [n][32];
 * d_flat (flat natural layout) ACCUMULATED.  One wave per 16 rows, the heads in registers on the packed / transposed packs of
 * neuray_pack_pass_weights / neuray_pack_pass_t_index_map, weight gradients on the MFMA. */
int neuray_dist_decoder_rows_backward(const float* feats_dev, const float* packed_weights_dev, const float* packed_t_dev, int n,
                                               int has_vis_head, float var_bias, const float* d_mean_dev, const float* d_var_dev,
                                               const float* d_aw_dev, const float* d_vis_dev, float* d_feats_dev, float* d_flat_dev,
                                               void* stream);
/* ---- backward of neuray_interpolate_feats: d_feats [b][c][fh][fw] += bilinear weights * d_out [b][n][c] (accumulated). */
int neuray_interpolate_feats_backward(const float* d_out_dev, const float* points_dev, const float* mask_dev, int b, int n, int c,
                                      int fh, int fw, int h_full, int w_full, int align_corners, float* d_feats_dev, void* stream);
/* The same through a channels-last staging map: the scatter adds a point's c channels to c CONSECUTIVE floats of a texel of
 * tmp_nhwc_zeroed_dev [b][fh][fw][c] (zeroed by the caller; one cache line per tap and wave instead of one per lane), then that map is
 * transposed and ADDED to d_feats_dev [b][c][fh][fw].  For many points per map (the generalisation renderer's 8192 depth-loss pixels per
 * view, renderer.py:280-316).  ABI 9. */
int neuray_interpolate_feats_backward_staged(const float* d_out_dev, const float* points_dev, const float* mask_dev, int b, int n, int c,
                                             int fh, int fw, int h_full, int w_full, int align_corners, float* tmp_nhwc_zeroed_dev,
                                             float* d_feats_dev, void* stream);

/* ---- a17: sample_fine_depth + torch.sort (render_ops.py:172-229, renderer.py:210-213) ------------------------
 * u_dev: externally drawn uniforms [rn][fdn] (training: the reference draws torch.rand on the CPU,
 * render_ops.py:205) or NULL for the deterministic stratified samples.  out [rn][fdn (+ dn if use_all)].
 * use_all: bit 0 = merge the coarse depths (fine_depth_use_all), bit 1 = skip the sort (the bare render_ops function),
 * bit 2 = inv_mode=False (interpolate the metric depths instead of the normalised inverse depths). */
int neuray_sample_fine_depth(const float* query_const_dev, const float* depth_dev, const float* hit_prob_dev,
                             const float* u_dev, int rn, int dn, int fdn, int use_all, float* out_dev, void* stream);
/* The same, also writing out what the samples were drawn from (tests; either pointer may be NULL): idx_out_dev int32 [rn][fdn] =
 * the `torch.searchsorted(cdf, u, right=True)` bin of every sample (render_ops.py:207) in the order of u, i.e. before the sort;
 * cdf_out_dev [rn][dn + 1] = the cdf (render_ops.py:193-196).  The pdf total is summed in numpy's pairwise order, so that on identical
 * inputs both equal the numpy oracle's bit for bit (tests/test_fine_index.py). */
int neuray_sample_fine_depth_traced(const float* query_const_dev, const float* depth_dev, const float* hit_prob_dev,
                                    const float* u_dev, int rn, int dn, int fdn, int use_all, float* out_dev, int* idx_out_dev,
                                    float* cdf_out_dev, void* stream);

/* ---- get_diff_feats of the depth init net (network/init_net.py:30-61, with depth2pts3d :13-28,
 * project_points_ref_views render_ops.py:117-130, interpolate_feats ops.py:14-34 and masked_mean_var ops.py:36-41 fused):
 * every pixel of every view lifted with its depth, projected into all rfn views, |rgb| and inverse-depth differences
 * reduced to masked mean / variance.  view_const_dev: neuray_setup_views; lift_const_dev [rfn][NEURAY_QUERY_CONST]:
 * neuray_setup_query of every view; rgbd_dev [rfn][h][w][4] = rgb + metric depth;
 * out_dev [rfn][h][w][8] = [rgb_mean 3, rgb_var 3, dpt_mean, dpt_var] (channels-last storage of the reference's [rfn,8,h,w]). */
int neuray_diff_feats(const float* view_const_dev, const float* lift_const_dev, const float* rgbd_dev, int rfn, int h, int w,
                      float* out_dev, void* stream);

/* ---- plane-sweep variance volume of the cost-volume init net (network/mvsnet/mvsnet.py:186-203
 * `construct_cost_volume_with_src` with `homo_warp`, network/mvsnet/modules.py:25-64, fused): for every reference view r,
 * depth plane d and feature pixel, the 32-channel features of the n_num source views nn_ids[r][j] (rows of src_feats,
 * all < sn) are read at the homography  transforms[r][j] = (src_proj_j @ inverse(ref_proj_r))[:3] (3x4 row-major,
 * computed by the caller as the reference does) - bilinear, zero padding, align_corners=True - and reduced together
 * with the reference's own feature to the per-channel variance.
 * ref_feats_dev [rfn][fh][fw][32], src_feats_dev [sn][fh][fw][32] (NHWC), depth_vals_dev [rfn][dn],
 * out_dev [rfn][32][dn][fh][fw]. */
int neuray_warp_variance(const float* ref_feats_dev, const float* src_feats_dev, const int* nn_ids_dev, const float* transforms_dev,
                         const float* depth_vals_dev, int rfn, int sn, int n_num, int dn, int fh, int fw, float* out_dev, void* stream);
/* The same with the output layout selectable: channels_last = 1 writes [rfn][dn][fh][fw][32] (a voxel's 32 channels in one 128-byte
 * line), the layout neuray_conv3d_c32_c8 reads. */
int neuray_warp_variance_layout(const float* ref_feats_dev, const float* src_feats_dev, const int* nn_ids_dev, const float* transforms_dev,
                                const float* depth_vals_dev, int rfn, int sn, int n_num, int dn, int fh, int fw, int channels_last,
                                float* out_dev, void* stream);

/* ---- f-3, MVSNet cost regularisation (network/mvsnet/mvsnet.py:29-69 CostRegNet, frozen / evaluation-only inside the cost-volume init
 * net, network/init_net.py:121-160): its first and last layer.
 * neuray_conv3d_c32_c8: `conv0` = leaky_relu(batch_norm(Conv3d(32, 8, 3, padding=1, bias=False)(x)), slope) with the frozen batch norm folded:
 *   x_ndhwc_dev [n][d][h][w][32] (channels-last), wpack_dev [3 kz][4 r][3 kx][2 q][64 lanes][4] = the folded weights as per-lane MFMA
 *   A fragments for TWO output rows per wave - row slot r = input row - first output row + 1, lane l (m = l & 15, g = l >> 4),
 *   component i: W'[m][8 g + 4 q + i][kz][r][kx] for m < 8 and r <= 2, W'[m - 8][8 g + 4 q + i][kz][r - 1][kx] for m >= 8 and r >= 1,
 *   else 0; W' = W * gamma / sqrt(var + eps) per output channel -, bias_dev [8] = beta - mean * gamma / sqrt(var + eps);
 *   out_dev [n][8][d][h][w].
 * neuray_conv3d_c8_c1: `prob` = Conv3d(8, 1, 3, padding=1): x_dev [n][8][d][h][w], w27_dev [3 ky][8 c][3 kz][3 kx] = W[0][c][kz][ky][kx] (the order
 *   the kernel walks them in; ABI 9 - [8][27] before), out_dev [n][d][h][w]. */
int neuray_conv3d_c32_c8(const float* x_ndhwc_dev, const float* wpack_dev, const float* bias_dev, float slope, int n, int d, int h, int w,
                         float* out_dev, void* stream);
int neuray_conv3d_c8_c1(const float* x_dev, const float* w27_dev, float bias, int n, int d, int h, int w, float* out_dev, void* stream);
/* neuray_convtranspose3d_c16_c8: the last decoder step, `c0 + conv11(x)` = skip + leaky_relu(batch_norm(ConvTranspose3d(16, 8, 3, stride=2,
 *   padding=1, output_padding=1, bias=False)(x)), slope) with the frozen batch norm folded (mvsnet.py:57-69): x_dev [n][16][d][h][w],
 *   wpack_dev [3 kz][3 ky][16 ci][8 co][3 kx] = W[ci][co][kz][ky][kx] * gamma[co] / sqrt(var[co] + eps), bias_dev [8], skip_dev
 *   [n][8][2d][2h][2w] or NULL, out_dev [n][8][2d][2h][2w]. */
int neuray_convtranspose3d_c16_c8(const float* x_dev, const float* wpack_dev, const float* bias_dev, float slope, const float* skip_dev,
                                  int n, int d, int h, int w, float* out_dev, void* stream);
/* neuray_convtranspose3d_bn_leaky: the same kernel for (C_in, C_out) = (16, 8) (conv11, as above) and (32, 16) (conv9: `c2 + conv9(x)`, one level
 *   down the decoder; mvsnet.py:57-69); other shapes return an error.  x_dev [n][C_in][d][h][w], wpack_dev [3 kz][3 ky][C_in][C_out][3 kx] (batch
 *   norm folded), bias_dev [C_out], skip_dev / out_dev [n][C_out][2d][2h][2w].  ABI 9. */
int neuray_convtranspose3d_bn_leaky(const float* x_dev, const float* wpack_dev, const float* bias_dev, float slope, const float* skip_dev,
                                    int n, int cin, int cout, int d, int h, int w, float* out_dev, void* stream);

/* neuray_conv3d_bn_leaky: the interior layers of the cost regularisation's encoder half, leaky_relu(batch_norm(Conv3d(C_in, C_out, 3, stride,
 *   padding=1, bias=False)(x)), slope) with the frozen batch norm folded - conv1 (8 -> 16, stride 2), conv2 (16 -> 16), conv3 (16 -> 32, stride 2),
 *   conv4 (32 -> 32) of network/mvsnet/mvsnet.py:29-69 (ConvBnReLU3D, modules.py:16-23) - and, on one-plane volumes (d = 1: the dz = 0 / 2 taps lie in
 *   the zero padding), the 3 x 3 stride-1 layers of the 2-D feature net (mvsnet.py:7-30: 3 -> 8, 8 -> 8, 16 -> 16, 32 -> 32).  Built for C_in rounded up
 *   to a multiple of 4 and C_out to a multiple of 16 in {(4 | 8 | 16, 16, s1), (32, 32, s1), (8, 16, s2), (16, 32, s2)}; wpack_dev / bias_dev are padded
 *   with zeros to those counts, x_dev / out_dev carry the true ones; other shapes return an error.  x_dev [n][C_in][d][h][w],
 *   wpack_dev [3 dz][C_in / 4 q][3 dy][3 dx][C_out / 16 mt][64 lanes] = per-lane MFMA A operands: lane l (m = l & 15, g = l >> 4) holds
 *   W[16 mt + m][4 q + g][dz][dy][dx] * gamma / sqrt(var + eps) of output channel 16 mt + m; bias_dev [C_out] = beta - mean * gamma / sqrt(var + eps);
 *   out_dev [n][C_out][(d - 1) / stride + 1][(h - 1) / stride + 1][(w - 1) / stride + 1].  ABI 9. */
int neuray_conv3d_bn_leaky(const float* x_dev, const float* wpack_dev, const float* bias_dev, float slope, int n, int cin, int cout, int stride,
                           int d, int h, int w, float* out_dev, void* stream);
/* ---- f-1: the 3 x 3 stride-1 convolutions of the per-image encoders at fp32 grade on the K = 32 bf16 MFMA (ABI 11).  Replaces
 * F.conv2d / nn.Conv2d(C_in, C_out, 3, padding_mode='reflect') on an input that already carries its reflection padding - the
 * BasicBlock / conv layers of ResUNetLight (network/ops.py:86-148,150-230), network/vis_encoder.py:6-21, the res_net of
 * network/init_net.py:13-61 - and, with pad = 2 and a transpose_flip pack, their data gradient.  Arithmetic: every weight and every
 * activation is split exactly into three bf16 parts, a product is the six MFMAs lh, hl, mm, mh, hm, hh with fp32 accumulation (the
 * dropped terms are below 2^-23 of the product): the AR_X3 arithmetic of neuray_render_points.
 * neuray_conv3x3_x3_pack_bytes: size of a pack, -1 unless C_in and C_out are multiples of 32.
 * neuray_conv3x3_x3_pack: w_dev [C_out][C_in][3][3] fp32 (the layer's weight) -> wpack_dev, the pack of the layer itself
 *   [9 taps][C_in / 32][C_out / 16][3 parts][64 lanes][4 dwords] (lane l = (m = l & 15, g = l >> 4), dword d: the bf16 pair of input channels
 *   32 kb + 8 g + 2 d, + 1 of output channel 16 mt + m), and / or wpack_t_dev, the pack of its DATA GRADIENT - the convolution from C_out to
 *   C_in channels with W'[i][o][dy][dx] = w[o][i][2 - dy][2 - dx], same size; either pointer may be NULL.  One launch.
 * neuray_conv3x3_x3: x_dev [n][C_in][h][w] NCHW fp32, `pad` rings of zeros around it (0: valid correlation of a pre-padded input;
 *   1: padding = 1 zeros; 2: full correlation), bias_dev [C_out] or NULL -> out_dev [n][C_out][h + 2 pad - 2][w + 2 pad - 2].
 *   The data gradient of a layer: x_dev = d_out, C_in / C_out exchanged, wpack_t_dev, pad = 2. */
long long neuray_conv3x3_x3_pack_bytes(int cin, int cout);
int neuray_conv3x3_x3_pack(const float* w_dev, int cout, int cin, void* wpack_dev, void* wpack_t_dev, void* stream);
int neuray_conv3x3_x3(const float* x_dev, const void* wpack_dev, const float* bias_dev, int n, int cin, int cout, int h, int w, int pad,
                      float* out_dev, void* stream);
/* neuray_conv3x3_x3_wrw: the WEIGHT gradient of the same layers in the same arithmetic, straight from the NCHW tensors (the contraction runs
 *   over positions, which NCHW stores contiguously: no transposed copies): dy_dev [n][C_out][hp - 2][wp - 2] (the gradient of the layer's
 *   output), xp_dev [n][C_in][hp][wp] (the layer's pre-padded input) -> dw_dev [C_out][C_in][3][3] (overwritten).  wp must be even.
 *   workspace_dev: neuray_conv3x3_x3_wrw_workspace_floats(...) floats (-1: shape not supported) - per-workgroup partial sums, added in a fixed
 *   order by a second launch (deterministic, no atomics).  torch.ops.aten.convolution_backward(..., output_mask=[False, True, False]). */
long long neuray_conv3x3_x3_wrw_workspace_floats(int n, int cin, int cout, int hp, int wp);
int neuray_conv3x3_x3_wrw(const float* dy_dev, const float* xp_dev, int n, int cin, int cout, int hp, int wp, float* workspace_dev, float* dw_dev,
                          void* stream);
/* neuray_scale_shift_leaky: MVSNet's frozen activated batch norm behind every convolution of the feature net and the cost regularisation
 *   (inplace_abn.ABN in evaluation mode; network/mvsnet/modules.py:7-23 `self.bn(self.conv(x))`, network/mvsnet/mvsnet.py:7-69) as ONE pass, in
 *   place on the convolution's output: x_dev [n][c][inner] (inner = h w or d h w) <- leaky_relu(x * scale_dev[c] + shift_dev[c], slope),
 *   scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale (the caller folds them). */
int neuray_scale_shift_leaky(float* x_dev, const float* scale_dev, const float* shift_dev, int n, int c, long long inner, float slope, void* stream);

/* ---- a7 standalone: interpolate_feats / interpolate_feature_map on NCHW maps (network/ops.py:14-34,
 * render_ops.py:54-70): bilinear, padding_mode='border'.  feats [b][c][fh][fw], points [b][n][2] pixel (x,y) in
 * units of the (w_full, h_full) image, mask [b][n] or NULL, out [b][n][c]. */
int neuray_interpolate_feats(const float* feats_dev, const float* points_dev, const float* mask_dev, int b, int n, int c,
                             int fh, int fw, int h_full, int w_full, int align_corners, float* out_dev, void* stream);

/* ---- stand-alone ops of the network.render_ops surface (same device code as the fused kernels) -----------------------
 * a2  coords2rays / depth2points (render_ops.py:4-39): centers, dirs [rn][3] (may be NULL); with pts != NULL also
 *     pts, que_dir [rn][dn][3] from depth [rn][dn]. */
int neuray_rays_points(const float* query_const_dev, const float* coords_dev, const float* depth_dev, int rn, int dn,
                       float* centers_dev, float* dirs_dev, float* pts_dev, float* que_dir_dev, void* stream);
/* a3  depth2dists (inverse = 0) / depth2inv_dists (inverse = 1, que_depth_range = [near, far]) (render_ops.py:41-52) */
int neuray_depth_dists(const float* depth_dev, const float* que_depth_range_dev, int inverse, int rows, int dn,
                       float* out_dev, void* stream);
/* a4-a6  project_points_ref_views (render_ops.py:82-130): pts [pn][3] -> dir [rfn][pn][3], pts2d [rfn][pn][2],
 *     depth [rfn][pn], mask [rfn][pn] (bytes 0/1; no z > 0 test, quirk A.9.1) */
int neuray_project_points(const float* view_const_dev, const float* pts_dev, int rfn, int pn, int h, int w, float* dir_dev,
                          float* pts2d_dev, float* depth_dev, unsigned char* mask_dev, void* stream);
/* a15 alpha_values2hit_prob (render_ops.py:72-80), rows x dn */
int neuray_alpha2hit_prob(const float* alpha_dev, int rows, int dn, float* out_dev, void* stream);

/* ---- a18 direct rendering, cfg['use_dr_prediction'] (renderer.py:85-125: predict_alpha_values_dr, predict_colors_dr,
 * direct_rendering; sph_solver.py:1-59: SphericalHarmonicsSolver.forward / predict).
 * neuray_direct_render_points, per sample point: alpha_dr [rn*dn] = the visibility-weighted mean of the views' alpha logits
 * (`ground` = cfg['alpha_value_ground_state'] where no view sees the point) and, unless color_dev is NULL
 * (cfg['use_nr_color_for_dr']), color_dr [rn*dn][3] = the weighted degree-3 spherical-harmonics fit of the views' colours over
 * their viewing directions, evaluated at the query direction.  view_rec_dev is NeurayPointsArgs.dbg_dev of the SAME pass
 * ([rn*dn][rfn][NEURAY_DBG_FIELDS]; fields 0 / 4 / 5 = mask / hit_prob / visibility per view), regs_dev [16] the solver's
 * `regs` buffer.
 * neuray_direct_render_rays, per ray: hit_prob_dr [rn][dn] = alpha_values2hit_prob(sigmoid(alpha_dr)), pixel_colors_dr [rn][3]
 * = sum_i hit_i colour_i with colour i at colors_dev[(ray * dn + i) * color_stride + color_first] (the SH colours: stride 3,
 * first 0; use_nr_color_for_dr: the point records, stride NEURAY_POINT_REC, first 16). */
int neuray_direct_render_points(const float* query_const_dev, const float* view_const_dev, const float* coords_dev,
                                const float* depth_dev, const float* rgba_dev, const float* view_rec_dev, const float* regs_dev,
                                int rfn, int rn, int dn, int h, int w, float ground, float* alpha_dev, float* color_dev, void* stream);
int neuray_direct_render_rays(const float* alpha_dev, const float* colors_dev, int color_stride, int color_first, int rn, int dn,
                              float* hit_prob_dev, float* pixel_dev, void* stream);

/* ---- f-1 encoders: fused InstanceNorm2d(affine=True, eps) + activation (+ residual add) + reflection padding, NCHW fp32
 * (network/ops.py:43-75,150-230 ResidualBlock / ResUNetLight: `conv -> norm -> relu [-> + skip -> relu]`, `conv -> norm -> elu`;
 * network/vis_encoder.py:6-21; the reflection padding is the `padding_mode='reflect'` of the NEXT convolution).
 * forward:  out_padded [n][c][h + 2 pad][w + 2 pad] = reflect_pad(act(gamma (x - mean) / sqrt(var + eps) + beta [+ res])), act 0 = none,
 *           1 = ReLU, 2 = ELU; res (NULL = none) is addressed with element strides (a view of another padded buffer);
 *           raw_zeroed [n*c][2] is scratch that must be zero on entry; stats [n*c][2] receives (mean, 1 / std) for the backward.
 *           out_stride_n: floats between consecutive images of out_padded (0 = c (h + 2 pad) (w + 2 pad), contiguous) - the output may
 *           be the leading channels of a wider buffer, e.g. the first half of a channel concatenation.
 * backward: d_out_padded is the gradient of the padded output (everything that consumed the padded tensor or its interior view);
 *           -> dx [n][c][h][w], d_res [n][c][h][w] (NULL = no residual).  raw_zeroed [n*c][2] (zero on entry) returns per plane
 *           (sum g, sum g xhat); d_gamma / d_beta [c] (both or neither; NULL = not wanted) receive their sums over the images: the
 *           gradients of the affine parameters.  out_padded / d_out_padded take image strides as the forward's output. */
int neuray_inorm_forward(const float* x_dev, const float* gamma_dev, const float* beta_dev, const float* res_dev, long long res_stride_n,
                         long long res_stride_c, long long res_stride_h, int n, int c, int h, int w, int pad, int act, float eps,
                         float* raw_zeroed_dev, float* stats_dev, float* out_padded_dev, long long out_stride_n, void* stream);
int neuray_inorm_backward(const float* x_dev, const float* out_padded_dev, long long out_stride_n, const float* d_out_padded_dev,
                          long long d_out_stride_n, const float* stats_dev, const float* gamma_dev, int n, int c, int h, int w, int pad,
                          int act, float* raw_zeroed_dev, float* dx_dev, float* d_res_dev, float* d_gamma_dev,
                          float* d_beta_dev, void* stream);

/* ---- f-1: bilinear x2 up-sampling + reflection padding of the image encoder's decoder half ------------------------------------
 * Replaces: F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) followed by the reflection padding of the next 3 x 3
 * convolution (network/ops.py:150-230, ResUNetLight.upconv3 / upconv2) and, in the backward, PyTorch's four float atomics per
 * output-gradient element.  x [planes][h][w] -> out [planes][2h + 2 pad][2w + 2 pad], pad 0 or 1; scale_y = fp32((h - 1) / (2h - 1)),
 * scale_x likewise (PyTorch's area_pixel_compute_scale): source index = scale * output index, truncated.
 * backward: a gather per input element over per-axis tables the caller builds with the same fp32 arithmetic - for input row y the
 * cnt_y[y] <= 8 padded output rows idx_y[y][k] that read it with weight wgt_y[y][k] (mirrored padding rows included), columns alike;
 * w <= 2047 (the combined rows of a workgroup are staged in LDS). */
int neuray_upsample2x_pad_forward(const float* x_dev, int planes, int h, int w, int pad, float scale_y, float scale_x,
                                  float* out_padded_dev, void* stream);
int neuray_upsample2x_pad_backward(const float* d_out_padded_dev, int planes, int h, int w, int pad, const int* cnt_y_dev,
                                   const int* idx_y_dev /*[h][8]*/, const float* wgt_y_dev /*[h][8]*/, const int* cnt_x_dev,
                                   const int* idx_x_dev /*[w][8]*/, const float* wgt_x_dev /*[w][8]*/, float* dx_dev, void* stream);

/* ---- a9 stand-alone: MixtureLogisticsDistDecoder.forward / predict_mean on arbitrary rows (dist_decoder.py:99-107,147-149).
 * feats [n][32] -> mean [n][2], var [n][2] (bias_val included), aw [n], vis [n] (vis only with a vis head, else NULL). */
int neuray_dist_decoder_rows(const float* feats_dev, const float* packed_weights_dev, int n, int has_vis_head, float var_bias,
                             float* mean_dev, float* var_dev, float* aw_dev, float* vis_dev, void* stream);
/* ---- a19: compute_prob(is_ref=False) of the query rays' own distributions (renderer.py:137-155, dist_decoder.py:39-46).
 * vis_dev NULL = the decoder's use_vis is False.  out [rn][dn]. */
int neuray_self_hit_prob(const float* query_const_dev, const float* depth_dev, const float* mean_dev, const float* var_dev,
                         const float* aw_dev, const float* vis_dev, int rn, int dn, float* out_dev, void* stream);

/* ---- f-4 host pipeline: the ray sampler's np.random.shuffle, off the interpreter -------------------------------------------
 * Replaces: the two `np.random.shuffle` calls of utils/base_utils.py:585-603 (sample_train_coords: a training step's rays, drawn over
 * the full pixel lists of the query image - 640 000 entries at 800 x 800).  Host-only: the permutation numpy's legacy RandomState
 * (MT19937) produces from the state { key[624], pos } (np.random.get_state()[1:3]), in place on a 1-D array of 4- or 8-byte items;
 * key / pos are advanced exactly as numpy advances them, so a seeded run draws the same rays.  Unlike numpy's it runs without the
 * interpreter lock (the host side of neuray_amd draws the NEXT step's rays on a worker thread while the current step is queued). */
int neuray_mt19937_shuffle(unsigned int* key624_host, int* pos_host, void* data_host, long long n, int itemsize);

/* ---- hardware self test of the MFMA operand layout the kernels assume (16x4 @ 4x16) ----------------------------- */
int neuray_mfma_selftest(const float* A_dev, const float* B_dev, float* D_dev, void* stream);
/* ---- hardware self test of the lane-group sum behind the vector rows (v_permlane16_swap / v_permlane32_swap):
 * y[l] = (x[c] + x[c+16]) + (x[c+32] + x[c+48]) with c = l % 16, for the 64 lanes of one wave. */
int neuray_group_sum_selftest(const float* x_dev, float* y_dev, void* stream);

/* ---- hardware self test of NEURAY_ARITH_X3: D [16][16] = A [16][32] @ B [32][16] through the point kernel's operand path - both operands
 * split into three bf16 parts on the device, six v_mfma_f32_16x16x32_bf16 products, fp32 accumulation.  parts_dev (may be NULL):
 * [3][16][32], the three parts of A as fp32 values (A = parts[0] + parts[1] + parts[2] exactly). */
int neuray_x3_selftest(const float* A_dev, const float* B_dev, float* D_dev, float* parts_dev, void* stream);
/* Workgroups of the point kernel (rfn = 7..8: four waves of two views, no vis head) the runtime keeps resident per compute unit for the given
 * NEURAY_ARITH_*: hipOccupancyMaxActiveBlocksPerMultiprocessor with the kernel's registers and LDS.  -1 = not built / not a device build. */
int neuray_points_resident_workgroups(int arith, int rfn);

#ifdef __cplusplus
}
#endif
#endif
