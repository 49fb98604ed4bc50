"""ctypes binding of include/neuray_hip.h.

The product path loads neuray_amd/libneuray_hip.so (hipcc, gfx950) and raises if it is missing:
there is NO CPU or eager-PyTorch fallback for the hot path.  `bind()` is also used by the CPU
test-suite to bind tests/emu/_build/libneuray_emu.so (the kernel sources compiled for a CPU fiber
emulator) - test infrastructure that the package itself never loads.
"""
import ctypes as C
import os

import torch  # noqa: F401  (first: the library must resolve libamdhip64 to the HIP runtime PyTorch already loaded -
#                      loaded before torch it binds the system copy and later launches find "no ROCm-capable device")

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NEURAY_HIP_LIB', os.path.join(HERE, 'libneuray_hip.so'))   # env override: A/B debugging only
BF16_LIB_PATH = os.path.join(HERE, 'libneuray_hip_bf16.so')     # the separately reported bf16-operand variant (inference only)
BF16X3_LIB_PATH = os.path.join(HERE, 'libneuray_hip_bf16x3.so') # the split variant: hi + lo bf16 operands, 3 bf16 MFMAs per fp32 quad (inference only)

PASS_TENSORS = 68
POINT_REC = 20
VIEW_CONST = 20
QUERY_CONST = 28
DBG_FIELDS = 16
MAX_VIEWS = 16
MAX_SAMPLES = 128
MAX_BACKWARD_SAMPLES = 128     # samples per ray and pass the backward kernels take (include/neuray_hip.h): = MAX_SAMPLES

ARITH_F32, ARITH_X3 = 0, 1      # NeurayPointsArgs.arith (include/neuray_hip.h NEURAY_ARITH_*)

c_float_p = C.POINTER(C.c_float)


class NeurayPointsArgs(C.Structure):
    _fields_ = [
        ('query_const_dev', C.c_void_p), ('view_const_dev', C.c_void_p), ('coords_dev', C.c_void_p),
        ('depth_dev', C.c_void_p), ('ray_feats_nhwc_dev', C.c_void_p), ('img_feats_nhwc_dev', C.c_void_p),
        ('rgba_dev', C.c_void_p), ('packed_weights_dev', C.c_void_p), ('point_out_dev', C.c_void_p),
        ('dbg_dev', C.c_void_p),
        ('rfn', C.c_int), ('rn', C.c_int), ('dn', C.c_int), ('h', C.c_int), ('w', C.c_int), ('fh', C.c_int),
        ('fw', C.c_int), ('has_vis_head', C.c_int), ('use_vis', C.c_int), ('var_bias', C.c_float),
        ('views_per_wave', C.c_int), ('saved_dev', C.c_void_p), ('folded', C.c_int), ('slot_stats_dev', C.c_void_p),
        ('arith', C.c_int),
    ]


class NeurayRaysArgs(C.Structure):
    _fields_ = [
        ('point_rec_dev', C.c_void_p), ('depth_dev', C.c_void_p), ('pos_enc_dev', C.c_void_p),
        ('packed_weights_dev', C.c_void_p), ('hit_prob_dev', C.c_void_p), ('pixel_dev', C.c_void_p),
        ('render_depth_dev', C.c_void_p), ('ray_mask_dev', C.c_void_p), ('density_dev', C.c_void_p),
        ('rn', C.c_int), ('dn', C.c_int), ('ray_mask_view_num', C.c_int), ('ray_mask_point_num', C.c_int),
        ('att_save_dev', C.c_void_p),
    ]


class NeurayRaysBwdArgs(C.Structure):
    _fields_ = [
        ('point_rec_dev', C.c_void_p), ('depth_dev', C.c_void_p), ('pos_enc_dev', C.c_void_p),
        ('packed_weights_dev', C.c_void_p), ('d_pixel_dev', C.c_void_p), ('d_hit_prob_dev', C.c_void_p),
        ('d_render_depth_dev', C.c_void_p), ('d_point_rec_dev', C.c_void_p), ('d_ray_weights_dev', C.c_void_p),
        ('rn', C.c_int), ('dn', C.c_int), ('att_saved_dev', C.c_void_p),
    ]


class NeurayPointsBwdArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'query_const_dev', 'view_const_dev', 'coords_dev', 'depth_dev', 'ray_feats_nhwc_dev', 'img_feats_nhwc_dev',
        'rgba_dev', 'flat_weights_dev', 'd_point_rec_dev', 'd_flat_weights_dev', 'd_ray_feats_nhwc_dev',
        'd_img_feats_nhwc_dev')] + \
        [(n, C.c_int) for n in ('rfn', 'rn', 'dn', 'h', 'w', 'fh', 'fw', 'has_vis_head', 'use_vis')] + \
        [('var_bias', C.c_float), ('packed_weights_dev', C.c_void_p), ('packed_t_weights_dev', C.c_void_p), ('saved_dev', C.c_void_p),
         ('handover_dev', C.c_void_p)]


PACKED_RAY_FLOATS = 1348
RAY_ATT_SAVE = 24            # NEURAY_RAY_ATT_SAVE
# (state_dict suffix under agg_net.agg_impl., offset, shape) of the ray-part weights inside d_ray_weights (include/neuray_hip.h)
RAY_WEIGHT_SLOTS = (
    ('ray_attention.w_qs.weight', 0, (16, 16)), ('ray_attention.w_ks.weight', 256, (16, 16)),
    ('ray_attention.w_vs.weight', 512, (16, 16)), ('ray_attention.fc.weight', 768, (16, 16)),
    ('ray_attention.layer_norm.weight', 1024, (16,)), ('ray_attention.layer_norm.bias', 1040, (16,)),
    ('out_geometry_fc.0.weight', 1056, (16, 16)), ('out_geometry_fc.0.bias', 1312, (16,)),
    ('out_geometry_fc.2.weight', 1328, (1, 16)), ('out_geometry_fc.2.bias', 1344, (1,)),
)


# every symbol include/neuray_hip.h declares: (restype, argtypes)
SYMBOLS = {
    'neuray_abi_version': (C.c_int, []),
    'neuray_last_error': (C.c_char_p, []),
    'neuray_is_device_build': (C.c_int, []),
    'neuray_operand_precision': (C.c_int, []),
    'neuray_packed_pass_floats': (C.c_size_t, []),
    'neuray_points_saved_floats': (C.c_size_t, [C.c_int]),
    'neuray_pack_pass_weights': (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    'neuray_pack_pass_weights_folded': (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    'neuray_packed_points_floats_x3': (C.c_size_t, []),
    'neuray_pack_pass_weights_x3': (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    'neuray_pack_pass_index_map': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_mt19937_shuffle': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_longlong, C.c_int]),
    'neuray_setup_views': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_setup_query': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_relayout_nhwc': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'neuray_sample_coarse_depth': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_sample_coarse_depth_jittered': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_render_points': (C.c_int, [C.POINTER(NeurayPointsArgs), C.c_void_p]),
    'neuray_render_rays': (C.c_int, [C.POINTER(NeurayRaysArgs), C.c_void_p]),
    'neuray_sample_fine_depth': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_sample_fine_depth_traced': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_interpolate_feats': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_rays_points': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    'neuray_depth_dists': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_project_points': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_alpha2hit_prob': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_inorm_forward': (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] * 3 + [C.c_int] * 6 + [C.c_float] + [C.c_void_p] * 3 + [C.c_longlong, C.c_void_p]),
    'neuray_inorm_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p] * 6),
    'neuray_upsample2x_pad_forward': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'neuray_upsample2x_pad_backward': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_direct_render_points': (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_direct_render_rays': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_dist_decoder_rows': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    'neuray_self_hit_prob': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    'neuray_mfma_selftest': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_x3_selftest': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_points_resident_workgroups': (C.c_int, [C.c_int, C.c_int]),
    'neuray_render_rays_backward': (C.c_int, [C.POINTER(NeurayRaysBwdArgs), C.c_void_p]),
    'neuray_flat_pass_floats': (C.c_size_t, []),
    'neuray_flat_tensor_offset': (C.c_size_t, [C.c_int]),
    'neuray_packed_t_floats': (C.c_size_t, []),
    'neuray_pack_pass_t_index_map': (C.c_int, [C.c_int, C.c_void_p]),
    'neuray_points_backward_handover_floats': (C.c_size_t, [C.c_int]),
    'neuray_packed_quad_ranges': (C.c_int, [C.c_int, C.c_void_p, C.c_int]),
    'neuray_render_points_backward': (C.c_int, [C.POINTER(NeurayPointsBwdArgs), C.c_void_p]),
    'neuray_self_hit_prob_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                         C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_dist_decoder_rows_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_interpolate_feats_backward_staged': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                         C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_interpolate_feats_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_group_sum_selftest': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_warp_variance': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_warp_variance_layout': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_conv3d_c32_c8': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_convtranspose3d_bn_leaky': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_conv3d_bn_leaky': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    'neuray_conv3x3_x3_pack_bytes': (C.c_longlong, [C.c_int, C.c_int]),
    'neuray_conv3x3_x3_pack': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_conv3x3_x3': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_conv3x3_x3_wrw_workspace_floats': (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'neuray_conv3x3_x3_wrw': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'neuray_scale_shift_leaky': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_float, C.c_void_p]),
    'neuray_conv3d_c8_c1': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_convtranspose3d_c16_c8': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'neuray_diff_feats': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
}


class NeurayLibError(RuntimeError):
    pass


def bind(path):
    """dlopen `path` and attach prototypes for every symbol of the C ABI (raises if one is missing)."""
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


_LIBS = {}


def load(variant='fp32'):
    """The product library (HIP, gfx950), or with variant='bf16' the bf16-operand build.  Raises loudly if it has not
    been built."""
    if variant not in _LIBS:
        path = {'fp32': LIB_PATH, 'bf16': BF16_LIB_PATH, 'bf16x3': BF16X3_LIB_PATH}[variant]
        if not os.path.exists(path):
            raise NeurayLibError(
                "neuray_amd: %s not found - build it with `python -m neuray_amd.build` (hipcc, gfx950). "
                "There is no CPU/eager fallback for the render path." % path)
        lib = bind(path)
        if lib.neuray_is_device_build() != 1:
            raise NeurayLibError("neuray_amd: %s is not a device build" % path)
        if lib.neuray_operand_precision() != {'fp32': 32, 'bf16': 16, 'bf16x3': 48}[variant]:
            raise NeurayLibError("neuray_amd: %s is not the %s build" % (path, variant))
        _LIBS[variant] = lib
    return _LIBS[variant]


def check(lib, rc):
    if rc != 0:
        raise RuntimeError("neuray_hip: " + lib.neuray_last_error().decode('utf-8', 'replace'))
