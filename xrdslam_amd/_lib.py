"""ctypes binding of the C-ABI shared library (include/xrdslam_hip.h).

The product path has NO CPU fallback: if ``libxrdslam_hip.so`` is missing or a
kernel launch fails, an exception is raised.  Build with
``python -m xrdslam_amd.build`` (hipcc, gfx950).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libxrdslam_hip.so')

_lib = None

c_float_p = C.c_void_p  # raw device pointers are passed as integers
i32, i64, f32, f64, vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


class XrdError(RuntimeError):
    pass


class NiceScene(C.Structure):
    """mirror of ``xrd_nice_scene``"""
    _fields_ = [('bound', f64 * 6), ('grid', vp * 4), ('gdim', i32 * 12),
                ('dec', vp * 4), ('gmask', vp * 4), ('n_samples', i32), ('n_surface', i32),
                ('t_uniform', vp), ('t_surface', vp),
                ('coarse_enlarge', f64)]


ADAM_MAX_SETS = 4   # XRD_ADAM_MAX_SETS


class AdamCellsSet(C.Structure):
    """xrd_adam_cells_set (include/xrdslam_hip.h)"""
    _fields_ = [('param', C.c_void_p), ('grad', C.c_void_p),
                ('m', C.c_void_p), ('v', C.c_void_p),
                ('cell_idx', C.c_void_p), ('n_cells', C.c_int64),
                ('lr', C.c_float), ('step_ticket', C.c_void_p),
                ('n_cells_dev', C.c_void_p)]


ADAM_DENSE_MAX_SETS = 8   # XRD_ADAM_DENSE_MAX_SETS


class AdamDenseSet(C.Structure):
    """xrd_adam_dense_set (include/xrdslam_hip.h)"""
    _fields_ = [('param', C.c_void_p), ('grad', C.c_void_p),
                ('m', C.c_void_p), ('v', C.c_void_p), ('n', C.c_int64),
                ('lr', C.c_float), ('weight_decay', C.c_float),
                ('step_ticket', C.c_void_p), ('advance', C.c_int32)]


class CoslamScene(C.Structure):
    """mirror of ``xrd_coslam_scene``"""
    _fields_ = [('bound', f64 * 6), ('lv_scale', f32 * 16),
                ('lv_res', C.c_uint32 * 16), ('lv_size', C.c_uint32 * 16),
                ('lv_offset', C.c_uint32 * 16), ('table', vp), ('pack', vp),
                ('t_near', vp), ('t_far', vp), ('t_uniform', vp),
                ('n_range_d', i32), ('n_sample_d', i32), ('perturb', i32),
                ('white_bkgd', i32), ('trunc', f32), ('sc_factor', f32)]


_SIGS = {
    'xrd_abi_version': (C.c_int, []),
    'xrd_last_error': (C.c_char_p, []),
    'xrd_nice_flat_len': (C.c_int, [C.c_int]),
    'xrd_nice_pack_len': (C.c_int, [C.c_int]),
    'xrd_nice_pack_index': (C.c_int, [C.c_int, vp]),
    'xrd_nice_render_fwd': (C.c_int, [C.POINTER(NiceScene), C.c_int, C.c_int,
                                      vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'xrd_nice_eval_points': (C.c_int, [C.POINTER(NiceScene), C.c_int, i64, vp,
                                       vp, vp]),
    'xrd_point_geo_fwd': (C.c_int, [i64] + [vp] * 7 + [f32, C.c_int] +
                          [vp] * 6),
    'xrd_point_geo_bwd': (C.c_int, [i64] + [vp] * 7 + [f32, C.c_int] +
                          [vp] * 7),
    'xrd_point_color_flat_len': (C.c_int, []),
    'xrd_point_color_grad_len': (C.c_int, []),
    'xrd_point_color_pack_len': (C.c_int, []),
    'xrd_point_color_pack_index': (C.c_int, [vp]),
    'xrd_point_color_ops_floats': (i64, [i64]),
    'xrd_point_color_ws_floats': (i64, []),
    'xrd_point_color_fwd': (C.c_int, [i64] + [vp] * 6 + [f32, C.c_int] +
                            [vp] * 7),
    'xrd_point_color_bwd': (C.c_int, [i64] + [vp] * 6 + [f32, C.c_int] +
                            [vp] * 12),
    'xrd_point_track_loss': (C.c_int, [C.c_int] * 3 + [f32] + [vp] * 10),
    'xrd_point_batch': (C.c_int, [C.c_int, C.c_int] + [vp] * 5 +
                        [C.c_int] * 5 + [i64, f32, f32] + [vp] * 7),
    'xrd_point_map_loss': (C.c_int, [C.c_int, C.c_int] + [vp] * 6 +
                           [f32, f32, C.c_int, vp, vp, vp]),
    'xrd_point_composite_fwd': (C.c_int, [C.c_int, C.c_int, vp, C.c_int, vp,
                                          C.c_int, vp, vp, f32, vp, vp, vp,
                                          vp]),
    'xrd_point_composite_bwd': (C.c_int, [C.c_int, C.c_int, vp, C.c_int, vp,
                                          C.c_int, vp, vp, f32, vp, vp, vp,
                                          vp, C.c_int, vp, C.c_int, vp]),
    'xrd_point_render_scratch_floats': (i64, [i64]),
    'xrd_point_render_fwd': (C.c_int, [C.c_int, C.c_int] + [vp] * 8 +
                             [f32, C.c_int] + [vp] * 5 + [f32] + [vp] * 11),
    'xrd_point_render_bwd': (C.c_int, [C.c_int, C.c_int] + [vp] * 8 +
                             [f32, C.c_int] + [vp] * 4 + [f32] + [vp] * 18),
    'xrd_nice_bwd_ws_floats': (i64, [C.c_int]),
    'xrd_nice_fwd_masks_words': (i64, [C.c_int]),
    'xrd_nice_render_fwd_masks': (C.c_int, [C.POINTER(NiceScene), C.c_int,
                                            C.c_int] + [vp] * 10),
    'xrd_nice_render_bwd_masks': (C.c_int, [C.POINTER(NiceScene), C.c_int,
                                            C.c_int] + [vp] * 13),
    'xrd_nice_coarse_ws_floats': (i64, [C.POINTER(NiceScene)]),
    'xrd_nice_render_bwd': (C.c_int, [C.POINTER(NiceScene), C.c_int, C.c_int,
                                      vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                      C.POINTER(vp * 4), C.POINTER(vp * 4), vp,
                                      vp]),
    'xrd_adam_cells': (C.c_int, [vp, vp, vp, vp, vp, i64, C.c_int, f32, f32,
                                 f32, f32, C.c_int, C.c_int, vp]),
    'xrd_adam_cells_devstep': (C.c_int, [vp, vp, vp, vp, vp, i64, C.c_int, f32,
                                         f32, f32, f32, vp, C.c_int, vp]),
    'xrd_adam_cells_devcount': (C.c_int, [vp, vp, vp, vp, vp, i64, C.c_int,
                                          f32, f32, f32, f32, vp, vp, C.c_int,
                                          vp]),
    'xrd_adam_cells_tick': (C.c_int, [vp, vp, vp, vp, vp, i64, C.c_int, f32,
                                      f32, f32, f32, vp, vp, C.c_int, vp]),
    'xrd_adam_cells_multi': (C.c_int, [C.c_int, C.POINTER(AdamCellsSet),
                                       C.c_int, f32, f32, f32, C.c_int, vp]),
    'xrd_nice_warmup': (C.c_int, []),
    'xrd_nice_map_ws_floats': (i64, [C.POINTER(NiceScene), C.c_int, C.c_int]),
    'xrd_nice_map_iter': (C.c_int, [C.POINTER(NiceScene), C.c_int, C.c_int,
                                    vp, vp, vp, vp, vp, vp, f32, vp, vp,
                                    C.POINTER(vp * 4), vp, vp, vp, vp]),
    'xrd_nice_map_iter_export': (C.c_int, [C.POINTER(NiceScene), C.c_int,
                                           C.c_int, vp, vp, vp, vp, vp, vp,
                                           f32, vp, vp, C.POINTER(vp * 4),
                                           vp, vp, vp, vp, vp, vp]),
    'xrd_nice_track_ws_floats': (i64, [C.c_int]),
    'xrd_nice_track_iter': (C.c_int, [C.POINTER(NiceScene), C.c_int, vp, vp,
                                      vp, vp, vp, vp, C.c_int, C.c_int, f32,
                                      vp, vp, vp, vp, vp]),
    'xrd_nice_map_warmup': (C.c_int, []),
    'xrd_nice_frustum_cells': (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_int,
                                         C.c_int, f32, f32, f32, f32, vp, vp,
                                         vp, vp, vp, vp]),
    'xrd_hashgrid_levels': (C.c_int, [C.c_int, C.c_int, f32, C.c_int, C.c_int,
                                      vp, vp, vp, vp, vp]),
    'xrd_hashgrid_fwd': (C.c_int, [C.c_int, vp, vp, vp, vp, i64, vp, vp, vp,
                                   vp]),
    'xrd_hashgrid_bwd': (C.c_int, [C.c_int, vp, vp, vp, vp, i64, vp, vp, vp,
                                   vp, vp, vp]),
    'xrd_hashgrid_tv': (C.c_int, [C.c_int, vp, vp, vp, vp, vp, C.c_int, vp,
                                  f64, f64, vp, vp, f32, vp, vp, vp, vp, vp]),
    'xrd_oneblob_fwd': (C.c_int, [i64, C.c_int, C.c_int, vp, vp, vp]),
    'xrd_oneblob_bwd': (C.c_int, [i64, C.c_int, C.c_int, vp, vp, vp, vp]),
    'xrd_octree_create': (vp, [C.c_int, C.c_int, f64]),
    'xrd_octree_destroy': (None, [vp]),
    'xrd_octree_reset_id_counter': (None, []),
    'xrd_octree_insert': (C.c_int, [vp, vp, i64, vp]),
    'xrd_octree_try_insert': (f64, [vp, vp, i64]),
    'xrd_octree_has_voxel': (C.c_int, [vp, vp]),
    'xrd_octree_count_nodes': (i64, [vp]),
    'xrd_octree_count_leaf_nodes': (i64, [vp]),
    'xrd_octree_export': (C.c_int, [vp, vp, vp, vp]),
    'xrd_octree_get_voxels': (i64, [vp, vp, i64]),
    'xrd_octree_get_leaf_voxels': (i64, [vp, vp, i64]),
    'xrd_svo_intersect': (C.c_int, [C.c_int, C.c_int, C.c_int, f32, C.c_int,
                                    C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                    vp]),
    'xrd_inverse_cdf_sampling': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int,
                                           f32, vp, vp, vp, vp, vp, vp, vp,
                                           vp, vp, vp]),
    'xrd_vox_flat_len': (C.c_int, []),
    'xrd_vox_pack_len': (C.c_int, []),
    'xrd_vox_pack_index': (C.c_int, [vp]),
    'xrd_vox_points_fwd': (C.c_int, [i64, vp, vp, vp, vp, vp, f32] +
                           [vp] * 11),
    'xrd_vox_points_bwd': (C.c_int, [i64, vp, vp, vp, vp, vp, f32] +
                           [vp] * 14),
    'xrd_vox_dw_ws_floats': (i64, []),
    'xrd_vox_dw': (C.c_int, [i64] + [vp] * 14),
    'xrd_vox_meta_len': (C.c_int, []),
    'xrd_vox_sample_rays': (C.c_int, [C.c_int, C.c_int, C.c_int, i64, C.c_int,
                                      vp, vp, f32, f32, f32, f32, f32] +
                            [vp] * 21),
    'xrd_vox_sample_rays_shard': (C.c_int, [C.c_int, C.c_int, C.c_int, i64,
                                            C.c_int, vp, vp, f32, f32, f32,
                                            f32, f32] + [vp] * 22),
    'xrd_vox_render_fwd': (C.c_int, [C.c_int, C.c_int, i64, f32, f32] +
                           [vp] * 14 + [f32] * 4 + [vp] * 3),
    'xrd_vox_render_bwd': (C.c_int, [C.c_int, C.c_int, i64, f32, f32] +
                           [vp] * 14),
    'xrd_vox_ray_grads': (C.c_int, [C.c_int, C.c_int, i64] + [vp] * 8),
    'xrd_comm_load': (C.c_int, [C.c_char_p]),
    'xrd_comm_unique_id_bytes': (C.c_int, []),
    'xrd_comm_unique_id': (C.c_int, [vp]),
    'xrd_comm_create': (vp, [vp, C.c_int, C.c_int]),
    'xrd_comm_world': (C.c_int, [vp]),
    'xrd_allreduce_grads': (C.c_int, [vp, vp, i64, vp]),
    'xrd_allreduce_max_i32': (C.c_int, [vp, vp, i64, vp]),
    'xrd_comm_destroy': (None, [vp]),
    'xrd_gs_preprocess': (C.c_int, [vp, C.c_int] + [vp] * 11),
    'xrd_gs_band_clip': (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    'xrd_gs_duplicate_keys': (C.c_int, [C.c_int, C.c_int, vp, vp, vp, vp, vp,
                                        vp]),
    'xrd_gs_tile_ranges': (C.c_int, [i64, vp, vp, vp]),
    'xrd_gs_bin_ws_bytes': (i64, [C.c_int, i64, C.c_int, C.c_int]),
    'xrd_gs_bin': (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, i64, vp,
                             vp, vp, vp, vp]),
    'xrd_gs_bin2': (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, i64, vp,
                              vp, vp, vp, vp, vp, vp]),
    'xrd_gs_blend_ckpt_floats': (i64, [i64, C.c_int, C.c_int]),
    'xrd_gs_blend_fwd': (C.c_int, [vp] * 15),
    'xrd_gs_blend_bwd': (C.c_int, [vp, C.c_int, i64] + [vp] * 22),
    'xrd_gs_prepare_fwd': (C.c_int, [C.c_int] + [vp] * 5 + [C.c_int] +
                           [vp] * 7),
    'xrd_gs_prepare_bwd': (C.c_int, [C.c_int] + [vp] * 5 + [C.c_int] +
                           [vp] * 13),
    'xrd_gs_loss_fwd': (C.c_int, [C.c_int] * 4 + [f32] * 4 + [vp] * 8),
    'xrd_gs_loss_bwd': (C.c_int, [C.c_int] * 4 + [f32] * 4 + [vp] * 10),
    'xrd_gs_render_fwd': (C.c_int, [vp] * 12),
    'xrd_gs_render_bwd': (C.c_int, [vp] * 14),
    'xrd_gs_render_fwd2': (C.c_int, [vp] * 14),
    'xrd_gs_render_bwd2': (C.c_int, [vp] * 17),
    'xrd_gs_preprocess_bwd': (C.c_int, [vp, C.c_int] + [vp] * 10),
    'xrd_knn_cell_ids': (C.c_int, [i64, vp, vp, f32, vp, vp, vp]),
    'xrd_knn_cell_ranges': (C.c_int, [i64, vp, vp, vp, vp]),
    'xrd_knn_search': (C.c_int, [i64, vp, vp, vp, vp, f32, vp, vp, vp,
                                 C.c_int, f32, vp, vp, vp]),
    'xrd_knn_search_count': (C.c_int, [i64, vp, vp, vp, vp, f32, vp, vp, vp,
                                       C.c_int, f32, vp, vp, vp, f32, vp,
                                       vp]),
    'xrd_sample_rays': (C.c_int, [C.c_int] * 5 + [f32] * 4 + [vp] * 12),
    'xrd_sample_rays_bwd': (C.c_int, [C.c_int] * 5 + [f32] * 4 + [vp] * 5),
    'xrd_sample_rays_multi': (C.c_int,
                              [C.c_int] * 6 + [f32] * 4 + [vp] * 14),
    'xrd_sample_rays_multi_bwd': (C.c_int,
                                  [C.c_int] * 6 + [f32] * 4 + [vp] * 7),
    'xrd_nice_loss': (C.c_int, [C.c_int] * 4 + [f32] + [vp] * 10),
    'xrd_pose_quat_fwd': (C.c_int, [vp] * 4),
    'xrd_pose_quat_bwd': (C.c_int, [vp] * 5),
    'xrd_pose_aa_fwd': (C.c_int, [C.c_int, vp, vp, vp, vp]),
    'xrd_pose_aa_bwd': (C.c_int, [C.c_int, vp, vp, vp, vp, vp]),
    'xrd_pose_from_matrix': (C.c_int, [C.c_int, vp, vp, vp]),
    'xrd_pose_from_matrix_checked': (C.c_int, [C.c_int, vp, vp, vp, vp]),
    'xrd_pose_predict': (C.c_int, [vp, vp, vp, vp]),
    'xrd_adam_dense': (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32,
                                 vp, vp]),
    'xrd_adam_dense_multi': (C.c_int, [C.c_int, C.POINTER(AdamDenseSet), f32,
                                       f32, f32, vp]),
    'xrd_adam_dense_tick': (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32,
                                      f32, vp, C.c_int, vp]),
    'xrd_track_best': (C.c_int, [vp] * 6),
    'xrd_coslam_flat_len': (C.c_int, []),
    'xrd_coslam_pack_len': (C.c_int, []),
    'xrd_coslam_dw_len': (C.c_int, []),
    'xrd_coslam_index': (C.c_int, [vp, vp]),
    'xrd_coslam_render_fwd': (C.c_int, [C.POINTER(CoslamScene), C.c_int] +
                              [vp] * 8),
    'xrd_coslam_bwd_ws_floats': (i64, [C.c_int]),
    'xrd_coslam_render_bwd': (C.c_int, [C.POINTER(CoslamScene), C.c_int] +
                              [vp] * 12),
    'xrd_coslam_bwd_ws_floats_extra': (i64, [C.c_int, i64]),
    'xrd_coslam_render_bwd_extra': (C.c_int, [C.POINTER(CoslamScene), C.c_int]
                                    + [vp] * 10 + [i64] + [vp] * 4),
    'xrd_sample_distinct': (C.c_int, [i64, C.c_int, vp, vp, vp]),
    'xrd_sample_distinct_dev': (C.c_int, [vp, C.c_int, vp, vp, vp]),
    'xrd_coslam_map_rows': (C.c_int, [C.c_int, vp, vp, C.c_int, C.c_int] +
                            [vp] * 8),
    'xrd_pose_rays_fwd': (C.c_int, [C.c_int, vp, C.c_int, vp, vp, vp, vp, vp]),
    'xrd_pose_rays_bwd': (C.c_int, [C.c_int, C.c_int, vp, C.c_int, vp, vp, vp,
                                    vp, vp]),
    'xrd_coslam_loss': (C.c_int, [C.c_int, C.c_int] + [f32] * 7 + [vp] * 10),
    'xrd_coslam_loss_live': (C.c_int, [C.c_int, C.c_int] + [f32] * 7 +
                             [vp] * 11),
    'xrd_coslam_loss_stats': (C.c_int, [C.c_int, C.c_int] + [f32] * 3 +
                              [vp] * 7),
    'xrd_coslam_loss_grads': (C.c_int, [C.c_int, C.c_int] + [f32] * 7 +
                              [vp] * 7 + [i64] + [vp] * 4),
    'xrd_ssim_fwd': (C.c_int, [C.c_int] * 3 + [vp] * 7),
    'xrd_ssim_bwd': (C.c_int, [C.c_int] * 3 + [vp] * 8),
    'xrd_selftest_mfma': (C.c_int, [vp, vp, vp, vp]),
    'xrd_compact_ws_ints': (i64, [i64]),
    'xrd_compact_rows': (C.c_int, [i64, vp, C.c_int] + [vp] * 6),
    'xrd_voxel_first_flags': (C.c_int, [i64, vp, vp, vp, vp, i64, vp, vp]),
    'xrd_point_dynamic_radius': (C.c_int, [C.c_int, C.c_int, vp] + [f64] * 4 +
                                 [vp] * 3),
    'xrd_point_sensor_points': (C.c_int, [i64] + [vp] * 5),
    'xrd_point_insert': (C.c_int, [C.c_int] + [vp] * 7 + [C.c_int, C.c_int,
                                                          f32, f32] + [vp] * 5),
    'xrd_point_frustum_mask': (C.c_int, [i64, vp, vp, vp, C.c_int, C.c_int] +
                               [f64] * 4 + [C.c_int] + [vp] * 5),
}


class GsCamera(C.Structure):
    """mirror of ``xrd_gs_camera``"""
    _fields_ = [('image_height', i32), ('image_width', i32),
                ('tanfovx', f32), ('tanfovy', f32), ('bg', f32 * 3),
                ('scale_modifier', f32), ('viewmatrix', f32 * 16),
                ('projmatrix', f32 * 16)]


def declared_symbols():
    """every entry point include/xrdslam_hip.h declares (kept in sync by
    tests/test_abi.py, which parses the header)."""
    return sorted(_SIGS)


def register(name, restype, argtypes):
    _SIGS[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype, fn.argtypes = restype, argtypes


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise XrdError(
                f'{LIB_PATH} not found: the HIP engine is not built. Run '
                '`python -m xrdslam_amd.build` (there is no CPU fallback).')
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = {1: 'bad argument', 2: 'HIP launch/runtime error',
               3: 'unsupported configuration'}.get(rc, f'error {rc}')
        detail = lib().xrd_last_error()
        raise XrdError(f'{what}: {msg}'
                       f'{" (" + detail.decode() + ")" if detail and rc == 2 else ""}')


def ptr(t):
    """device/host pointer of a torch tensor (or None) as void*"""
    if t is None:
        return None
    import torch
    dense = t.numel() == 0 or t.is_contiguous() or (
        t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d))
    assert dense, 'engine tensors must be dense'
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
