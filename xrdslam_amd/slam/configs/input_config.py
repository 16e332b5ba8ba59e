"""Per-algorithm default configurations with the reference's field names and
values (slam/configs/input_config.py:45-493).  Only the hot-path algorithms are
registered; tracker/mapper cadence values that the benchmark loop needs are
kept next to the algorithm config."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict

from ..algorithms.coslam import CoSLAMConfig
from ..algorithms.nice_slam import NiceSLAMConfig
from ..algorithms.point_slam import PointSLAMConfig
from ..algorithms.splatam import SplaTAMConfig
from ..algorithms.voxfusion import VoxFusionConfig
from ..engine.optimizers import AdamOptimizerConfig
from ..engine.schedulers import (LRconfig, NiceSLAMSchedulerConfig,
                                 PointSLAMSchedulerConfig)
from ..models.conv_onet import ConvOnetConfig
from ..models.conv_onet_pointslam import ConvOnet2Config
from ..models.gaussian_splatting import GaussianSplattingConfig
from ..models.joint_encoding import JointEncodingConfig
from ..models.sparse_voxel import SparseVoxelConfig


@dataclass
class PipelineCadence:
    """the TrackerConfig/MapperConfig numbers the loop depends on"""
    map_every: int = 5
    keyframe_every: int = 50
    render_freq: int = 50
    use_relative_pose: bool = False
    init_pose_offset: int = 0
    lazy_start: int = -1


def _sched(**lr):
    return NiceSLAMSchedulerConfig(stage_lr=LRconfig(**lr))


def nice_slam_config(bound=None) -> NiceSLAMConfig:
    """algorithm_configs['nice-slam'] (input_config.py:45-156), office0"""
    bound = bound or [[-5.5, 5.9], [-6.7, 5.4], [-4.7, 5.3]]
    z = dict(coarse=0.0, middle=0.0, fine=0.0, color=0.0)
    return NiceSLAMConfig(
        coarse=True, tracking_n_iters=10, mapping_n_iters=60,
        mapping_first_n_iters=1500, mapping_window_size=5,
        tracking_sample=200, mapping_sample=1000, min_sample_pixels=200,
        ray_batch_size=100000, tracking_Wedge=100, tracking_Hedge=100,
        mapping_bound=[list(b) for b in bound],
        marching_cubes_bound=[list(b) for b in bound],
        mapping_middle_iter_ratio=0.4, mapping_fine_iter_ratio=0.6,
        mapping_lr_factor=1.0, mapping_lr_first_factor=5.0,
        model=ConvOnetConfig(points_batch_size=100000,
                             mapping_frustum_feature_selection=True),
        optimizers={
            'decoder': {'optimizer': AdamOptimizerConfig(),
                        'scheduler': _sched(**{**z, 'color': 0.005})},
            'grid_coarse': {'optimizer': AdamOptimizerConfig(),
                            'scheduler': _sched(**{**z, 'coarse': 0.001})},
            'grid_middle': {'optimizer': AdamOptimizerConfig(),
                            'scheduler': _sched(coarse=0.0, middle=0.1,
                                                fine=0.005, color=0.005)},
            'grid_fine': {'optimizer': AdamOptimizerConfig(),
                          'scheduler': _sched(coarse=0.0, middle=0.0,
                                              fine=0.005, color=0.005)},
            'grid_color': {'optimizer': AdamOptimizerConfig(),
                           'scheduler': _sched(**{**z, 'color': 0.005})},
            'tracking_pose': {'optimizer': AdamOptimizerConfig(lr=1e-3),
                              'scheduler': None},
            'mapping_pose': {'optimizer': AdamOptimizerConfig(),
                             'scheduler': _sched(**{**z, 'color': 0.001})},
        })


def coslam_config(bound=None) -> CoSLAMConfig:
    """algorithm_configs['co-slam'] (input_config.py:203-295), office0"""
    bound = bound or [[-3, 3], [-4, 2.5], [-2, 2.5]]
    adam = AdamOptimizerConfig
    return CoSLAMConfig(
        separate_LR=True, retain_graph=True, rot_rep='axis_angle',
        tracking_n_iters=10, mapping_n_iters=10, mapping_first_n_iters=200,
        keyframe_selection_method='all', mapping_sample=2048,
        tracking_sample=1024, min_sample_pixels=100, ray_batch_size=30000,
        tracking_Wedge=20, tracking_Hedge=20,
        mapping_bound=[list(b) for b in bound],
        marching_cubes_bound=[list(b) for b in bound],
        model=JointEncodingConfig(cam_depth_trunc=100.0, tcnn_encoding=True),
        optimizers={
            'decoder': {'optimizer': adam(lr=1e-2, weight_decay=1e-6,
                                          betas=(0.9, 0.99)),
                        'scheduler': None},
            'embed_fn': {'optimizer': adam(lr=1e-2, eps=1e-15,
                                           betas=(0.9, 0.99)),
                         'scheduler': None},
            'embed_fn_color': {'optimizer': adam(lr=1e-2, eps=1e-15,
                                                 betas=(0.9, 0.99)),
                               'scheduler': None},
            'tracking_pose_r': {'optimizer': adam(lr=1e-3), 'scheduler': None},
            'tracking_pose_t': {'optimizer': adam(lr=1e-3), 'scheduler': None},
            'mapping_pose_r': {'optimizer': adam(lr=1e-3, accum_step=5),
                               'scheduler': None},
            'mapping_pose_t': {'optimizer': adam(lr=1e-3, accum_step=5),
                               'scheduler': None},
        })


def voxfusion_config() -> VoxFusionConfig:
    """algorithm_configs['vox-fusion'] (input_config.py:159-201)"""
    adam = AdamOptimizerConfig
    return VoxFusionConfig(
        keyframe_selection_method='random', tracking_n_iters=30,
        mapping_n_iters=15, mapping_first_n_iters=30, mapping_window_size=5,
        mapping_sample=1024, tracking_sample=1024, ray_batch_size=3000,
        model=SparseVoxelConfig(),
        optimizers={
            'decoder': {'optimizer': adam(lr=5e-3), 'scheduler': None},
            'embeddings': {'optimizer': adam(lr=5e-3), 'scheduler': None},
            'tracking_pose': {'optimizer': adam(lr=1e-2), 'scheduler': None},
            'mapping_pose': {'optimizer': adam(lr=1e-3), 'scheduler': None},
        })


def pointslam_config() -> PointSLAMConfig:
    """algorithm_configs['point-slam'] (input_config.py:298-375)"""
    adam = AdamOptimizerConfig

    def sched(a, b):
        return {'optimizer': adam(),
                'scheduler': PointSLAMSchedulerConfig(start_lr=a, end_lr=b)}
    return PointSLAMConfig(
        separate_LR=True, tracking_n_iters=40, mapping_n_iters=300,
        mapping_first_n_iters=1500, mapping_window_size=12,
        tracking_sample=1500, mapping_sample=5000, min_sample_pixels=40,
        ray_batch_size=3000, tracking_Wedge=100, tracking_Hedge=100,
        mapping_BA=False, mapping_frustum_feature_selection=True,
        mapping_pixels_based_on_color_grad=1000,
        model=ConvOnet2Config(cuda_id=0, points_batch_size=500000),
        optimizers={
            'decoder': sched(0.001, 0.005), 'geometry': sched(0.03, 0.005),
            'color': sched(0.0, 0.005),
            'tracking_pose_r': {'optimizer': adam(lr=0.002 * 0.2),
                                'scheduler': None},
            'tracking_pose_t': {'optimizer': adam(lr=0.002),
                                'scheduler': None},
            'mapping_pose_r': {'optimizer': adam(lr=0.0002),
                               'scheduler': None},
            'mapping_pose_t': {'optimizer': adam(lr=0.0002),
                               'scheduler': None}})


def splatam_config() -> SplaTAMConfig:
    """algorithm_configs['splaTAM'] (input_config.py:377-431)"""
    def adam(lr, eps=1e-8):
        return {'optimizer': AdamOptimizerConfig(lr=lr, eps=eps),
                'scheduler': None}
    return SplaTAMConfig(
        retain_graph=False, separate_LR=True, keyframe_use_ray_sample=False,
        tracking_n_iters=40, mapping_n_iters=60, mapping_first_n_iters=60,
        mapping_window_size=24, model=GaussianSplattingConfig(),
        optimizers={'means3D': adam(0.0001, 1e-15),
                    'rgb_colors': adam(0.0025, 1e-15),
                    'unnorm_rotations': adam(0.001, 1e-15),
                    'logit_opacities': adam(0.05, 1e-15),
                    'log_scales': adam(0.001, 1e-15),
                    'tracking_pose_r': adam(0.0004),
                    'tracking_pose_t': adam(0.002)})


algorithm_configs: Dict[str, object] = {'nice-slam': nice_slam_config,
                                        'co-slam': coslam_config,
                                        'vox-fusion': voxfusion_config,
                                        'splaTAM': splatam_config,
                                        'point-slam': pointslam_config}
cadence: Dict[str, PipelineCadence] = {
    'nice-slam': PipelineCadence(),
    'co-slam': PipelineCadence(map_every=5, keyframe_every=5),
    'vox-fusion': PipelineCadence(map_every=1, keyframe_every=50,
                                  use_relative_pose=True,
                                  init_pose_offset=10),
    'splaTAM': PipelineCadence(map_every=1, keyframe_every=5,
                               use_relative_pose=True),
    'point-slam': PipelineCadence(map_every=5, keyframe_every=20,
                                  lazy_start=20)}
