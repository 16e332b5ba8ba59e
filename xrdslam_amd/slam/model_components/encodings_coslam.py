"""``get_encoder`` of Co-SLAM (reference:
slam/model_components/encodings_coslam.py:9-95) on the HIP encodings: same
arguments, same tiny-cuda-nn configuration dictionaries, evaluated by
xrdslam_amd.compat.tinycudann (hash grid / dense grid / OneBlob)."""
import numpy as np
import torch

from ...compat import tinycudann as tcnn


def get_encoder(encoding, input_dim=3, degree=4, n_bins=16, n_frequencies=12,
                n_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=512):
    name = encoding.lower()
    if 'dense' in name or 'hash' in name or 'tiled' in name:
        dense = 'dense' in name
        if dense:
            n_levels = 4
        pls = np.exp2(np.log2(desired_resolution / base_resolution) /
                      (n_levels - 1))
        cfg = {'n_levels': n_levels, 'n_features_per_level': level_dim,
               'base_resolution': base_resolution, 'per_level_scale': pls}
        if dense:
            cfg.update({'otype': 'Grid', 'type': 'Dense',
                        'interpolation': 'Linear'})
        else:
            cfg.update({'otype': 'HashGrid',
                        'log2_hashmap_size': log2_hashmap_size})
        embed = tcnn.Encoding(n_input_dims=input_dim, encoding_config=cfg,
                              dtype=torch.float)
    elif 'blob' in name:
        embed = tcnn.Encoding(n_input_dims=input_dim,
                              encoding_config={'otype': 'OneBlob',
                                               'n_bins': n_bins},
                              dtype=torch.float)
    else:
        raise NotImplementedError(
            f"encoding '{encoding}': only HashGrid/Dense/OneBlob are built "
            '(Co-SLAM defaults: HashGrid + OneBlob)')
    return embed, embed.n_output_dims
