"""Vox-Fusion implicit decoder (reference:
slam/model_components/decoder_voxfusion.py:85-149): trilinear voxel features
-> [positional encoding] -> ``depth`` ReLU layers -> (sdf, sdf feature) ->
colour head (sdf feature + encoded input -> ReLU -> sigmoid rgb).  Parameter
names match the reference's ``state_dict`` (``pts_linears.N``, ``sdf_out``,
``color_out.0/.2``).  Reference defaults for Vox-Fusion
(sparse_voxel.py:59-62): depth 2, width 128, in_dim 16, embedder 'none'."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Identity(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.embedding_size = dim

    def forward(self, x):
        return x


class _NerfEncoding(nn.Module):
    """[x, sin(2^k x), cos(2^k x)]_k, k = 0..multires-1"""

    def __init__(self, dim, multires):
        super().__init__()
        self.n_freqs = multires
        self.embedding_size = dim * (2 * multires + 1)

    def forward(self, x):
        assert x.dim() == 2
        bands = 2.**torch.linspace(0., self.n_freqs - 1, steps=self.n_freqs)
        parts = [x]
        for f in bands:
            parts += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(parts, 1)


class _GaussianFourier(nn.Module):
    def __init__(self, dim, mapping_size=93, scale=25):
        super().__init__()
        self._B = nn.Parameter(torch.randn(dim, mapping_size) * scale)
        self.embedding_size = mapping_size

    def forward(self, x):
        assert x.dim() == 2
        return torch.sin(x @ self._B.to(x.device))


class Decoder(nn.Module):
    def __init__(self, depth=8, width=256, in_dim=3, sdf_dim=128, skips=(4, ),
                 multires=6, embedder='nerf', local_coord=False, **kwargs):
        super().__init__()
        self.D, self.W, self.skips = depth, width, list(skips)
        if embedder == 'nerf':
            self.pe = _NerfEncoding(in_dim, multires)
        elif embedder == 'none':
            self.pe = _Identity(in_dim)
        elif embedder == 'gaussian':
            self.pe = _GaussianFourier(in_dim)
        else:
            raise NotImplementedError('unknown positional encoder')
        e = self.pe.embedding_size
        layers = [nn.Linear(e, width)]
        for i in range(depth - 1):
            layers.append(nn.Linear(width + (e if i in self.skips else 0),
                                    width))
        self.pts_linears = nn.ModuleList(layers)
        self.sdf_out = nn.Linear(width, 1 + sdf_dim)
        self.color_out = nn.Sequential(nn.Linear(sdf_dim + e, width),
                                       nn.ReLU(), nn.Linear(width, 3),
                                       nn.Sigmoid())

    def get_values(self, x):
        """[N, in_dim] -> [N, 4] = (rgb in [0,1], sdf)"""
        x = self.pe(x)
        h = x
        for i, layer in enumerate(self.pts_linears):
            h = F.relu(layer(h))
            if i in self.skips:
                h = torch.cat([x, h], -1)
        out = self.sdf_out(h)
        rgb = self.color_out(torch.cat([out[:, 1:], x], -1))
        return torch.cat([rgb, out[:, :1]], -1)

    def get_sdf(self, inputs):
        return self.get_values(inputs['emb'])[:, 3]

    def forward(self, inputs):
        out = self.get_values(inputs['emb'])
        return {'color': out[:, :3], 'sdf': out[:, 3]}
