"""Co-SLAM decoders (reference: slam/model_components/decoder_coslam.py): two
bias-free MLPs.  SDF net: [hash features, OneBlob] -> hidden -> (sdf, geo
feature); colour net: [OneBlob, geo feature] -> hidden -> rgb.  Only the
plain-PyTorch variant is built (``tcnn_network=False`` is the reference
default, joint_encoding.py:35); parameter names/shapes match the reference's
``state_dict`` (``color_net.model.0.weight`` ...)."""
import torch
import torch.nn as nn


def _mlp(in_dim, hidden, out_dim, n_layers):
    layers = []
    for k in range(n_layers):
        last = k == n_layers - 1
        layers.append(nn.Linear(in_dim if k == 0 else hidden,
                                out_dim if last else hidden, bias=False))
        if not last:
            layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class ColorNet(nn.Module):
    def __init__(self, config, input_ch=4):
        super().__init__()
        if config.tcnn_network:
            raise NotImplementedError('tcnn FullyFusedMLP is not built')
        self.model = _mlp(input_ch + config.geo_feat_dim,
                          config.hidden_dim_color, 3, config.num_layers_color)

    def forward(self, x):
        return self.model(x)


class SDFNet(nn.Module):
    def __init__(self, config, input_ch=3):
        super().__init__()
        if config.tcnn_network:
            raise NotImplementedError('tcnn FullyFusedMLP is not built')
        self.model = _mlp(input_ch, config.hidden_dim,
                          1 + config.geo_feat_dim, config.num_layers)

    def forward(self, x, return_geo=True):
        out = self.model(x)
        return out if return_geo else out[..., :1]


class ColorSDFNet(nn.Module):
    """separate colour grid (oneGrid=False)"""

    def __init__(self, config, input_ch=3, input_ch_pos=12):
        super().__init__()
        self.color_net = ColorNet(config, input_ch=input_ch + input_ch_pos)
        self.sdf_net = SDFNet(config, input_ch=input_ch + input_ch_pos)

    def forward(self, embed, embed_pos, embed_color):
        h = self.sdf_net(torch.cat([embed, embed_pos], -1))
        sdf, geo = h[..., :1], h[..., 1:]
        rgb = self.color_net(torch.cat([embed_pos, embed_color, geo], -1))
        return torch.cat([rgb, sdf], -1)


class ColorSDFNet_v2(nn.Module):
    """one shared grid (the default)"""

    def __init__(self, config, input_ch=3, input_ch_pos=12):
        super().__init__()
        self.color_net = ColorNet(config, input_ch=input_ch_pos)
        self.sdf_net = SDFNet(config, input_ch=input_ch + input_ch_pos)

    def forward(self, embed, embed_pos):
        h = self.sdf_net(torch.cat([embed, embed_pos], -1))
        sdf, geo = h[..., :1], h[..., 1:]
        rgb = self.color_net(torch.cat([embed_pos, geo], -1))
        return torch.cat([rgb, sdf], -1)
