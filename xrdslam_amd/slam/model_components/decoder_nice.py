"""NICE decoders as flat parameter vectors for the fused HIP render.

The reference keeps each decoder as a tree of ``nn.Linear`` modules
(slam/model_components/decoder_nice.py:103-234,237-320,323-384); the kernels
consume one packed buffer per decoder, so every decoder here owns ONE flat
``nn.Parameter`` in the reference's ``state_dict`` order (what
``xrd_nice_pack_index`` expects) and exposes the reference's key names through
``state_dict()`` / ``load_state_dict()`` so reference checkpoints
(pretrained/nice_slam/*.pt) load unchanged.  Initialisation draws from the
torch RNG in the same order and with the same initialisers as the reference,
so equal seeds give equal weights.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn as nn

from ...engine import nice as _en


def _linear_default(out_dim, in_dim):
    """nn.Linear.reset_parameters"""
    w = torch.empty(out_dim, in_dim)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1 / math.sqrt(in_dim) if in_dim > 0 else 0
    b = torch.empty(out_dim)
    nn.init.uniform_(b, -bound, bound)
    return w, b


def _dense(out_dim, in_dim, activation):
    """DenseLayer.reset_parameters (decoder_nice.py:86-91)"""
    w = torch.empty(out_dim, in_dim)
    nn.init.xavier_uniform_(w, gain=nn.init.calculate_gain(activation))
    return w, torch.zeros(out_dim)


class FlatDecoder(nn.Module):
    """one decoder = one flat parameter; ``kind`` in coarse/middle/fine/color"""

    def __init__(self, kind: str):
        super().__init__()
        self.kind = kind
        self.shapes = _en.param_shapes(kind)
        vals = OrderedDict()
        if kind == 'coarse':  # MLP_no_xyz: only DenseLayers
            ins = [32, 32, 32, 64, 32]
            for i in range(5):
                w, b = _dense(32, ins[i], 'relu')
                vals[f'pts_linears.{i}.weight'] = w
                vals[f'pts_linears.{i}.bias'] = b
            w, b = _dense(1, 32, 'linear')
        else:
            c_dim = 64 if kind == 'fine' else 32
            out_dim = 4 if kind == 'color' else 1
            for i in range(5):
                w, b = _linear_default(32, c_dim)
                vals[f'fc_c.{i}.weight'] = w
                vals[f'fc_c.{i}.bias'] = b
            vals['embedder._B'] = torch.randn((3, 93)) * 25
            ins = [93, 32, 32, 125, 32]
            for i in range(5):
                w, b = _dense(32, ins[i], 'relu')
                vals[f'pts_linears.{i}.weight'] = w
                vals[f'pts_linears.{i}.bias'] = b
            w, b = _dense(out_dim, 32, 'linear')
        vals['output_linear.weight'] = w
        vals['output_linear.bias'] = b
        self.flat = nn.Parameter(_en.flatten_state_dict(vals, kind))
        self.bound = None

    def named_views(self):
        out, off = OrderedDict(), 0
        for name, shape in self.shapes:
            n = int(torch.tensor(shape).prod())
            out[name] = self.flat.detach()[off:off + n].view(shape)
            off += n
        return out

    def state_dict(self, *a, **k):  # reference key names
        return OrderedDict((n, v.clone()) for n, v in
                           self.named_views().items())

    def load_state_dict(self, sd, strict=True):
        with torch.no_grad():
            self.flat.copy_(_en.flatten_state_dict(
                {k: v.to(self.flat.device) for k, v in sd.items()},
                self.kind))


class NICE(nn.Module):
    """container with the reference's attribute names
    (decoder_nice.py:337-384); evaluation happens in the fused kernel."""

    def __init__(self, coarse=False, **kwargs):
        super().__init__()
        if coarse:
            self.coarse_decoder = FlatDecoder('coarse')
        self.middle_decoder = FlatDecoder('middle')
        self.fine_decoder = FlatDecoder('fine')
        self.color_decoder = FlatDecoder('color')
        self.bound = None

    def decoders(self):
        d = {'middle': self.middle_decoder, 'fine': self.fine_decoder,
             'color': self.color_decoder}
        if hasattr(self, 'coarse_decoder'):
            d['coarse'] = self.coarse_decoder
        return d
