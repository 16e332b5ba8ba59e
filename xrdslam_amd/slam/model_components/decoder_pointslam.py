"""Point-SLAM decoders (reference:
slam/model_components/decoder_pointslam.py): features are interpolated from
the <= 8 nearest neural points inside the query radius (inverse squared
distance weights), then an MLP on Fourier features of the position adds the
interpolated feature to every hidden layer.  Geometry: 5 x 32 ReLU -> occupancy
logit.  Colour: neighbour features first pass through F_theta together with a
Fourier embedding of the relative position; 5 x 128 softplus -> sigmoid rgb.
Parameter names match the reference's ``state_dict``.

The neighbour search is the exact grid kNN (``xrd_knn_*``) through
``NeuralPointCloud.find_neighbors_faiss``; missing neighbours (index -1) get
zero weight."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.init as init


class GaussianFourierFeatureTransform(nn.Module):
    def __init__(self, num_input_channels, mapping_size=93, scale=25,
                 learnable=False, concat=True):
        super().__init__()
        self.concat, self.mapping_size = concat, mapping_size
        self.scale, self.learnable = scale, learnable
        B = torch.randn((num_input_channels, mapping_size)) * scale
        self._B = nn.Parameter(B) if learnable else B

    def forward(self, x):
        x = x.squeeze(0)
        assert x.dim() == 2
        x = (2 * math.pi * x) @ self._B.to(x.device)
        return torch.cat((torch.sin(x), torch.cos(x)), -1) if self.concat \
            else torch.sin(x)


class DenseLayer(nn.Linear):
    def __init__(self, in_dim, out_dim, activation='relu', *args, **kwargs):
        self.activation = activation
        super().__init__(in_dim, out_dim, *args, **kwargs)

    def reset_parameters(self):
        init.xavier_uniform_(self.weight,
                             gain=init.calculate_gain(self.activation))
        if self.bias is not None:
            init.zeros_(self.bias)


class Same(nn.Module):
    def __init__(self, mapping_size=3):
        super().__init__()
        self.mapping_size = mapping_size

    def forward(self, x):
        return x.squeeze(0)


class MLP_col_neighbor(nn.Module):
    """F_theta: [rel-pos embedding, colour feature] -> colour feature"""

    def __init__(self, c_dim, embedding_size_rel, hidden_size):
        super().__init__()
        self.linear1 = nn.Linear(c_dim + embedding_size_rel, hidden_size)
        self.linear2 = nn.Linear(hidden_size, c_dim)
        self.act_fn = nn.Softplus(beta=100)
        init.xavier_uniform_(self.linear1.weight)
        init.xavier_uniform_(self.linear2.weight)

    def forward(self, x):
        return self.linear2(self.act_fn(self.linear1(x)))


class MLP_exposure(nn.Module):
    def __init__(self, latent_dim, hidden_size):
        super().__init__()
        self.linear1 = nn.Linear(latent_dim, hidden_size)
        self.linear2 = nn.Linear(hidden_size, 12)
        self.act_fn = nn.Softplus(beta=100)
        init.normal_(self.linear1.weight, mean=0, std=0.01)
        init.normal_(self.linear2.weight, mean=0, std=0.01)

    def forward(self, x):
        return self.linear2(self.act_fn(self.linear1(x)))


def _empty_feature(c_dim, device):
    """the random feature given to sample points without enough neighbours
    (one vector per call, decoder_pointslam.py:215-217); replaceable for
    parity tests"""
    return torch.zeros([c_dim], device=device).normal_(mean=0, std=0.01)


class _PointMLP(nn.Module):
    """what MLP_geometry and MLP_color share: neighbour interpolation and the
    skip-connected trunk with per-layer feature injection"""

    def _build_trunk(self, embedding_input, hidden_size, n_blocks, c_dim,
                     out_dim, out_act):
        self.fc_c = nn.ModuleList(
            [nn.Linear(c_dim, hidden_size) for _ in range(n_blocks)])
        layers = [DenseLayer(embedding_input, hidden_size, activation='relu')]
        for i in range(n_blocks - 1):
            extra = embedding_input if i in self.skips else 0
            layers.append(DenseLayer(hidden_size + extra, hidden_size,
                                     activation='relu'))
        self.pts_linears = nn.ModuleList(layers)
        self.output_linear = DenseLayer(hidden_size, out_dim,
                                        activation=out_act)

    def _interpolate(self, npc, p, npc_feats, is_tracker, dynamic_r_query,
                     transform=None, neighbors=None):
        """-> (feature [n,c_dim], has_neighbors [n]).  ``neighbors``: the
        (D, I, n_nb) of a search already made for these points (POINT.forward
        searches once for both decoders; the reference searches twice with
        the same result, decoder_pointslam.py:181,426)"""
        cloud = npc.cloud_tensor(p.device)
        p = p.reshape(-1, 3)
        D, I, n_nb = neighbors if neighbors is not None else \
            npc.find_neighbors_faiss(p.detach().clone(), step='query',
                                     dynamic_radius=dynamic_r_query)
        missing = I < 0
        I = I.clamp(min=0)
        bound = npc.get_radius_query()**2 if not self.use_dynamic_radius \
            else dynamic_r_query.reshape(-1, 1)**2
        if is_tracker:
            # distances recomputed so that they carry the pose gradient
            D = torch.sum(torch.square(cloud[I] - p.reshape(-1, 1, 3)), -1)
            D = torch.where(missing, torch.full_like(D, float('inf')), D)
        has = n_nb > self.min_nn_num - 1
        w = 1.0 / (D + 1e-10) if self.weighting == 'distance' \
            else torch.exp(-20 * torch.sqrt(D))
        w = torch.where((D > bound) | missing, torch.zeros_like(w), w)
        w = F.normalize(w, p=1, dim=1).unsqueeze(-1)
        feats = npc_feats[I]
        if transform is not None:
            feats = transform(cloud[I] - p[:, None, :], feats)
        c = (w * feats).sum(1).reshape(-1, self.c_dim)
        c[~has] = self.empty_feature_fn(self.c_dim, p.device)
        return c, has

    def _trunk(self, embedded, c, act):
        h = embedded
        for i, layer in enumerate(self.pts_linears):
            h = act(layer(h)) + self.fc_c[i](c)
            if i in self.skips:
                h = torch.cat([embedded, h], -1)
        return self.output_linear(h)


class MLP_geometry(_PointMLP):
    def __init__(self, use_dynamic_radius, pointcloud_nn_weighting,
                 pointcloud_min_nn_num, rendering_n_surface, c_dim=32,
                 hidden_size=128, n_blocks=5, leaky=False,
                 sample_mode='bilinear', skips=(2, ),
                 pos_embedding_method='fourier', concat_feature=False,
                 use_view_direction=False):
        super().__init__()
        if pos_embedding_method != 'fourier':
            raise NotImplementedError('only the fourier embedding is built')
        self.c_dim, self.skips = c_dim, list(skips)
        self.weighting = pointcloud_nn_weighting
        self.use_dynamic_radius = use_dynamic_radius
        self.min_nn_num, self.N_surface = pointcloud_min_nn_num, \
            rendering_n_surface
        self.empty_feature_fn = _empty_feature
        self.embedder = GaussianFourierFeatureTransform(
            3, mapping_size=93, scale=25, concat=False, learnable=True)
        self.embedder_rel_pos = GaussianFourierFeatureTransform(
            3, mapping_size=10, scale=32, learnable=True)
        self.mlp_col_neighbor = MLP_col_neighbor(c_dim, 20, hidden_size)
        self._build_trunk(93, hidden_size, n_blocks, c_dim, 1, 'relu')

    use_fused = True   # neighbour interpolation + trunk as one kernel each way
    map_gradients = True   # False (tracking): no gradient to the map features

    def forward(self, p, npc, pts_num=16, is_tracker=False, pts_views_d=None,
                dynamic_r_query=None, neighbors=None):
        if self.use_fused and p.is_cuda and is_tracker and \
                not self.output_linear.weight.requires_grad:
            # (a decoder that is being trained keeps the torch path: the
            # kernels return no weight gradients)
            from ...engine import point as _pt
            if _pt.supported(self):
                flat = p.reshape(-1, 3)
                if neighbors is None:
                    neighbors = npc.find_neighbors_faiss(
                        flat.detach(), step='query',
                        dynamic_radius=dynamic_r_query)
                occ, has = _pt.geometry(self, flat, neighbors, npc,
                                        dynamic_r_query)
                valid_ray = ~(torch.sum(has.view(-1, pts_num), 1) <
                              int(self.N_surface / 2 + 1))
                return occ, valid_ray, has
        c, has = self._interpolate(npc, p, npc.get_geo_feats(), is_tracker,
                                   dynamic_r_query, neighbors=neighbors)
        # a ray is valid when at least half of its samples have neighbours
        valid_ray = ~(torch.sum(has.view(-1, pts_num), 1) <
                      int(self.N_surface / 2 + 1))
        emb = self.embedder(p.float().reshape(1, -1, 3))
        return self._trunk(emb, c, F.relu).squeeze(-1), valid_ray, has


class MLP_color(_PointMLP):
    def __init__(self, use_dynamic_radius, pointcloud_nn_weighting,
                 pointcloud_min_nn_num, rendering_n_surface,
                 model_encode_rel_pos_in_col, model_encode_exposure,
                 model_encode_viewd, model_exposure_dim, c_dim=32,
                 hidden_size=128, n_blocks=5, leaky=False,
                 sample_mode='bilinear', skips=(2, ),
                 pos_embedding_method='fourier', concat_feature=False,
                 use_view_direction=False):
        super().__init__()
        if pos_embedding_method != 'fourier':
            raise NotImplementedError('only the fourier embedding is built')
        self.c_dim, self.skips = c_dim, list(skips)
        self.weighting = pointcloud_nn_weighting
        self.use_dynamic_radius = use_dynamic_radius
        self.min_nn_num, self.N_surface = pointcloud_min_nn_num, \
            rendering_n_surface
        self.use_view_direction = use_view_direction
        self.encode_rel_pos_in_col = model_encode_rel_pos_in_col
        self.encode_exposure, self.encode_viewd = model_encode_exposure, \
            model_encode_viewd
        self.empty_feature_fn = _empty_feature
        self.embedder = GaussianFourierFeatureTransform(3, mapping_size=20,
                                                        scale=32)
        embedding_input = 40
        if use_view_direction:
            self.embedder_view_direction = GaussianFourierFeatureTransform(
                3, mapping_size=20, scale=32) if model_encode_viewd \
                else Same(mapping_size=3)
            embedding_input += (2 if model_encode_viewd else 1) * \
                self.embedder_view_direction.mapping_size
        self.embedder_rel_pos = GaussianFourierFeatureTransform(
            3, mapping_size=10, scale=32, learnable=True)
        self.mlp_col_neighbor = MLP_col_neighbor(c_dim, 20, hidden_size)
        if model_encode_exposure:
            self.mlp_exposure = MLP_exposure(model_exposure_dim, hidden_size)
        self._build_trunk(embedding_input, hidden_size, n_blocks, c_dim, 3,
                          'linear')
        self.actvn = (lambda x: F.leaky_relu(x, 0.2)) if leaky \
            else nn.Softplus(beta=100)

    def _f_theta(self, rel_pos, feats):
        n = rel_pos.shape[0]
        emb = self.embedder_rel_pos(rel_pos.reshape(-1, 3)).reshape(n, -1, 20)
        return self.mlp_col_neighbor(torch.cat([emb, feats], -1))

    use_fused = True   # F_theta + interpolation + trunk as one kernel each way
    map_gradients = True   # False (tracking): no gradient to features / weights

    def forward(self, p, npc, is_tracker=False, pts_views_d=None,
                dynamic_r_query=None, exposure_feat=None, neighbors=None):
        if self.use_fused and p.is_cuda and is_tracker:
            from ...engine import point as _pt
            if _pt.color_supported(self):
                flat = p.reshape(-1, 3)
                if neighbors is None:
                    neighbors = npc.find_neighbors_faiss(
                        flat.detach(), step='query',
                        dynamic_radius=dynamic_r_query)
                return _pt.color(self, flat, neighbors, npc, dynamic_r_query)
        c, _ = self._interpolate(
            npc, p, npc.col_feats, is_tracker, dynamic_r_query,
            transform=self._f_theta if self.encode_rel_pos_in_col else None,
            neighbors=neighbors)
        emb = self.embedder(p.float().reshape(1, -1, 3))
        if self.use_view_direction:
            emb = torch.cat([emb, self.embedder_view_direction(
                F.normalize(pts_views_d, p=2, dim=1))], -1)
        out = self._trunk(emb, c, self.actvn)
        if self.encode_exposure:
            if exposure_feat is None:
                return out  # compensation applied by the caller
            aff = self.mlp_exposure(exposure_feat)
            out = torch.matmul(out, aff[:9].reshape(3, 3)) + aff[-3:]
        return torch.sigmoid(out)


class POINT(nn.Module):
    def __init__(self, use_dynamic_radius, pointcloud_nn_weighting,
                 pointcloud_min_nn_num, rendering_n_surface,
                 model_encode_rel_pos_in_col, model_encode_exposure,
                 model_encode_viewd, model_exposure_dim, c_dim=32,
                 hidden_size=128, pos_embedding_method='fourier',
                 use_view_direction=False):
        super().__init__()
        common = dict(use_dynamic_radius=use_dynamic_radius,
                      pointcloud_min_nn_num=pointcloud_min_nn_num,
                      pointcloud_nn_weighting=pointcloud_nn_weighting,
                      rendering_n_surface=rendering_n_surface, c_dim=c_dim,
                      skips=[2], n_blocks=5,
                      pos_embedding_method=pos_embedding_method)
        self.geo_decoder = MLP_geometry(hidden_size=32, **common)
        self.color_decoder = MLP_color(
            model_encode_rel_pos_in_col=model_encode_rel_pos_in_col,
            model_encode_exposure=model_encode_exposure,
            model_encode_viewd=model_encode_viewd,
            model_exposure_dim=model_exposure_dim, hidden_size=hidden_size,
            use_view_direction=use_view_direction, **common)

    def forward(self, p, npc, stage, pts_num=16, is_tracker=False,
                pts_views_d=None, dynamic_r_query=None, exposure_feat=None):
        # one neighbour search serves both decoders
        nb = None
        if stage != 'geometry':
            nb = npc.find_neighbors_faiss(
                p.reshape(-1, 3).detach(), step='query',
                dynamic_radius=dynamic_r_query)
        occ, ray_mask, point_mask = self.geo_decoder(
            p, npc, pts_num=pts_num, is_tracker=is_tracker,
            dynamic_r_query=dynamic_r_query, neighbors=nb)
        if stage == 'geometry':
            raw = torch.zeros(occ.shape[0], 4, dtype=torch.float,
                              device=occ.device)
            raw[..., -1] = occ
            return raw, ray_mask, point_mask
        rgb = self.color_decoder(p=p, npc=npc, is_tracker=is_tracker,
                                 pts_views_d=pts_views_d,
                                 dynamic_r_query=dynamic_r_query,
                                 exposure_feat=exposure_feat, neighbors=nb)
        return torch.cat([rgb, occ.unsqueeze(-1)], -1), ray_mask, point_mask
