"""``Mesher`` (reference: slam/common/mesher.py:20-263): evaluate the model on
a uniform lattice over ``marching_cubes_bound``, extract the level set, colour
the vertices by a second query.

The lattice (``get_grid_uniform``, numpy ``meshgrid`` order and all), the
out-of-bound override of ``eval_points`` and the batching are the reference's.
The iso-surface extraction is NOT skimage's Lewiner marching cubes (a
third-party dependency that, like open3d and trimesh, is absent offline —
parity unpinned): it is marching tetrahedra on the same lattice, evaluated as
tensor operations on the device — every lattice cube is split into six
tetrahedra around its main diagonal (the split is consistent across faces, so
the mesh is watertight), a vertex sits where the level crosses a tetrahedron
edge (linear interpolation, like marching cubes on the cube edges), vertices
shared between tetrahedra are merged by their lattice edge.  ``use_mask``
(convex hull of an open3d TSDF fusion of the keyframes) is not built.
``Mesh`` stands in for ``trimesh.Trimesh`` (vertices, faces, vertex_colors,
``export`` to .ply / .obj)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Type

import numpy as np
import torch

from ..configs.base_config import InstantiateConfig


@dataclass
class MesherConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: Mesher)
    points_batch_size: int = 500000
    resolution: int = 130
    level_set: int = 0
    remove_small_geometry_threshold: float = 0.2
    clean_mesh_bound_scale: float = 1.02
    get_largest_components: bool = False


@dataclass
class Mesh:
    vertices: np.ndarray                      # [V,3] float
    faces: np.ndarray                         # [F,3] int
    vertex_colors: Optional[np.ndarray] = None   # [V,3] uint8

    def export(self, path):
        v, f, c = self.vertices, self.faces, self.vertex_colors
        if str(path).endswith('.obj'):
            with open(path, 'w') as fh:
                for i in range(v.shape[0]):
                    col = '' if c is None else ' %.4f %.4f %.4f' % tuple(
                        c[i, :3] / 255.0)
                    fh.write('v %.7g %.7g %.7g%s\n' % (*v[i], col))
                for t in f:
                    fh.write('f %d %d %d\n' % (t[0] + 1, t[1] + 1, t[2] + 1))
            return
        with open(path, 'wb') as fh:      # binary little-endian PLY
            hdr = ['ply', 'format binary_little_endian 1.0',
                   f'element vertex {v.shape[0]}', 'property float x',
                   'property float y', 'property float z']
            if c is not None:
                hdr += ['property uchar red', 'property uchar green',
                        'property uchar blue']
            hdr += [f'element face {f.shape[0]}',
                    'property list uchar int vertex_indices', 'end_header']
            fh.write(('\n'.join(hdr) + '\n').encode())
            if c is None:
                fh.write(v.astype('<f4').tobytes())
            else:
                rec = np.empty(v.shape[0], dtype=[('p', '<f4', 3),
                                                  ('c', 'u1', 3)])
                rec['p'], rec['c'] = v, c[:, :3]
                fh.write(rec.tobytes())
            rec = np.empty(f.shape[0], dtype=[('n', 'u1'), ('i', '<i4', 3)])
            rec['n'], rec['i'] = 3, f
            fh.write(rec.tobytes())


# the six tetrahedra of a cube around its main diagonal 0-6; corner numbering
# 0:(0,0,0) 1:(1,0,0) 2:(1,1,0) 3:(0,1,0) 4:(0,0,1) 5:(1,0,1) 6:(1,1,1) 7:(0,1,1)
_CORNERS = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1),
            (1, 1, 1), (0, 1, 1)]
_TETS = [(0, 1, 2, 6), (0, 2, 3, 6), (0, 3, 7, 6), (0, 7, 4, 6), (0, 4, 5, 6),
         (0, 5, 1, 6)]
# tetrahedron edges and, per inside-mask, the triangles as edge triples
_TET_EDGES = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
_TET_TRIS = {1: [(0, 1, 2)], 2: [(0, 3, 4)], 4: [(1, 3, 5)], 8: [(2, 4, 5)],
             3: [(1, 2, 4), (1, 4, 3)], 5: [(0, 2, 5), (0, 5, 3)],
             9: [(0, 1, 5), (0, 5, 4)]}
for _m in (1, 2, 4, 8, 3, 5, 9):
    _TET_TRIS[15 - _m] = _TET_TRIS[_m]


@torch.no_grad()
def marching_tetrahedra(volume: torch.Tensor, level=0.0,
                        spacing=(1.0, 1.0, 1.0), descent=True):
    """level set of ``volume`` [X,Y,Z] -> (vertices [V,3] float64 in units of
    ``spacing`` from the lattice origin, faces [F,3] int64).  ``descent``:
    the object is where the values exceed the level (skimage's default
    gradient_direction), faces are wound so that normals point out of it."""
    vol = volume.double()
    dev = vol.device
    X, Y, Z = vol.shape
    if min(X, Y, Z) < 2:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64)
    lin = torch.arange(X * Y * Z, device=dev).reshape(X, Y, Z)
    sp = torch.tensor(spacing, dtype=torch.float64, device=dev)
    cv, cid = [], []
    for (dx, dy, dz) in _CORNERS:
        sl = (slice(dx, X - 1 + dx), slice(dy, Y - 1 + dy),
              slice(dz, Z - 1 + dz))
        cv.append(vol[sl].reshape(-1))
        cid.append(lin[sl].reshape(-1))
    inside = [v > level for v in cv]
    # only cubes the level crosses
    n_in = sum(i.long() for i in inside)
    cross = (n_in > 0) & (n_in < 8) & torch.isfinite(sum(cv))
    sel = torch.nonzero(cross).flatten()
    cv = [v[sel] for v in cv]
    cid = [c[sel] for c in cid]
    inside = [i[sel] for i in inside]
    tri_pos, tri_key = [], []
    N = X * Y * Z

    def unravel(idx):
        return torch.stack([idx // (Y * Z), (idx // Z) % Y, idx % Z],
                           -1).double()

    for tet in _TETS:
        mask = sum(inside[c].long() << k for k, c in enumerate(tet))
        for m, tris in _TET_TRIS.items():
            rows = torch.nonzero(mask == m).flatten()
            if rows.numel() == 0:
                continue
            pts, keys = {}, {}
            for e in {e for t in tris for e in t}:
                a, b = (tet[k] for k in _TET_EDGES[e])
                va, vb = cv[a][rows], cv[b][rows]
                ia, ib = cid[a][rows], cid[b][rows]
                t = ((level - va) / (vb - va)).clamp(0.0, 1.0).unsqueeze(-1)
                pa, pb = unravel(ia), unravel(ib)
                pts[e] = (pa + t * (pb - pa)) * sp
                lo, hi = torch.minimum(ia, ib), torch.maximum(ia, ib)
                k = lo * N + (hi - lo)
                # a crossing AT a lattice point (value == level) belongs to
                # that point, whichever edge reaches it: the triangles that
                # collapse there get a repeated index and are dropped below
                k = torch.where(t.squeeze(-1) >= 1.0, ib * N, k)
                k = torch.where(t.squeeze(-1) <= 0.0, ia * N, k)
                keys[e] = k
            # every triangle of a tetrahedron lies in a level plane of its
            # linear interpolant, i.e. is perpendicular to the interpolant's
            # gradient g; sum_k (v_k - mean v)(x_k - mean x) = C g with C the
            # (positive definite) covariance of the corners, so its dot with
            # a triangle normal has the sign of normal . g — never near zero
            xs = [unravel(cid[c][rows]) * sp for c in tet]
            vs = [cv[c][rows] for c in tet]
            xm, vm = sum(xs) / 4.0, sum(vs) / 4.0
            uphill = sum((v - vm).unsqueeze(-1) * (x - xm)
                         for v, x in zip(vs, xs))
            # descent: the object is where the values are high, normals point
            # out of it = downhill
            outward = -uphill if descent else uphill
            for (e0, e1, e2) in tris:
                p0, p1, p2 = pts[e0], pts[e1], pts[e2]
                nrm = torch.cross(p1 - p0, p2 - p0, dim=-1)
                flip = (nrm * outward).sum(-1) < 0
                k1 = torch.where(flip, keys[e2], keys[e1])
                k2 = torch.where(flip, keys[e1], keys[e2])
                q1 = torch.where(flip.unsqueeze(-1), p2, p1)
                q2 = torch.where(flip.unsqueeze(-1), p1, p2)
                tri_pos.append(torch.stack([p0, q1, q2], 1))
                tri_key.append(torch.stack([keys[e0], k1, k2], 1))
    if not tri_pos:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64)
    pos = torch.cat(tri_pos).reshape(-1, 3)
    key = torch.cat(tri_key).reshape(-1)
    uniq, inv = torch.unique(key, return_inverse=True)
    verts = torch.zeros(uniq.numel(), 3, dtype=torch.float64, device=dev)
    verts[inv] = pos          # a merged vertex has one position
    faces = inv.reshape(-1, 3)
    ok = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & \
        (faces[:, 0] != faces[:, 2])
    return verts.cpu().numpy(), faces[ok].cpu().numpy()


class Mesher:
    def __init__(self, config: MesherConfig, camera, bounding_box,
                 marching_cubes_bound) -> None:
        self.config = config
        self.bounding_box = bounding_box
        self.marching_cubes_bound = marching_cubes_bound
        self.camera = camera
        self.scale = 1.0

    def get_grid_uniform(self, resolution):
        """lattice points in numpy ``meshgrid`` ('xy') order (:45-67)"""
        b = self.marching_cubes_bound
        b = b.tolist() if hasattr(b, 'tolist') else b
        x = np.linspace(b[0][0], b[0][1], resolution)
        y = np.linspace(b[1][0], b[1][1], resolution)
        z = np.linspace(b[2][0], b[2][1], resolution)
        xx, yy, zz = np.meshgrid(x, y, z)
        pts = torch.tensor(np.vstack([xx.ravel(), yy.ravel(), zz.ravel()]).T,
                           dtype=torch.float)
        return {'grid_points': pts, 'xyz': [x, y, z]}

    def eval_points(self, p, query_fn, boundingbox, device='cuda:0'):
        """model values of ``p`` in batches; 100 outside the bound (:137-164)"""
        bound = boundingbox
        rets = []
        for pi in torch.split(p, self.config.points_batch_size):
            mask = (pi[:, 0] < bound[0][1]) & (pi[:, 0] > bound[0][0]) & \
                (pi[:, 1] < bound[1][1]) & (pi[:, 1] > bound[1][0]) & \
                (pi[:, 2] < bound[2][1]) & (pi[:, 2] > bound[2][0])
            ret = query_fn(pi)
            ret[~mask, :] = 100
            rets.append(ret)
        return torch.cat(rets, dim=0)

    def get_mesh(self, keyframe_graph, query_fn, color_func=None,
                 device='cuda:0', use_mask=False):
        if use_mask:
            raise NotImplementedError(
                'use_mask needs the open3d TSDF convex hull (:69-135), not '
                'available offline')
        with torch.no_grad():
            grid = self.get_grid_uniform(self.config.resolution)
            points = grid['grid_points'].to(device)
            z = torch.cat([
                self.eval_points(pnts, query_fn, self.bounding_box,
                                 device).reshape(pnts.shape[0], -1)[:, -1]
                for pnts in torch.split(points,
                                        self.config.points_batch_size)])
            x, y, zz = grid['xyz']
            # meshgrid('xy') order is [y, x, z]; the volume is indexed [x,y,z]
            vol = z.float().reshape(y.shape[0], x.shape[0],
                                    zz.shape[0]).permute(1, 0, 2)
            verts, faces = marching_tetrahedra(
                vol, level=float(self.config.level_set),
                spacing=(x[2] - x[1], y[2] - y[1], zz[2] - zz[1]))
            if verts.shape[0] == 0:
                print('marching: no surface extracted from the level set.')
                return None
            vertices = verts + np.array([x[0], y[0], zz[0]])
            colors = None
            if color_func is not None:
                vp = torch.from_numpy(vertices).float()
                cols = torch.cat([
                    self.eval_points(pn.to(device), color_func,
                                     self.bounding_box, device).cpu()[..., :3]
                    for pn in torch.split(vp, self.config.points_batch_size)])
                colors = (np.clip(cols.numpy(), 0, 1) * 255).astype(np.uint8)
            vertices = vertices / self.scale
            return Mesh(vertices, faces, colors)
