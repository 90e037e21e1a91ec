"""Geodesic basis for Mip-NeRF 360's integrated positional encoding: unit vectors of a
twice-tesselated icosahedron with mirror duplicates removed (21 directions), as
`generate_basis("icosahedron", 2)` builds them (models/mipnerf360/helper.py:396-531).
Product code: the drop-in module registers it as the `pos_basis_t` buffer."""
import numpy as np
import torch

_PHI = (np.sqrt(5.0) + 1.0) / 2.0
_VERTS = np.array([(-1, 0, _PHI), (1, 0, _PHI), (-1, 0, -_PHI), (1, 0, -_PHI), (0, _PHI, 1), (0, _PHI, -1),
                   (0, -_PHI, 1), (0, -_PHI, -1), (_PHI, 1, 0), (-_PHI, 1, 0), (_PHI, -1, 0), (-_PHI, -1, 0)])
_FACES = ((0, 4, 1), (0, 9, 4), (9, 5, 4), (4, 5, 8), (4, 8, 1), (8, 10, 1), (8, 3, 10), (5, 3, 8), (5, 2, 3),
          (2, 7, 3), (7, 10, 3), (7, 6, 10), (7, 11, 6), (11, 0, 6), (0, 1, 6), (6, 1, 10), (9, 0, 11), (9, 11, 2),
          (9, 2, 5), (7, 2, 11))


def _pair_sq_dist(a, b):
    return np.maximum(0.0, (a ** 2).sum(1)[:, None] + (b ** 2).sum(1)[None, :] - 2.0 * a @ b.T)


def icosahedron_basis(tesselation=2, tol=1e-4):
    """(3, n) fp32; n = 21 for tesselation 2.  Column order and the xyz -> zyx flip follow
    the reference so checkpoints' `pos_basis_t` buffers agree."""
    corners = _VERTS / np.sqrt(_PHI + 2.0)
    w = np.array([(i, j, tesselation - i - j) for i in range(tesselation + 1)
                  for j in range(tesselation + 1 - i)], dtype=np.float64) / tesselation
    cloud = np.concatenate([w @ corners[list(f)] for f in _FACES], axis=0)
    cloud /= np.linalg.norm(cloud, axis=1, keepdims=True)
    owner = np.array([np.flatnonzero(row <= tol)[0] for row in _pair_sq_dist(cloud, cloud)])
    cloud = cloud[np.unique(owner)]
    antipodal = _pair_sq_dist(cloud, -cloud) < tol
    cloud = cloud[np.triu(antipodal).any(axis=1)]
    return torch.from_numpy(np.ascontiguousarray(cloud[:, ::-1].T)).to(torch.float32)
