"""Oracle: sinusoidal positional encoding.  Test infrastructure (oracle/__init__.py)."""
import numpy as np
import torch


def pos_enc(x, min_deg, max_deg):
    """[x | sin(x 2^k) for k | sin(x 2^k + pi/2) for k], k-major / channel-minor.

    Follows neo360/helper.py:121-125 == vanilla_nerf/helper.py:445-449.  The
    phase is added in fp32 (python float 0.5*pi rounded to fp32 by the add), and
    the shifted sine stands in for cosine.  Width = C*(2*(max_deg-min_deg)+1).
    """
    freq = torch.tensor([2 ** k for k in range(min_deg, max_deg)]).type_as(x)
    scaled = (x[..., None, :] * freq[:, None]).reshape(list(x.shape[:-1]) + [-1])
    waves = torch.sin(torch.cat([scaled, scaled + 0.5 * np.pi], dim=-1))
    return torch.cat([x, waves], dim=-1)
