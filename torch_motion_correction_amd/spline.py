"""Host tables for uniform cubic spline grids (the torch_cubic_spline_grids
semantics the reference relies on, deformation_field_utils.py:30-38).

A coordinate u in [0,1] on an axis with n samples lies in interval il (found with
searchsorted on linspace(0,1,n), clipped so that u == 1 uses the last interval); the
value is [1,s,s^2,s^3] @ M @ (4 control samples il-1..il+2), where samples outside
the axis are linear extrapolations.  That extrapolation is folded into the weights
here, so the device kernel only sees 4 in-range (index, weight) taps per axis.
Axes with a single sample are constant.
"""

from __future__ import annotations

import torch

_M = {
    "bspline": (1.0 / 6.0)
    * torch.tensor([[1, 4, 1, 0], [-3, 0, 3, 0], [3, -6, 3, 0], [-1, 3, -3, 1]], dtype=torch.float32),
    "catmull_rom": 0.5
    * torch.tensor([[0, 2, 0, 0], [-1, 0, 1, 0], [2, -5, 4, -1], [-1, 3, -3, 1]], dtype=torch.float32),
}


def basis_matrix(grid_type: str) -> torch.Tensor:
    if grid_type not in _M:
        raise ValueError(f"unknown grid_type {grid_type!r} (expected 'catmull_rom' or 'bspline')")
    return _M[grid_type]


def axis_taps(n: int, u: torch.Tensor, grid_type: str):
    """u: (m,) float32 CPU coordinates in [0,1].  Returns idx (m,4) int32 into the real
    axis of length n and w (m,4) float32."""
    M = basis_matrix(grid_type)
    u = u.detach().to(torch.float32).cpu().contiguous()
    n_eff = max(n, 2)
    pos = torch.linspace(0, 1, steps=n_eff)
    iu = torch.searchsorted(pos, u, side="right")
    il = torch.clamp(iu - 1, 0, n_eff - 2)
    s = (u - pos[il]) * float(n_eff - 1)
    basis = torch.stack([torch.ones_like(s), s, s * s, s * s * s], dim=-1) @ M  # (m,4)
    w = basis.clone()
    # tap k sits at padded position il-1+k; fold p=-1 -> 2*d[0]-d[1], p=n -> 2*d[n-1]-d[n-2]
    lo = il == 0
    w[lo, 1] += 2 * basis[lo, 0]
    w[lo, 2] -= basis[lo, 0]
    w[lo, 0] = 0
    hi = il + 2 == n_eff
    w[hi, 2] += 2 * basis[hi, 3]
    w[hi, 1] -= basis[hi, 3]
    w[hi, 3] = 0
    idx = torch.clamp(il[:, None] - 1 + torch.arange(4)[None, :], 0, n_eff - 1)
    idx = torch.clamp(idx, max=n - 1)  # duplicated single-sample axis -> sample 0
    return idx.to(torch.int32).contiguous(), w.contiguous()
