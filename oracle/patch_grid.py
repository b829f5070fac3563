"""Oracle restatement of the reference's patch lattice and lazy patch gather.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED: tests/golden/
patch_grid_*.npz were produced by the reference's own patch_grid package
(oracle/make_goldens.py) and tests/test_oracle_patch_grid.py checks this file
against them.

Follows:
  patch_grid/_patch_grid_centers.py:70-111   (centres, "distribute" rounding)
  patch_grid/_patch_grid_indices.py:75-98    (centre -> index rows)
  patch_grid/_patch_grid.py:264-300,336-347,390-478  (int-key gather, 50-entry
      memo that hands out the *same tensor* again, evict-half rule)
"""

from __future__ import annotations

import torch


def centers_1d(dim_length: int, patch_length: int, patch_step: int, distribute: bool = True):
    lo = patch_length // 2
    hi = max(dim_length - lo - 1, lo)
    c = torch.arange(lo, hi + 1, step=patch_step)
    if distribute:
        slack = hi - c[-1]
        c = c + torch.round(torch.linspace(0, slack, steps=len(c))).long()
    return c


def index_rows(centers: torch.Tensor, patch_length: int):
    """(k,) centres -> (k, patch_length) pixel indices  c + arange(p) - p//2."""
    return centers[:, None] + (torch.arange(patch_length) - patch_length // 2)[None, :]


def centers_3d(image_shape, patch_shape, patch_step, distribute=True):
    """(gd, gh, gw, 3) integer (t, y, x) centres."""
    axes = [
        centers_1d(n, p, s, distribute)
        for n, p, s in zip(image_shape[-3:], patch_shape, patch_step)
    ]
    gd, gh, gw = (len(a) for a in axes)
    out = torch.empty((gd, gh, gw, 3), dtype=torch.long)
    out[..., 0] = axes[0][:, None, None]
    out[..., 1] = axes[1][None, :, None]
    out[..., 2] = axes[2][None, None, :]
    return out


class LazyPatches:
    """lazy[frame] -> (1, gh, gw, 1, ph, pw) gather with the reference's memo:
    a repeated key returns the *same* tensor object (so in-place edits by the caller
    persist), and when more than 50 keys are held, the first half of
    ``list(set_of_keys)`` is dropped."""

    LIMIT = 50

    def __init__(self, images, patch_shape, patch_step, distribute=True):
        assert len(patch_shape) == 3 and patch_shape[0] == 1
        self.images = images
        self.centers = centers_3d(images.shape, patch_shape, patch_step, distribute)
        self.rows_h = index_rows(self.centers[0, :, 0, 1], patch_shape[1])  # (gh, ph)
        self.rows_w = index_rows(self.centers[0, 0, :, 2], patch_shape[2])  # (gw, pw)
        self.memo: dict[int, torch.Tensor] = {}
        self.keys: set[int] = set()

    def __getitem__(self, frame: int) -> torch.Tensor:
        if frame in self.memo:
            return self.memo[frame]
        img = self.images[frame]
        gathered = img[self.rows_h[:, None, :, None], self.rows_w[None, :, None, :]]
        gathered = gathered[None, :, :, None]  # (1, gh, gw, 1, ph, pw)
        self.memo[frame] = gathered
        self.keys.add(frame)
        if len(self.memo) > self.LIMIT:
            for k in list(self.keys)[: len(self.keys) // 2]:
                self.memo.pop(k, None)
                self.keys.discard(k)
        return gathered


def patch_grid_lazy(images, patch_shape, patch_step, distribute_patches=True):
    lazy = LazyPatches(images, patch_shape, patch_step, distribute_patches)
    return lazy, lazy.centers
