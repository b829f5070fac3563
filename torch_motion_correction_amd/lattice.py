"""Patch lattice and the reference's lazy-patch memo schedule (host, integers only).

* ``patch_centers_1d`` follows patch_grid/_patch_grid_centers.py:70-111 (centres at
  p//2 + k*step, then spread over the slack with rounded linspace offsets).
* ``mask_schedule`` replays the bookkeeping of LazyPatchGrid
  (patch_grid/_patch_grid.py:264-300, 336-347: memoised gathers handed out by
  reference, more than 50 keys -> drop the first half of ``list(set_of_keys)``)
  under the access pattern of the reference's frame loop
  (estimate_motion_xc.py:297-346), where ``patches *= mask`` mutates whatever the
  memo handed out.  No pixel data is involved: the result is, for every processed
  frame, how many times each memo entry had already been multiplied by the mask
  when it was read.  CPython's ``set`` ordering of small ints is
  what decides the eviction order, so a real ``set`` is used here too.
"""

from __future__ import annotations

import numpy as np
import torch

MEMO_LIMIT = 50


def patch_centers_1d(dim_length: int, patch_length: int, patch_step: int) -> np.ndarray:
    first = patch_length // 2
    last = max(dim_length - first - 1, first)
    centers = torch.arange(first, last + 1, step=patch_step)
    slack = last - centers[-1]
    centers = centers + torch.round(torch.linspace(0, slack, steps=len(centers))).long()
    return centers.numpy().astype(np.int64)


def patch_grid_centers(t: int, h: int, w: int, p: int) -> tuple[np.ndarray, np.ndarray]:
    """50 %-overlap lattice of p x p patches (estimate_motion_xc.py:251-257)."""
    return patch_centers_1d(h, p, p // 2), patch_centers_1d(w, p, p // 2)


def centers_tensor(t: int, cy: np.ndarray, cx: np.ndarray) -> torch.Tensor:
    """(t, gh, gw, 3) int64 (frame, y, x) centres as the reference returns them."""
    out = torch.empty((t, len(cy), len(cx), 3), dtype=torch.int64)
    out[..., 0] = torch.arange(t)[:, None, None]
    out[..., 1] = torch.from_numpy(cy)[None, :, None]
    out[..., 2] = torch.from_numpy(cx)[None, None, :]
    return out


class _Memo:
    """exponent bookkeeping of one LazyPatchGrid"""

    def __init__(self):
        self.expo: dict[int, int] = {}
        self.keys: set[int] = set()

    def get(self, frame: int) -> int:
        """touch entry `frame` (gathering it fresh if absent); return its exponent"""
        if frame in self.expo:
            return self.expo[frame]
        self.expo[frame] = 0
        self.keys.add(frame)
        if len(self.expo) > MEMO_LIMIT:
            for k in list(self.keys)[: len(self.keys) // 2]:
                self.expo.pop(k, None)
                self.keys.discard(k)
        return 0


def mask_schedule(t: int, reference_strategy: str, reference_frame: int, with_ref_reads: bool = False):
    """`reference_frame` is the memo KEY as the caller gave it: the reference's memo is a dict keyed
    by the raw Python int, so -1 and t-1 are two entries (gathered from the same frame) with
    exponents of their own, and a negative key never equals a loop index (no frame is skipped).
    Returns (ref_expo, cur_expo, processed) and, `with_ref_reads`, also ref_read[f] = exponent of
    the reference entry when frame f read it (middle_frame; -1 = not read):
    ref_expo[f][o]  exponent of entry o when it was read to build frame f's reference
                    (-1 = not read);
    cur_expo[f]     exponent of entry f when it was read as the current frame;
    processed       list of frames the loop handles, in order."""
    memo = _Memo()
    ref_expo = np.full((t, t), -1, dtype=np.int64)
    cur_expo = np.full((t,), -1, dtype=np.int64)
    processed = []
    ref_read = np.full((t,), -1, dtype=np.int64)
    ref_col = reference_frame % t if t > 0 else 0
    # tensors handed out by the memo stay alive (and keep being mutated) while the
    # caller holds them even if the memo evicted them: model them as boxes.
    boxes: dict[int, list[int]] = {}

    def fetch(frame: int) -> list[int]:
        had = frame in memo.expo
        memo.get(frame)
        if not had:
            boxes[frame] = [0]  # fresh gather from the (unmodified) image
        box = boxes[frame]
        if frame not in memo.expo:
            # evicted immediately by its own insertion: the caller still holds the box,
            # but the next fetch will gather a fresh one
            pass
        return box

    for f in range(t):
        if reference_strategy == "middle_frame":
            if f == reference_frame:
                continue
            rbox = fetch(reference_frame)
            ref_expo[f, ref_col] = rbox[0]
            ref_read[f] = rbox[0]
        elif reference_strategy == "mean_except_current":
            rbox = None
            for o in range(t):
                if o != f:
                    ref_expo[f, o] = fetch(o)[0]
        else:
            raise ValueError(f"Unknown reference_strategy: {reference_strategy}")
        cbox = fetch(f)
        cur_expo[f] = cbox[0]
        if rbox is not None:
            rbox[0] += 1  # ref_patches *= mask on the memo's own tensor
        cbox[0] += 1  # frame_patches *= mask on the memo's own tensor
        processed.append(f)
    if with_ref_reads:
        return ref_expo, cur_expo, processed, ref_read
    return ref_expo, cur_expo, processed


def leave_one_out_schedule(ref_expo: np.ndarray):
    """Turn the (t,t) exponent table of `mask_schedule` (mean_except_current) into the
    incremental schedule libmcorr's reference-spectrum kernel consumes: S_f = {o != f :
    ref_expo[f,o] == 1}; frame f lists the members to ADD to S_{f-1} when S_f is a
    superset of it, else (an entry was reset by the memo eviction) all of S_f with
    rebuild=1.  Returns (ptr[t+1], idx[...], rebuild[t]) as int32/int32/uint8 arrays."""
    t = ref_expo.shape[0]
    ptr, idx, rebuild = [0], [], []
    prev: set[int] = set()
    for f in range(t):
        cur = {o for o in range(t) if o != f and ref_expo[f, o] == 1}
        if f > 0 and f not in prev and prev <= cur:
            add = sorted(cur - prev)
            rebuild.append(0)
        else:
            add = sorted(cur)
            rebuild.append(1)
        idx.extend(add)
        ptr.append(len(idx))
        prev = cur
    return (np.asarray(ptr, dtype=np.int32), np.asarray(idx if idx else [0], dtype=np.int32),
            np.asarray(rebuild, dtype=np.uint8))
