"""Deformation-field CSV interchange, wire-compatible with the
reference's data_io.py:10-141: header ``t,h,w,y_shift,x_shift``, one row per control point
in (t, h, w) order, channel 0 = y shift, channel 1 = x shift.  Host-side only (a field is a
few hundred numbers); written without the reference's per-element Python loops.
"""

from __future__ import annotations

from pathlib import Path
from typing import Union

import numpy as np
import torch

from .api import _say


def write_deformation_field_to_csv(deformation_field: torch.Tensor, output_path: Union[str, Path]) -> None:
    """(2, t, h, w) field -> CSV (data_io.py:10-72).  Values are written as the float64
    images of the float32 entries, exactly what ``.item()`` + pandas give in the reference."""
    import pandas as pd

    if deformation_field.dim() != 4 or deformation_field.shape[0] != 2:
        raise ValueError(f"expected a (2, t, h, w) deformation field, got {tuple(deformation_field.shape)}")
    f = deformation_field.detach().to("cpu", torch.float32).numpy().astype(np.float64)
    _, t, h, w = f.shape
    ti, hi, wi = np.meshgrid(np.arange(t), np.arange(h), np.arange(w), indexing="ij")
    df = pd.DataFrame({"t": ti.ravel(), "h": hi.ravel(), "w": wi.ravel(),
                       "y_shift": f[0].ravel(), "x_shift": f[1].ravel()})
    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    df.to_csv(output_path, index=False)
    _say(f"Deformation field written to {output_path}")
    _say(f"CSV contains {len(df)} rows with {t} time points and {h * w} spatial positions per time point")


def read_deformation_field_from_csv(csv_path: Union[str, Path], device: torch.device = None) -> torch.Tensor:
    """CSV -> (2, t, h, w) float32 field on `device` (default CPU, as the reference:
    data_io.py:75-141).  Index values need not be contiguous: like the reference, each
    axis is the sorted set of the values that occur; missing combinations stay zero."""
    import pandas as pd

    if device is None:
        device = torch.device("cpu")
    df = pd.read_csv(csv_path)
    missing = {"t", "h", "w", "y_shift", "x_shift"} - set(df.columns)
    if missing:
        raise KeyError(f"{csv_path}: missing column(s) {sorted(missing)}")
    axes = [np.unique(df[c].to_numpy()) for c in ("t", "h", "w")]
    pos = [np.searchsorted(a, df[c].to_numpy()) for a, c in zip(axes, ("t", "h", "w"))]
    field = np.zeros((2, len(axes[0]), len(axes[1]), len(axes[2])), dtype=np.float32)
    field[0, pos[0], pos[1], pos[2]] = df["y_shift"].to_numpy(dtype=np.float32)
    field[1, pos[0], pos[1], pos[2]] = df["x_shift"].to_numpy(dtype=np.float32)
    _say(f"Detected dimensions: t={field.shape[1]}, h={field.shape[2]}, w={field.shape[3]}")
    return torch.from_numpy(field).to(device)
