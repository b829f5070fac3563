"""Optimisation trajectory bookkeeping of estimate_local_motion -- same classes, fields and
JSON layout as the reference's optimization_state.py:6-144."""

from __future__ import annotations

import json

import torch


class OptimizationState:
    """One checkpoint: the (2, nt, nh, nw) field (moved to the CPU), its loss and step."""

    def __init__(self, deformation_field: torch.Tensor, loss: float, step: int):
        self.deformation_field = deformation_field.cpu()
        self.loss = loss
        self.step = step

    def as_dict(self) -> dict:
        return {"deformation_field": self.deformation_field.tolist(), "loss": self.loss, "step": self.step}


class OptimizationTracker:
    def __init__(self, sample_every_n_steps: int, total_steps: int):
        self.checkpoints: list[OptimizationState] = []
        self.sample_every_n_steps = sample_every_n_steps
        self.total_steps = total_steps

    def sample_this_step(self, step: int) -> bool:
        return step % self.sample_every_n_steps == 0 or step == self.total_steps - 1

    def add_checkpoint(self, deformation_field: torch.Tensor, loss: float, step: int) -> None:
        self.checkpoints.append(OptimizationState(deformation_field, loss, step))

    def as_dict(self) -> dict:
        return {
            "optimization_checkpoints": [cp.as_dict() for cp in self.checkpoints],
            "sample_every_n_steps": self.sample_every_n_steps,
            "total_steps": self.total_steps,
        }

    def to_json(self, filepath: str) -> None:
        with open(filepath, "w") as f:
            json.dump(self.as_dict(), f)
