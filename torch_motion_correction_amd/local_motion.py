"""estimate_local_motion: spline-field refinement by gradient descent on the agreement of
Fourier-shifted patches (reference: estimate_motion_optimizer.py:28-439).

What the reference recomputes in every iteration -- rfftn of every masked patch, the
filters -- is done once here (pruned HIP transforms of engine._forward_spectra); an
iteration is then one or two launches of csrc/local_motion.hip over those spectra plus a
handful of tiny tensor operations (spline basis product, torch.optim step).  The analytic
gradient replaces autograd; the optimisers are torch.optim's own, fed through a
torch.autograd.Function, so 'adam', 'sgd', 'rmsprop' and 'lbfgs' behave as in the reference
(same defaults, estimate_motion_optimizer.py:517-608).

Patch order: the reference shuffles the patches of every pass with ``random.shuffle`` on Python's
global ``random`` state (patch_utils.py:160-164: once per pass, once per LBFGS closure evaluation).
The order only enters through which patches share the last, smaller batch of a pass (a batch's loss
is the mean over its own patches) and, with ``lbfgs_patch_subsample``, through which patches are
used at all.  The same ``random.shuffle`` call is made here, so for the same ``random.seed`` the
per-patch weights are the reference's and the global random state advances as it does there.
"""

from __future__ import annotations

import math
import random

import numpy as np
import torch

from . import _lib, engine, lattice
from . import plan as planmod
from ._lib import check, ptr, stream_ptr
from .optimization_state import OptimizationTracker

BATCH = 8  # estimate_motion_optimizer.py:361


def setup_optimizer(optimizer_type: str, parameters, **kw):
    """Same names and defaults as estimate_motion_optimizer.py:517-608."""
    name = optimizer_type.lower()
    if name == "adam":
        return torch.optim.Adam(parameters, lr=kw.get("lr", 0.01), betas=kw.get("betas", (0.9, 0.999)),
                                eps=kw.get("eps", 1e-08), weight_decay=kw.get("weight_decay", 0),
                                amsgrad=kw.get("amsgrad", False))
    if name == "sgd":
        return torch.optim.SGD(parameters, lr=kw.get("lr", 0.01), momentum=kw.get("momentum", 0.9),
                               weight_decay=kw.get("weight_decay", 0), dampening=kw.get("dampening", 0),
                               nesterov=kw.get("nesterov", True))
    if name == "rmsprop":
        return torch.optim.RMSprop(parameters, lr=kw.get("lr", 0.01), alpha=kw.get("alpha", 0.99),
                                   eps=kw.get("eps", 1e-08), weight_decay=kw.get("weight_decay", 0),
                                   momentum=kw.get("momentum", 0), centered=kw.get("centered", False))
    if name == "lbfgs":
        max_iter = int(kw.get("max_iter", 1))
        max_eval = kw.get("max_eval", None)
        if max_eval is None:
            max_eval = max(1, int(max_iter * 1.25))
        return torch.optim.LBFGS(parameters, lr=kw.get("lr", 1), max_iter=max_iter, max_eval=max_eval,
                                 tolerance_grad=kw.get("tolerance_grad", 1e-11),
                                 tolerance_change=kw.get("tolerance_change", 1e-11),
                                 history_size=kw.get("history_size", 5),
                                 line_search_fn=kw.get("line_search_fn", "strong_wolfe"))
    raise ValueError(f"Unsupported optimizer: {optimizer_type}. Choose 'adam', 'sgd', 'rmsprop', or 'lbfgs'.")


class LocalMotionProblem:
    """Iteration-invariant state: pruned patch spectra on the device, the bins' frequencies,
    the spline basis at the patch centres."""

    def __init__(self, img: torch.Tensor, pixel_spacing: float, patch_shape, resolution, grid_type: str,
                 b_factor: float = 500, frequency_range=(300, 10)):
        self.lib = _lib.load()
        dev = img.device
        t, h, w = img.shape
        ph, pw = int(patch_shape[0]), int(patch_shape[1])
        if t < 2:
            raise ValueError("estimate_local_motion needs at least 2 frames (patch centres are normalised by t - 1)")
        if ph > h or pw > w:
            raise ValueError(f"Patch size {(ph, pw)} too large for image of shape {(t, h, w)}")
        if t > 512:
            raise NotImplementedError(f"{t} frames: the loss kernels keep the per-frame shifts of a patch in LDS "
                                      "(mc_local_loss_sums: at most 512 frames)")
        self.t, self.h, self.w, self.ph, self.pw, self.dev = t, h, w, ph, pw, dev
        self.ps = float(pixel_spacing)
        self.res = tuple(int(r) for r in resolution)
        pl = planmod.get_xc_plan(ph, pw, self.ps, b_factor, frequency_range, dev,
                                 mask_radius=pw / 4, mask_smoothing=pw / 4)  # :162-167
        g = pl.geom
        self.nkx, self.nky = g.nkx, g.nky
        cy = lattice.patch_centers_1d(h, ph, ph // 2)  # :116-122
        cx = lattice.patch_centers_1d(w, pw, pw // 2)
        self.cy, self.cx = cy, cx
        self.gh, self.gw = len(cy), len(cx)
        self.npatch = self.gh * self.gw
        origin = ((cy[:, None] - ph // 2) * w + (cx[None, :] - pw // 2)).reshape(-1).astype(np.int64)
        off = (origin[:, None] + np.arange(t, dtype=np.int64)[None, :] * (h * w)).reshape(-1)  # job = patch * t + frame
        ones = torch.ones(off.size, dtype=torch.int32, device=dev)
        stats = engine.central_box_stats(img)  # normalize_image (:113) is applied inside the row pass
        self.spectra = engine._forward_spectra(img, engine._i64(off, dev), w, ones, pl, stats, min_expo=1)
        rows = np.concatenate([np.arange(g.kyp), np.arange(ph - g.kyn, ph)])
        kk = np.where(rows < (ph + 1) // 2, rows, rows - ph).astype(np.float32)
        self.fy = torch.from_numpy(kk * np.float32(1.0 / ph)).to(dev)
        kx = np.arange(g.nkx)
        self.fx = torch.from_numpy(kx.astype(np.float32) * np.float32(1.0 / pw)).to(dev)
        herm = np.where((kx == 0) | ((pw % 2 == 0) & (kx == pw // 2)), 1.0, 2.0).astype(np.float32)
        self.hx = torch.from_numpy(herm).to(dev)
        nt = torch.zeros(1, dtype=torch.int32)
        check(self.lib.mc_local_loss_tiles(g.nkx, g.nky, nt.data_ptr()), "mc_local_loss_tiles")
        self.ntiles = int(nt.item())
        # spline basis at the normalised patch centres (patch_utils.py:89-93): A[(b, f), ctrl]
        nctrl = self.res[0] * self.res[1] * self.res[2]
        eye = torch.eye(nctrl, dtype=torch.float32, device=dev).reshape(nctrl, *self.res)
        ut = torch.arange(t, dtype=torch.float32) / float(t - 1)
        uy = torch.from_numpy(cy.astype(np.float32)) / float(h - 1)
        ux = torch.from_numpy(cx.astype(np.float32)) / float(w - 1)
        basis = engine.spline_lattice(eye, ut, uy, ux, grid_type)  # (nctrl, t, gh, gw)
        self.A = basis.permute(2, 3, 1, 0).reshape(self.npatch * t, nctrl).contiguous()

    # ---- kernels
    def sums(self, shifts_px: torch.Tensor, hermitian: bool) -> torch.Tensor:
        """(npatch, t, 6) sums of csrc/local_motion.hip for (npatch, t, 2) pixel shifts."""
        s = shifts_px.detach().to(torch.float32).contiguous()
        part = torch.empty((self.npatch, self.ntiles, self.t, 6), dtype=torch.float32, device=self.dev)
        check(self.lib.mc_local_loss_sums(ptr(self.spectra), ptr(s), ptr(self.fy), ptr(self.fx),
                                          ptr(self.hx) if hermitian else None, self.npatch, self.t,
                                          self.nkx, self.nky, ptr(part), stream_ptr(self.dev)),
              "mc_local_loss_sums")
        return part.sum(dim=1, dtype=torch.float64)

    def ncc_grad_sums(self, shifts_px: torch.Tensor, ab: torch.Tensor) -> torch.Tensor:
        s = shifts_px.detach().to(torch.float32).contiguous()
        ab = ab.to(torch.float32).contiguous()
        part = torch.empty((self.npatch, self.ntiles, self.t, 2), dtype=torch.float32, device=self.dev)
        check(self.lib.mc_local_ncc_grad(ptr(self.spectra), ptr(s), ptr(self.fy), ptr(self.fx), ptr(self.hx),
                                         ptr(ab), self.npatch, self.t, self.nkx, self.nky, ptr(part),
                                         stream_ptr(self.dev)), "mc_local_ncc_grad")
        return part.sum(dim=1, dtype=torch.float64)

    def loss_and_grad(self, shifts_px: torch.Tensor, wb: torch.Tensor, loss_type: str):
        """Loss (0-d float64 tensor) = sum_b wb[b] * (per-patch mean loss of the reference's
        _compute_loss for a batch of one), and its gradient (npatch, t, 2) w.r.t. the pixel shifts.
        wb (npatch,) float64 carries the batch normalisation (1 / batch size, 0 = unused patch)."""
        t, n = self.t, self.ph * self.pw
        if loss_type == "mse":  # :626
            q = self.sums(shifts_px, hermitian=False)
            c = wb / (t * self.ph * (self.pw // 2 + 1) * n)
            loss = (c * q[:, :, 2].sum(1)).sum() / (t - 1) ** 2
            grad = (-4.0 * math.pi * t / (t - 1) ** 2) * c[:, None, None] * q[:, :, 0:2]
        elif loss_type == "cc":  # :657-671
            q = self.sums(shifts_px, hermitian=True)
            c = wb / (t * n * (t - 1))
            loss = -(c * q[:, :, 3].sum(1)).sum()
            grad = (-4.0 * math.pi) * c[:, None, None] * q[:, :, 0:2]
        elif loss_type == "ncc":  # :627-656 (the band-pass removes DC, so the means are zero)
            q = self.sums(shifts_px, hermitian=True)
            ex = q[:, :, 5] / n
            eps = 1e-8
            with torch.enable_grad():  # dL/dnum, dL/dey of a (npatch, t) formula: tiny, let torch do it
                num = (q[:, :, 3] / ((t - 1) * n)).requires_grad_(True)
                ey = (q[:, :, 4] / ((t - 1) ** 2 * n)).requires_grad_(True)
                loss = -((wb / t)[:, None] * num / torch.sqrt((ex + eps) * (ey + eps))).sum()
                a, b = torch.autograd.grad(loss, (num, ey))
            g = self.ncc_grad_sums(shifts_px, torch.stack([a, b], dim=-1))
            grad = (2.0 * math.pi / n) * g
            loss = loss.detach()
        else:
            raise ValueError(f"Unknown loss_type: {loss_type}. Must be 'mse', 'cc' or 'ncc'.")
        return loss, grad

    def shifts_px(self, new: torch.Tensor, init: torch.Tensor) -> torch.Tensor:
        """:466-472 -- (npatch, t, 2) = -(new(c) + initial(c)) / pixel_spacing at the patch centres."""
        val = (new + init).reshape(2, -1) @ self.A.t()  # (2, npatch * t)
        return (-val / self.ps).t().reshape(self.npatch, self.t, 2)


class _Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, shifts_px, prob, wb, loss_type):
        loss, grad = prob.loss_and_grad(shifts_px, wb, loss_type)
        ctx.save_for_backward(grad.to(shifts_px.dtype))
        return loss.to(shifts_px.dtype)

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return gout * grad, None, None, None


def estimate_local_motion(img: torch.Tensor, pixel_spacing, patch_shape, deformation_field_resolution,
                          init_data: torch.Tensor, n_iterations, b_factor, frequency_range, optimizer_type,
                          grid_type, loss_type, optimizer_kwargs, trajectory: OptimizationTracker | None):
    """Driver loop of estimate_motion_optimizer.py:219-432 on device tensors.  `init_data` is the
    (2, nt, nh, nw) initial field already resampled and mean-subtracted (or zeros)."""
    if loss_type not in ("mse", "cc", "ncc"):
        raise ValueError(f"Unknown loss_type: {loss_type}. Must be 'mse', 'cc' or 'ncc'.")
    prob = LocalMotionProblem(img, pixel_spacing, patch_shape, deformation_field_resolution, grid_type,
                              b_factor, frequency_range)
    dev = img.device
    new = torch.zeros((2, *prob.res), dtype=torch.float32, device=dev, requires_grad=True)
    okw = dict(optimizer_kwargs or {})
    opt = setup_optimizer(optimizer_type, [new], **okw)
    lbfgs = optimizer_type.lower() == "lbfgs"
    npatch = prob.npatch
    # Per-patch weights of one pass.  The reference shuffles the patch list (random.shuffle, once
    # per pass / closure evaluation) and cuts it into batches: a patch's weight is 1 / (size of the
    # batch its POSITION in the shuffled list falls into); under LBFGS the first `used` positions
    # are single-patch batches averaged, the rest is skipped.
    if lbfgs:  # closure: batches of one patch, averaged (:287-336)
        sub = okw.get("lbfgs_patch_subsample", None)
        used = npatch if sub is None else max(0, min(int(sub), npatch))
        by_pos = np.zeros(npatch, dtype=np.float64)
        if used:
            by_pos[:used] = 1.0 / used
        nbatch = 1
    else:  # batches of 8, each a mean over its own patches, gradients accumulated (:361-417)
        sizes = np.minimum(BATCH, npatch - (np.arange(npatch) // BATCH) * BATCH)
        by_pos = 1.0 / sizes.astype(np.float64)
        nbatch = (npatch + BATCH - 1) // BATCH
    uniform = bool(np.all(by_pos == by_pos[0]))
    wb_const = torch.from_numpy(by_pos).to(dev)

    def pass_weights():
        order = list(range(npatch))
        random.shuffle(order)  # always: the global random state must advance as in the reference
        if uniform:
            return wb_const
        w = np.empty(npatch, dtype=np.float64)
        w[np.asarray(order, dtype=np.int64)] = by_pos
        return torch.from_numpy(w).to(dev)

    for it in range(int(n_iterations)):
        if lbfgs:
            def closure():
                opt.zero_grad()
                wb = pass_weights()
                if not used:
                    return torch.tensor(0.0, device=dev, requires_grad=True)
                loss = _Loss.apply(prob.shifts_px(new, init_data), prob, wb, loss_type)
                loss.backward()
                return loss
            avg = opt.step(closure)
        else:
            wb = pass_weights()
            loss = _Loss.apply(prob.shifts_px(new, init_data), prob, wb, loss_type)
            loss.backward()
            opt.step()
            opt.zero_grad()
            avg = loss.detach() / nbatch
        if trajectory is not None and trajectory.sample_this_step(it):
            trajectory.add_checkpoint(new.detach().clone(), float(avg.detach()), it)
    final = new.detach() + init_data  # :430-432
    return final - torch.mean(final)
