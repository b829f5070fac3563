"""CPU restatement of ``estimate_local_motion`` (estimate_motion_optimizer.py:28-439) --
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference optimises a (2, nt, nh, nw) spline grid of shifts so that, patch by patch,
every Fourier-shifted frame agrees with the mean of the other frames.  Per iteration and
per batch of patches (estimate_motion_optimizer.py:361-417):

    P      = rfftn(patch * circle(radius pw/4, soft edge pw/4))          (b, t, ph, pw/2+1)
    s      = -(new(c) + initial(c)) / pixel_spacing                      (b, t, 2) px
    G      = fourier_shift(P, s) * bandpass * b_envelope                 :466-514
    ref_t  = (sum_t' G_t' - G_t) / (t - 1)                               :391-399
    loss   = _compute_loss(G, ref, ph, pw, loss_type)                    :611-671
    loss.backward()                      (gradients accumulate over the batches)

followed by one optimiser step.  This file follows that with torch autograd on the CPU.

Patch order: the reference draws the patches of every pass with ``random.shuffle`` on Python's
global ``random`` state (patch_utils.py:160-164; one call per pass, one per LBFGS closure
evaluation).  The same call is made here, so for the same ``random.seed`` the batches -- and with
``lbfgs_patch_subsample`` the patches that are used at all -- are the reference's
(tests/golden/reference_local_helpers.npz holds the reference's own order for a seed).

Deliberate difference, stated where it matters:
  * the parameters of the un-vendored spline package start at zero (its documented
    default); the third-party semantics of circle / b_envelope / bandpass / fourier
    shift / spline evaluation are the ones in oracle/thirdparty_semantics.py --
    **parity unpinned** there, as for the rest of the oracle.
"""

from __future__ import annotations

import random

import torch

from oracle import patch_grid as pg
from oracle import thirdparty_semantics as tp
from oracle.motion import normalize_image, prepare_bandpass_filter, resample_deformation_field


class OptimizationState:
    """optimization_state.py:6-49"""

    def __init__(self, deformation_field, loss, step):
        self.deformation_field = deformation_field.cpu()
        self.loss = loss
        self.step = step

    def as_dict(self):
        return {"deformation_field": self.deformation_field.tolist(), "loss": self.loss, "step": self.step}


class OptimizationTracker:
    """optimization_state.py:52-144"""

    def __init__(self, sample_every_n_steps, total_steps):
        self.checkpoints = []
        self.sample_every_n_steps = sample_every_n_steps
        self.total_steps = total_steps

    def sample_this_step(self, step):
        return step % self.sample_every_n_steps == 0 or step == self.total_steps - 1

    def add_checkpoint(self, deformation_field, loss, step):
        self.checkpoints.append(OptimizationState(deformation_field, loss, step))

    def as_dict(self):
        return {"optimization_checkpoints": [cp.as_dict() for cp in self.checkpoints],
                "sample_every_n_steps": self.sample_every_n_steps, "total_steps": self.total_steps}


def compute_loss(shifted, reference, ph, pw, loss_type="mse"):
    """estimate_motion_optimizer.py:611-671"""
    if loss_type == "mse":
        return torch.mean((shifted - reference).abs() ** 2) / (ph * pw)
    x = torch.fft.irfftn(shifted, s=(ph, pw), dim=(-2, -1))
    y = torch.fft.irfftn(reference, s=(ph, pw), dim=(-2, -1))
    if loss_type == "ncc":
        eps = 1e-8
        xc = x - x.mean(dim=(-2, -1), keepdim=True)
        yc = y - y.mean(dim=(-2, -1), keepdim=True)
        num = (xc * yc).sum(dim=(-2, -1))
        den = torch.sqrt((xc.square().sum(dim=(-2, -1)) + eps) * (yc.square().sum(dim=(-2, -1)) + eps))
        return -(num / den).mean()
    if loss_type == "cc":
        return -(x * y).sum(dim=(-2, -1)).mean()
    return None  # the reference falls off the end of the function for other names


def setup_optimizer(optimizer_type, parameters, **kw):
    """estimate_motion_optimizer.py:517-608 (same defaults)"""
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
    """Everything of estimate_local_motion that does not change between iterations."""

    def __init__(self, image, pixel_spacing, patch_shape, b_factor=500, frequency_range=(300, 10)):
        image = image.detach().to(torch.float32).cpu()
        self.t, self.h, self.w = image.shape
        self.ph, self.pw = patch_shape
        self.ps = float(pixel_spacing)
        img = normalize_image(image)  # :113
        centers = pg.centers_3d((self.t, self.h, self.w), (1, self.ph, self.pw),
                                (1, self.ph // 2, self.pw // 2), True)  # :116-122 -> (t, gh, gw, 3)
        self.centers = centers
        self.gh, self.gw = centers.shape[1:3]
        cn = centers.clone().float()  # patch_utils.py:89-93
        cn[..., 0] /= float(self.t - 1)
        cn[..., 1] /= float(self.h - 1)
        cn[..., 2] /= float(self.w - 1)
        self.centers_norm = cn.reshape(self.t, -1, 3)  # (t, npatch, 3)
        mask = tp.circle(self.pw / 4, (self.ph, self.pw), smoothing_radius=self.pw / 4)  # :162-167
        env = tp.b_envelope(b_factor, (self.ph, self.pw), self.ps, rfft=True, fftshift=False)  # :169-176
        band = prepare_bandpass_filter(frequency_range, (self.ph, self.pw), self.ps)  # :178-184
        self.filt = band * env
        pts = centers[0].reshape(-1, 3)
        patches = []
        for cp in pts:  # patch_utils.py:172-186
            y, x = int(cp[1]), int(cp[2])
            y0, x0 = y - self.ph // 2, x - self.pw // 2
            patches.append(img[:, y0:y0 + self.ph, x0:x0 + self.pw])
        self.spectra = torch.fft.rfftn(torch.stack(patches) * mask, dim=(-2, -1))  # (npatch, t, ph, pw/2+1)
        self.npatch = len(patches)

    def shifts_px(self, new_data, init_data, grid_type, idx):
        """:466-472 -- (b, t, 2) pixel shifts of patches `idx` for the current parameters."""
        c = self.centers_norm[:, idx]  # (t, b, 3)
        val = (tp.cubic_spline_grid_3d(new_data, c, grid_type, differentiable=True)
               + tp.cubic_spline_grid_3d(init_data, c, grid_type))
        return (-1 * val).transpose(0, 1) / self.ps

    def batch_loss(self, new_data, init_data, grid_type, idx, loss_type):
        s = self.shifts_px(new_data, init_data, grid_type, idx)
        G = tp.fourier_shift_dft_2d(self.spectra[idx], (self.ph, self.pw), s) * self.filt  # :475-489
        total = G.sum(dim=1, keepdim=True)
        ref = (total - G) / (self.t - 1) if self.t > 1 else G  # :391-399
        return compute_loss(G, ref, self.ph, self.pw, loss_type)


def estimate_local_motion(image, pixel_spacing, patch_shape, deformation_field_resolution,
                          initial_deformation_field=None, device=None, n_iterations=100, b_factor=500,
                          frequency_range=(300, 10), optimizer_type="adam", grid_type="catmull_rom",
                          loss_type="mse", optimizer_kwargs=None, return_trajectory=False,
                          trajectory_kwargs=None):
    """estimate_motion_optimizer.py:28-439, the reference's shuffled batches (module header)."""
    if grid_type not in ("catmull_rom", "bspline"):
        raise ValueError(f"Invalid grid type: {grid_type}. Must be 'catmull_rom' or 'bspline'.")
    prob = LocalMotionProblem(image, pixel_spacing, patch_shape, b_factor, frequency_range)
    res = tuple(int(r) for r in deformation_field_resolution)
    if return_trajectory:
        tk = dict(trajectory_kwargs or {})
        tk.setdefault("sample_every_n_steps", 1)
        tk.setdefault("total_steps", n_iterations)
        trajectory = OptimizationTracker(**tk)
    new = torch.zeros((2, *res), dtype=torch.float32, requires_grad=True)
    if initial_deformation_field is None:
        init = torch.zeros((2, *res), dtype=torch.float32)
    else:
        init = resample_deformation_field(initial_deformation_field.detach().cpu(), res).contiguous()
        init = init - torch.mean(init)  # :148
    okw = dict(optimizer_kwargs or {})
    opt = setup_optimizer(optimizer_type, [new], **okw)
    lbfgs = optimizer_type.lower() == "lbfgs"
    sub = okw.get("lbfgs_patch_subsample", None) if lbfgs else None
    for it in range(n_iterations):
        if lbfgs:
            def closure():
                opt.zero_grad()
                tot, n = None, 0
                order = list(range(prob.npatch))
                random.shuffle(order)  # get_iterator(batch_size=1, randomized=True), :295-297
                for pos, b in enumerate(order):  # :287-324, batch_size=1
                    if sub is not None and pos >= sub:
                        break
                    l = prob.batch_loss(new, init, grid_type, [b], loss_type)
                    tot = l if tot is None else tot + l
                    n += 1
                if n == 0:
                    return torch.tensor(0.0, requires_grad=True)
                avg = tot / n
                avg.backward()
                return avg
            avg = float(opt.step(closure).detach())
        else:
            total, n = 0.0, 0
            order = list(range(prob.npatch))
            random.shuffle(order)  # get_iterator(batch_size=8), randomized by default (:362)
            for a in range(0, prob.npatch, 8):  # :361 batch_size=8
                idx = order[a : a + 8]
                l = prob.batch_loss(new, init, grid_type, idx, loss_type)
                l.backward()
                total += l.item()
                n += 1
            opt.step()
            opt.zero_grad()
            avg = total / n if n else 0.0
        if return_trajectory and trajectory.sample_this_step(it):
            trajectory.add_checkpoint(new.detach().clone(), avg, it)
    final = new.detach() + init  # :430-432
    final = final - torch.mean(final)
    return (final, trajectory) if return_trajectory else final
