"""Movie-level software pipeline on one GPU.

The reference processes one movie at a time: estimate_global_motion -> correct_motion
(-> sum) (examples/ttMotion.py:284-398).  Movies are independent, so when several are
processed back to back the estimator of movie k+1 (latency- and issue-bound FFT kernels)
can run on a second HIP stream underneath the HBM-bound warp of movie k.  Results are
identical to calling the two API functions one after the other; only the enqueue order
differs.  No host synchronisation happens here: the caller's stream waits on both
pipeline streams at the end of ``run``.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterable, Optional

import torch

from . import engine
from ._lib import device_scope, require_gpu


@dataclass
class MovieResult:
    field: torch.Tensor  # (2, t, 1, 1) Angstrom, as estimate_global_motion returns it
    total: torch.Tensor  # (h, w) sum of the corrected frames
    frames: Optional[torch.Tensor]  # (t, h, w) corrected frames or None


class MoviePipeline:
    """estimate_global_motion -> correct_motion (+ fused frame sum) over a sequence of
    movies, two HIP streams deep.  Arguments mirror estimate_global_motion
    (estimate_motion_xc.py:21-28) and correct_motion (correct_motion.py:18-25)."""

    def __init__(self, device=None, pixel_spacing: float = 1.0, reference_frame: Optional[int] = None,
                 b_factor: float = 500, frequency_range=(300, 10), grid_type: str = "catmull_rom",
                 return_frames: bool = True, overlap: bool = True):
        self.device = require_gpu(device)
        self.pixel_spacing = float(pixel_spacing)
        self.reference_frame = reference_frame
        self.b_factor = float(b_factor)
        self.frequency_range = tuple(frequency_range)
        self.grid_type = grid_type
        self.return_frames = return_frames
        self.overlap = overlap
        # the estimator's stream gets the higher priority: its short, latency-bound kernels then
        # slot in between the waves of the long HBM-bound warp instead of queueing behind them
        # (measured: 19.4-19.7 k -> 20.1-20.4 k frames/s on 40 x 4096^2 stacks)
        import os

        pe, pw = (int(x) for x in os.environ.get("MC_PIPE_PRIORITIES", "-1,0").split(","))
        self._s_est = torch.cuda.Stream(self.device, priority=pe) if overlap else None
        self._s_warp = torch.cuda.Stream(self.device, priority=pw) if overlap else None

    # the two stages, each enqueued on whatever stream is current
    def _estimate(self, img: torch.Tensor) -> torch.Tensor:
        t = img.shape[0]
        ref = t // 2 if self.reference_frame is None else int(self.reference_frame)
        shifts = engine.global_shifts(img, ref, self.pixel_spacing, self.b_factor, self.frequency_range)
        return (shifts * self.pixel_spacing).transpose(0, 1)[:, :, None, None]  # dfu.py:129-162

    def _prepare(self, img: torch.Tensor, field: torch.Tensor):
        """Everything of correct_motion that only needs the field: the per-frame lattices and the rigid
        warp's weight tables (six launches of a few microseconds each).  Enqueued on the ESTIMATOR's
        stream, behind its last kernel: they then run under the previous movie's warp instead of
        holding up this movie's (the warp stream idled 56 us between two launches, 3 % of a step)."""
        lat = engine.frame_lattices(field.contiguous(), img.shape[0], self.grid_type)
        return engine.rigid_tables(img, lat, self.pixel_spacing)

    def _correct(self, img: torch.Tensor, tables):
        return engine.warp(img, None, self.pixel_spacing, want_frames=self.return_frames, want_sum=True,
                           rigid=True, tables=tables)

    def iterate(self, movies: Iterable[torch.Tensor],
                around_warp: Optional[Callable[[Callable[[], object]], object]] = None):
        """Generator form of ``run``: yields a MovieResult as soon as the movie's work has been
        ENQUEUED.  A yielded result is ordered on the pipeline's warp stream; the caller's
        stream only waits for the pipeline once the generator is exhausted, so use the tensors
        after the loop (or drop them: memory returns to the warp stream's pool, which the
        next movie reuses).  `around_warp(fn)` (optional, for instrumentation) must call fn()
        and return its result; it runs with the warp stream current."""
        dev = self.device
        call = around_warp if around_warp is not None else (lambda fn: fn())
        # every enqueue happens with the pipeline's GPU as the current HIP device (libmcorr launches
        # on the current device); the scope is left again before control returns to the caller
        if not self.overlap:
            for img in movies:
                with device_scope(dev):
                    img = self._check(img)
                    field = self._estimate(img)
                    tables = self._prepare(img, field)
                    frames, total = call(lambda: self._correct(img, tables))
                yield MovieResult(field, total, frames)
            return
        with device_scope(dev):
            caller = torch.cuda.current_stream(dev)
            start = torch.cuda.Event()
            start.record(caller)
            self._s_est.wait_event(start)
            self._s_warp.wait_event(start)
        try:
            for img in movies:
                with device_scope(dev):
                    img = self._check(img)
                    img.record_stream(self._s_est)
                    img.record_stream(self._s_warp)
                    with torch.cuda.stream(self._s_est):
                        field = self._estimate(img)
                        tables = self._prepare(img, field)
                        ready = torch.cuda.Event()
                        ready.record(self._s_est)
                    with torch.cuda.stream(self._s_warp):
                        self._s_warp.wait_event(ready)
                        field.record_stream(self._s_warp)
                        for x in tables:
                            if isinstance(x, torch.Tensor):
                                x.record_stream(self._s_warp)
                        frames, total = call(lambda: self._correct(img, tables))
                yield MovieResult(field, total, frames)
        finally:
            with device_scope(dev):
                for s in (self._s_est, self._s_warp):
                    done = torch.cuda.Event()
                    done.record(s)
                    caller.wait_event(done)

    def run(self, movies: Iterable[torch.Tensor],
            around_warp: Optional[Callable[[Callable[[], object]], object]] = None) -> list[MovieResult]:
        """Process `movies` ((t,h,w) float32 or float16 tensors on the pipeline's device) in order and
        return every result, ready for use on the caller's current stream."""
        results = list(self.iterate(movies, around_warp))
        if self.overlap:
            caller = torch.cuda.current_stream(self.device)
            for r in results:
                for x in (r.field, r.total, r.frames):
                    if x is not None:
                        x.record_stream(caller)
        return results

    def _check(self, img: torch.Tensor) -> torch.Tensor:
        if img.dim() != 3:
            raise ValueError(f"expected a (t, h, w) stack, got shape {tuple(img.shape)}")
        # fp16 stacks stay fp16: K1 (4096-column frames) and the rigid warp read the 16-bit samples
        dtype = torch.float16 if img.dtype == torch.float16 else torch.float32
        if img.device != self.device or img.dtype != dtype or not img.is_contiguous():
            img = img.detach().to(device=self.device, dtype=dtype).contiguous()
        return img


class RawMoviePipeline(MoviePipeline):
    """The same two-stream pipeline for RAW uint8 / int16 movies and a gain reference (N2): per movie one
    statistics pass over the raw bytes, then the estimator's row transform and the rigid warp condition the
    samples on the fly (``raw * gain - frame mean``, examples/ttMotion.py:90-121, 180-199) -- no conditioned
    fp32 movie is allocated.  Results equal ``condition_movie`` followed by the MoviePipeline.  Frame shapes
    without a fused kernel raise McorrUnsupported (use ``motion_correct_raw``, which falls back)."""

    def __init__(self, gain, device=None, pixel_spacing: float = 1.0, reference_frame: Optional[int] = None,
                 b_factor: float = 500, frequency_range=(300, 10), grid_type: str = "catmull_rom",
                 return_frames: bool = True, overlap: bool = True, mean_zero: bool = True):
        super().__init__(device, pixel_spacing, reference_frame, b_factor, frequency_range, grid_type,
                         return_frames, overlap)
        self.gain = None if gain is None else gain.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self.mean_zero = bool(mean_zero)
        self._rm = {}

    def _check(self, img: torch.Tensor) -> torch.Tensor:
        if img.dim() != 3:
            raise ValueError(f"expected a (t, h, w) stack, got shape {tuple(img.shape)}")
        if img.dtype not in (torch.uint8, torch.int16):
            raise TypeError(f"RawMoviePipeline reads uint8 or int16 movies, got {img.dtype}")
        if img.device != self.device or not img.is_contiguous():
            img = img.detach().to(device=self.device).contiguous()
        return img

    def _estimate(self, img: torch.Tensor) -> torch.Tensor:
        t = img.shape[0]
        ref = t // 2 if self.reference_frame is None else int(self.reference_frame)
        rm = engine.RawMovie(img, self.gain, mean_zero=self.mean_zero)  # the statistics pass, on the estimator's stream
        self._rm[id(img)] = rm
        shifts = engine.global_shifts_raw(rm, ref, self.pixel_spacing, self.b_factor, self.frequency_range)
        return (shifts * self.pixel_spacing).transpose(0, 1)[:, :, None, None]

    def _prepare(self, img: torch.Tensor, field: torch.Tensor):
        lat = engine.frame_lattices(field.contiguous(), img.shape[0], self.grid_type)
        rm = self._rm.pop(id(img))
        shifts_px, scratch = engine.rigid_tables(img, lat, self.pixel_spacing)
        return shifts_px, scratch, rm.mu, rm

    def _correct(self, img: torch.Tensor, tables):
        shifts_px, scratch, _, rm = tables
        return engine.warp_rigid_raw(rm, None, self.pixel_spacing, want_frames=self.return_frames, want_sum=True,
                                     tables=(shifts_px, scratch))


def motion_correct_movies(movies: Iterable[torch.Tensor], pixel_spacing: float, reference_frame=None,
                          b_factor=500, frequency_range=(300, 10), grid_type="catmull_rom",
                          return_frames=False, device=None, overlap=True) -> list[MovieResult]:
    """Global (rigid) motion correction of several movies: for each one the field of
    estimate_global_motion and the aligned frame sum (and the corrected frames when asked).
    Equivalent to ``[ (f := estimate_global_motion(m, ps, ...), motion_correct_sum(m, f, ps)) ]``
    with the two stages of consecutive movies overlapped on the GPU."""
    first = None
    movies = list(movies)
    if movies:
        first = movies[0]
    dev = require_gpu(device if device is not None else (first.device if first is not None else None))
    pipe = MoviePipeline(dev, pixel_spacing, reference_frame, b_factor, frequency_range, grid_type,
                         return_frames, overlap)
    return pipe.run(movies)
