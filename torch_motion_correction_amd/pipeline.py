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
        self._s_rest = None  # third stream of the three-stage schedule, created on first use
        self._tables = {}

    # the two stages, each enqueued on whatever stream is current
    def _estimate(self, img: torch.Tensor) -> torch.Tensor:
        t = img.shape[0]
        ref = t // 2 if self.reference_frame is None else int(self.reference_frame)
        shifts = engine.global_shifts(img, ref, self.pixel_spacing, self.b_factor, self.frequency_range)
        return self._field_and_tables(img, shifts)

    def _field_and_tables(self, img: torch.Tensor, shifts: torch.Tensor) -> torch.Tensor:
        """(t,2) px shifts -> the (2,t,1,1) Angstrom field (dfu.py:129-162); everything of correct_motion that
        only needs the field -- the per-frame lattice values and the rigid warp's weight tables -- is built
        right here, on the ESTIMATOR's stream, in two launches (mc_rigid_tables_from_shifts; it used to be
        seven, each waiting for a wave slot under the previous movie's warp), and handed to ``_prepare``."""
        field, tables = engine.rigid_tables_from_shifts(shifts, tuple(img.shape), self.pixel_spacing, self.grid_type)
        self._tables[id(img)] = tables
        return field

    def _prepare(self, img: torch.Tensor, field: torch.Tensor):
        return self._tables.pop(id(img))

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
        import os

        # MC_PIPE_SCHEDULE: k1first (default), abc (three stages, the warp alone), all (round 2: the whole
        # estimator under the previous warp), k2first -- measured on 40 x 4096^2 (same box): 1.70 / 1.92 /
        # 1.76-1.79 / 1.72-1.73 ms per step
        if os.environ.get("MC_PIPE_SCHEDULE", "k1first") == "abc" and type(self) is MoviePipeline:
            yield from self._iterate_three_stage(movies, call, caller)
            return
        sched = os.environ.get("MC_PIPE_SCHEDULE", "k1first")
        k1_first = sched in ("k1first", "k2first", "k3first")
        engine.HOOK_AFTER_K2 = sched == "k2first"
        from . import _lib
        lib = _lib.load()
        try:
            pending = None  # k1first: (img, field, tables, ready) of the movie whose warp is not enqueued yet
            warp_done = None
            for img in movies:
                with device_scope(dev):
                    img = self._check(img)
                    img.record_stream(self._s_est)
                    img.record_stream(self._s_warp)
                    k1_done = torch.cuda.Event() if k1_first else None
                    with torch.cuda.stream(self._s_est):
                        if sched == "k3first":
                            # the previous warp waits until every machine-filling kernel of this estimate (K1,
                            # K2, the near-window column pass) has run; the C library records the event.  And
                            # this estimate starts once the warp before that one has finished: strict alternation
                            # [K1 K2 K3n](k+1) -> [warp(k) || rest of the search(k+1)] -> [K1 K2 K3n](k+2) ...
                            if warp_done is not None:
                                self._s_est.wait_event(warp_done)
                            k1_done.record(self._s_est)  # creates the hipEvent
                            lib.mc_xc_after_k3n_event(k1_done.cuda_event)
                        elif k1_first:
                            engine.AFTER_K1_HOOK = lambda: k1_done.record(self._s_est)
                        try:
                            field = self._estimate(img)
                        finally:
                            engine.AFTER_K1_HOOK = None
                            if sched == "k3first":
                                lib.mc_xc_after_k3n_event(None)
                        tables = self._prepare(img, field)
                        ready = torch.cuda.Event()
                        ready.record(self._s_est)

                    def enqueue_warp(img_, field_, tables_, ready_, extra=None):
                        with torch.cuda.stream(self._s_warp):
                            self._s_warp.wait_event(ready_)
                            if extra is not None:
                                self._s_warp.wait_event(extra)  # the NEXT movie's K1 has left the HBM to this warp
                            field_.record_stream(self._s_warp)
                            for x in tables_:
                                if isinstance(x, torch.Tensor):
                                    x.record_stream(self._s_warp)
                            return call(lambda: self._correct(img_, tables_))

                    if not k1_first:
                        frames, total = enqueue_warp(img, field, tables, ready)
                        res = MovieResult(field, total, frames)
                    else:
                        res = None
                        if pending is not None:
                            pimg, pfield, ptables, pready = pending
                            frames, total = enqueue_warp(pimg, pfield, ptables, pready, k1_done)
                            res = MovieResult(pfield, total, frames)
                            if sched == "k3first":
                                warp_done = torch.cuda.Event()
                                warp_done.record(self._s_warp)
                        pending = (img, field, tables, ready)
                if res is not None:
                    yield res
            if k1_first and pending is not None:
                with device_scope(dev):
                    pimg, pfield, ptables, pready = pending
                    with torch.cuda.stream(self._s_warp):
                        self._s_warp.wait_event(pready)
                        pfield.record_stream(self._s_warp)
                        for x in ptables:
                            if isinstance(x, torch.Tensor):
                                x.record_stream(self._s_warp)
                        frames, total = call(lambda: self._correct(pimg, ptables))
                yield MovieResult(pfield, total, frames)
        finally:
            with device_scope(dev):
                for s in (self._s_est, self._s_warp):
                    done = torch.cuda.Event()
                    done.record(s)
                    caller.wait_event(done)

    def _iterate_three_stage(self, movies, call, caller):
        """Three stages on three streams: A = K1 (the estimator's HBM-bound row pass), B = the rest of the
        estimator and the warp's tables (latency-bound kernels with 32 KB of LDS per workgroup), C = the
        warp.  Per period   [ A(m) || B(m-1) ]  then  [ C(m-1) ] :  the warp has the chip to itself (under
        the two-stream overlap its launch stretched from 1.09 to 1.44-1.65 ms: K1 took its HBM share, K2-K4
        only fit into a CU once a warp tile had retired), and B's workgroups fit beside K1's, which hold
        64 KB of LDS per CU.  Events only; no host synchronisation."""
        dev = self.device
        if self._s_rest is None:
            self._s_rest = torch.cuda.Stream(dev)
        s_a, s_b, s_c = self._s_est, self._s_rest, self._s_warp
        with device_scope(dev):
            start = torch.cuda.Event()
            start.record(caller)
            s_b.wait_event(start)
        t_ref = lambda img: (img.shape[0] // 2 if self.reference_frame is None else int(self.reference_frame))  # noqa: E731
        pending = None   # (img, state, evA) of the movie whose stages B and C are not enqueued yet
        ev_c = None      # end of the last enqueued warp

        def stage_bc(p, ev_a_next):
            nonlocal ev_c
            pimg, pstate, pev_a = p
            with torch.cuda.stream(s_b):
                s_b.wait_event(pev_a)
                if ev_c is not None:
                    s_b.wait_event(ev_c)
                for x in engine.stage_tensors(pstate):
                    x.record_stream(s_b)
                shifts = engine.global_stage_b(pstate, t_ref(pimg))
                field = self._field_and_tables(pimg, shifts)
                tables = self._prepare(pimg, field)
                ev_b = torch.cuda.Event()
                ev_b.record(s_b)
            with torch.cuda.stream(s_c):
                s_c.wait_event(ev_b)
                if ev_a_next is not None:
                    s_c.wait_event(ev_a_next)  # the next movie's K1 has left the HBM to this warp
                field.record_stream(s_c)
                for x in tables:
                    if isinstance(x, torch.Tensor):
                        x.record_stream(s_c)
                frames, total = call(lambda: self._correct(pimg, tables))
                ev_c = torch.cuda.Event()
                ev_c.record(s_c)
            return MovieResult(field, total, frames)

        try:
            for img in movies:
                res = None
                with device_scope(dev):
                    img = self._check(img)
                    for s_ in (s_a, s_b, s_c):
                        img.record_stream(s_)
                    with torch.cuda.stream(s_a):
                        if ev_c is not None:
                            s_a.wait_event(ev_c)
                        state = engine.global_stage_a(img, self.pixel_spacing, self.b_factor, self.frequency_range)
                        ev_a = torch.cuda.Event()
                        ev_a.record(s_a)
                    if pending is not None:
                        res = stage_bc(pending, ev_a)
                    pending = (img, state, ev_a)
                if res is not None:
                    yield res
            if pending is not None:
                with device_scope(dev):
                    res = stage_bc(pending, None)
                yield res
        finally:
            with device_scope(dev):
                for s_ in (s_a, s_b, s_c):
                    done = torch.cuda.Event()
                    done.record(s_)
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
        return self._field_and_tables(img, shifts)

    def _prepare(self, img: torch.Tensor, field: torch.Tensor):
        rm = self._rm.pop(id(img))
        shifts_px, scratch = self._tables.pop(id(img))
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
