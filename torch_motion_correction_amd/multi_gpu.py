"""Movie-level sharding across the GPUs of one node.

Movies are independent (the reference processes one (t,h,w) stack per call and has no
cross-stack state, estimate_motion_xc.py:21, correct_motion.py:18), so the path shards
by movie with NO data-path collective: one process per GPU, movie i -> rank i mod N.
torch.distributed is only used by callers for the start/stop barrier and to gather
small per-rank results (timings, shift tables)."""

from __future__ import annotations

from typing import Callable, Iterable, Sequence


def movies_for_rank(n_movies: int, rank: int, world_size: int) -> list[int]:
    """Round-robin assignment; every movie index appears on exactly one rank."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    return list(range(rank, n_movies, world_size))


def assignment(n_movies: int, world_size: int) -> list[list[int]]:
    return [movies_for_rank(n_movies, r, world_size) for r in range(world_size)]


def process_shard(movie_ids: Sequence[int], load: Callable[[int], object],
                  work: Callable[[object], object]) -> dict[int, object]:
    """Run `work(load(i))` for the movies of this rank, in order."""
    return {i: work(load(i)) for i in movie_ids}


def gather_results(local: dict[int, object], world_size: int, group=None) -> dict[int, object]:
    """All-gather small per-movie results (python objects) from every rank."""
    if world_size == 1:
        return dict(local)
    import torch.distributed as dist

    parts: list = [None] * world_size
    dist.all_gather_object(parts, local, group=group)
    out: dict[int, object] = {}
    for p in parts:
        overlap = set(out) & set(p)
        if overlap:
            raise RuntimeError(f"movies processed twice: {sorted(overlap)}")
        out.update(p)
    return out


def motion_correct_movies_sharded(movies: Sequence, rank: int, world_size: int, pixel_spacing: float,
                                  reference_frame=None, b_factor=500, frequency_range=(300, 10),
                                  grid_type="catmull_rom", device=None, load: Callable = None,
                                  keep_sums: bool = True, group=None, pipeline_factory: Callable = None):
    """Global (rigid) motion correction of a list of movies on ONE rank of a one-process-per-GPU job:
    this rank runs ``MoviePipeline`` (estimate of movie k+1 under the warp of movie k) over its
    round-robin shard and the small per-movie results -- the (2, t, 1, 1) Angstrom fields of
    ``estimate_global_motion`` -- are gathered so that every rank returns the fields of ALL movies.
    The aligned sums stay on the rank (and device) that computed them.  No data-path collective:
    the only communication is one all_gather_object of the shift tables.

    movies            sequence of (t, h, w) stacks, or of anything ``load`` turns into one (paths,
                      ids ...); only this rank's entries are touched
    load              optional ``load(item) -> (t, h, w) tensor`` applied lazily to this rank's items
    pipeline_factory  optional ``() -> object with .iterate(iterable_of_stacks)`` yielding results
                      with .field / .total (tests substitute a CPU stand-in; default MoviePipeline)

    Returns (fields, sums): fields = {movie index: (2, t, 1, 1) CPU tensor} for every movie of the
    job, sums = {movie index: (h, w) tensor on this rank's device} for this rank's movies."""
    mine = movies_for_rank(len(movies), rank, world_size)
    if pipeline_factory is None:
        from .pipeline import MoviePipeline

        def pipeline_factory():
            return MoviePipeline(device, pixel_spacing, reference_frame, b_factor, frequency_range, grid_type,
                                 return_frames=False, overlap=True)

    pipe = pipeline_factory()
    stacks = ((load(movies[i]) if load is not None else movies[i]) for i in mine)
    local_fields: dict[int, object] = {}
    sums: dict[int, object] = {}
    results = []
    for i, res in zip(mine, pipe.iterate(stacks)):
        results.append((i, res))  # ordered on the pipeline's streams; read after the loop
    for i, res in results:
        local_fields[i] = res.field.detach().cpu()
        if keep_sums:
            sums[i] = res.total
    fields = gather_results(local_fields, world_size, group=group)
    if len(fields) != len(movies):
        raise RuntimeError(f"{len(movies) - len(fields)} movies were processed by no rank")
    return fields, sums
