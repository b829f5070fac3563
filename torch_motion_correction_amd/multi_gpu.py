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
