"""MI355X-native (gfx950) drop-in for the estimate -> correct hot path of
teamtomo/torch-motion-correction: same public function names and signatures
(reference: src/torch_motion_correction/__init__.py:12-44), hand-written HIP kernels
behind a C ABI (include/mcorr.h, libmcorr.so).  There is no CPU fallback."""

from .api import (  # noqa: F401
    condition_movie,
    correct_motion,
    correct_motion_fast,
    correct_motion_slow,
    correct_motion_two_grids,
    dose_weighted_sum,
    estimate_global_motion,
    estimate_local_motion,
    estimate_motion,
    estimate_motion_cross_correlation_patches,
    evaluate_deformation_field,
    evaluate_deformation_field_at_t,
    get_pixel_shifts,
    image_shifts_to_deformation_field,
    motion_correct_raw,
    motion_correct_sum,
    resample_deformation_field,
)
from ._lib import McorrError  # noqa: F401
from .data_io import read_deformation_field_from_csv, write_deformation_field_to_csv  # noqa: F401
from .optimization_state import OptimizationState, OptimizationTracker  # noqa: F401
from .multi_gpu import motion_correct_movies_sharded  # noqa: F401
from .pipeline import MoviePipeline, MovieResult, RawMoviePipeline, motion_correct_movies  # noqa: F401

__all__ = [
    "correct_motion",
    "correct_motion_fast",
    "correct_motion_slow",
    "correct_motion_two_grids",
    "get_pixel_shifts",
    "evaluate_deformation_field",
    "estimate_global_motion",
    "estimate_motion_cross_correlation_patches",
    "estimate_local_motion",
    "OptimizationState",
    "OptimizationTracker",
    "estimate_motion",
    "motion_correct_sum",
    "motion_correct_raw",
    "dose_weighted_sum",
    "condition_movie",
    "evaluate_deformation_field_at_t",
    "resample_deformation_field",
    "image_shifts_to_deformation_field",
    "McorrError",
    "write_deformation_field_to_csv",
    "read_deformation_field_from_csv",
    "motion_correct_movies",
    "motion_correct_movies_sharded",
    "MoviePipeline",
    "RawMoviePipeline",
    "MovieResult",
]
__version__ = "0.1.0"
