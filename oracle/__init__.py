"""CPU oracle for the motion-correction hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU / numpy / scipy) of the algorithm of
teamtomo/torch-motion-correction's cross-correlation estimate -> deformation-field
warp -> frame-sum path.  It exists to *check* the HIP implementation; it is never
the thing that is shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``torch_motion_correction_amd``) must never import anything from here.

Pinning status (see DESIGN.md section "Oracle"):
  * patch lattice, gather indices and the LazyPatchGrid cache/eviction replay are
    PINNED against the reference's own ``patch_grid`` sub-package (importable in
    the build container; goldens in tests/golden/patch_grid_*.npz were produced by
    the reference code itself, see oracle/make_goldens.py).
  * normalize_image, array_to_grid_sample, image_shifts_to_deformation_field, the
    sub-pixel refinement, outlier rejection, temporal smoothing and get_pixel_shifts are
    PINNED bit for bit against the reference's own functions (its modules import once the
    absent packages are bound to stubs that raise when called; outputs on seeded inputs in
    tests/golden/reference_helpers.npz, see oracle/make_goldens.py).
  * ``scipy.signal.savgol_filter`` is the real dependency (scipy is installed).
  * everything that the reference delegates to the five un-vendored teamtomo
    packages (torch_grid_utils, torch_fourier_filter, torch_fourier_shift,
    torch_cubic_spline_grids, torch_image_interpolation) is restated from their
    published behaviour in ``oracle/thirdparty_semantics.py`` -- those packages
    are not installed, not vendored and not pinned by the reference
    (pyproject.toml:36-46), and the reference's tests hold no numeric golden
    values, so at those boundaries the oracle is **parity unpinned**.
"""

from oracle.motion import (  # noqa: F401
    correct_motion,
    correct_motion_fast,
    correct_motion_slow,
    correct_motion_two_grids,
    dose_weighted_sum,
    estimate_global_motion,
    estimate_motion_cross_correlation_patches,
    evaluate_deformation_field,
    evaluate_deformation_field_at_t,
    get_pixel_shifts,
    image_shifts_to_deformation_field,
    normalize_image,
    prepare_bandpass_filter,
    resample_deformation_field,
)
from oracle.local_motion import (  # noqa: F401,E402
    LocalMotionProblem,
    OptimizationTracker,
    estimate_local_motion,
)
