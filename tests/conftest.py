"""Shared test plumbing.

Markers: ``gpu`` = needs a real MI355X (run by the driver with ``-m gpu``); everything else must pass on a
CPU-only box.  Only tests (and bench.py's cpu_baseline leg / smoke()) may import ``oracle``.
"""
import functools
import glob
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (HIP kernels through the C ABI)")
    config.addinivalue_line("markers", "slow: longer CPU-side oracle checks")


@pytest.fixture(scope="session", autouse=True)
def _library_present(request):
    """GPU runs from a bare checkout: the shared library is git-ignored, so compile it once if it is missing
    (hipcc is part of the image).  CPU-only runs build it too -- the C-ABI symbol check needs it."""
    try:
        from radar_processor_amd.build import ensure_built
        ensure_built(verbose=False)
    except Exception as exc:      # no hipcc: the tests that need the library report it themselves
        print(f"[conftest] could not build libradargrid_hip.so: {exc}")


def load_golden(name):
    """Load one fixture: returns (meta dict, {array name: ndarray})."""
    path = os.path.join(GOLDEN, name if name.endswith(".npz") else name + ".npz")
    with np.load(path) as z:
        arrays = {k: z[k] for k in z.files if k != "meta"}
        meta = json.loads(bytes(z["meta"]).decode())
    return meta, arrays


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


@functools.lru_cache(maxsize=4)
def golden_volume(n_elev, n_az, n_gates, seed, fields, flatten_z=False):
    """Regenerate the synthetic volume a fixture was computed from (digest-checked by the caller)."""
    from radar_processor_amd import synthetic
    vol = synthetic.make_volume(n_elev=n_elev, n_az=n_az, n_gates=n_gates, seed=seed, fields=fields)
    if flatten_z:
        vol.gate_z = np.zeros_like(vol.gate_z)
    return vol


def volume_for(meta):
    v = meta["volume"]
    vol = golden_volume(v["n_elev"], v["n_az"], v["n_gates"], v["seed"], tuple(meta["fields"]),
                        bool(v.get("flatten_z", False)))
    assert vol.digest() == meta["digest"], "synthetic generator drifted from the one the fixture was made with"
    return vol


def builder_kwargs(meta):
    return dict(radar_altitude=meta.get("radar_altitude", 0.0), min_radius=meta.get("min_radius", 250.0),
                beam_factor=meta.get("beam_factor", 0.01746), weighting=meta["weighting"], toa=meta["toa"])


def grid_spec(meta):
    shape = tuple(meta["grid_shape"])
    limits = tuple(tuple(float(x) for x in lim) for lim in meta["grid_limits"])
    return shape, limits


def reference_indices(name, meta, arrays):
    """Non-barnes2 G3 fixtures share gate_indices with their barnes2 sibling (same neighbour sets)."""
    if "gate_indices" in arrays:
        return arrays["gate_indices"]
    sibling = name.rsplit("_", 1)[0] + "_barnes2"
    _, sib = load_golden(sibling)
    return sib["gate_indices"]


@pytest.fixture(scope="session")
def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# Absolute floor of the gridded-value comparisons, as a fraction of max|field|: a weighted mean of mixed-sign data that
# cancels to ~0 has no relative floor, so "<= 1e-5 relative" (north_star) needs one.  The largest absolute error ever
# measured between this build and the reference / the oracle is 6.1e-7 * max|field| (round 3,
# gpurun_out/parity_relerr_fullsize.json); the floor is set three times above that, not at the 1e-5 of rounds 1-3, which
# was 16x looser than the code.  A failure at this floor is a finding to report, not a reason to loosen it.
ATOL_FRAC = 2e-6
RTOL = 1e-5


def assert_same_to_rounding(got, want, scale, fill=None, rtol=RTOL, atol_frac=ATOL_FRAC):
    """Two grids summed in different float32 orders (the row-wise kernel against the tile kernels / the oracle): the same
    voxels filled, and every value within ``rtol * |want| + atol_frac * scale`` (``scale`` = the largest |value| of the
    field).  Accepts numpy arrays or torch tensors."""
    if hasattr(got, "detach"):
        got = got.detach().cpu().numpy()
    if hasattr(want, "detach"):
        want = want.detach().cpu().numpy()
    if fill is None or np.isnan(fill):
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    else:
        np.testing.assert_array_equal(got == np.float32(fill), want == np.float32(fill))
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol_frac * float(scale), equal_nan=True)
