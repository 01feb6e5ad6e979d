#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container (where /root/reference is mounted); the GPU box and the test-suite
only ever read the .npz files this script wrote.  The reference package's ``__init__`` eagerly imports
rasterio (absent), so the hot-path submodules are imported individually under a stub parent package
(SURVEY.md §8(c)).  Nothing from the reference is copied: fixtures hold inputs' digests, kwargs and the
reference's *outputs*.

    python tests/golden/make_golden.py [--only g3 ...]

Inputs are regenerated from seeds by ``radar_processor_amd.synthetic`` at test time; each fixture stores
the sha256 digest of the volume it was computed from so generator drift is detected, plus the NumPy /
SciPy versions (the reference's intermediates are float32 only under NumPy >= 2).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import tempfile
import time
import types
import warnings

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/src/radar_grid"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


def load_reference():
    if not os.path.isdir(REF_SRC):
        raise SystemExit("reference not mounted; golden vectors can only be regenerated in the build container")
    pkg = types.ModuleType("radar_grid")
    pkg.__path__ = [REF_SRC]
    sys.modules["radar_grid"] = pkg
    mods = {}
    for name in ("geometry", "filters", "interpolate", "products", "compute", "utils"):
        mods[name] = importlib.import_module("radar_grid." + name)
    return types.SimpleNamespace(**mods)


def window_limits(center_xy, shape, spacing_xy, z_limits):
    nz, ny, nx = shape
    cx, cy = center_xy
    hx = 0.5 * (nx - 1) * spacing_xy
    hy = 0.5 * (ny - 1) * spacing_xy
    return (tuple(float(v) for v in z_limits), (cy - hy, cy + hy), (cx - hx, cx + hx))


def meta_blob(**kw):
    kw.update(numpy=np.__version__, scipy=scipy.__version__, generated=time.strftime("%Y-%m-%d"))
    return np.frombuffer(json.dumps(kw, sort_keys=True).encode(), dtype=np.uint8)


def run_window(ref, vol, shape, limits, weighting, radar_altitude=0.0, toa=17000.0, n_workers=4,
               fields=("DBZH",), qc=None, extra_fill=None, min_radius=250.0, beam_factor=0.01746):
    """Reference builder + apply on one window; returns dict of arrays for the fixture."""
    with tempfile.TemporaryDirectory() as tmp:
        geom = ref.compute.compute_grid_geometry(
            vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, tmp, radar_altitude=radar_altitude,
            min_radius=min_radius, beam_factor=beam_factor, weighting=weighting, toa=toa, n_workers=n_workers)
    out = dict(indptr=geom.indptr, gate_indices=geom.gate_indices, weights=geom.weights)
    radar = vol.as_radar()
    gf = None
    if qc is not None:
        gf = ref.filters.GateFilter(radar)
        gf.exclude_below(qc[0], qc[1])
        out["qc_excluded_count"] = np.array([int(gf.n_excluded())])
    for name in fields:
        fdata = ref.utils.get_field_data(radar, name)
        # the reference needs a full-array mask
        fdata = np.ma.array(np.ma.getdata(fdata), mask=np.ma.getmaskarray(fdata))
        out[f"grid_{name}"] = ref.interpolate.apply_geometry(geom, fdata)
        if gf is not None:
            out[f"grid_{name}_qc"] = ref.interpolate.apply_geometry(geom, fdata, additional_filters=[gf])
        if extra_fill is not None:
            out[f"grid_{name}_fill"] = ref.interpolate.apply_geometry(geom, fdata, fill_value=extra_fill)
    return geom, out


def products_block(ref, grid, geom, prefix, out):
    """CAPPI / column products of the reference on one gridded field."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        P = ref.products
        out[f"{prefix}_cappi4000_linear"] = np.array(P.constant_altitude_ppi(grid, geom, 4000.0, "linear"))
        out[f"{prefix}_cappi4000_nearest"] = np.array(P.constant_altitude_ppi(grid, geom, 4000.0, "nearest"))
        out[f"{prefix}_cappi2500_linear"] = np.array(P.constant_altitude_ppi(grid, geom, 2500.0, "linear"))
        out[f"{prefix}_cappi_above"] = np.array(P.constant_altitude_ppi(grid, geom, 99000.0, "linear"))
        out[f"{prefix}_colmax"] = P.column_max(grid)
        out[f"{prefix}_colmax_alt"] = P.column_max(grid, z_min_alt=1000, z_max_alt=8000, geometry=geom)
        out[f"{prefix}_colmax_idx"] = P.column_max(grid, z_min_idx=2, z_max_idx=6)
        out[f"{prefix}_colmin"] = P.column_min(grid)
        out[f"{prefix}_colmean"] = P.column_mean(grid)
        out[f"{prefix}_colmean_alt"] = P.column_mean(grid, z_min_alt=1000, z_max_alt=8000, geometry=geom)


def gen_g2(ref, synth):
    """C1: single PPI sweep 360x500 flattened to z=0, 2-D grid windows (1,48,48) at three ranges."""
    vol = synth.make_volume(n_elev=1, n_az=360, n_gates=500, seed=1, fields=("DBZH",))
    vol.gate_z = np.zeros_like(vol.gate_z)
    for tag, centre in (("near", (6e3, -4e3)), ("mid", (70e3, 55e3)), ("far", (-150e3, 160e3))):
        shape = (1, 48, 48)
        limits = window_limits(centre, shape, 960.0, (0.0, 0.0))
        geom, out = run_window(ref, vol, shape, limits, "barnes2", n_workers=1, extra_fill=-9999.0)
        out["meta"] = meta_blob(case="G2", volume=dict(n_elev=1, n_az=360, n_gates=500, seed=1, flatten_z=True),
                                digest=vol.digest(), grid_shape=shape, grid_limits=limits, weighting="barnes2",
                                toa=17000.0, radar_altitude=0.0, fields=["DBZH"])
        np.savez_compressed(os.path.join(HERE, f"g2_c1_{tag}.npz"), **out)
        print("g2", tag, geom)


def gen_g3(ref, synth):
    """C2 volume (12x360x1000), windows (20,12,12) at true 480 m spacing, z 0..15 km."""
    fields = ("DBZH", "ZDR", "RHOHV")
    vol = synth.make_volume(n_elev=12, n_az=360, n_gates=1000, seed=0, fields=fields)
    shape = (20, 12, 12)
    cases = [
        ("r010", (7.1e3, 7.3e3), ("barnes2", "cressman", "nearest")),
        ("r060", (-42e3, 43e3), ("barnes2",)),
        ("r150", (106e3, -106e3), ("barnes2", "cressman")),
        ("r235", (-166e3, -166.4e3), ("barnes2",)),
        ("corner", (237e3, 237e3), ("barnes2",)),
    ]
    for tag, centre, weightings in cases:
        limits = window_limits(centre, shape, 480.0, (0.0, 15000.0))
        for wname in weightings:
            geom, out = run_window(ref, vol, shape, limits, wname, fields=fields, qc=("RHOHV", 0.8),
                                   extra_fill=-9999.0 if wname == "barnes2" else None)
            if wname == "barnes2":
                products_block(ref, out["grid_DBZH"], geom, "DBZH", out)
            if wname != "barnes2":
                # identical neighbour sets and KD-tree order: keep only the weights
                del out["gate_indices"]
            out["meta"] = meta_blob(case="G3", volume=dict(n_elev=12, n_az=360, n_gates=1000, seed=0),
                                    digest=vol.digest(), grid_shape=shape, grid_limits=limits, weighting=wname,
                                    toa=17000.0, radar_altitude=0.0, fields=list(fields), qc=["RHOHV", 0.8])
            np.savez_compressed(os.path.join(HERE, f"g3_c2_{tag}_{wname}.npz"), **out)
            print("g3", tag, wname, geom)


def gen_g4(ref, synth):
    """C4 volume (14x720x2000, 120 m gates), windows (40,6,6) at 240 m spacing."""
    vol = synth.make_volume(n_elev=14, n_az=720, n_gates=2000, seed=4, fields=("DBZH",))
    shape = (40, 6, 6)
    for tag, centre in (("r030", (21e3, -21.5e3)), ("r180", (-127e3, 128e3))):
        limits = window_limits(centre, shape, 240.0, (0.0, 15000.0))
        geom, out = run_window(ref, vol, shape, limits, "barnes2", n_workers=4)
        out["meta"] = meta_blob(case="G4", volume=dict(n_elev=14, n_az=720, n_gates=2000, seed=4),
                                digest=vol.digest(), grid_shape=shape, grid_limits=limits, weighting="barnes2",
                                toa=17000.0, radar_altitude=0.0, fields=["DBZH"])
        np.savez_compressed(os.path.join(HERE, f"g4_c4_{tag}.npz"), **out)
        print("g4", tag, geom)


def gen_g5(ref, synth):
    """Products on a dense synthetic 3-D grid, incl. the exact-level CAPPI case (z 0..19 km, nz=20)."""
    rng = np.random.default_rng(55)
    grid = (rng.normal(15.0, 18.0, size=(20, 40, 56))).astype(np.float32)
    grid[rng.random(grid.shape) < 0.25] = np.nan
    grid[:, :6, :7] = np.nan                       # all-NaN columns
    grid[3:9, 10:14, 20:30] = np.float32(41.5)     # ties for the argmax contract
    grid[5, 30:34, 40:44] = np.float32(-0.0)
    for tag, zlim in (("z15", (0.0, 15000.0)), ("z19", (0.0, 19000.0))):
        geom = ref.geometry.GridGeometry(grid_shape=grid.shape, grid_limits=(zlim, (-1e4, 1e4), (-1.4e4, 1.4e4)),
                                         indptr=np.zeros(grid.size + 1, dtype=np.int32),
                                         gate_indices=np.zeros(0, dtype=np.int32),
                                         weights=np.zeros(0, dtype=np.float32), toa=17000.0)
        out = dict(grid=grid)
        products_block(ref, grid, geom, "P", out)
        out["meta"] = meta_blob(case="G5", grid_shape=grid.shape, z_limits=zlim)
        np.savez_compressed(os.path.join(HERE, f"g5_products_{tag}.npz"), **out)
        print("g5", tag)


def gen_g7(ref, synth):
    """Constant-elevation PPI + beam-height helpers on a grid wide enough for the beam to cross all levels."""
    rng = np.random.default_rng(77)
    shape = (20, 48, 64)
    grid = rng.normal(12.0, 15.0, size=shape).astype(np.float32)
    grid[rng.random(shape) < 0.2] = np.nan
    limits = ((0.0, 15000.0), (-110e3, 110e3), (-150e3, 150e3))
    geom = ref.geometry.GridGeometry(grid_shape=shape, grid_limits=limits, indptr=np.zeros(grid.size + 1, dtype=np.int32),
                                     gate_indices=np.zeros(0, dtype=np.int32), weights=np.zeros(0, dtype=np.float32),
                                     toa=17000.0, radar_altitude=312.0)
    out = dict(grid=grid)
    P = ref.products
    for elev in (0.0, 0.5, 2.0, 10.0, 45.0):
        for interp in ("linear", "nearest"):
            for curved in (True, False):
                key = f"ppi_e{elev}_{interp}_{'curved' if curved else 'flat'}"
                out[key] = np.array(P.constant_elevation_ppi(grid, geom, elev, interpolation=interp, earth_curvature=curved))
    out["ppi_e2.0_linear_ke1"] = np.array(P.constant_elevation_ppi(grid, geom, 2.0, ke=1.0))
    d = np.array([0.0, 1.0, 10000.0, 20000.0, 50000.0, 237000.5])
    out["bh_dist"] = d
    out["bh_curved"] = P.compute_beam_height(d, 2.0, 100.0)
    out["bh_simple"] = P.compute_beam_height_simple(d, 2.0, 100.0)
    out["bh_flat"] = P.compute_beam_height_flat(d, 2.0, 100.0)
    out["bh_difference"] = P.get_beam_height_difference(geom, 1.5, radar_altitude=312.0)
    out["elev_from_z_curved"] = P.get_elevation_from_z_level(3000.0, geom, radar_altitude=312.0)
    out["elev_from_z_flat"] = P.get_elevation_from_z_level(3000.0, geom, radar_altitude=312.0, earth_curvature=False)
    out["meta"] = meta_blob(case="G7", grid_shape=shape, grid_limits=limits, radar_altitude=312.0)
    np.savez_compressed(os.path.join(HERE, "g7_ppi.npz"), **out)
    print("g7")


def gen_g6(ref, synth):
    """radar_altitude != 0 pins the compute.py:182 subtraction and the toa cut; low toa drops sweeps."""
    vol = synth.make_volume(n_elev=12, n_az=360, n_gates=1000, seed=6, fields=("DBZH",))
    shape = (20, 10, 10)
    limits = window_limits((-61e3, 88e3), shape, 480.0, (0.0, 15000.0))
    geom, out = run_window(ref, vol, shape, limits, "barnes2", radar_altitude=438.5, toa=9000.0, n_workers=4,
                           min_radius=400.0, beam_factor=0.02)
    out["meta"] = meta_blob(case="G6", volume=dict(n_elev=12, n_az=360, n_gates=1000, seed=6),
                            digest=vol.digest(), grid_shape=shape, grid_limits=limits, weighting="barnes2",
                            toa=9000.0, radar_altitude=438.5, min_radius=400.0, beam_factor=0.02, fields=["DBZH"])
    np.savez_compressed(os.path.join(HERE, "g6_altitude.npz"), **out)
    print("g6", geom)


def _sha(a) -> str:
    import hashlib
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest() + f":{a.dtype.str}:{a.shape}"


def gen_g8(ref, synth):
    """``.npz`` interchange (geometry.py:94-150), both directions:

    * ``g8_ref_saved_geometry.npz`` is the file the REFERENCE's ``save_geometry`` wrote for a small geometry built by
      the reference's ``compute_grid_geometry`` (an output file of the reference = data); the tests load it with this
      build's ``load_geometry`` and grid through it;
    * this build's ``save_geometry`` writes the same geometry, the REFERENCE's ``load_geometry`` reads that file back and
      the outcome (field-by-field equality with the reference's own object, per-array digests, the key sets of the two
      files) is recorded in ``g8_interchange.npz`` together with the reference's gridded field.
    """
    import radar_processor_amd as rg
    vol = synth.make_volume(n_elev=12, n_az=360, n_gates=1000, seed=8, fields=("DBZH",))
    shape = (6, 12, 12)
    limits = window_limits((47e3, -72e3), shape, 480.0, (500.0, 8000.0))
    geom, out = run_window(ref, vol, shape, limits, "barnes2", radar_altitude=0.0, toa=12000.0, n_workers=2)
    geom.radar_altitude = 312.5           # compute_grid_geometry drops it (compute.py:277-284); pin the key's round trip
    ref_path = os.path.join(HERE, "g8_ref_saved_geometry.npz")
    ref.geometry.save_geometry(geom, ref_path)                       # <- written by the reference
    with np.load(ref_path) as z:
        ref_keys = sorted(z.files)
        ref_dtypes = {k: z[k].dtype.str for k in z.files}
    with tempfile.TemporaryDirectory() as tmp:
        ours = rg.GridGeometry(shape, limits, geom.indptr.copy(), geom.gate_indices.copy(), geom.weights.copy(),
                               toa=12000.0, radar_altitude=312.5)
        our_path = os.path.join(tmp, "written_by_this_build.npz")
        rg.save_geometry(ours, our_path)                             # <- written by this build
        with np.load(our_path) as z:
            our_keys = sorted(z.files)
            our_dtypes = {k: z[k].dtype.str for k in z.files}
        back = ref.geometry.load_geometry(our_path)                  # <- read by the reference
        radar = vol.as_radar()
        fdata = ref.utils.get_field_data(radar, "DBZH")
        fdata = np.ma.array(np.ma.getdata(fdata), mask=np.ma.getmaskarray(fdata))
        grid_back = ref.interpolate.apply_geometry(back, fdata)
    same = dict(
        grid_shape=tuple(int(v) for v in back.grid_shape) == tuple(shape),
        grid_limits=tuple(tuple(float(x) for x in lim) for lim in back.grid_limits)
        == tuple(tuple(float(x) for x in lim) for lim in limits),
        indptr=bool(np.array_equal(back.indptr, geom.indptr) and back.indptr.dtype == geom.indptr.dtype),
        gate_indices=bool(np.array_equal(back.gate_indices, geom.gate_indices)
                          and back.gate_indices.dtype == geom.gate_indices.dtype),
        weights=bool(np.array_equal(back.weights, geom.weights) and back.weights.dtype == geom.weights.dtype),
        toa=float(back.toa) == 12000.0, radar_altitude=float(back.radar_altitude) == 312.5,
        gridded_equal=bool(np.array_equal(grid_back, out["grid_DBZH"], equal_nan=True)))
    blob = dict(grid_DBZH=out["grid_DBZH"], indptr=geom.indptr, gate_indices=geom.gate_indices, weights=geom.weights)
    blob["meta"] = meta_blob(case="G8", volume=dict(n_elev=12, n_az=360, n_gates=1000, seed=8), digest=vol.digest(),
                             grid_shape=shape, grid_limits=limits, weighting="barnes2", toa=12000.0,
                             radar_altitude=312.5, fields=["DBZH"],
                             reference_read_our_file=same, ref_file_keys=ref_keys, our_file_keys=our_keys,
                             ref_file_dtypes=ref_dtypes, our_file_dtypes=our_dtypes,
                             digests=dict(indptr=_sha(geom.indptr), gate_indices=_sha(geom.gate_indices),
                                          weights=_sha(geom.weights)))
    np.savez_compressed(os.path.join(HERE, "g8_interchange.npz"), **blob)
    print("g8", same, ref_keys == our_keys)


def gen_g9(ref, synth):
    """a1, the part the reference itself can pin: the HEIGHT of a gate.  The reference's gate coordinates are PyART's
    (absent), but ``products.compute_beam_height`` (products.py:70-89) evaluates the same 4/3-earth height
    ``sqrt(r^2 + (ke*Re)^2 + 2*r*ke*Re*sin(el)) - ke*Re`` with the same constants (products.py:19-20) from a GROUND
    distance it first turns back into a slant range, ``s / max(cos(el), 0.01)``.  Fed ``s = r*cos(el)`` it therefore
    returns the z of the gate at slant range r (float64); stored for every sweep elevation and the range tables of
    BASELINE configs 1, 2 and 4.  x / y stay unpinned (PyART's arc-length formula has no counterpart in the tree)."""
    out = {}
    tables = {}
    for tag, (n_elev, n_gates) in dict(c1=(1, 500), c2=(12, 1000), c4=(14, 2000)).items():
        elev, _, rng_m = synth.sweep_geometry(n_elev, 4, n_gates)
        elev = np.asarray(elev, dtype=np.float64)
        rng_m = np.asarray(rng_m, dtype=np.float64)
        z = np.stack([ref.products.compute_beam_height(rng_m * np.cos(np.radians(e)), float(e), 0.0) for e in elev])
        assert z.dtype == np.float64 and z.shape == (n_elev, n_gates)
        out[f"z_{tag}"] = z
        tables[tag] = dict(n_elev=n_elev, n_gates=n_gates, elevations=[float(e) for e in elev],
                           first_range=float(rng_m[0]), range_step=float(rng_m[1] - rng_m[0]))
    out["meta"] = meta_blob(case="G9", tables=tables, radar_altitude=0.0,
                            what="reference products.compute_beam_height(r*cos(el), el, 0.0), float64")
    np.savez_compressed(os.path.join(HERE, "g9_beam_z.npz"), **out)
    print("g9", {k: v.shape for k, v in out.items() if k != "meta"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    ref = load_reference()
    from radar_processor_amd import synthetic as synth
    gens = dict(g2=gen_g2, g3=gen_g3, g4=gen_g4, g5=gen_g5, g6=gen_g6, g7=gen_g7, g8=gen_g8, g9=gen_g9)
    for name, fn in gens.items():
        if args.only and name not in args.only:
            continue
        t0 = time.time()
        fn(ref, synth)
        print(f"[{name}] {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
