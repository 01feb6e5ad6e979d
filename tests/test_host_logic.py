"""CPU-side checks (no GPU needed): the geometry container and its .npz format, GateFilter predicates, the
reference's error behaviour, the C-ABI library (loads, exports every symbol include/radargrid_hip.h declares),
and that the product path refuses to run without a HIP device instead of falling back to the CPU.

Modelled on the reference's tests/test_radar_grid_geometry.py and tests/test_radar_grid_filters.py.
"""
import ctypes
import json
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest

import radar_processor_amd as rg
from conftest import REPO
from oracle import radar_grid_oracle as oracle
from radar_processor_amd import _native


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.fixture
def geometry():
    nz, ny, nx = 4, 5, 6
    n = nz * ny * nx
    return rg.GridGeometry(
        grid_shape=(nz, ny, nx),
        grid_limits=((0.0, 3000.0), (-2000.0, 2000.0), (-2500.0, 2500.0)),
        indptr=np.arange(0, 2 * n + 1, 2, dtype=np.int32),
        gate_indices=np.arange(2 * n, dtype=np.int32),
        weights=np.full(2 * n, 0.5, dtype=np.float32),
        toa=12000.0, radar_altitude=150.0)


class TestGridGeometry:           # reference: tests/test_radar_grid_geometry.py
    def test_counters(self, geometry):
        assert geometry.n_grid_points() == 120
        assert geometry.n_pairs() == 240
        assert geometry.avg_neighbors() == 2.0
        assert geometry.memory_usage_mb() == pytest.approx((121 * 4 + 240 * 4 + 240 * 4) / 1e6)
        assert not geometry.is_device_resident

    def test_z_levels(self, geometry):
        np.testing.assert_array_equal(geometry.z_levels(), np.linspace(0.0, 3000.0, 4))
        np.testing.assert_array_equal(geometry.z_levels_absolute(), np.linspace(0.0, 3000.0, 4) + 150.0)

    def test_default_radar_altitude_and_repr(self, geometry):
        g = rg.GridGeometry((1, 1, 1), ((0, 1), (0, 1), (0, 1)), np.zeros(2, dtype=np.int32),
                            np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32), toa=1.0)
        assert g.radar_altitude == 0.0
        text = repr(geometry)
        assert text.startswith("GridGeometry(") and "n_pairs=240" in text and "avg_neighbors=2.0" in text
        assert "toa=12000.0m" in text and "radar_altitude=150.0m" in text

    def test_npz_roundtrip_and_keys(self, geometry, tmp_path):
        path = str(tmp_path / "geom.npz")
        rg.save_geometry(geometry, path)
        with np.load(path) as z:
            assert sorted(z.files) == sorted(["grid_shape", "grid_limits_z", "grid_limits_y", "grid_limits_x", "indptr",
                                              "gate_indices", "weights", "toa", "radar_altitude"])
        back = rg.load_geometry(path)
        assert back == geometry
        assert back.grid_shape == (4, 5, 6) and back.toa == 12000.0 and back.radar_altitude == 150.0
        assert back.grid_limits == geometry.grid_limits

    def test_load_without_optional_keys(self, tmp_path):
        """Old files lack toa / radar_altitude (geometry.py:146-147)."""
        path = str(tmp_path / "old.npz")
        np.savez_compressed(path, grid_shape=np.array([1, 1, 2]), grid_limits_z=np.array([0.0, 0.0]),
                            grid_limits_y=np.array([0.0, 0.0]), grid_limits_x=np.array([0.0, 1.0]),
                            indptr=np.array([0, 1, 2], dtype=np.int32), gate_indices=np.array([0, 1], dtype=np.int32),
                            weights=np.ones(2, dtype=np.float32))
        g = rg.load_geometry(path)
        assert g.toa == np.inf and g.radar_altitude == 0.0 and g.n_pairs() == 2


class TestNpzInterchangeWithTheReference:
    """geometry.py:94-150 both ways, pinned by tests/golden/make_golden.py::gen_g8 (run in the build container, where
    the reference is importable): a file written by the REFERENCE's save_geometry is read by this build, and the
    reference read a file written by this build (outcome recorded in the fixture)."""

    def test_reference_written_file_loads(self, tmp_path):
        import hashlib
        from conftest import GOLDEN, load_golden
        meta, arr = load_golden("g8_interchange")
        geom = rg.load_geometry(os.path.join(GOLDEN, "g8_ref_saved_geometry.npz"))
        assert tuple(int(v) for v in geom.grid_shape) == tuple(meta["grid_shape"])
        assert tuple(tuple(float(x) for x in lim) for lim in geom.grid_limits) == \
            tuple(tuple(float(x) for x in lim) for lim in meta["grid_limits"])
        assert geom.toa == meta["toa"] and geom.radar_altitude == meta["radar_altitude"]
        for name in ("indptr", "gate_indices", "weights"):
            got = getattr(geom, name)
            assert got.dtype == arr[name].dtype and np.array_equal(got, arr[name]), name
            digest = hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest()
            assert meta["digests"][name].startswith(digest + ":"), name
        assert geom.n_pairs() == len(arr["gate_indices"]) and geom.indptr.dtype == np.int32
        # re-saving what we loaded reproduces the reference file's keys, dtypes and contents
        path = str(tmp_path / "resaved.npz")
        rg.save_geometry(geom, path)
        with np.load(path) as ours, np.load(os.path.join(GOLDEN, "g8_ref_saved_geometry.npz")) as theirs:
            assert sorted(ours.files) == sorted(theirs.files) == meta["ref_file_keys"]
            for k in theirs.files:
                assert ours[k].dtype == theirs[k].dtype and ours[k].shape == theirs[k].shape, k
                assert np.array_equal(ours[k], theirs[k]), k

    def test_reference_read_our_file(self):
        from conftest import load_golden
        meta, _ = load_golden("g8_interchange")
        outcome = meta["reference_read_our_file"]
        assert outcome and all(outcome.values()), outcome       # every field equal, and the reference gridded the same bits
        assert meta["ref_file_keys"] == meta["our_file_keys"]
        assert meta["ref_file_dtypes"] == meta["our_file_dtypes"]


@pytest.fixture
def radar():
    rng = np.random.default_rng(0)
    nrays, ngates = 100, 500
    dbz = rng.normal(20, 15, size=(nrays, ngates)).astype(np.float32)
    dbz[3, 10:20] = np.nan
    dbz[4, 5] = np.inf
    rho = rng.uniform(0.5, 1.0, size=(nrays, ngates)).astype(np.float32)
    r = SimpleNamespace(nrays=nrays, ngates=ngates)
    r.fields = {"DBZH": {"data": np.ma.array(dbz, mask=dbz < -10)}, "RHOHV": {"data": rho}}
    r.gate_altitude = {"data": rng.uniform(0, 15000, size=(nrays, ngates))}
    r.range = {"data": np.arange(ngates) * 250.0}
    r.elevation = {"data": np.repeat(np.array([0.5, 1.5, 3.0, 10.0, 30.0]), 20)}
    return r


class TestGateFilter:             # reference: tests/test_radar_grid_filters.py
    def test_starts_empty(self, radar):
        gf = rg.GateFilter(radar)
        assert gf.n_gates == 50000 and gf.n_excluded() == 0 and gf.n_included() == 50000
        assert gf.gate_included.all()

    @pytest.mark.parametrize("method,args,op", [
        ("exclude_below", (5.0,), ("below", 5.0, 0)), ("exclude_above", (40.0,), ("above", 40.0, 0)),
        ("exclude_between", (0.0, 10.0), ("between", 0.0, 10.0)), ("exclude_outside", (0.0, 30.0), ("outside", 0.0, 30.0)),
        ("exclude_equal", (20.0, 0.5), ("equal", 20.0, 0.5)), ("exclude_invalid", (), ("invalid", 0, 0))])
    def test_threshold_predicates(self, radar, method, args, op):
        gf = getattr(rg.GateFilter(radar), method)("DBZH", *args)
        raw = np.ma.getdata(radar.fields["DBZH"]["data"]).ravel()
        np.testing.assert_array_equal(gf.gate_excluded, oracle.gate_mask(op[0], raw, op[1], op[2]))

    def test_nan_not_excluded_by_threshold(self, radar):
        gf = rg.GateFilter(radar).exclude_below("DBZH", 1e9)
        assert not gf.gate_excluded[3 * 500 + 10]           # NaN < x is False (filters.py:134)

    def test_missing_field_is_a_warning_noop(self, radar, caplog):
        with caplog.at_level("WARNING", logger="radar_grid.filters"):
            gf = rg.GateFilter(radar).exclude_below("KDP", 1.0)
        assert gf.n_excluded() == 0 and "not found in radar" in caplog.text

    def test_masked_and_all_invalid(self, radar):
        m = np.ma.getmaskarray(radar.fields["DBZH"]["data"]).ravel()
        np.testing.assert_array_equal(rg.GateFilter(radar).exclude_masked("DBZH").gate_excluded, m)
        inv = np.ma.getmaskarray(np.ma.masked_invalid(radar.fields["DBZH"]["data"])).ravel()
        np.testing.assert_array_equal(rg.GateFilter(radar).exclude_all_invalid("DBZH").gate_excluded, inv)
        assert rg.GateFilter(radar).exclude_masked("RHOHV").n_excluded() == 0   # plain ndarray: nothing masked

    def test_geometric_predicates(self, radar):
        alt = radar.gate_altitude["data"].ravel()
        np.testing.assert_array_equal(rg.GateFilter(radar).exclude_below_altitude(2000.0).gate_excluded, alt < 2000.0)
        np.testing.assert_array_equal(rg.GateFilter(radar).exclude_above_altitude(9000.0).gate_excluded, alt > 9000.0)
        rng2d = np.broadcast_to(radar.range["data"], (100, 500)).ravel()
        np.testing.assert_array_equal(rg.GateFilter(radar).exclude_below_range(5000.0).gate_excluded, rng2d < 5000.0)
        np.testing.assert_array_equal(rg.GateFilter(radar).exclude_above_range(100000.0).gate_excluded, rng2d > 100000.0)
        el = np.repeat(radar.elevation["data"], 500)
        np.testing.assert_array_equal(rg.GateFilter(radar).exclude_below_elevation_angle(2.0).gate_excluded, el < 2.0)
        np.testing.assert_array_equal(rg.GateFilter(radar).exclude_above_elevation_angle(20.0).gate_excluded, el > 20.0)
        np.testing.assert_array_equal(rg.GateFilter(radar).exclude_outside_elevation_range(1.0, 20.0).gate_excluded,
                                      (el < 1.0) | (el > 20.0))

    def test_chaining_copy_reset_custom(self, radar):
        gf = rg.GateFilter(radar).exclude_below("DBZH", 0.0).exclude_below("RHOHV", 0.8)
        assert len(gf._filter_history) == 2 and "Filters applied (2)" in gf.summary()
        dup = gf.copy()
        dup.exclude_all()
        assert dup.n_excluded() == 50000 and gf.n_excluded() < 50000
        assert gf.reset().n_excluded() == 0 and gf.include_all() is gf
        with pytest.raises(ValueError, match="doesn't match n_gates"):
            gf.exclude_where(np.zeros(7, dtype=bool))
        gf.exclude_where(np.ones((100, 500), dtype=bool), "everything")
        assert gf.n_excluded() == 50000
        gf2 = rg.GateFilter(radar).exclude_by_function("RHOHV", lambda x: x < 0.75, "low rho")
        assert "RHOHV: low rho" in gf2._filter_history[0]

    def test_create_mask_from_filter(self, radar):
        gf = rg.GateFilter(radar).exclude_below("RHOHV", 0.8)
        data, mask = rg.create_mask_from_filter(radar, "DBZH", gf)
        assert data.dtype == np.float32 and data.shape == (50000,)
        inv = np.ma.getmaskarray(np.ma.masked_invalid(radar.fields["DBZH"]["data"])).ravel()
        np.testing.assert_array_equal(mask, inv | gf.gate_excluded)
        _, mask0 = rg.create_mask_from_filter(radar, "DBZH")
        np.testing.assert_array_equal(mask0, inv)


class TestAdaptors:
    def test_field_and_coordinates(self):
        from radar_processor_amd import synthetic
        vol = synthetic.make_volume(n_elev=2, n_az=8, n_gates=16, seed=3, fields=("DBZH",))
        radar = vol.as_radar()
        gx, gy, gz = rg.get_gate_coordinates(radar)
        assert gx.dtype == np.float32 and gx.shape == (2 * 8 * 16,)
        np.testing.assert_array_equal(gz, vol.gate_z)
        f = rg.get_field_data(radar, "DBZH")
        assert isinstance(f, np.ma.MaskedArray) and f.dtype == np.float32 and f.shape == gx.shape
        assert rg.get_available_fields(radar) == ["DBZH"]
        info = rg.get_radar_info(radar)
        assert info["nrays"] == 16 and info["ngates"] == 16 and info["total_gates"] == 256 and info["volume_nr"] == "03"


class TestErrorBehaviourBeforeTheGpu:
    """Argument errors are raised by host code, identically with or without a device."""

    def test_bad_filter_type(self, geometry):
        with pytest.raises(ValueError, match="additional_filters must be a list of GateFilter objects"):
            rg.apply_geometry(geometry, np.ma.zeros(240), additional_filters=42)

    def test_builder_validation(self, tmp_path):
        z = np.zeros(3, dtype=np.float32)
        with pytest.raises(ValueError, match="temp_dir does not exist"):
            rg.compute_grid_geometry(z, z, z, (1, 1, 1), ((0, 0), (0, 0), (0, 0)), str(tmp_path / "missing"))
        with pytest.raises(ValueError, match="Unknown weighting function: idw"):
            rg.compute_grid_geometry(z, z, z, (1, 1, 1), ((0, 0), (0, 0), (0, 0)), str(tmp_path), weighting="idw")

    def test_cappi_scalar_paths_need_no_gpu(self, geometry, caplog):
        grid = np.arange(120, dtype=np.float32).reshape(4, 5, 6)
        with caplog.at_level("WARNING", logger="radar_grid.products"):
            out = rg.constant_altitude_ppi(grid, geometry, 99999.0)
        assert out.dtype == np.float32 and np.all(np.isnan(out)) and "outside grid range" in caplog.text
        level = rg.constant_altitude_ppi(grid, geometry, 1000.0)           # exact level 1: a view
        assert np.shares_memory(level, grid) and np.array_equal(level, grid[1])
        assert np.array_equal(rg.constant_altitude_ppi(grid, geometry, 1400.0, "nearest"), grid[1])
        with pytest.raises(ValueError, match="Unknown interpolation method: cubic"):
            rg.constant_altitude_ppi(grid, geometry, 1400.0, "cubic")
        with pytest.raises(ValueError, match="geometry is required when using altitude-based limits"):
            rg.column_max(grid, z_min_alt=10.0)


class TestNativeLibrary:
    def test_header_symbols_are_exported(self):
        """Every function declared in include/radargrid_hip.h is exported by the built library, and the
        ctypes table binds exactly that set."""
        header = open(os.path.join(REPO, "include", "radargrid_hip.h")).read()
        declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(rg_\w+)\s*\(", header, flags=re.M))
        assert len(declared) >= 15
        assert os.path.exists(_native.LIB_PATH), "build with `python -m radar_processor_amd.build`"
        lib = ctypes.CDLL(_native.LIB_PATH)
        for name in sorted(declared):
            assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert declared == set(_native.SIGNATURES)

    def test_version_and_constants(self):
        lib = rg.load_library(require_device=False)
        assert lib.rg_version() == _native.ABI_VERSION == 104
        header = open(os.path.join(REPO, "include", "radargrid_hip.h")).read()
        assert f"0x{_native.RG_EXCLUDED_BITS:08X}" in header.upper().replace("0X", "0x")
        assert np.isnan(np.array([_native.RG_EXCLUDED_BITS], dtype=np.uint32).view(np.float32)[0])

    def test_argument_validation_without_compute(self):
        """Entry points reject bad arguments before touching the device."""
        lib = rg.load_library(require_device=False)
        assert lib.rg_csr_apply_f32(None, 0, None, None, 4, 0, 0, None, 1, 1, 0, 0.0, None, None) == _native.RG_EINVAL
        assert b"null" in lib.rg_last_error()
        assert lib.rg_csr_apply_f32(1 << 12, 0, None, None, 4, 0, 0, None, 9, 8, 0, 0.0, 1 << 12, None) == _native.RG_EUNSUPPORTED
        # n_vox must be a whole number of grid lines; the compact chunk count follows the documented formula
        assert lib.rg_csr_apply_f32(1 << 12, 0, None, None, 10, 0, 4, None, 1, 1, 0, 0.0, 1 << 12, None) == _native.RG_EINVAL
        assert lib.rg_csr_compact_chunks(40 * 2000 * 2000, 2000, 2000) == 40 * 500 * 32
        assert lib.rg_csr_compact_chunks(3 * 17 * 29, 29, 17) == 3 * 5 * 1
        assert lib.rg_csr_compact_chunks(1000, 0, 0) == 16 and lib.rg_csr_compact_chunks(10, 4, 0) == _native.RG_EINVAL
        assert lib.rg_column_reduce_f32(1 << 12, 4, 16, 0, 7, 0, 1 << 12, None, None) == _native.RG_EINVAL
        assert lib.rg_gate_mask_f32(1 << 12, 8, 99, 0.0, 0.0, 1 << 12, None) == _native.RG_EINVAL
        assert lib.rg_geom_bin_workspace_bytes(-1, 4, 4) == _native.RG_EINVAL
        assert lib.rg_geom_bin_workspace_bytes(1000, 4, 4) > 4 * 1000 * 4
        # per-level gate lists: the level count must be the grid's, the reach bound needs 0 <= beam_factor < 0.5
        assert lib.rg_geom_bin_levels_workspace_bytes(-1, 10, 16) == _native.RG_EINVAL
        assert lib.rg_geom_bin_levels_workspace_bytes(1000, 9000, 5 * 16) > 4 * 9000 * 4
        cells = _native.CellGrid(x0=0.0, y0=0.0, inv_cx=1.0, inv_cy=1.0, z_lo=0.0, z_hi=1.0, ncx=4, ncy=4, levels=5, level0=0)
        args = (1 << 12, 1 << 12, 1 << 12, 10, 0.0, 1.0, cells, 1 << 12)
        assert lib.rg_geom_bin_levels_count(*args, 4, 250.0, 0.01746, 1 << 12, None) == _native.RG_EINVAL        # 4 levels != 5
        assert lib.rg_geom_bin_levels_count(*args, 5, 250.0, 0.6, 1 << 12, None) == _native.RG_EUNSUPPORTED
        assert b"beam_factor" in lib.rg_last_error()
        assert lib.rg_geom_bin_levels_count(*args, 5, 250.0, 0.01746, None, None) == _native.RG_EINVAL          # no total
        cells.level0 = 2                                                    # a slab view is for the search kernels only
        assert lib.rg_geom_bin_gates_levels_f32(*args, 5, 250.0, 0.01746, 10, 1 << 12, 1 << 12, 1 << 12, 1 << 20,
                                                None) == _native.RG_EINVAL
        # the search kernels check that a slab of levels lies inside the binned levels
        cells.level0 = 3
        assert lib.rg_geom_count_f32(1 << 12, 1 << 12, cells, 1 << 12, 1 << 12, 1 << 12, 3, 4, 4, 250.0, 0.01746, 1 << 12,
                                     None) == _native.RG_EINVAL and b"levels 3 .. 5" in lib.rg_last_error()

    def test_product_library_refuses_experiment_codes(self):
        """Timing-only kernels (results wrong by construction) and tuning variants are not in the shipped library: their
        tile / variant codes are RG_EINVAL (validation only, nothing is launched), and no DIAG instantiation of the row-wise
        kernel is linked in."""
        import subprocess
        lib = rg.load_library(require_device=False)
        p = 1 << 12                                   # never dereferenced: every call fails validation first
        for tile in (2100, 2102, 2116, 2199, 2201, 2264, 2265, 1999, 385):
            st = lib.rg_csr_compact_apply_packed_f32(p, 0, p, p, _native.RG_REC_ORDER_DISPATCH, 120 << 23, p, p, 64, 0, 64, 1,
                                                     p, 1, 1, 64, 0.0, p, 256, tile, None)
            assert st == _native.RG_EINVAL, (tile, st)
        for tile in (901, 903, 909, 5384, 1000):
            st = lib.rg_csr_compact_apply_f32(p, 0, p, p, p, p, 64, 0, 64, 1, p, 1, 1, 64, 0.0, p, 256, tile, None)
            assert st == _native.RG_EINVAL, (tile, st)
        for variant in (8, 9, 16, 19, 22, 28, 640):
            st = lib.rg_csr_apply_f32_ex(p, 0, p, p, 64, 0, 64, p, 1, 1, 64, 0.0, p, variant, None)
            assert st == _native.RG_EINVAL, (variant, st)
        import shutil
        nm_exe = shutil.which("nm")
        assert nm_exe, "binutils nm is part of the image"
        nm = subprocess.run([nm_exe, "-C", "--defined-only", _native.LIB_PATH], capture_output=True, text=True)
        if nm.returncode == 0:       # host-side kernel stubs carry the template arguments
            rows = [l for l in nm.stdout.splitlines() if "csr_compact_rowwise_kernel<" in l]
            assert rows, "row-wise kernel not found in the symbol table"
            for l in rows:           # <IndT, NF, STRIDE, DIAG, COLS, REGS>: DIAG must be 0 everywhere
                args = l[l.index("csr_compact_rowwise_kernel<") + len("csr_compact_rowwise_kernel<"):].split(">(")[0].split(",")
                assert len(args) == 6 and args[3].strip() == "0", l

    def test_ensure_built_rebuilds_a_stale_library(self, tmp_path, monkeypatch):
        """ensure_built() compiles when the library is missing AND when it was built from other sources or headers than
        the tree holds (content digest in the .stamp file next to it) -- and leaves an up-to-date library alone."""
        from radar_processor_amd import build as rg_build
        src, hdr = tmp_path / "k.hip", tmp_path / "k.hpp"
        src.write_text("// kernel\n")
        hdr.write_text("// header v1\n")
        lib = tmp_path / "lib.so"
        calls = []

        def fake_build(force=False, verbose=True):
            calls.append(force)
            lib.write_text("built")
            (tmp_path / "lib.so.stamp").write_text(rg_build.source_digest() + "\n")
            return str(lib)
        monkeypatch.setattr(rg_build, "sources_and_headers", lambda: [str(src), str(hdr)])
        monkeypatch.setattr(rg_build, "LIB_PATH", str(lib))
        monkeypatch.setattr(rg_build, "STAMP_PATH", str(lib) + ".stamp")
        monkeypatch.setattr(rg_build, "CSRC", str(tmp_path))
        monkeypatch.setattr(rg_build, "build", fake_build)
        assert rg_build.is_stale()
        rg_build.ensure_built(verbose=False)
        assert calls == [False] and not rg_build.is_stale()
        rg_build.ensure_built(verbose=False)                      # up to date: nothing happens
        assert calls == [False]
        os.utime(hdr, (1, 1))                                     # time stamps alone do not matter (snapshots reorder them)
        assert not rg_build.is_stale()
        hdr.write_text("// header v2: a signature changed\n")    # a header edit does
        assert rg_build.is_stale()
        rg_build.ensure_built(verbose=False)
        assert calls == [False, True] and not rg_build.is_stale()
        (tmp_path / "lib.so.stamp").unlink()                      # a library of unknown provenance is rebuilt too
        assert rg_build.is_stale()

    def test_the_in_tree_library_matches_its_sources(self):
        from radar_processor_amd import build as rg_build
        assert not rg_build.is_stale(), "libradargrid_hip.so was built from other sources: python -m radar_processor_amd.build"

    def test_loader_refuses_a_library_of_another_abi_version(self, monkeypatch):
        """A stale library whose signatures differ must not be called with shifted arguments: the loader
        compares rg_version() with the version this binding was written for."""
        monkeypatch.setattr(_native, "_lib", None)
        monkeypatch.setattr(_native, "ABI_VERSION", _native.ABI_VERSION + 1)
        with pytest.raises(rg.NativeUnavailable, match="header version"):
            _native.load_library(require_device=False)
        monkeypatch.undo()
        assert _native.load_library(require_device=False).rg_version() == _native.ABI_VERSION
        assert "RG_LIBRARY" not in open(_native.__file__).read().replace("no environment override", "")

    def test_columns_entry_point_validates_before_touching_the_device(self):
        """rg_csr_compact_apply_columns_f32 / rg_csr_columns_workspace_bytes: argument checks only (nothing is launched)."""
        lib = rg.load_library(require_device=False)
        p = 1 << 12

        def call(**kw):
            a = dict(indptr=p, is64=0, rec=p, rec_ptr=p, order=_native.RG_REC_ORDER_DISPATCH, w_base=120 << 23, dict_ptr=p, dict=p,
                     n_vox=2 * 4 * 64, n_pairs=0, nx=64, ny=4, packed=p, nf=3, stride=4, n_gates=100, fill=0.0, out=p, planes=None,
                     keep_lo=0, n_keep=0, cmax=None, carg=None, lo=0, hi=1, window=256, pieces=1, wg_order=None, ws=None,
                     ws_bytes=0, hint=0)
            a.update(kw)
            return lib.rg_csr_compact_apply_columns_f32(*a.values(), None)
        assert call(out=None) == _native.RG_EINVAL and b"nothing to produce" in lib.rg_last_error()
        assert call(pieces=3) == _native.RG_EINVAL and call(pieces=0) == _native.RG_EINVAL          # 2 planes
        assert call(planes=p, keep_lo=1, n_keep=2) == _native.RG_EINVAL
        assert call(carg=p) == _native.RG_EINVAL
        assert call(cmax=p, lo=1, hi=0) == _native.RG_EINVAL and call(cmax=p, hi=2) == _native.RG_EINVAL
        assert call(cmax=p, pieces=2) == _native.RG_EWORKSPACE and b"workspace" in lib.rg_last_error()
        assert call(hint=3) == _native.RG_EINVAL and call(hint=1000) == _native.RG_EINVAL     # no experiment variants here
        assert call(nf=5, stride=8) == _native.RG_EUNSUPPORTED and call(stride=2) == _native.RG_EINVAL
        assert call(w_base=1) == _native.RG_EINVAL and call(order=7) == _native.RG_EINVAL
        assert call(n_vox=100) == _native.RG_EINVAL                                             # not planes x lines x rows
        assert call(packed=p + 4) == _native.RG_EALIGN
        assert lib.rg_csr_columns_workspace_bytes(2000, 2000, 4, 3) == 3 * 4 * 4_000_000 * 8
        assert lib.rg_csr_columns_workspace_bytes(4, 64, 3, 1) == 0
        assert lib.rg_csr_columns_workspace_bytes(4, 64, 5, 2) == _native.RG_EINVAL

    def test_plane_products_request_and_layout_digest(self):
        """PlaneProducts validates like the reference's functions do; the sidecar key changes with the CSR and with the
        chunk layout constants, not with anything else."""
        from radar_processor_amd import grid_geometry
        with pytest.raises(ValueError, match="Unknown interpolation method"):
            rg.PlaneProducts(cappi=(1000.0,), interpolation="cubic")
        spec = rg.PlaneProducts(colmax=False, argmax=True, cappi=[2500, 4000.0], z_min_alt=1000.0, fused=True)
        assert spec.colmax and spec.argmax and spec.cappi == (2500.0, 4000.0) and spec.window == (None, None, 1000.0, None)
        assert spec.fused is True and rg.PlaneProducts().fused is None
        g = rg.GridGeometry((1, 2, 2), ((0.0, 0.0), (0.0, 1.0), (0.0, 1.0)), np.array([0, 1, 2, 2, 3], dtype=np.int32),
                            np.array([0, 1, 2], dtype=np.int32), np.array([1.0, 0.5, 0.25], dtype=np.float32), toa=17000.0)
        key = grid_geometry._reference_arrays_digest(g)
        assert key == grid_geometry._reference_arrays_digest(g) and len(key) == 64
        g2 = rg.GridGeometry(g.grid_shape, ((0.0, 9.0), (0.0, 1.0), (0.0, 1.0)), g.indptr.copy(), g.gate_indices.copy(),
                             g.weights.copy(), toa=1.0)
        assert grid_geometry._reference_arrays_digest(g2) == key          # limits / toa are not what the copy is derived from
        g2.weights = np.array([1.0, 0.5, 0.125], dtype=np.float32)
        assert grid_geometry._reference_arrays_digest(g2) != key

    @pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a box without a GPU")
    def test_no_cpu_fallback(self, geometry):
        """Without a HIP device the product path raises instead of computing on the CPU."""
        field = np.ma.masked_invalid(np.arange(240, dtype=np.float32))
        with pytest.raises(rg.NativeUnavailable):
            rg.apply_geometry(geometry, field)
        with pytest.raises(rg.NativeUnavailable):
            rg.column_max(np.zeros((4, 5, 6), dtype=np.float32))
        with pytest.raises(rg.NativeUnavailable):
            rg.constant_altitude_ppi(np.zeros((4, 5, 6), dtype=np.float32), geometry, 1500.0)
        # the raster stage and the 2-D filters have no CPU path either
        plane = np.ma.masked_invalid(np.linspace(-30, 60, 30, dtype=np.float32).reshape(5, 6))
        with pytest.raises(rg.NativeUnavailable):
            rg.collapse_field_3d_to_2d(np.zeros((4, 5, 6), dtype=np.float32), "colmax")
        with pytest.raises(rg.NativeUnavailable):
            rg.apply_filter_masks(plane, [type("F", (), {"field": "DBZH", "min": 0.0, "max": None})()], [], "DBZH", {"qc": {}})
        with pytest.raises(rg.NativeUnavailable):
            rg.apply_colormap_to_array(plane.filled(np.nan), np.array([[0, 0, 0], [1, 1, 1]], dtype=float), 0.0, 1.0)
        with pytest.raises(rg.NativeUnavailable):
            rg.GridFilter().apply_below(plane.filled(np.nan), 15.0)
        import torch
        with pytest.raises(rg.NativeUnavailable):                       # the products-only path has no CPU route either
            rg.grid_products_device(geometry, [torch.zeros(240)], products=rg.PlaneProducts(cappi=(1500.0,)))
        with pytest.raises(rg.NativeUnavailable):
            rg.compute_grid_geometry(np.zeros(4, np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32), (1, 2, 2),
                                     ((0.0, 0.0), (0.0, 1.0), (0.0, 1.0)), ".", layout="compact")
        # asking how many fields a pass over this geometry fuses is a question about its DEVICE copy: no CPU answer
        from radar_processor_amd import batch, gridding
        with pytest.raises(rg.NativeUnavailable):
            gridding.fields_per_pass(geometry)
        with pytest.raises(rg.NativeUnavailable):
            batch.VolumeBatch(geometry, ["DBZH"]).volumes_per_pass

    def test_product_package_never_imports_the_oracle(self):
        pkg_dir = os.path.dirname(rg.__file__)
        for root, _, files in os.walk(pkg_dir):
            for f in files:
                if f.endswith(".py"):
                    text = open(os.path.join(root, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f"{f} imports the oracle"


class TestBenchLauncher:
    """bench.py --gpus N: never a smaller run labelled N.  CPU-only checks of the refusals; the
    launches themselves are covered on the GPU box (tests/test_gpu_batch.py)."""

    def _run(self, args, env_extra):
        import subprocess
        import sys
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        env.update(env_extra)
        return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], capture_output=True, env=env,
                              timeout=300, cwd=REPO)

    def test_world_size_must_match_gpus(self):
        res = self._run(["--gpus", "4"], dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
        assert res.returncode == 2 and b"WORLD_SIZE=1 but --gpus 4" in res.stderr and not res.stdout.strip()

    def test_launcher_failure_path(self):
        """A rank that dies before the rendezvous must not leave its siblings (and the parent) waiting for the
        process-group timeout: the launcher stops them and returns non-zero within seconds, without a JSON line."""
        import time
        t0 = time.time()
        res = self._run(["--gpus", "2", "--share-device", "--rendezvous-only", "--fail-rank", "1", "--pg-timeout", "300"], {})
        took = time.time() - t0
        assert res.returncode != 0 and not res.stdout.strip(), res.stderr[-600:]
        assert b"rank 1 exited with code 3: stopping the other ranks" in res.stderr
        assert took < 30, f"launcher needed {took:.0f} s to notice a dead rank"

    def test_launcher_passes_rank0_line_through(self):
        res = self._run(["--gpus", "2", "--share-device", "--rendezvous-only"], {})
        assert res.returncode == 0, res.stderr[-600:]
        line = json.loads(res.stdout.decode().strip().splitlines()[-1])
        # one entry per rank, in rank order (the gather every real N-GPU line uses for per_rank_kernel_ms)
        assert line == {"rendezvous": "ok", "n_gpus": 2, "per_rank_kernel_ms": [10.0, 11.0]}

    def test_per_rank_list_has_one_entry_per_rank(self):
        res = self._run(["--gpus", "3", "--share-device", "--rendezvous-only"], {})
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        line = json.loads(res.stdout.decode().strip().splitlines()[-1])
        assert line["n_gpus"] == 3 and line["per_rank_kernel_ms"] == [10.0, 11.0, 12.0]

    def test_parent_refuses_more_ranks_than_gpus(self):
        import torch
        if torch.cuda.device_count() >= 8:
            pytest.skip("box has 8 GPUs")
        res = self._run(["--gpus", "8"], {})
        assert res.returncode == 2 and b"refusing to label a smaller run" in res.stderr and not res.stdout.strip()


class TestCompactLayoutHostLogic:
    """Chunk / segment arithmetic of the compact CSR copy (grid_geometry.CompactCSR) -- the host side must cut lines and
    number chunks exactly as csrc/rg_csr_compact.hip does (checked against rg_csr_compact_chunks, no GPU needed)."""

    def test_segments_are_balanced_and_cover_the_line(self):
        from radar_processor_amd.grid_geometry import CompactCSR
        for nx in (1, 5, 63, 64, 65, 128, 255, 300, 1000, 2000, 4097):
            st = CompactCSR.segment_starts(nx)
            widths = np.diff(st)
            assert st[0] == 0 and st[-1] == nx and len(widths) == (nx + 63) // 64
            assert widths.max() <= 64 and widths.max() - widths.min() <= 1 and (np.diff(widths) <= 0).all()

    def test_chunk_numbering_matches_the_library(self):
        import torch
        from radar_processor_amd.grid_geometry import CompactCSR
        lib = rg.load_library(require_device=False)
        for shape in ((1, 1, 1), (3, 17, 29), (2, 5, 300), (40, 2000, 2000), (20, 1000, 1000), (1, 4, 64)):
            nz, ny, nx = shape
            nsx, nyg, n_chunks = CompactCSR.layout(shape)
            assert n_chunks == lib.rg_csr_compact_chunks(nz * ny * nx, nx, ny)
            assert nsx == (nx + 63) // 64 and nyg == (ny + _native.RG_COMPACT_LINES - 1) // _native.RG_COMPACT_LINES
        shape = (2, 6, 150)                          # 3 segments of 50 rows, 2 line groups (4 + 2 lines), 2 planes
        rows = torch.arange(2 * 6 * 150)
        chunk = CompactCSR.chunk_of_rows(rows, shape).view(2, 6, 150)
        assert chunk.max().item() + 1 == CompactCSR.layout(shape)[2] == 2 * 2 * 3
        for z in range(2):
            for y in range(6):
                for sx in range(3):
                    want = (z * 2 + y // 4) * 3 + sx
                    assert bool((chunk[z, y, sx * 50:(sx + 1) * 50] == want).all())

    def test_dispatch_order_slots_invert_the_block_rotation(self):
        """RG_REC_ORDER_DISPATCH: slot = block * H + wavefront for the block -> chunk rotation of the apply kernels
        (block_chunk in csrc/rg_csr_compact.hip), restated here in plain Python; every segment gets exactly one slot,
        the slots of one workgroup are neighbours, and record_pointers lays the records out in slot order."""
        import torch
        from radar_processor_amd.grid_geometry import CompactCSR
        H, rot = _native.RG_COMPACT_LINES, _native.RG_COMPACT_ROTATION
        for shape in ((1, 1, 1), (3, 17, 29), (2, 6, 150), (2, 9, 700), (3, 8, 2000), (1, 4, 64)):
            nz, ny, nx = shape
            nsx, nyg, n_chunks = CompactCSR.layout(shape)
            want = {}
            for bid in range(n_chunks):                       # the kernel's map, forwards
                grp, col = divmod(bid, nsx)
                sx = (col + ((grp * rot) & 0xFFFFFFFF) % nsx) % nsx
                plane, yg = divmod(grp, nyg)
                for w in range(H):
                    y = yg * H + w
                    if y < ny:
                        want[(plane * ny + y, sx)] = bid * H + w
            line = torch.arange(nz * ny)[:, None].expand(nz * ny, nsx)
            sx = torch.arange(nsx)[None, :].expand(nz * ny, nsx)
            got = CompactCSR.slot_of_segments(line, sx, shape, _native.RG_REC_ORDER_DISPATCH)
            assert got.shape == (nz * ny, nsx) and len(want) == nz * ny * nsx
            assert all(int(got[l, s]) == v for (l, s), v in want.items())
            assert len(set(got.reshape(-1).tolist())) == got.numel()          # one slot per segment
            seg = CompactCSR.slot_of_segments(line, sx, shape, _native.RG_REC_ORDER_SEGMENT)
            assert torch.equal(seg, line * nsx + sx)
            # record_pointers: ceil(pairs / 3) records per segment, in slot order, empty slots past a plane's last line
            rng = np.random.default_rng(nx)
            lens = rng.integers(0, 9, nz * ny * nx)
            indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
            st = CompactCSR.segment_starts(nx)
            for order in (_native.RG_REC_ORDER_SEGMENT, _native.RG_REC_ORDER_DISPATCH):
                rp = CompactCSR.record_pointers(indptr, shape, order)
                slots = CompactCSR.slot_of_segments(line, sx, shape, order)
                assert rp.numel() == (nz * ny * nsx if order == _native.RG_REC_ORDER_SEGMENT else n_chunks * H) + 1
                n_rec = rp[1:] - rp[:-1]
                total = 0
                for l in range(nz * ny):
                    for s in range(nsx):
                        pairs = int(indptr[l * nx + st[s + 1]] - indptr[l * nx + st[s]])
                        assert int(n_rec[slots[l, s]]) == (pairs + 2) // 3
                        total += (pairs + 2) // 3
                assert int(rp[-1]) == total and int(rp[0]) == 0

    def test_window_choice(self):
        import torch
        from radar_processor_amd.grid_geometry import CompactCSR
        c = CompactCSR(torch.zeros(0, dtype=torch.int16), torch.zeros(5, dtype=torch.int64), torch.zeros(0, dtype=torch.int32),
                       900, 768, (1, 16, 64), chunk_pairs=torch.tensor([100, 100, 100, 700]),
                       chunk_counts=torch.tensor([10, 300, 800, 900]))
        assert c.window_for(1) == 768 and c.window_for(2) == 768 and c.window_for(3) == 768 and c.window_for(8) == 768
        c.window_cap = 4096
        assert c.window_for(1) == 4096 and c.window_for(3) == 2688 and c.window_for(4) == 2048 and c.window_for(8) == 1024
        assert [c.entry_bytes(n) for n in (1, 2, 3, 4, 5, 8)] == [8, 8, 12, 16, 32, 32]      # 32 KiB of LDS
        assert [c.entry_bytes(n, rowwise=True) for n in (1, 2, 3, 4, 5, 8)] == [8, 8, 16, 20, 40, 40]
        assert c.window_for(8, rowwise=True) == 1216 and c.window_for(3, rowwise=True) == 3072 and c.window_for(4, rowwise=True) == 2432
        assert c.fallback_fraction(1000) == 0.0 and abs(c.fallback_fraction(850) - 0.7) < 1e-9
        assert abs(c.fallback_fraction(256) - 0.9) < 1e-9


class TestGeometryAssignment:
    """Plain attribute assignment works as on the reference dataclass: nothing is lost, caches are dropped."""

    def test_setters_keep_the_other_arrays(self):
        ip = np.array([0, 1, 3], dtype=np.int32)
        g = rg.GridGeometry((1, 1, 2), ((0, 0), (0, 0), (0, 1)), ip, np.array([0, 1, 2], dtype=np.int32),
                            np.ones(3, dtype=np.float32), toa=10.0)
        g._gridders = {"stale": object()}
        g._compact = ("stale", None)
        g.weights = np.array([0.5, 0.25, 0.25], dtype=np.float32)
        assert np.array_equal(g.indptr, ip) and np.array_equal(g.gate_indices, [0, 1, 2])
        assert np.array_equal(g.weights, [0.5, 0.25, 0.25])
        assert "_gridders" not in g.__dict__ and "_compact" not in g.__dict__ and not g.is_device_resident
        g.gate_indices = np.array([2, 1, 0], dtype=np.int32)
        g.indptr = np.array([0, 2, 3], dtype=np.int32)
        assert g.n_pairs() == 3 and np.array_equal(g.indptr, [0, 2, 3]) and np.array_equal(g.weights, [0.5, 0.25, 0.25])
        g.invalidate_device()                     # a no-op without a device copy
        assert np.array_equal(g.gate_indices, [2, 1, 0])

    def test_cappi_plan(self):
        from radar_processor_amd.grid_products import cappi_plan
        assert cappi_plan((0.0, 15000.0), 20, 99000.0) == ("outside",)
        assert cappi_plan((0.0, 19000.0), 20, 4000.0) == ("level", 4)                      # exact level: a view
        kind, k, w0, w1 = cappi_plan((0.0, 15000.0), 20, 4000.0)
        assert (kind, k) == ("blend", 5) and abs(w1 - (4000.0 / (15000.0 / 19) - 5)) < 1e-12 and abs(w0 + w1 - 1) < 1e-12
        assert cappi_plan((0.0, 15000.0), 20, 4000.0, "nearest") == ("level", 5)
        for alt in (0.0, 1234.5, 7777.0, 15000.0):
            mine, ref = cappi_plan((0.0, 15000.0), 20, alt), oracle.cappi_plan((0.0, 15000.0), 20, alt)
            assert mine[1:] == ref[1:]
        with pytest.raises(ValueError, match="Unknown interpolation method"):
            cappi_plan((0.0, 1.0), 2, 0.5, "cubic")
