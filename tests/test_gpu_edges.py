"""Edge cases of the HIP path against the CPU oracle on small *random point clouds* (the builder and the fused
gridder take arbitrary gate coordinates, not only polar volumes): degenerate grid shapes, grids that are not a
multiple of the wavefront, search boxes spanning more than 64 cell rows, constant ROI, ROI larger than the grid,
no gate in reach, everything masked, top-of-atmosphere cuts, non-finite coordinates."""
import numpy as np
import pytest

from conftest import assert_same_to_rounding
from oracle import radar_grid_oracle as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rg():
    import radar_processor_amd as pkg
    pkg.load_library()
    return pkg


def _cloud(seed, n, extent=20e3, zmax=9e3):
    rng = np.random.default_rng(seed)
    gx = rng.uniform(-extent, extent, n).astype(np.float32)
    gy = rng.uniform(-extent, extent, n).astype(np.float32)
    gz = rng.uniform(0, zmax, n).astype(np.float32)
    val = rng.normal(20, 10, n).astype(np.float32)
    mask = rng.random(n) < 0.15
    return gx, gy, gz, val, mask


def _check(rg, gx, gy, gz, val, mask, shape, limits, tmp_path, cell_size=None, **kw):
    import torch
    weighting = kw.pop("weighting", "barnes2")
    o_ip, o_idx, o_w = oracle.build_geometry(gx, gy, gz, shape, limits, weighting=weighting, **kw)
    want = oracle.csr_apply(o_ip, o_idx, o_w, val, mask, shape)
    # CSR path
    geom = rg.compute_grid_geometry(gx, gy, gz, shape, limits, str(tmp_path), weighting=weighting, **kw)
    ip, idx, w = oracle.canonical_rows(geom.indptr, geom.gate_indices, geom.weights)
    np.testing.assert_array_equal(ip, o_ip)
    np.testing.assert_array_equal(idx, o_idx)
    if weighting == "barnes2":
        assert np.abs(w.view(np.int32).astype(np.int64) - o_w.view(np.int32).astype(np.int64)).max(initial=0) <= 1
    else:
        np.testing.assert_array_equal(w, o_w)
    got = rg.apply_geometry(geom, np.ma.array(val, mask=mask))
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5 * 60, equal_nan=True)
    # fused path (optionally with a forced cell size)
    kw2 = {k: v for k, v in kw.items()}
    search = rg.RoiSearch(gx, gy, gz, shape, limits, cell_size=cell_size, **kw2)
    f = torch.from_numpy(val).to(search.dev)
    m = torch.from_numpy(mask.astype(np.uint8)).to(search.dev)
    fused = rg.roi_grid_fields_device(search, [f], [m], weighting=weighting)[0].cpu().numpy()
    np.testing.assert_array_equal(np.isnan(fused), np.isnan(want))
    np.testing.assert_allclose(fused, want, rtol=1e-5, atol=1e-5 * 60, equal_nan=True)
    return geom


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 1, 65), (2, 3, 1), (3, 1, 130), (1, 7, 63), (4, 5, 64), (2, 9, 200)])
def test_degenerate_and_ragged_grid_shapes(rg, tmp_path, shape):
    gx, gy, gz, val, mask = _cloud(1, 4000)
    nz, ny, nx = shape
    limits = ((500.0, 500.0 if nz == 1 else 6000.0), (-3e3, -3e3 if ny == 1 else 9e3), (-15e3, -15e3 if nx == 1 else 15e3))
    _check(rg, gx, gy, gz, val, mask, shape, limits, tmp_path, min_radius=900.0, beam_factor=0.05)


@pytest.mark.parametrize("weighting", ["barnes2", "cressman", "nearest"])
def test_search_box_spanning_more_than_64_cell_rows(rg, tmp_path, weighting):
    """ROI 3 km over 40 m cells = 150 cell rows per voxel: exercises the chunked row-bounds loop of the fused
    gridder (and long runs in the builder)."""
    gx, gy, gz, val, mask = _cloud(2, 6000, extent=8e3, zmax=3e3)
    shape, limits = (2, 6, 70), ((0.0, 2000.0), (-2e3, 2e3), (-5e3, 5e3))
    _check(rg, gx, gy, gz, val, mask, shape, limits, tmp_path, cell_size=40.0, min_radius=3000.0, beam_factor=0.0,
           weighting=weighting)


def test_constant_roi_and_roi_larger_than_the_grid(rg, tmp_path):
    gx, gy, gz, val, mask = _cloud(3, 1500, extent=5e3, zmax=2e3)
    shape, limits = (2, 4, 9), ((0.0, 1000.0), (-1e3, 1e3), (-2e3, 2e3))
    geom = _check(rg, gx, gy, gz, val, mask, shape, limits, tmp_path, min_radius=50e3, beam_factor=0.0)
    assert geom.n_pairs() == 1500 * 72                     # every gate is a neighbour of every voxel


def test_no_gate_in_reach_all_masked_and_toa(rg, tmp_path):
    import torch
    gx, gy, gz, val, mask = _cloud(4, 3000)
    shape, limits = (3, 8, 70), ((0.0, 8000.0), (-10e3, 10e3), (-12e3, 12e3))
    # gates far away from the grid: empty geometry, all fill
    far = rg.compute_grid_geometry(gx + 5e5, gy, gz, shape, limits, str(tmp_path))
    assert far.n_pairs() == 0 and np.all(rg.apply_geometry(far, np.ma.array(val, mask=mask), fill_value=-1.0) == -1.0)
    # every gate masked -> all NaN although the geometry is dense
    geom = _check(rg, gx, gy, gz, val, np.ones_like(mask), shape, limits, tmp_path, min_radius=1500.0)
    assert geom.n_pairs() > 0
    # toa removes the upper gates (float32 comparison of z - radar_altitude, compute.py:182,193)
    _check(rg, gx, gy, gz, val, mask, shape, limits, tmp_path, min_radius=1500.0, toa=4000.0, radar_altitude=250.0)
    # toa = inf (load_geometry default) keeps everything
    _check(rg, gx, gy, gz, val, mask, shape, limits, tmp_path, min_radius=1500.0, toa=float("inf"))
    # non-finite gate coordinates are dropped, not propagated
    bad = gx.copy()
    bad[::50] = np.nan
    good = np.isfinite(bad)
    a = rg.compute_grid_geometry(bad, gy, gz, shape, limits, str(tmp_path), min_radius=1500.0)
    o_ip, o_idx, _ = oracle.build_geometry(gx[good], gy[good], gz[good], shape, limits, min_radius=1500.0)
    remap = np.nonzero(good)[0]
    ip, idx, _ = oracle.canonical_rows(a.indptr, a.gate_indices, a.weights)
    np.testing.assert_array_equal(ip, o_ip)
    np.testing.assert_array_equal(idx, remap[o_idx])


def test_zero_gates(rg, tmp_path):
    z = np.zeros(0, dtype=np.float32)
    geom = rg.compute_grid_geometry(z, z, z, (2, 3, 4), ((0.0, 1000.0), (-1e3, 1e3), (-1e3, 1e3)), str(tmp_path))
    assert geom.n_pairs() == 0 and geom.indptr.shape == (25,)
    out = rg.apply_geometry(geom, np.ma.array(z, mask=np.zeros(0, dtype=bool)))
    assert out.shape == (2, 3, 4) and np.all(np.isnan(out))


def test_unmasked_nan_reaches_only_its_neighbours_in_the_fused_gridder(rg):
    """Voxel blocking shares one gather between 4 voxels: a NaN gate must poison the voxels it belongs to and no
    others (compare with the CSR path, which has no blocking)."""
    import torch
    gx, gy, gz, val, mask = _cloud(5, 5000, extent=6e3, zmax=1e3)
    val[int(np.argmin(gx * gx + gy * gy + (gz - 400.0) ** 2))] = np.nan     # a gate in reach of the grid, NOT masked
    mask[:] = False
    shape, limits = (1, 12, 80), ((400.0, 400.0), (-3e3, 3e3), (-5e3, 5e3))
    o_ip, o_idx, o_w = oracle.build_geometry(gx, gy, gz, shape, limits, min_radius=700.0, beam_factor=0.0)
    want = oracle.csr_apply(o_ip, o_idx, o_w, val, mask, shape)
    search = rg.RoiSearch(gx, gy, gz, shape, limits, min_radius=700.0, beam_factor=0.0)
    fused = rg.roi_grid_fields_device(search, [torch.from_numpy(val).to(search.dev)], [None])[0].cpu().numpy()
    np.testing.assert_array_equal(np.isnan(fused), np.isnan(want))
    assert 0 < np.isnan(fused).sum() < fused.size // 2
    np.testing.assert_allclose(fused, want, rtol=1e-5, atol=6e-4, equal_nan=True)


def test_closest_gate_mode_and_processor_seam_package(rg):
    """rg_roi_grid_f32 in closest-gate mode + the GRID3D cache package of radar_processor/processor.py:170-179.
    PyART is absent: checked against this build's own brute-force statement of the rule (parity unpinned)."""
    from types import SimpleNamespace
    from radar_processor_amd import processor_seam as seam
    rng = np.random.default_rng(12)
    nrays, ngates = 60, 80
    n = nrays * ngates
    gx = rng.uniform(-9e3, 9e3, n).astype(np.float32)
    gy = rng.uniform(-9e3, 9e3, n).astype(np.float32)
    gz = rng.uniform(0, 3e3, n).astype(np.float32)
    data = rng.normal(15, 12, (nrays, ngates)).astype(np.float32)
    data[rng.random((nrays, ngates)) < 0.1] = np.nan
    radar = SimpleNamespace(nrays=nrays, ngates=ngates,
                            fields={"DBZH": {"data": np.ma.masked_invalid(data), "units": "dBZ", "long_name": "refl"}},
                            gate_x={"data": gx.reshape(nrays, ngates)}, gate_y={"data": gy.reshape(nrays, ngates)},
                            gate_z={"data": gz.reshape(nrays, ngates)},
                            latitude={"data": np.array([-31.4])}, longitude={"data": np.array([-64.2])})
    excluded = rng.random(n) < 0.2
    zl, yl, xl, res = (0.0, 2400.0), (-8000.0, 8000.0), (-6000.0, 6000.0), 400.0
    pkg = seam.build_grid3d_package(radar, "DBZH", zl, yl, xl, res, gate_excluded=excluded)
    shape = seam.grid3d_shape(zl, yl, xl, res)
    assert shape == (7, 40, 30) and pkg["arr3d"].shape == shape and isinstance(pkg["arr3d"], np.ma.MaskedArray)
    assert seam.constant_roi_for(res, yl) == max(600.0, 800 + 0.08 * 400)
    assert pkg["x"].shape == (30,) and pkg["y"].shape == (40,) and pkg["z"].shape == (7,)
    assert pkg["field_name"] == "DBZH" and pkg["field_metadata"] == {"units": "dBZ", "long_name": "refl"}
    assert pkg["projection"]["proj"] == "pyart_aeqd"
    drop = excluded | ~np.isfinite(data.ravel())
    want, gap = oracle.closest_gate_grid(gx, gy, gz, data.ravel(), drop, shape, (zl, yl, xl), seam.constant_roi_for(res, yl))
    got = np.ma.filled(pkg["arr3d"], np.nan)
    clear = gap > 1e-2                      # skip voxels whose two closest gates are closer than float32 can order
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(got[clear], want[clear])
    assert clear.mean() > 0.99 and np.isfinite(want).mean() > 0.5


def test_argument_validation_on_the_device_layer(rg):
    """Bad CSRs and bad tensors are rejected by host code before anything is launched."""
    import torch
    dev = torch.device("cuda", 0)
    lim = ((0, 1), (0, 1), (0, 1))
    field = np.ma.masked_invalid(np.arange(4, dtype=np.float32))
    bad = [
        (np.array([0, 2, 1, 4, 4], dtype=np.int32), np.arange(4, dtype=np.int32), "non-decreasing"),       # not monotone
        (np.array([1, 2, 3, 4, 4], dtype=np.int32), np.arange(4, dtype=np.int32), "start at 0"),            # indptr[0] != 0
        (np.array([0, 1, 2, 3], dtype=np.int32), np.arange(4, dtype=np.int32), "entries"),                  # wrong length
        (np.array([0, 1, 2, 3, 4], dtype=np.int32), np.array([0, -1, 2, 3], dtype=np.int32), "negative"),   # negative index
    ]
    for indptr, idx, msg in bad:
        g = rg.GridGeometry((1, 2, 2), lim, indptr, idx, np.ones(4, dtype=np.float32), toa=1.0)
        with pytest.raises(ValueError, match=msg):
            rg.apply_geometry(g, field)
    g = rg.GridGeometry((1, 2, 2), lim, np.arange(5, dtype=np.int32), np.arange(4, dtype=np.int32), np.ones(3, dtype=np.float32), toa=1.0)
    with pytest.raises(ValueError, match="differ in length"):
        rg.apply_geometry(g, field)
    good = rg.GridGeometry((1, 2, 2), lim, np.arange(5, dtype=np.int32), np.arange(4, dtype=np.int32), np.ones(4, dtype=np.float32), toa=1.0)
    f32 = torch.arange(4, dtype=torch.float32, device=dev)
    with pytest.raises(ValueError, match="contiguous cuda float32"):
        rg.grid_fields_device(good, [f32.double()])
    with pytest.raises(ValueError, match="contiguous cuda float32"):
        rg.grid_fields_device(good, [torch.arange(8, dtype=torch.float32, device=dev)[::2]])
    with pytest.raises(ValueError, match="uint8"):
        rg.grid_fields_device(good, [f32], [torch.zeros(4, dtype=torch.bool, device=dev)])
    with pytest.raises(ValueError, match="one entry"):
        rg.grid_fields_device(good, [f32], [None, None])
    with pytest.raises(ValueError, match="no fields"):
        rg.grid_fields_device(good, [])
    with pytest.raises(rg.NativeUnavailable):
        rg.grid_fields_device(good, [f32.cpu()])
    out = rg.grid_fields_device(good, [f32])
    assert out.shape == (1, 1, 2, 2) and torch.equal(out.view(-1), f32)
    with pytest.raises(ValueError, match="unknown gate predicate"):
        rg.device_gate_mask(f32, "sideways")
    with pytest.raises(ValueError, match="device grids must be float32"):
        rg.column_max(torch.zeros((2, 2, 2), dtype=torch.float64, device=dev))
    with pytest.raises(ValueError, match="Unknown weighting function"):
        rg.roi_grid_fields_device(rg.RoiSearch(np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros(1, np.float32),
                                               (1, 1, 1), ((0, 0), (0, 0), (0, 0))), [torch.zeros(1, device=dev)],
                                  weighting="gaussian")


def test_device_copies_are_cached_whatever_the_device_spelling(rg, tmp_path):
    """torch.device("cuda"), "cuda:0", 0 and None name the same GPU: one device CSR, one compact copy (an index-less
    device compares unequal to a tensor's cuda:0, which used to replicate the CSR on every call)."""
    import torch
    gx, gy, gz, val, mask = _cloud(3, 2000)
    geom = rg.compute_grid_geometry(gx, gy, gz, (1, 8, 70), ((500.0, 500.0), (-3e3, 9e3), (-15e3, 15e3)), str(tmp_path),
                                    min_radius=1500.0, beam_factor=0.05)
    first = geom.device_csr(torch.device("cuda"))
    for spelling in (torch.device("cuda", 0), "cuda:0", 0, None, torch.device("cuda")):
        assert geom.device_csr(spelling) is first
    compact = geom.device_compact(torch.device("cuda"))
    assert compact is not None and geom.device_compact(torch.device("cuda", 0)) is compact


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 3, 255), (2, 1, 257), (1, 5, 512), (3, 7, 300)])
def test_compact_csr_edge_shapes(rg, tmp_path, shape):
    """Compact copy on grids whose lines are not a multiple of 64 rows and whose planes are not a multiple of 4 lines
    (short segments, chunks with fewer than 4 live wavefronts), with int32 and int64 row pointers, empty rows and a
    custom fill value.  The tile kernel (plain arrays, and the packed records with tile=384) is bit-identical to the
    standard kernel for 1-8 fused fields, both with the LDS window and on the per-pair fallback; the row-wise kernel over
    the packed records (the default for 1-4 fields) agrees to float32 rounding, and bit for bit with itself whatever the
    window."""
    import torch
    from radar_processor_amd import _native
    from radar_processor_amd.gridding import CsrGridder
    from radar_processor_amd.grid_geometry import DeviceCSR, GridGeometry
    gx, gy, gz, val, mask = _cloud(7, 6000)
    nz, ny, nx = shape
    limits = ((500.0, 500.0 if nz == 1 else 6000.0), (-3e3, -3e3 if ny == 1 else 9e3), (-15e3, -15e3 if nx == 1 else 15e3))
    geom = rg.compute_grid_geometry(gx, gy, gz, shape, limits, str(tmp_path), min_radius=1500.0, beam_factor=0.05)
    dev = torch.device("cuda")
    f = torch.from_numpy(val).to(dev)
    m = torch.from_numpy(mask.astype(np.uint8)).to(dev)
    for as_i64 in (False, True):
        csr = geom.device_csr(dev)
        if as_i64:
            g2 = GridGeometry.from_device(shape, limits, DeviceCSR(csr.indptr.to(torch.int64), csr.gate_indices, csr.weights,
                                                                   csr.max_gate), 17000.0)
        else:
            g2 = geom
        g_c = CsrGridder(g2, f.numel(), 1, device=dev, compact=True, packed=False)
        g_s = CsrGridder(g2, f.numel(), 1, device=dev)
        if g_s.csr.n_pairs == 0:
            assert g_c.compact is None
            continue
        assert g_c.compact is not None and not g_c.packed_stream
        g_c.pack([f], [m]); g_s.pack([f], [m])
        want = torch.empty((1, g_s.n_vox), dtype=torch.float32, device=dev)
        got = torch.empty_like(want)
        g_s.apply(want, fill_value=-1.0)
        g_c.apply(got, fill_value=-1.0)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        c, k = g_c.compact, g_c.csr
        assert torch.equal(c.decode(k), k.gate_indices)
        lib = _native.load_library()

        def compact_apply(gr, out, window, tile):
            _native.check(lib.rg_csr_compact_apply_f32(
                _native.ptr(k.indptr), int(k.is_i64), _native.ptr(c.local_idx), _native.ptr(k.weights),
                _native.ptr(c.dict_ptr), _native.ptr(c.dict), gr.n_vox, k.n_pairs, nx, ny, _native.ptr(gr.packed),
                gr.n_fields, gr.stride, gr.n_gates, -1.0, _native.ptr(out), window, tile, _native.stream_ptr()),
                "rg_csr_compact_apply_f32")

        compact_apply(g_c, got, 0, 0)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))     # per-pair fallback, same tile size
        compact_apply(g_c, got, c.window_cap, 256)
        # another tile size regroups the float32 partial sums: equal to rounding, not bit for bit -- but equal, bit for
        # bit, to the standard kernel run with the same tile
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=1e-5)
        g_t = CsrGridder(g2, f.numel(), 1, device=dev, tile=256)
        g_t.pack([f], [m])
        want_t = torch.empty_like(want)
        g_t.apply(want_t, fill_value=-1.0)
        assert torch.equal(got.view(torch.int32), want_t.view(torch.int32))
        # fused multi-field passes through the compact copy: every stride, windowed and per-pair paths, bit for bit
        extra = [torch.from_numpy(np.roll(val, 7 * (i + 1)) * np.float32(1.0 + i)).to(dev) for i in range(7)]
        emask = [torch.from_numpy(np.roll(mask, 13 * (i + 1)).astype(np.uint8)).to(dev) if i % 2 else None for i in range(7)]
        for nf in (2, 3, 4, 5, 8):
            fl, ml = [f] + extra[:nf - 1], [m] + emask[:nf - 1]
            gm_c = CsrGridder(g2, f.numel(), nf, device=dev, compact=True, packed=False)
            gm_s = CsrGridder(g2, f.numel(), nf, device=dev)
            assert gm_s.compact is None
            gm_c.compact, gm_c.window = c, c.window_for(nf)   # even if the policy would have preferred the standard kernel
            gm_c.pack(fl, ml); gm_s.pack(fl, ml)
            want_m = torch.empty((nf, gm_s.n_vox), dtype=torch.float32, device=dev)
            got_m = torch.full_like(want_m, 3.0)
            gm_s.apply(want_m, fill_value=-1.0)
            gm_c.apply(got_m, fill_value=-1.0)
            assert torch.equal(got_m.view(torch.int32), want_m.view(torch.int32)), nf
            got_m.fill_(3.0)
            compact_apply(gm_c, got_m, 0, 0)
            assert torch.equal(got_m.view(torch.int32), want_m.view(torch.int32)), nf
            if nf > 4:
                continue
            # the packed records: tile kernel (tile=384) bit for bit, row-wise kernel (default) to rounding -- and the
            # row-wise kernel bit for bit with itself on the per-pair path (window 0) and from run to run
            gm_r = CsrGridder(g2, f.numel(), nf, device=dev, compact=True)
            assert gm_r.compact is c and gm_r.packed_stream
            gm_r.packed = gm_s.packed
            gm_r.tile = 384
            got_m.fill_(3.0)
            gm_r.apply(got_m, fill_value=-1.0)
            assert torch.equal(got_m.view(torch.int32), want_m.view(torch.int32)), nf
            gm_r.tile = 0
            got_m.fill_(3.0)
            gm_r.apply(got_m, fill_value=-1.0)
            assert_same_to_rounding(got_m, want_m, scale=float(np.abs(val).max()) * nf, fill=-1.0)
            again = torch.full_like(got_m, 3.0)
            gm_r.window = 0
            gm_r.apply(again, fill_value=-1.0)
            assert torch.equal(again.view(torch.int32), got_m.view(torch.int32)), nf
        g_r = CsrGridder(g2, f.numel(), 1, device=dev, compact=True)      # one field, row-wise
        assert g_r.packed_stream
        g_r.pack([f], [m])
        g_r.apply(got, fill_value=-1.0)
        assert_same_to_rounding(got, want, scale=float(np.abs(val).max()), fill=-1.0)


def test_compact_csr_rich_chunks(rg):
    """Chunks far richer than the radar geometries produce: ~45 000 distinct gates per chunk of 4 x 64 rows force the LDS hash
    set of the builder through 8-16 rounds and every chunk of the kernel onto the per-pair fallback; a chunk with more
    than 65 536 distinct gates makes the geometry non-compactable.  Hand-made CSR, compared bit for bit with the
    standard kernel and decoded back to the original indices."""
    import torch
    from radar_processor_amd.gridding import CsrGridder
    from radar_processor_amd.grid_geometry import CompactCSR, DeviceCSR, GridGeometry
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(5)
    n_gates, n_rows = 300_000, 1000                      # 4 lines of 250 rows = one chunk group of 4 segments each
    lengths = torch.randint(150, 260, (n_rows,), device=dev, generator=gen)
    lengths[17] = 0
    lengths[300:310] = 0
    indptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(lengths, 0)
    n_pairs = int(indptr[-1])
    gidx = torch.randint(0, n_gates, (n_pairs,), device=dev, generator=gen, dtype=torch.int32)
    gidx[:5000] = gidx[0]                                # plus one heavily repeated gate
    wts = torch.rand(n_pairs, device=dev, generator=gen) + 0.01
    shape, limits = (1, 4, n_rows // 4), ((0.0, 0.0), (0.0, 1.0), (0.0, 1.0))
    csr = DeviceCSR(indptr.to(torch.int32), gidx, wts, int(gidx.max()))
    geom = GridGeometry.from_device(shape, limits, csr, 17000.0)
    compact = geom.device_compact(dev)
    assert compact is not None and compact.max_dict > 40000 and compact.window_cap == 8192
    assert torch.equal(compact.decode(csr), gidx)
    values = torch.randn(n_gates, device=dev, generator=gen)
    mask = (torch.rand(n_gates, device=dev, generator=gen) < 0.2).to(torch.uint8)
    g_c = CsrGridder(geom, n_gates, 1, device=dev)
    g_c.compact, g_c.window = compact, 8192              # force the copy although hardly any chunk fits its window
    g_s = CsrGridder(geom, n_gates, 1, device=dev)
    g_c.pack([values], [mask]); g_s.pack([values], [mask])
    want = torch.empty((1, n_rows), dtype=torch.float32, device=dev)
    got = torch.empty_like(want)
    g_s.apply(want); g_c.apply(got)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    # more than 65536 distinct gates in one chunk (76 800 here) but fewer in each of its four segments: the chunk is
    # stored SPLIT -- one dictionary per wavefront behind a header -- and still decodes and grids bit for bit
    rich = torch.arange(256 * 300, device=dev, dtype=torch.int32)
    rich = rich[torch.randperm(rich.numel(), device=dev, generator=gen)].contiguous()
    ip2 = torch.arange(0, 256 * 300 + 1, 300, device=dev, dtype=torch.int64)
    csr2 = DeviceCSR(ip2.to(torch.int32), rich, torch.rand(rich.numel(), device=dev, generator=gen) + 0.01, int(rich.max()))
    split = CompactCSR.build(csr2, (1, 4, 64))
    assert split is not None and split.max_dict == 4 + 76800 and split.dict_ptr.tolist() == [0, 4 + 76800]
    assert split.dict[:4].tolist() == [4, 4 + 19200, 4 + 38400, 4 + 57600]
    assert torch.equal(split.decode(csr2), rich)
    assert torch.equal(split.decode(csr2, 70, 200), rich[70 * 300:200 * 300])
    geom2 = GridGeometry.from_device((1, 4, 64), limits, csr2, 17000.0, compact=split)
    vals2 = torch.randn(76800, device=dev, generator=gen)
    mask2 = (torch.rand(76800, device=dev, generator=gen) < 0.3).to(torch.uint8)
    for nf in (1, 3):
        g_c2 = CsrGridder(geom2, 76800, nf, device=dev)
        g_c2.compact, g_c2.window = split, split.window_for(nf)
        g_s2 = CsrGridder(geom2, 76800, nf, device=dev)
        assert g_s2.compact is None
        fl = [vals2, vals2 * 2.0, -vals2][:nf]
        g_c2.pack(fl, [mask2] * nf); g_s2.pack(fl, [mask2] * nf)
        want2 = torch.empty((nf, 256), dtype=torch.float32, device=dev)
        got2 = torch.full_like(want2, 5.0)
        g_s2.apply(want2); g_c2.apply(got2)
        assert torch.equal(got2.view(torch.int32), want2.view(torch.int32)), nf
        # the packed records of a split chunk: tile kernel bit for bit, row-wise kernel to rounding
        assert split.ensure_packed(csr2)
        g_c2.packed_stream = True
        for tile, exact in ((384, True), (0, False)):
            g_c2.tile = tile
            got2.fill_(5.0)
            g_c2.apply(got2)
            if exact:
                assert torch.equal(got2.view(torch.int32), want2.view(torch.int32)), nf
            else:
                assert_same_to_rounding(got2, want2, scale=2.0 * float(vals2.abs().max()))
    four = CompactCSR.build(csr2, (4, 1, 64))                   # the same rows as four one-line chunks: 19 200 gates each
    assert four is not None and four.max_dict == 19200 and torch.equal(four.decode(csr2), rich)
    # a single 64-row segment with more than 65536 distinct gates: not compactable, the standard kernel stays in charge
    big = torch.arange(64 * 1100, device=dev, dtype=torch.int32)
    ip3 = torch.arange(0, 64 * 1100 + 1, 1100, device=dev, dtype=torch.int64)
    csr3 = DeviceCSR(ip3.to(torch.int32), big, torch.ones(big.numel(), device=dev), int(big.max()))
    assert CompactCSR.build(csr3, (1, 1, 64)) is None


@pytest.mark.parametrize("seed", range(16))
def test_compact_and_packed_kernels_fuzz(rg, seed):
    """Random hand-made CSRs -- random grid shapes (lines that are not multiples of 64 rows, planes that are not
    multiples of 4 lines), empty rows, rows longer than a tile, few or many distinct gates per chunk, int32 / int64 row
    pointers, 1-4 fused fields -- through the plain compact kernel, the tile kernel over the packed stream and the
    packed-only decode, all bit for bit against the standard kernel / the original arrays; and through the row-wise kernel
    over the packed stream (the default), equal to float32 rounding and bit for bit with itself on the per-pair path."""
    import torch
    from radar_processor_amd.gridding import CsrGridder
    from radar_processor_amd.grid_geometry import CompactCSR, DeviceCSR, GridGeometry
    dev = torch.device("cuda")
    rng = np.random.default_rng(1000 + seed)
    nz, ny, nx = int(rng.integers(1, 4)), int(rng.integers(1, 11)), int(rng.integers(1, 400))
    n_vox = nz * ny * nx
    n_gates = int(rng.integers(50, 200_000))
    lengths = rng.integers(0, int(rng.choice([3, 40, 130])), size=n_vox)
    lengths[rng.random(n_vox) < rng.choice([0.0, 0.3, 0.9])] = 0
    for r in rng.integers(0, n_vox, size=3):
        lengths[r] = int(rng.integers(400, 1500))                      # rows longer than a tile
    indptr = np.zeros(n_vox + 1, dtype=np.int64)
    np.cumsum(lengths, out=indptr[1:])
    n_pairs = int(indptr[-1])
    if n_pairs == 0:
        pytest.skip("empty case")
    spread = int(rng.choice([30, 2000, n_gates]))                      # gate locality: tiny or huge dictionaries
    base = rng.integers(0, n_gates, size=n_vox)
    row_of_pair = np.repeat(np.arange(n_vox), lengths)
    gidx = ((base[row_of_pair] + rng.integers(0, spread, size=n_pairs)) % n_gates).astype(np.int32)
    wts = np.exp(-4.0 * rng.random(n_pairs)).astype(np.float32) + np.float32(1e-5)      # Barnes range: codable
    as_i64 = bool(seed % 2)
    ip_t = torch.from_numpy(indptr if as_i64 else indptr.astype(np.int32)).to(dev)
    csr = DeviceCSR(ip_t, torch.from_numpy(gidx).to(dev), torch.from_numpy(wts).to(dev), int(gidx.max()))
    shape, limits = (nz, ny, nx), ((0.0, 1.0), (0.0, 1.0), (0.0, 1.0))
    geom = GridGeometry.from_device(shape, limits, csr, 17000.0)
    compact = geom.device_compact(dev)
    assert compact is not None and torch.equal(compact.decode(csr), csr.gate_indices)
    assert compact.ensure_packed(csr)
    # the records decode to the positions and the weights they were packed from
    pos, w_back = compact._record_fields(csr, 0, n_vox)
    assert torch.equal(pos, compact.local_idx.to(torch.int64) & 0xFFFF)
    assert torch.equal(w_back.view(torch.int32), csr.weights.view(torch.int32))
    fields = [torch.from_numpy(rng.normal(10, 20, n_gates).astype(np.float32)).to(dev) for _ in range(8)]
    masks = [torch.from_numpy((rng.random(n_gates) < 0.2).astype(np.uint8)).to(dev) if k % 2 == 0 else None for k in range(8)]
    fields[6][4::9] = float("nan")
    fields[5][::23] = float("inf")
    fields[1][::7] = float("nan")                                      # unmasked NaN propagates like in NumPy
    fields[1][3::11] = float("inf")                                    # ... and so do unmasked infinities (Inf - Inf = NaN)
    fields[1][5::13] = float("-inf")
    fields[1][1::17] = 1e-40                                           # a denormal value: the product is not flushed
    fields[0][2::19] = -0.0
    for nf in (1, 2, 3, 4, 5, 6, 7, 8):      # 5-8 fields: the row-wise kernel only (the tile kernel over the records takes 1-4)
        g_s = CsrGridder(geom, n_gates, nf, device=dev)
        g_p = CsrGridder(geom, n_gates, nf, device=dev)
        g_p.compact, g_p.window, g_p.packed_stream, g_p.tile = compact, compact.window_for(nf), True, 384
        g_r = CsrGridder(geom, n_gates, nf, device=dev)
        g_r.compact, g_r.window, g_r.packed_stream = compact, compact.window_for(nf, rowwise=True), True
        g_c = CsrGridder(geom, n_gates, nf, device=dev)
        g_c.compact, g_c.window, g_c.packed_stream = compact, compact.window_for(nf), False
        want = torch.empty((nf, n_vox), dtype=torch.float32, device=dev)
        for gr in (g_s, g_p, g_c, g_r):
            gr.pack(fields[:nf], masks[:nf])
        g_s.apply(want, fill_value=-3.0)
        for name, gr in (("packed", g_p), ("compact", g_c)):
            if nf > 4 and name == "packed":
                with pytest.raises(rg.NativeError, match="tile kernel"):
                    gr.apply(torch.full_like(want, 9.0), fill_value=-3.0)
                continue
            got = torch.full_like(want, 9.0)
            gr.apply(got, fill_value=-3.0)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (name, nf, shape)
        # window too small for most chunks: the per-pair path of the packed kernel
        got = torch.full_like(want, 9.0)
        if nf <= 4:
            g_p.window = 0
            g_p.apply(got, fill_value=-3.0)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), ("packed, no window", nf, shape)
        # the row-wise kernel: another order of the float32 adds
        row = torch.full_like(want, 9.0)
        g_r.apply(row, fill_value=-3.0)
        assert_same_to_rounding(row, want, scale=100.0, fill=-3.0)
        g_r.window = 0
        got.fill_(9.0)
        g_r.apply(got, fill_value=-3.0)
        assert torch.equal(got.view(torch.int32), row.view(torch.int32)), ("row-wise, no window", nf, shape)
        # ... and it is exactly the order the kernel documents: bit for bit against the oracle's restatement of that order,
        # for the shipped lane split and for diagnostic ones (1 lane per row ... 64 lanes per row, other targets)
        f_np = [t.cpu().numpy() for t in fields[:nf]]
        m_np = [None if t is None else t.cpu().numpy().astype(bool) for t in masks[:nf]]
        g_r.window = compact.window_for(nf, rowwise=True)
        for hint in ((0, 1, 8, 64, 71, 99) if nf in (1, 3, 8) and seed < 4 else (0,)):
            g_r.tile = 2000 + hint if hint else 0
            got.fill_(9.0)
            g_r.apply(got, fill_value=-3.0)
            emu = oracle.csr_apply_rowwise_order(indptr, gidx, wts, f_np, m_np, shape, fill_value=-3.0,
                                                 lanes_hint=hint).reshape(nf, n_vox)
            got_np = got.cpu().numpy()
            np.testing.assert_array_equal(np.isnan(got_np), np.isnan(emu))
            live = ~np.isnan(emu)
            assert np.array_equal(got_np.view(np.int32)[live], emu.view(np.int32)[live]), ("row-wise order", nf, hint, shape)
        g_r.tile = 0
        rows_by_order = row
        # the OTHER record order (line-major segments instead of dispatch order): other slots, the same records, the same bits
        # from both kernels, and the same positions and weights decoded
        if nf in (1, 3):
            from radar_processor_amd import _native, grid_geometry
            other = CompactCSR(compact.local_idx, compact.dict_ptr, compact.dict, compact.max_dict, compact.window_cap,
                               compact.grid_shape, compact.chunk_pairs, compact.chunk_counts)
            assert compact.rec_order == grid_geometry.DEFAULT_REC_ORDER == _native.RG_REC_ORDER_DISPATCH
            other.rec_order = _native.RG_REC_ORDER_SEGMENT
            assert other.ensure_packed(csr) and other.rec.shape == compact.rec.shape
            assert other.rec_ptr.numel() == nz * ny * ((nx + 63) // 64) + 1
            pos_o, w_o = other._record_fields(csr, 0, n_vox)
            assert torch.equal(pos_o, pos) and torch.equal(w_o.view(torch.int32), w_back.view(torch.int32))
            g_o = CsrGridder(geom, n_gates, nf, device=dev)
            g_o.compact, g_o.window, g_o.packed_stream, g_o.packed = other, other.window_for(nf), True, g_r.packed
            got.fill_(9.0)
            g_o.apply(got, fill_value=-3.0)
            assert torch.equal(got.view(torch.int32), rows_by_order.view(torch.int32)), ("segment order, row-wise", nf, shape)
            g_o.tile = 384
            got.fill_(9.0)
            g_o.apply(got, fill_value=-3.0)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), ("segment order, tile kernel", nf, shape)
    # and against the float64 oracle (tolerance: float32 accumulation)
    data = fields[0].cpu().numpy()
    want64 = oracle.csr_apply_f64(indptr, gidx, wts, data, masks[0].cpu().numpy().astype(bool), shape, fill_value=-3.0)
    g1 = CsrGridder(geom, n_gates, 1, device=dev, compact=True)
    g1.pack(fields[:1], masks[:1])
    out1 = torch.empty((1, n_vox), dtype=torch.float32, device=dev)
    g1.apply(out1, fill_value=-3.0)
    np.testing.assert_allclose(out1.cpu().numpy().reshape(shape), want64, rtol=2e-6, atol=2e-6 * 100.0)


def test_nonfinite_and_denormal_values_bit_for_bit_across_kernels(rg):
    """Unmasked +/-Inf, NaN, -0.0 and denormal gate values (interpolate.py:78-82 lets them all through: only MASKED gates
    are dropped) on rows of one or two pairs -- where the order of the float32 adds cannot matter -- so every kernel
    must return the same BITS as NumPy's arithmetic: K1, the tile kernel over the packed records and the row-wise kernel
    (whose masked product is v_mul_legacy_f32 with one select: 0 * x = +0 only for an excluded gate, the IEEE product
    otherwise -- the packed weights are never 0: a zero weight is not codable and falls back to the plain arrays)."""
    import torch
    from radar_processor_amd.gridding import CsrGridder
    from radar_processor_amd.grid_geometry import DeviceCSR, GridGeometry
    dev = torch.device("cuda")
    inf, nan = np.float32("inf"), np.float32("nan")
    den = np.float32(1e-40)                                            # subnormal in float32
    assert 0 < den < np.finfo(np.float32).tiny
    #            0     1     2      3     4     5     6     7          8
    vals = np.array([inf, -inf, nan, -0.0, den, 5.0, -den, 3.0e38, 2.5], dtype=np.float32)
    mask = np.zeros(9, dtype=bool)
    mask[8] = True                                                     # an excluded gate next to the special ones
    rows = [[0], [1], [0, 1], [2, 5], [3], [3, 3], [4], [4, 6], [4, 4], [7, 7], [0, 8], [8], [5, 8], [4, 8], [6], []]
    lengths = np.array([len(r) for r in rows])
    indptr = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(lengths, out=indptr[1:])
    gidx = np.array([g for r in rows for g in r], dtype=np.int32)
    wts = np.array([1.0, 0.5, 0.25, 0.75, 1.0 + 1e-5, 0.0183, 0.3][: 7] * 4, dtype=np.float32)[: gidx.size]
    shape = (1, 1, len(rows))
    want = oracle.csr_apply(indptr, gidx, wts, vals, mask, shape, fill_value=-9.0).reshape(-1)
    with np.errstate(all="ignore"):
        assert np.isposinf(want[0]) and np.isneginf(want[1]) and np.isnan(want[2]) and np.isnan(want[3])
        assert want[4] == 0 and np.signbit(want[4]) and 0 < want[6] < np.finfo(np.float32).tiny
        assert want[11] == -9.0 and want[15] == -9.0
    csr = DeviceCSR(torch.from_numpy(indptr.astype(np.int32)).to(dev), torch.from_numpy(gidx).to(dev),
                    torch.from_numpy(wts).to(dev), int(gidx.max()))
    geom = GridGeometry.from_device(shape, ((0.0, 1.0),) * 3, csr, 17000.0)
    compact = geom.device_compact(dev)
    assert compact is not None and compact.ensure_packed(csr)
    f_t, m_t = torch.from_numpy(vals).to(dev), torch.from_numpy(mask.astype(np.uint8)).to(dev)
    got = {}
    for name, tile, use_compact in (("K1", 0, False), ("tile", 384, True), ("rowwise", 0, True)):
        g = CsrGridder(geom, vals.size, 1, device=dev)
        if use_compact:
            g.compact, g.window, g.packed_stream, g.tile = compact, compact.window_for(1), True, tile
        g.pack([f_t], [m_t])
        out = torch.full((1, len(rows)), 7.0, dtype=torch.float32, device=dev)
        g.apply(out, fill_value=-9.0)
        got[name] = out.cpu().numpy().reshape(-1)
    # One documented bit-level deviation (found by this test in round 4): the sign of an exactly-zero mean.  NumPy's
    # reduceat starts a row's sum from its first product, so a row whose unmasked products are all -0.0 sums to -0.0; the
    # kernels start every running sum from +0.0, and +0.0 + -0.0 = +0.0.  The two are equal under ==, i.e. inside any
    # tolerance; initialising with -0.0 (the true identity of IEEE addition) would not make it exact either, because
    # masked pairs and lanes without a pair add +0.0 terms NumPy does not have.  Everything else -- infinities, NaN,
    # denormal products, the fill value -- is the same bits.
    zero = want == 0
    assert zero.sum() == 2 and np.signbit(want[zero]).all()
    for name, arr in got.items():
        np.testing.assert_array_equal(np.isnan(arr), np.isnan(want), err_msg=name)
        live = ~np.isnan(want) & ~zero
        assert np.array_equal(arr.view(np.int32)[live], want.view(np.int32)[live]), (name, arr, want)
        assert (arr[zero] == 0).all() and not np.signbit(arr[zero]).any(), (name, arr[zero])


def test_product_library_refuses_timing_only_and_experiment_tile_codes(rg):
    """The shipped library computes right answers or refuses: the timing-only variants of the row-wise kernel (tile = 2100 +
    bits: no store, no record loads, ...), several chunks per workgroup (2201 .. 2264), the ablations of the tile kernel
    (901 .. 909, block-rotation overrides) and K1's tuning variants exist only in -DRG_EXPERIMENTS builds
    (tools/build_experiments.py)."""
    import torch
    from radar_processor_amd import _native
    lib = rg.load_library()
    dev = torch.device("cuda")
    ip = torch.zeros(65, dtype=torch.int32, device=dev)
    buf = torch.zeros(64, dtype=torch.float32, device=dev)
    i64 = torch.zeros(8, dtype=torch.int64, device=dev)
    rec = torch.zeros((4, 4), dtype=torch.int32, device=dev)
    P = _native.ptr
    for tile in (2100, 2102, 2116, 2199, 2201, 2204, 2264, 2265, 1999, 385):
        st = lib.rg_csr_compact_apply_packed_f32(P(ip), 0, P(rec), P(i64), _native.RG_REC_ORDER_DISPATCH, 120 << 23, P(i64),
                                                 P(ip), 64, 0, 64, 1, P(buf), 1, 1, 64, 0.0, P(buf), 256, tile, 0)
        assert st == _native.RG_EINVAL, (tile, st)
    for tile in (901, 903, 909, 5384, 1000):
        st = lib.rg_csr_compact_apply_f32(P(ip), 0, P(ip), P(buf), P(i64), P(ip), 64, 0, 64, 1, P(buf), 1, 1, 64, 0.0, P(buf),
                                          256, tile, 0)
        assert st == _native.RG_EINVAL, (tile, st)
    for variant in (8, 9, 16, 19, 22, 28, 640):
        st = lib.rg_csr_apply_f32_ex(P(ip), 0, P(ip), P(buf), 64, 0, 64, P(buf), 1, 1, 64, 0.0, P(buf), variant, 0)
        assert st == _native.RG_EINVAL, (variant, st)
    torch.cuda.synchronize()


def test_device_layout_sidecar_round_trip(rg, tmp_path):
    """The compact copy (dictionaries + packed records) saved next to the reference .npz and attached to a freshly loaded
    geometry instead of being derived again: the same bits out of the row-wise and the tile kernel; a sidecar derived from
    another geometry is refused (geometry.py:94-150 stays the interchange format; the sidecar is this build's own)."""
    import torch
    from radar_processor_amd.gridding import CsrGridder
    gx, gy, gz, val, mask = _cloud(3, 6000)
    shape, limits = (3, 9, 130), ((0.0, 8e3), (-20e3, 20e3), (-20e3, 20e3))
    geom = rg.compute_grid_geometry(gx, gy, gz, shape, limits, str(tmp_path), min_radius=900.0, beam_factor=0.05)
    dev = torch.device("cuda")
    npz, side = str(tmp_path / "g.npz"), str(tmp_path / "g.layout.npz")
    rg.save_geometry(geom, npz)
    assert rg.save_device_layout(geom, side)
    f_t, m_t = torch.from_numpy(val).to(dev), torch.from_numpy(mask.astype(np.uint8)).to(dev)

    def grids(g, attach):
        if attach is not None:
            assert rg.load_device_layout(g, attach) is True
        compact = g.device_compact(dev)
        assert compact is not None and compact.ensure_packed(g.device_csr(dev))
        gr = CsrGridder(g, val.size, 1, device=dev)
        gr.compact, gr.window, gr.packed_stream = compact, compact.window_for(1), True
        gr.pack([f_t], [m_t])
        out = []
        for tile in (0, 384):
            gr.tile = tile
            o = torch.empty((1, gr.n_vox), dtype=torch.float32, device=dev)
            gr.apply(o)
            out.append(o)
        return out, compact
    want, c0 = grids(geom, None)
    back = rg.load_geometry(npz)
    got, c1 = grids(back, side)
    assert c1 is not c0 and torch.equal(c1.rec, c0.rec) and torch.equal(c1.dict, c0.dict) and c1.w_base == c0.w_base
    for a, b in zip(got, want):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    other = rg.compute_grid_geometry(gx, gy, gz, shape, limits, str(tmp_path), min_radius=700.0, beam_factor=0.05)
    assert rg.load_device_layout(other, side) is False                      # another CSR: key mismatch
    assert rg.load_device_layout(back, str(tmp_path / "missing.npz")) is False


@pytest.mark.parametrize("weighting", ["barnes2", "nearest"])
def test_per_level_gate_lists_give_the_same_geometry(rg, weighting):
    """RoiSearch keeps one cell-sorted gate list per grid level (a gate under the levels within the largest radius of influence
    any neighbouring voxel can have) -- a candidate filter, nothing else: against the single list for all levels the builder returns
    the same rows (same gates, same float32 weights) for every voxel, whatever the cell size, a non-zero radar altitude and
    a min_radius that dominates included; the CSR-free gridder agrees to float32 rounding; a slab of levels addresses its lists
    through cells_from(level0); beam factors the bound does not cover fall back to the single list."""
    import torch
    from radar_processor_amd import synthetic
    from radar_processor_amd.roi_grid import roi_grid_fields_device
    dev = torch.device("cuda", 0)
    vol = synthetic.make_volume(n_elev=7, n_az=120, n_gates=240, seed=5, fields=("DBZH",))
    shape, limits = (9, 37, 41), ((0.0, 12000.0), (-70e3, 70e3), (-60e3, 80e3))
    f = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
    m = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
    for kw in (dict(), dict(radar_altitude=350.0, min_radius=2500.0), dict(beam_factor=0.05, toa=9000.0)):
        ref = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, device=dev, per_level=False, **kw)
        assert not ref.per_level
        want = ref.build_csr(weighting)
        w_ip, w_idx, w_w = oracle.canonical_rows(want.indptr.cpu().numpy(), want.gate_indices.cpu().numpy(),
                                                 want.weights.cpu().numpy())
        grid_ref = roi_grid_fields_device(ref, [f], [m], weighting=weighting)
        for cell in (None, 900.0, 7000.0):
            s = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, device=dev, cell_size=cell, **kw)
            assert s.per_level and s.cells.levels == shape[0] and s.n_binned > ref.n_binned
            assert s.count_pairs() == want.n_pairs
            got = s.build_csr(weighting)
            g_ip, g_idx, g_w = oracle.canonical_rows(got.indptr.cpu().numpy(), got.gate_indices.cpu().numpy(),
                                                     got.weights.cpu().numpy())
            np.testing.assert_array_equal(g_ip, w_ip)
            np.testing.assert_array_equal(g_idx, w_idx)
            np.testing.assert_array_equal(g_w.view(np.int32), w_w.view(np.int32))
            grid = roi_grid_fields_device(s, [f], [m], weighting=weighting)
            assert_same_to_rounding(grid, grid_ref, scale=80.0)
            assert s.cells_from(0) is s.cells and s.cells_from(4).level0 == 4 and s.cells_from(4).levels == shape[0]
    wide = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, device=dev, beam_factor=0.7)
    assert not wide.per_level and wide.cells.levels == 0
    with pytest.raises(rg.NativeError, match="beam_factor"):
        rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, device=dev, beam_factor=0.7, per_level=True)
