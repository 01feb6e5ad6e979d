"""2-D raster stage on the GPU against the CPU oracle: processor-style collapse
('ppi' / 'cappi' / 'colmax' + re-mask), the filter masks of phases 10-11, and colormap -> RGBA.  Everything here is
integer / selection work or IEEE arithmetic in a fixed order, so the bar is bit-exact.

The reference side of these functions cannot be imported in the build container; the oracle is a restatement pinned
by the reference's own test expectations (tests/test_oracle_golden.py) and, for the colormap, it calls matplotlib --
the same third-party code the reference calls."""
import numpy as np
import pytest

from oracle import radar_grid_oracle as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rg():
    import radar_processor_amd as pkg
    pkg.load_library()
    return pkg


class _Filter:
    def __init__(self, field, lo, hi):
        self.field, self.min, self.max = field, lo, hi


class _Grid:
    """Duck-typed pyart.core.Grid: fields / x / y / z dictionaries."""

    def __init__(self, data3d, x, y, z, field="DBZH"):
        self.fields = {field: {"data": data3d, "units": "dBZ"}}
        self.x, self.y, self.z = {"data": x}, {"data": y}, {"data": z}


def _masked_grid(seed, shape, frac_masked=0.3):
    rng = np.random.default_rng(seed)
    data = rng.normal(10, 25, shape).astype(np.float32)
    return np.ma.array(data, mask=rng.random(shape) < frac_masked)


def _axes(shape, extent=120e3, top=15e3):
    nz, ny, nx = shape
    return np.linspace(-extent, extent, nx), np.linspace(-extent, extent, ny), np.linspace(0.0, top, nz)


# ------------------------------------------------------------------------------------------------
# collapse
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(10, 20, 30), (31, 257, 129), (1, 5, 7), (16, 1, 1000)])
@pytest.mark.parametrize("elev", [0.5, 3.2, 19.5, -0.3])
def test_ppi_levels_and_values_bit_exact(rg, shape, elev):
    import torch
    data = _masked_grid(5, shape)
    x, y, z = _axes(shape)
    want_level = oracle.ppi_levels(x, y, z, elev)
    want = oracle.collapse_3d_to_2d(data, "ppi", x, y, z, elevation_deg=elev)
    dev = torch.device("cuda")
    grid_t = torch.from_numpy(data.filled(np.nan)).to(dev)
    plane_t, level_t = rg.collapse_plane_device(grid_t, "ppi", x_coords=x, y_coords=y, z_levels=z, elevation_deg=elev,
                                                return_level=True)
    np.testing.assert_array_equal(level_t.cpu().numpy(), want_level)
    got = rg.collapse_field_3d_to_2d(data, "ppi", x_coords=x, y_coords=y, z_levels=z, elevation_deg=elev)
    assert isinstance(got, np.ma.MaskedArray) and got.dtype == np.float32
    np.testing.assert_array_equal(np.ma.getmaskarray(got), np.ma.getmaskarray(want))
    np.testing.assert_array_equal(got.filled(-9999.0), want.filled(-9999.0))
    np.testing.assert_array_equal(np.isnan(plane_t.cpu().numpy()), np.ma.getmaskarray(want))


def test_ppi_uneven_levels_and_ties(rg):
    # z levels that are not evenly spaced, and a target exactly half way between two levels (first one wins)
    shape = (4, 3, 3)
    data = np.arange(36, dtype=np.float32).reshape(shape)
    z = np.array([0.0, 100.0, 200.0, 1000.0])
    x = np.array([-1000.0, 0.0, 1000.0])
    y = np.array([-1000.0, 0.0, 1000.0])
    elev = float(np.rad2deg(np.arcsin(0.05)))   # 1000 m ground range -> ~50 m: the tie between levels 0 and 1
    want = oracle.collapse_3d_to_2d(data, "ppi", x, y, z, elevation_deg=elev)
    got = rg.collapse_field_3d_to_2d(data, "ppi", x_coords=x, y_coords=y, z_levels=z, elevation_deg=elev)
    np.testing.assert_array_equal(np.asarray(got), np.asarray(want))


@pytest.mark.parametrize("shape", [(10, 20, 30), (40, 130, 67)])
def test_cappi_and_colmax_bit_exact(rg, shape):
    data = _masked_grid(9, shape)
    x, y, z = _axes(shape)
    for height in (0.0, 3100.0, 5000.0, 99000.0):
        want = oracle.collapse_3d_to_2d(data, "cappi", z_levels=z, target_height_m=height)
        got = rg.collapse_field_3d_to_2d(data, "cappi", z_levels=z, target_height_m=height)
        np.testing.assert_array_equal(np.ma.getmaskarray(got), np.ma.getmaskarray(want))
        np.testing.assert_array_equal(got.filled(-1.0), want.filled(-1.0))
    want = oracle.collapse_3d_to_2d(data, "colmax")
    got = rg.collapse_field_3d_to_2d(data, "colmax")
    np.testing.assert_array_equal(np.ma.getmaskarray(got), np.ma.getmaskarray(want))
    np.testing.assert_array_equal(got.filled(-1.0), want.filled(-1.0))
    # the reference's own expectation on plain float64 input (tests/test_utils.py:209-238)
    plain = np.random.default_rng(0).random(shape)
    np.testing.assert_array_equal(rg.collapse_field_3d_to_2d(plain, "colmax").data, plain.max(axis=0).astype(np.float32))
    with pytest.raises(ValueError):
        rg.collapse_field_3d_to_2d(plain, "rhi")


@pytest.mark.parametrize("field,vmin", [("DBZH", -30.0), ("composite_reflectivity", 5.0), ("ZDR", -2.0), ("KDP", 0.0),
                                        ("RHOHV", -30.0)])
@pytest.mark.parametrize("product", ["ppi", "cappi", "colmax"])
def test_collapse_grid_to_2d_matches_reference_semantics(rg, field, vmin, product):
    shape = (12, 50, 70)
    data = _masked_grid(11, shape)
    raw = np.ma.getdata(data)
    raw[3, 4, 5] = np.inf                      # masked_invalid must catch +-inf as well as NaN
    raw[:, 7, 8] = vmin                        # exactly on the threshold: <= vs < matters
    data.mask[:, 7, 8] = False
    data.mask[3, 4, 5] = False
    x, y, z = _axes(shape)
    grid = _Grid(data.copy(), x, y, z, field)
    rg.collapse_grid_to_2d(grid, field, product, elevation_deg=1.3, target_height_m=3000.0, vmin=vmin)
    got = grid.fields[field]["data"]
    plane = oracle.collapse_3d_to_2d(data, product, x, y, z, elevation_deg=1.3, target_height_m=3000.0)
    want = oracle.collapse_remask(plane, field, vmin)
    assert got.shape == (1, 50, 70)
    assert grid.fields[field]["_FillValue"] == -9999.0
    np.testing.assert_array_equal(grid.z["data"], [0.0])
    np.testing.assert_array_equal(np.ma.getmaskarray(got[0]), np.ma.getmaskarray(want))
    np.testing.assert_array_equal(got[0].filled(-9999.0), want.filled(-9999.0))


# ------------------------------------------------------------------------------------------------
# filter masks
# ------------------------------------------------------------------------------------------------
def _planes(seed, shape=(173, 211)):
    rng = np.random.default_rng(seed)
    main = np.ma.array(rng.normal(15, 20, shape).astype(np.float32), mask=rng.random(shape) < 0.2)
    rho = np.ma.array(rng.uniform(0.3, 1.0, shape).astype(np.float32), mask=rng.random(shape) < 0.1)
    zdr = rng.normal(0.5, 2.0, shape).astype(np.float32)
    zdr[rng.random(shape) < 0.05] = np.nan
    return main, {"RHOHV": rho, "ZDR": zdr, "MISSING": None}


@pytest.mark.parametrize("case", range(7))
def test_apply_filter_masks_matches_oracle(rg, case):
    main, qc = _planes(case)
    cases = [
        ([_Filter("DBZH", -20, None)], [], "DBZH"),
        ([_Filter("dbzh", None, 50), _Filter("DBZH", 0.0, 40.0)], [], "DBZH"),
        ([_Filter("RHOHV", 0.8, None), _Filter("ZDR", -1.0, 3.0)], [], "DBZH"),                 # cross-field visual
        ([], [_Filter("RHOHV", 0.85, None), _Filter("ZDR", None, 2.5), _Filter("NOPE", 0, 1)], "DBZH"),
        ([_Filter("RHOHV", 0.3, 0.99)], [_Filter("MISSING", 0, 1)], "RHOHV"),                   # processor.py:849
        ([_Filter("", 0, 1), _Filter(None, 0, 1), _Filter("DBZH", None, None)], [_Filter("ZDR", 0.0, None)], "DBZH"),
        ([_Filter("DBZH", float(k), None) for k in range(-20, -6)], [_Filter("RHOHV", 0.5, None)], "DBZH"),  # > 12 tests
    ]
    visual, qcf, field = cases[case]
    src = qc["RHOHV"] if field == "RHOHV" else main
    want = oracle.filter_masks(src.copy(), visual, qcf, field, qc)
    got = rg.apply_filter_masks(src.copy(), visual, qcf, field, {"qc": qc})
    np.testing.assert_array_equal(np.ma.getmaskarray(got), np.ma.getmaskarray(want))
    np.testing.assert_array_equal(np.ma.getdata(got), np.ma.getdata(src))       # values untouched
    assert got is not src


def test_apply_filter_masks_reference_expectations(rg):
    # tests/test_processor_phases.py:265-357 through the HIP path
    plane = np.ma.array(np.linspace(-30, 60, 1000).reshape(50, 20), mask=np.zeros((50, 20), dtype=bool))
    f32 = plane.data.astype(np.float32)
    out = rg.apply_filter_masks(plane.copy(), [_Filter("DBZH", -20, None)], [], "DBZH", {"qc": {}})
    np.testing.assert_array_equal(out.mask, f32 < np.float32(-20))
    out = rg.apply_filter_masks(plane.copy(), [_Filter("DBZH", -20, 50)], [], "DBZH", {"qc": {}})
    np.testing.assert_array_equal(out.data, plane.data)
    rho = np.ma.array(np.linspace(0.5, 1.0, 1000).reshape(50, 20), mask=np.zeros((50, 20), dtype=bool))
    out = rg.apply_filter_masks(plane.copy(), [], [_Filter("RHOHV", 0.8, None)], "DBZH", {"qc": {"RHOHV": rho}})
    np.testing.assert_array_equal(out.mask, rho.data.astype(np.float32) < np.float32(0.8))
    same = rg.apply_filter_masks(plane, [], [], "DBZH", {"qc": {}})
    assert same is plane                                                        # processor.py:826-827


def test_plane_filter_device_contract(rg):
    import torch
    dev = torch.device("cuda")
    src = torch.tensor([1.0, float("nan"), 5.0, -3.0, float("inf")], device=dev)
    other = torch.tensor([0.0, 0.0, 9.0, 0.0, 0.0], device=dev)
    vals, mask = rg.plane_filter_device(src, [rg.PlaneTest(lo=0.0), rg.PlaneTest(plane=other, hi=8.0)], want_mask=True)
    np.testing.assert_array_equal(mask.cpu().numpy(), [0, 1, 1, 1, 0])
    np.testing.assert_array_equal(np.isnan(vals.cpu().numpy()), [False, True, True, True, False])
    _, mask = rg.plane_filter_device(src, [rg.PlaneTest(nonfinite=True)], want_values=False, want_mask=True)
    np.testing.assert_array_equal(mask.cpu().numpy(), [0, 1, 0, 0, 1])
    explicit = torch.tensor([1, 0, 0, 0, 0], dtype=torch.uint8, device=dev)
    _, mask = rg.plane_filter_device(src, [], src_mask=explicit, want_values=False, want_mask=True)
    np.testing.assert_array_equal(mask.cpu().numpy(), [1, 0, 0, 0, 0])          # explicit mask: NaN is not masked
    with pytest.raises(ValueError):
        rg.plane_filter_device(src, [rg.PlaneTest(lo=0.0)] * 13)


# ------------------------------------------------------------------------------------------------
# colormap
# ------------------------------------------------------------------------------------------------
def _product_plane(seed, shape=(311, 203), dtype=np.float32):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    data = (40 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + rng.normal(0, 6, shape)).astype(dtype)
    data[rng.random(shape) < 0.2] = np.nan
    return data


@pytest.mark.parametrize("cmap", ["viridis", "jet", "turbo", "gray", "tab10"])
@pytest.mark.parametrize("limits", [(None, None), (0, 70), (-10.0, 45.5), (np.float32(-5), np.float32(30)),
                                    (None, 20.0), (np.float32(3), None), (7.0, 7.0)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_colormap_bit_exact(rg, cmap, limits, dtype):
    data = _product_plane(3, dtype=dtype)
    want = oracle.colormap_rgba(data, cmap, limits[0], limits[1])
    got = rg.apply_colormap_to_array(data, cmap, vmin=limits[0], vmax=limits[1])
    assert got.shape == data.shape + (4,) and got.dtype == np.uint8
    np.testing.assert_array_equal(got, want)


def test_colormap_reference_expectations(rg):
    # tests/test_geotiff_generation.py:61-127 through the HIP path
    import matplotlib.pyplot as plt
    xx, yy = np.meshgrid(np.linspace(0, 1, 100), np.linspace(0, 1, 100))
    data = 50 * (xx + yy) / 2
    data[45:55, 45:55] = np.nan
    out = rg.apply_colormap_to_array(data, "viridis")
    assert out.shape == (100, 100, 4) and out.dtype == np.uint8
    assert np.all(out[45:55, 45:55, 3] == 0) and np.all(out[0:10, 0:10, 3] == 255)
    np.testing.assert_array_equal(out, oracle.colormap_rgba(data, "viridis"))
    np.testing.assert_array_equal(rg.apply_colormap_to_array(data, plt.get_cmap("jet")),
                                  oracle.colormap_rgba(data, plt.get_cmap("jet")))
    filled = data.copy()
    filled[45:55, 45:55] = -9999.0
    out = rg.apply_colormap_to_array(filled, "viridis", fill_value=-9999.0)
    assert np.all(out[45:55, 45:55, 3] == 0)
    np.testing.assert_array_equal(out, oracle.colormap_rgba(filled, "viridis", fill_value=-9999.0))
    np.testing.assert_array_equal(data[45:55, 45:55], np.full((10, 10), np.nan))  # input untouched


def test_colormap_edge_cases(rg):
    from matplotlib.colors import LinearSegmentedColormap, ListedColormap
    data = _product_plane(8)
    listed = ListedColormap(["#102030", "#ff0000", "#00ff7f", "#0000ff", "#fefefe", "#7f7f00", "#123456"])
    listed.set_under("#010203")
    listed.set_over((0.9, 0.8, 0.7, 0.5))
    listed.set_bad((0.2, 0.4, 0.6, 0.8))
    big = LinearSegmentedColormap.from_list("big", ["black", "orange", "white"], N=1000)
    for cmap in (listed, big):
        for kw in ({}, {"vmin": -20, "vmax": 20}, {"fill_value": float(data[0, 0])}, {"vmin": 1e9, "vmax": 2e9}):
            np.testing.assert_array_equal(rg.apply_colormap_to_array(data, cmap, **kw),
                                          oracle.colormap_rgba(data, cmap, **kw), err_msg=str(kw))
    # nothing valid at all: limits fall back to 0 / 1 (geotiff.py:126-130)
    empty = np.full((17, 9), np.nan, dtype=np.float32)
    np.testing.assert_array_equal(rg.apply_colormap_to_array(empty, "viridis"), oracle.colormap_rgba(empty, "viridis"))
    # fill_value set and every other pixel NaN: np.nanmin hands matplotlib NaN limits
    import warnings
    odd = np.full((5, 6), np.nan, dtype=np.float32)
    odd[0, :3] = -9999.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = oracle.colormap_rgba(odd, "viridis", fill_value=-9999.0)
    np.testing.assert_array_equal(rg.apply_colormap_to_array(odd, "viridis", fill_value=-9999.0), want)
    # a plain colour table instead of a matplotlib object
    table = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
    np.testing.assert_array_equal(rg.apply_colormap_to_array(data, table, vmin=-30, vmax=30),
                                  oracle.colormap_rgba(data, ListedColormap(table), vmin=-30, vmax=30))
    with pytest.raises(ValueError):
        rg.apply_colormap_to_array(data, "viridis", vmin=5, vmax=1)
    # values outside +-inf clip and zero-size input
    wild = np.array([[-np.inf, np.inf, -1e30, 1e30, 0.0, np.nan]], dtype=np.float32)
    np.testing.assert_array_equal(rg.apply_colormap_to_array(wild, "jet", vmin=-1, vmax=1),
                                  oracle.colormap_rgba(wild, "jet", vmin=-1, vmax=1))
    assert rg.apply_colormap_to_array(np.zeros((0, 5), dtype=np.float32), "jet", vmin=0, vmax=1).shape == (0, 5, 4)


def test_colormap_full_size_device_resident(rg):
    """2000 x 2000 product plane kept in HBM end to end (column max -> colormap): every pixel's colour is one of the
    table's entries, alpha is 0 exactly on the NaN pixels, and a 1/64 sample is bit-exact against the oracle."""
    import torch
    dev = torch.device("cuda")
    rng = np.random.default_rng(21)
    grid = rng.normal(20, 15, (6, 2000, 2000)).astype(np.float32)
    grid[:, rng.random((2000, 2000)) < 0.3] = np.nan
    plane_t = rg.column_max(torch.from_numpy(grid).to(dev))
    rgba_t = rg.apply_colormap_to_array(plane_t, "turbo", vmin=-10.0, vmax=70.0)
    assert rgba_t.is_cuda and rgba_t.shape == (2000, 2000, 4) and rgba_t.dtype == torch.uint8
    rgba = rgba_t.cpu().numpy()
    plane = plane_t.cpu().numpy()
    np.testing.assert_array_equal(rgba[..., 3] == 0, np.isnan(plane))
    lut = rg.colormap_lut("turbo")
    codes = set(map(tuple, lut[:, :3]))
    assert set(map(tuple, np.unique(rgba[~np.isnan(plane)][:, :3], axis=0))) <= codes
    np.testing.assert_array_equal(rgba[::8, ::8], oracle.colormap_rgba(plane[::8, ::8], "turbo", -10.0, 70.0))


# ------------------------------------------------------------------------------------------------
# GridFilter (radar_grid/filters.py:609-780): the reference's tests/test_grid_filters.py, through rg_grid_filter
# ------------------------------------------------------------------------------------------------
def _ref_grid_filter(grid, kind, *args, fill_value=np.nan):
    """The reference's arithmetic, restated: copy, boolean mask, masked assignment (filters.py:655-779)."""
    out = grid.copy()
    if kind == "below":
        out[out < args[0]] = fill_value
    elif kind == "above":
        out[out > args[0]] = fill_value
    elif kind == "outside":
        out[(out < args[0]) | (out > args[1])] = fill_value
    elif kind == "invalid":
        out[np.isnan(out) | np.isinf(out)] = fill_value
    else:
        out[args[0](out)] = fill_value
    return out


def test_grid_filter_reference_tests(rg):
    gf = rg.GridFilter()
    grid = np.array([[10.0, 20.0, 30.0, 40.0], [15.0, 25.0, 35.0, 45.0], [12.0, 22.0, 32.0, 42.0]])
    keep = grid.copy()
    out = gf.apply_below(grid, 15)                                   # test_grid_filters.py:29-52
    assert np.isnan(out[0, 0]) and np.isnan(out[2, 0]) and out[0, 1] == 20.0 and out[1, 0] == 15.0
    np.testing.assert_array_equal(grid, keep)
    colmax = np.array([[8.0, 18.0, 28.0, 38.0, 48.0], [12.0, 22.0, 32.0, 42.0, 52.0], [10.0, 20.0, 30.0, 40.0, 50.0],
                       [14.0, 24.0, 34.0, 44.0, 54.0]])
    out = gf.apply_above(colmax, 40)                                 # :84-96
    assert np.isnan(out[0, 4]) and np.isnan(out[3, 4]) and out[0, 0] == 8.0 and out[0, 3] == 38.0
    out = gf.apply_outside_range(colmax, 15, 45)                     # :98-113
    assert np.isnan(out[0, 0]) and np.isnan(out[1, 4]) and out[0, 1] == 18.0 and out[0, 3] == 38.0
    out = gf.apply_invalid(np.array([[10.0, np.inf, 30.0], [15.0, -np.inf, np.nan]]))   # :119-155
    assert np.isnan(out[0, 1]) and np.isnan(out[1, 1]) and np.isnan(out[1, 2]) and out[0, 0] == 10.0
    out = gf.apply_custom(np.array([[5.0, 10.0, 15.0, 20.0], [25.0, 30.0, 35.0, 40.0]]),
                          lambda x: (x.astype(int) % 2) != 0)        # :176-195
    np.testing.assert_array_equal(np.isnan(out), [[True, False, True, False], [True, False, True, False]])
    row = np.array([[5.0, 15.0, 25.0]])
    np.testing.assert_array_equal(gf.apply_below(row, 15, fill_value=-9999), [[-9999, 15.0, 25.0]])   # :201-210
    np.testing.assert_array_equal(gf.apply_above(row, 15, fill_value=-999), [[5.0, 15.0, -999]])     # :212-221
    chained = gf.apply_above(gf.apply_below(np.array([[5.0, 15.0, 25.0, 35.0], [10.0, 20.0, 30.0, 40.0]]), 12), 35)
    np.testing.assert_array_equal(np.isnan(chained), [[True, False, False, False], [True, False, False, True]])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_grid_filter_bit_exact(rg, dtype):
    import torch
    rng = np.random.default_rng(17)
    plane = rng.uniform(5, 50, (257, 131)).astype(dtype)
    plane[rng.random(plane.shape) < 0.1] = np.nan
    plane[3, 4], plane[5, 6] = np.inf, -np.inf
    thr = 15.000001                     # not representable in float32: the comparison dtype matters
    gf = rg.GridFilter()
    cases = [("below", (thr,), {}), ("above", (40.3,), {}), ("outside", (thr, 40.3), {}), ("invalid", (), {}),
             ("below", (thr,), {"fill_value": -9999.0}), ("invalid", (), {"fill_value": 0.0}),
             ("custom", (lambda x: np.abs(x - 30) < 3,), {"fill_value": -1.0})]
    for kind, args, kw in cases:
        want = _ref_grid_filter(plane, kind, *args, **kw)
        fn = {"below": gf.apply_below, "above": gf.apply_above, "outside": gf.apply_outside_range,
              "invalid": gf.apply_invalid, "custom": gf.apply_custom}[kind]
        got = fn(plane, *args, **kw)
        assert got.dtype == dtype and got.shape == plane.shape
        np.testing.assert_array_equal(got, want, err_msg=f"{kind} {kw}")
    # device-resident input stays on the device
    t = torch.from_numpy(plane).cuda()
    out = gf.apply_outside_range(t, thr, 40.3)
    assert out.is_cuda and out.dtype == t.dtype
    np.testing.assert_array_equal(out.cpu().numpy(), _ref_grid_filter(plane, "outside", thr, 40.3))
    out = gf.apply_custom(t, lambda x: x > 45)
    np.testing.assert_array_equal(out.cpu().numpy(), _ref_grid_filter(plane, "custom", lambda x: x > 45))
