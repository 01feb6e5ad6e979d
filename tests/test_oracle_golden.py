"""Pin the CPU oracle (oracle/radar_grid_oracle.py) before anything trusts it:

 1. the literal known answers of the reference's own tests
    (/root/reference/tests/test_radar_grid_interpolate.py, test_radar_grid_products.py), and
 2. the golden vectors produced by running the reference's modules (tests/golden/make_golden.py).

CPU only; no HIP code is touched here.
"""
import numpy as np
import pytest

from conftest import (builder_kwargs, golden_names, grid_spec, load_golden, reference_indices, volume_for)
from oracle import radar_grid_oracle as oracle


def _apply(indptr, idx, w, values, mask=None, shape=(1, 1, 1), fill=np.nan):
    values = np.asarray(values, dtype=np.float32)
    if mask is None:
        mask = ~np.isfinite(values)          # what np.ma.masked_invalid does in the reference's tests
    return oracle.csr_apply(np.asarray(indptr), np.asarray(idx, dtype=np.int32), np.asarray(w, dtype=np.float32),
                            values, mask, shape, fill)


# ------------------------------------------------------------------------------------------------
# 1. reference unit-test known answers
# ------------------------------------------------------------------------------------------------
class TestReferenceKnownAnswers:
    def test_weighted_average_17(self):          # test_radar_grid_interpolate.py:75-93
        out = _apply([0, 2], [0, 1], [0.3, 0.7], [10.0, 20.0])
        np.testing.assert_almost_equal(out[0, 0, 0], 17.0, decimal=5)

    def test_single_point_21(self):              # :236-254
        out = _apply([0, 3], [0, 1, 2], [0.2, 0.5, 0.3], [10.0, 20.0, 30.0])
        np.testing.assert_almost_equal(out[0, 0, 0], 21.0, decimal=5)

    def test_mixed_valid_invalid_20(self):       # :256-277
        out = _apply([0, 3], [0, 1, 2], [0.3, 0.4, 0.3], [10.0, np.nan, 30.0])
        np.testing.assert_almost_equal(out[0, 0, 0], 20.0, decimal=5)

    def test_inf_excluded(self):                 # :283-299
        out = _apply([0, 3], [0, 1, 2], [0.3, 0.4, 0.3], [10.0, np.inf, 30.0])
        assert np.isfinite(out[0, 0, 0])

    def test_negative_inf_excluded_10(self):     # :301-317
        out = _apply([0, 2], [0, 1], [0.5, 0.5], [10.0, -np.inf])
        np.testing.assert_almost_equal(out[0, 0, 0], 10.0, decimal=5)

    def test_fill_value_on_empty_row(self):      # :116-132
        out = _apply([0, 0], [], [], [10.0], fill=-9999.0)
        assert out[0, 0, 0] == -9999.0

    def test_all_masked_is_nan(self):            # :134-153
        out = _apply([0, 2], [0, 1], [0.5, 0.5], [10.0, 20.0], mask=np.array([True, True]))
        assert np.isnan(out[0, 0, 0])

    def test_empty_geometry_all_fill(self):      # :218-234
        out = _apply(np.zeros(9, dtype=np.int32), [], [], [10.0], shape=(2, 2, 2))
        assert out.shape == (2, 2, 2) and np.all(np.isnan(out))

    def test_nan_gates_first_voxel(self):        # :50-59
        vals = np.ones(16, dtype=np.float32) * 10.0
        vals[0:4] = np.nan
        out = _apply(np.arange(0, 17, 2), np.arange(16), np.ones(16), vals, shape=(2, 2, 2))
        assert out.dtype == np.float32 and np.isnan(out.ravel()[0]) and out.ravel()[2] == 10.0

    def test_column_known_answers(self):         # test_radar_grid_products.py:284-357
        data = np.zeros((10, 50, 50), dtype=np.float32)
        for z in range(10):
            data[z] = z * 10.0
        data[:, 0:5, 0:5] = np.nan
        cmax, cmin, cmean = (f(data, 0, 9) for f in (oracle.column_max, oracle.column_min, oracle.column_mean))
        assert cmax[10, 10] == 90.0 and np.isnan(cmax[0, 0])
        assert cmin[10, 10] == 0.0 and np.isnan(cmin[0, 0])
        np.testing.assert_almost_equal(cmean[10, 10], 45.0, decimal=1)
        assert np.isnan(cmean[0, 0])
        arg = oracle.column_argmax(data, 0, 9)
        assert arg[10, 10] == 9 and arg[0, 0] == -1 and arg.dtype == np.int32
        part = np.ones((10, 50, 50), dtype=np.float32) * 10.0
        part[0:3] = np.nan
        assert np.all(oracle.column_max(part, 0, 9)[10:20, 10:20] == 10.0)
        assert np.all(oracle.column_argmax(part, 0, 9) == 3)   # first level attaining the max

    def test_cappi_out_of_range_and_dtype(self):  # test_radar_grid_products.py:190-226
        grid = np.random.default_rng(0).random((10, 50, 50)).astype(np.float32)
        assert np.all(np.isnan(oracle.cappi(grid, (0.0, 10000.0), 15000.0)))
        c = oracle.cappi(grid, (0.0, 10000.0), 5000.0)
        assert c.shape == (50, 50) and c.dtype == np.float32


# ------------------------------------------------------------------------------------------------
# 2. golden vectors from the reference itself
# ------------------------------------------------------------------------------------------------
GEOM_CASES = golden_names("g2_") + golden_names("g3_") + golden_names("g4_") + golden_names("g6_")


def _canon(indptr, idx, w):
    return oracle.canonical_rows(indptr, idx, w)


@pytest.mark.parametrize("name", GEOM_CASES)
def test_builder_matches_reference_csr(name):
    """Neighbour sets identical; weights bit-identical for cressman/nearest and within 1 float32 ulp for
    barnes2 (both sides evaluate exp() in float64, then round)."""
    meta, ref = load_golden(name)
    vol = volume_for(meta)
    shape, limits = grid_spec(meta)
    indptr, idx, w = oracle.build_geometry(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, **builder_kwargs(meta))
    r_ip, r_idx, r_w = _canon(ref["indptr"], reference_indices(name, meta, ref), ref["weights"])
    np.testing.assert_array_equal(indptr, r_ip.astype(np.int64))
    np.testing.assert_array_equal(idx, r_idx)
    if meta["weighting"] == "barnes2":
        ulp = np.abs(w.view(np.int32).astype(np.int64) - r_w.view(np.int32).astype(np.int64))
        assert ulp.max(initial=0) <= 1
        assert (ulp > 0).mean() < 1e-4 if ulp.size else True
    else:
        np.testing.assert_array_equal(w, r_w)


@pytest.mark.parametrize("name", golden_names("g2_") + golden_names("g3_") + golden_names("g6_"))
def test_csr_apply_matches_reference_grids(name):
    """Oracle apply on the REFERENCE's CSR (same row order) reproduces the reference grids bit for bit;
    with the QC filter and with a custom fill value too."""
    meta, ref = load_golden(name)
    vol = volume_for(meta)
    shape, _ = grid_spec(meta)
    idx = reference_indices(name, meta, ref)
    qc = None
    if "qc" in meta:
        qc = oracle.gate_mask("below", np.ma.getdata(vol.fields[meta["qc"][0]]), meta["qc"][1])
        assert int(qc.sum()) == int(ref["qc_excluded_count"][0])
    for fname in meta["fields"]:
        data, mask = oracle.merge_masks(vol.fields[fname])
        got = oracle.csr_apply(ref["indptr"], idx, ref["weights"], data, mask, shape)
        np.testing.assert_array_equal(got, ref[f"grid_{fname}"])
        if qc is not None:
            _, mask_qc = oracle.merge_masks(vol.fields[fname], [qc])
            got = oracle.csr_apply(ref["indptr"], idx, ref["weights"], data, mask_qc, shape)
            np.testing.assert_array_equal(got, ref[f"grid_{fname}_qc"])
        if f"grid_{fname}_fill" in ref:
            got = oracle.csr_apply(ref["indptr"], idx, ref["weights"], data, mask, shape, fill_value=-9999.0)
            np.testing.assert_array_equal(got, ref[f"grid_{fname}_fill"])


@pytest.mark.parametrize("name", golden_names("g3_c2_r150") + golden_names("g3_c2_r060"))
def test_f64_yardstick_within_tolerance(name):
    """The float64-summing yardstick stays within the parity tolerance of the reference's float32 pairwise sums: rtol
    1e-5 with an absolute floor of 1e-5 * max|field| (the kernels sum in float32 in their own orders)."""
    meta, ref = load_golden(name)
    vol = volume_for(meta)
    shape, _ = grid_spec(meta)
    idx = reference_indices(name, meta, ref)
    for fname in meta["fields"]:
        data, mask = oracle.merge_masks(vol.fields[fname])
        got = oracle.csr_apply_f64(ref["indptr"], idx, ref["weights"], data, mask, shape)
        want = ref[f"grid_{fname}"]
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        atol = 1e-5 * float(np.nanmax(np.abs(data[~mask])))
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=atol, equal_nan=True)


@pytest.mark.parametrize("name", golden_names("g2_") + golden_names("g3_c2_r150") + golden_names("g3_c2_r060")
                         + golden_names("g6_"))
def test_rowwise_order_restatement_is_the_reference_mean(name):
    """oracle.csr_apply_rowwise_order -- the row-wise kernel's documented order of float32 additions, against which the
    GPU tests check that kernel bit for bit -- is itself pinned here: on the REFERENCE's CSR it reproduces the reference's
    grids (same voxels filled, rtol 1e-5 + 1e-5 * max|field|), fused fields equal single-field passes of the same lane
    split bit for bit, and every lane split gives the same values to rounding."""
    meta, ref = load_golden(name)
    vol = volume_for(meta)
    shape, _ = grid_spec(meta)
    idx = reference_indices(name, meta, ref)
    pairs = [oracle.merge_masks(vol.fields[f]) for f in meta["fields"]]
    data, masks = [p[0] for p in pairs], [p[1] for p in pairs]
    fused = oracle.csr_apply_rowwise_order(ref["indptr"], idx, ref["weights"], data, masks, shape)
    for k, fname in enumerate(meta["fields"]):
        want = ref[f"grid_{fname}"]
        atol = 1e-5 * float(np.nanmax(np.abs(data[k][~masks[k]])))
        np.testing.assert_array_equal(np.isnan(fused[k]), np.isnan(want))
        np.testing.assert_allclose(fused[k], want, rtol=1e-5, atol=atol, equal_nan=True)
        hint = 70 + oracle.ROWWISE_TARGET[len(data)]          # a single-field pass with the fused pass's lane split
        single = oracle.csr_apply_rowwise_order(ref["indptr"], idx, ref["weights"], data[k:k + 1], masks[k:k + 1], shape,
                                                lanes_hint=hint, kpre=oracle.ROWWISE_KPRE[len(data)])[0]   # and batch
        np.testing.assert_array_equal(single, fused[k])
        for lanes in (1, 64):
            other = oracle.csr_apply_rowwise_order(ref["indptr"], idx, ref["weights"], data[k:k + 1], masks[k:k + 1],
                                                   shape, lanes_hint=lanes)[0]
            np.testing.assert_allclose(other, want, rtol=1e-5, atol=atol, equal_nan=True)


@pytest.mark.parametrize("name", golden_names("g3_c2_r060")[:1] + golden_names("g2_")[:1])
def test_rowwise_order_restatement_for_five_to_eight_fields(name):
    """Passes of 5-8 fields (eight volumes of a batch in one pass): the restatement keeps ONE chain for the weight sums
    (ROWWISE_WEIGHT_CHAINS) and the eight-field lane split; on the REFERENCE's CSR every field of such a pass still
    reproduces the reference's grid within the bar, equals a single-field pass of the same lane split / batch / chain
    counts bit for bit, and differs from the two-chain form only in float32 rounding."""
    meta, ref = load_golden(name)
    vol = volume_for(meta)
    shape, _ = grid_spec(meta)
    idx = reference_indices(name, meta, ref)
    base = [oracle.merge_masks(vol.fields[f]) for f in meta["fields"]]
    for nf in (5, 8):
        pairs = [base[i % len(base)] for i in range(nf)]
        data, masks = [p[0] for p in pairs], [p[1] for p in pairs]
        assert oracle.ROWWISE_WEIGHT_CHAINS[nf] == 1 and oracle.ROWWISE_KPRE[nf] == 3
        fused = oracle.csr_apply_rowwise_order(ref["indptr"], idx, ref["weights"], data, masks, shape)
        two = oracle.csr_apply_rowwise_order(ref["indptr"], idx, ref["weights"], data, masks, shape, weight_chains=2)
        for k in range(nf):
            fname = meta["fields"][k % len(base)]
            want = ref[f"grid_{fname}"]
            atol = 1e-5 * float(np.nanmax(np.abs(data[k][~masks[k]])))
            np.testing.assert_array_equal(np.isnan(fused[k]), np.isnan(want))
            np.testing.assert_allclose(fused[k], want, rtol=1e-5, atol=atol, equal_nan=True)
            np.testing.assert_allclose(fused[k], two[k], rtol=2e-6, atol=atol, equal_nan=True)
        single = oracle.csr_apply_rowwise_order(ref["indptr"], idx, ref["weights"], data[1:2], masks[1:2], shape,
                                                lanes_hint=70 + oracle.ROWWISE_TARGET[nf], kpre=oracle.ROWWISE_KPRE[nf],
                                                weight_chains=1)[0]
        np.testing.assert_array_equal(single, fused[1])


def test_rowwise_order_restatement_known_answers():
    """The reference's unit-test answers (test_radar_grid_interpolate.py:75-93, :236-277, :116-153) through the restatement."""
    def run(indptr, idx, w, values, fill=np.nan):
        values = np.asarray(values, dtype=np.float32)
        return oracle.csr_apply_rowwise_order(np.asarray(indptr), np.asarray(idx, dtype=np.int32),
                                              np.asarray(w, dtype=np.float32), [values], [~np.isfinite(values)], (1, 1, 1),
                                              fill)[0, 0, 0, 0]
    np.testing.assert_almost_equal(run([0, 2], [0, 1], [0.3, 0.7], [10.0, 20.0]), 17.0, decimal=5)
    np.testing.assert_almost_equal(run([0, 3], [0, 1, 2], [0.2, 0.5, 0.3], [10.0, 20.0, 30.0]), 21.0, decimal=5)
    np.testing.assert_almost_equal(run([0, 3], [0, 1, 2], [0.3, 0.4, 0.3], [10.0, np.nan, 30.0]), 20.0, decimal=5)
    assert run([0, 0], [], [], [1.0], fill=-9999.0) == np.float32(-9999.0)
    assert np.isnan(run([0, 2], [0, 1], [0.5, 0.5], [np.nan, np.nan]))


def _product_checks(prefix, grid, z_limits, ref):
    nz = grid.shape[0]
    eq = np.testing.assert_array_equal
    eq(oracle.cappi(grid, z_limits, 4000.0, "linear"), ref[f"{prefix}_cappi4000_linear"])
    eq(oracle.cappi(grid, z_limits, 4000.0, "nearest"), ref[f"{prefix}_cappi4000_nearest"])
    eq(oracle.cappi(grid, z_limits, 2500.0, "linear"), ref[f"{prefix}_cappi2500_linear"])
    eq(oracle.cappi(grid, z_limits, 99000.0, "linear"), ref[f"{prefix}_cappi_above"])
    lo, hi = oracle.column_range(nz)
    eq(oracle.column_max(grid, lo, hi), ref[f"{prefix}_colmax"])
    eq(oracle.column_min(grid, lo, hi), ref[f"{prefix}_colmin"])
    eq(oracle.column_mean(grid, lo, hi), ref[f"{prefix}_colmean"])
    lo, hi = oracle.column_range(nz, z_min_alt=1000, z_max_alt=8000, z_limits=z_limits)
    eq(oracle.column_max(grid, lo, hi), ref[f"{prefix}_colmax_alt"])
    eq(oracle.column_mean(grid, lo, hi), ref[f"{prefix}_colmean_alt"])
    lo, hi = oracle.column_range(nz, z_min_idx=2, z_max_idx=6)
    eq(oracle.column_max(grid, lo, hi), ref[f"{prefix}_colmax_idx"])
    # argmax contract is build-defined; it must at least select the reference's max values
    arg = oracle.column_argmax(grid, 0, nz - 1)
    cmax = ref[f"{prefix}_colmax"]
    has = arg >= 0
    eq(has, ~np.isnan(cmax))
    yy, xx = np.nonzero(has)
    eq(grid[arg[yy, xx], yy, xx], cmax[yy, xx])


@pytest.mark.parametrize("name", golden_names("g5_"))
def test_products_match_reference(name):
    meta, ref = load_golden(name)
    _product_checks("P", ref["grid"], tuple(meta["z_limits"]), ref)


@pytest.mark.parametrize("name", [n for n in golden_names("g3_") if n.endswith("barnes2")])
def test_products_on_gridded_windows(name):
    meta, ref = load_golden(name)
    _, limits = grid_spec(meta)
    _product_checks("DBZH", ref["grid_DBZH"], limits[0], ref)


def test_exact_level_cappi_plan():
    """z 0..19 km / 20 levels puts 4000 m exactly on level 4 (view, no arithmetic); z 0..15 km does not."""
    assert oracle.cappi_plan((0.0, 19000.0), 20, 4000.0) == ("level", 4)
    plan = oracle.cappi_plan((0.0, 15000.0), 20, 4000.0)
    assert plan[0] == "lerp" and plan[1] == 5 and abs(plan[3] - (4000.0 / (15000.0 / 19) - 5)) < 1e-12


def test_antenna_transform_agrees_with_package():
    """Two independent restatements of PyART's published model (parity unpinned vs PyART itself)."""
    from radar_processor_amd import synthetic
    r = np.linspace(120.0, 240e3, 50)[None, :]
    az = np.array([0.0, 33.0, 181.5, 359.0])[:, None]
    el = np.array([0.5, 3.0, 11.8, 25.0])[:, None]
    for a, b in zip(oracle.antenna_to_cartesian(r, az, el), synthetic.antenna_to_cartesian(r, az, el)):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-6)


def _g9_tables():
    """(tag, elevations, ranges, reference z) of tests/golden/g9_beam_z.npz -- the reference's own
    ``products.compute_beam_height(r*cos(el), el, 0)`` (products.py:70-89) on the synthetic sweep tables."""
    from radar_processor_amd import synthetic
    meta, arr = load_golden("g9_beam_z")
    for tag, t in meta["tables"].items():
        elev, _, rng_m = synthetic.sweep_geometry(t["n_elev"], 4, t["n_gates"])
        assert [float(e) for e in elev] == t["elevations"] and float(rng_m[0]) == t["first_range"]
        assert float(rng_m[1] - rng_m[0]) == t["range_step"], "sweep tables drifted from the ones the fixture was made with"
        yield tag, np.asarray(elev, dtype=np.float64), np.asarray(rng_m, dtype=np.float64), arr[f"z_{tag}"]


def test_gate_height_is_pinned_by_the_reference_beam_height():
    """a1: the z coordinate of the gate transform against a REFERENCE-held formula.  The reference's x/y/z come from
    PyART (absent: x and y stay parity-unpinned), but its ``compute_beam_height`` evaluates the same 4/3-earth height
    with the same constants; g9 stores its float64 output for every sweep elevation and the range tables of configs 1,
    2 and 4.  The oracle's and the package's float64 z agree with it to 2e-9 m (the reference goes through
    ``r*cos/cos``), hence to the same float32 except for rare 1-ulp double roundings."""
    from radar_processor_amd import synthetic
    for tag, elev, rng_m, z_ref in _g9_tables():
        for fn in (oracle.antenna_to_cartesian, synthetic.antenna_to_cartesian):
            z = fn(rng_m[None, :], np.zeros((1, 1)), elev[:, None])[2]
            assert z.shape == z_ref.shape
            np.testing.assert_allclose(z, z_ref, rtol=0, atol=2e-9)
            ulp = np.abs(z.astype(np.float32).view(np.int32).astype(np.int64)
                         - z_ref.astype(np.float32).view(np.int32).astype(np.int64))
            assert ulp.max() <= 1, tag
        np.testing.assert_allclose(oracle.beam_height(rng_m[None, :] * np.cos(np.radians(elev))[:, None], elev[:, None]),
                                   z_ref, rtol=0, atol=0)       # the oracle's restatement of products.py:70-89 itself


# ------------------------------------------------------------------------------------------------
# constant-elevation PPI + beam-height helpers
# ------------------------------------------------------------------------------------------------
def _ppi_cases(ref):
    for key in sorted(k for k in ref if k.startswith("ppi_e") and not k.endswith("_ke1")):
        _, e, interp, curv = key.split("_")
        yield key, float(e[1:]), interp, curv == "curved"


def test_elevation_ppi_oracle_matches_reference():
    meta, ref = load_golden("g7_ppi")
    limits = tuple(tuple(v) for v in meta["grid_limits"])
    for key, elev, interp, curved in _ppi_cases(ref):
        got = oracle.elevation_ppi(ref["grid"], limits, elev, interp, curved)
        assert got.dtype == ref[key].dtype == (np.float64 if interp == "linear" else np.float32)
        np.testing.assert_array_equal(got, ref[key], err_msg=key)
    np.testing.assert_array_equal(oracle.elevation_ppi(ref["grid"], limits, 2.0, ke=1.0), ref["ppi_e2.0_linear_ke1"])
    np.testing.assert_array_equal(oracle.beam_height(ref["bh_dist"], 2.0, 100.0), ref["bh_curved"])
    np.testing.assert_array_equal(oracle.beam_height_flat(ref["bh_dist"], 2.0, 100.0), ref["bh_flat"])


def test_beam_height_helpers_match_reference():
    """Host-side helpers of the product package (plain NumPy formulas, no GPU involved)."""
    import radar_processor_amd as rg
    meta, ref = load_golden("g7_ppi")
    limits = tuple(tuple(v) for v in meta["grid_limits"])
    geom = rg.GridGeometry(tuple(meta["grid_shape"]), limits, np.zeros(2, dtype=np.int32), np.zeros(0, dtype=np.int32),
                           np.zeros(0, dtype=np.float32), toa=17000.0, radar_altitude=meta["radar_altitude"])
    d = ref["bh_dist"]
    np.testing.assert_array_equal(rg.compute_beam_height(d, 2.0, 100.0), ref["bh_curved"])
    np.testing.assert_array_equal(rg.compute_beam_height_simple(d, 2.0, 100.0), ref["bh_simple"])
    np.testing.assert_array_equal(rg.compute_beam_height_flat(d, 2.0, 100.0), ref["bh_flat"])
    diff = rg.get_beam_height_difference(geom, 1.5, radar_altitude=312.0)
    assert diff.dtype == np.float64
    np.testing.assert_array_equal(diff, ref["bh_difference"])
    np.testing.assert_array_equal(rg.get_elevation_from_z_level(3000.0, geom, radar_altitude=312.0), ref["elev_from_z_curved"])
    np.testing.assert_array_equal(rg.get_elevation_from_z_level(3000.0, geom, radar_altitude=312.0, earth_curvature=False),
                                  ref["elev_from_z_flat"])
    # reference unit tests (tests/test_radar_grid_products.py:30-91): monotone in range, flat-earth linearity
    h = rg.compute_beam_height(np.array([10000.0, 20000.0, 50000.0]), 2.0, 100.0)
    assert np.all(h > 100.0) and h[0] < h[1] < h[2]
    assert rg.compute_beam_height(np.array([10000.0]), 45.0, 0.0)[0] > 7000.0
    hf = rg.compute_beam_height_flat(np.array([10000.0, 20000.0]), 2.0, 100.0)
    np.testing.assert_almost_equal(hf[1] - hf[0], 10000.0 * np.tan(np.radians(2.0)), decimal=1)
    with pytest.raises(ValueError, match="Unknown interpolation method"):
        rg.constant_elevation_ppi(ref["grid"], geom, 2.0, interpolation="cubic")


# ------------------------------------------------------------------------------------------------
# 3. raster stage: no golden vectors exist (radar_processor / geotiff do not import
#    here), so the oracle is pinned by the expectations of the reference's own tests, restated below.
# ------------------------------------------------------------------------------------------------
class _Filter:                                   # the MockFilter of test_processor_phases.py:277-284
    def __init__(self, field, lo, hi):
        self.field, self.min, self.max = field, lo, hi


class TestRasterStageReferenceExpectations:
    def test_collapse_colmax(self):              # tests/test_utils.py:209-219
        data3d = np.random.default_rng(1).random((10, 20, 30))
        out = oracle.collapse_3d_to_2d(data3d, "colmax")
        assert out.shape == (20, 30)
        np.testing.assert_array_almost_equal(out.data, data3d.max(axis=0))

    def test_collapse_cappi(self):               # tests/test_utils.py:222-238
        data3d = np.random.default_rng(2).random((10, 20, 30))
        z = np.linspace(0, 10000, 10)
        out = oracle.collapse_3d_to_2d(data3d, "cappi", z_levels=z, target_height_m=5000.0)
        assert out.shape == (20, 30)
        np.testing.assert_array_almost_equal(out.data, data3d[np.abs(z - 5000.0).argmin()])

    def test_collapse_ppi(self):                 # tests/test_utils.py:241-259
        data3d = np.random.default_rng(3).random((10, 20, 30))
        out = oracle.collapse_3d_to_2d(data3d, "ppi", x_coords=np.linspace(-10000, 10000, 30),
                                       y_coords=np.linspace(-10000, 10000, 20), z_levels=np.linspace(0, 10000, 10),
                                       elevation_deg=0.5)
        assert out.shape == (20, 30) and isinstance(out, np.ma.MaskedArray)
        # 10 km at 0.5 degrees is ~100 m up: every pixel of this grid must take the lowest level
        np.testing.assert_array_equal(out.data, data3d[0].astype(np.float32))

    def test_collapse_unknown_product(self):     # utils.py:385
        with pytest.raises(ValueError):
            oracle.collapse_3d_to_2d(np.zeros((2, 2, 2)), "rhi")

    @pytest.fixture
    def plane(self):                             # test_processor_phases.py:268-274
        return np.ma.array(np.linspace(-30, 60, 1000).reshape(50, 20), mask=np.zeros((50, 20), dtype=bool))

    def test_filter_min(self, plane):            # :286-298
        out = oracle.filter_masks(plane.copy(), [_Filter("DBZH", -20, None)], [], "DBZH", {})
        np.testing.assert_array_equal(out.mask, plane.data < -20)
        assert out.mask.sum() > 0

    def test_filter_max(self, plane):            # :300-312
        out = oracle.filter_masks(plane.copy(), [_Filter("DBZH", None, 50)], [], "DBZH", {})
        np.testing.assert_array_equal(out.mask, plane.data > 50)
        assert out.mask.sum() > 0

    def test_filters_cumulative(self, plane):    # :314-327
        fl = [_Filter("DBZH", -20, None), _Filter("DBZH", None, 50)]
        out = oracle.filter_masks(plane.copy(), fl, [], "DBZH", {})
        np.testing.assert_array_equal(out.mask, (plane.data < -20) | (plane.data > 50))

    def test_qc_filter(self, plane):             # :329-345
        rho = np.ma.array(np.linspace(0.5, 1.0, 1000).reshape(50, 20), mask=np.zeros((50, 20), dtype=bool))
        out = oracle.filter_masks(plane.copy(), [], [_Filter("RHOHV", 0.8, None)], "DBZH", {"RHOHV": rho})
        np.testing.assert_array_equal(out.mask, rho.data < 0.8)
        assert out.mask.sum() > 0

    def test_filter_keeps_values(self, plane):   # :347-357
        out = oracle.filter_masks(plane.copy(), [_Filter("DBZH", -20, 50)], [], "DBZH", {})
        np.testing.assert_array_equal(out.data, plane.data)

    def test_rhohv_low_minimum_is_skipped(self, plane):   # processor.py:849
        rho = np.ma.array(np.linspace(0.0, 1.0, 1000).reshape(50, 20))
        out = oracle.filter_masks(rho, [_Filter("RHOHV", 0.3, None)], [], "RHOHV", {})
        assert not np.ma.getmaskarray(out).any()

    def test_remask_rules(self):                 # processor.py:541-546
        plane = np.array([[-31.0, -30.0, -29.0, np.nan, np.inf]])
        np.testing.assert_array_equal(np.ma.getmaskarray(oracle.collapse_remask(plane, "DBZH")),
                                      [[True, True, False, True, True]])
        np.testing.assert_array_equal(np.ma.getmaskarray(oracle.collapse_remask(plane, "ZDR")),
                                      [[True, False, False, True, True]])
        np.testing.assert_array_equal(np.ma.getmaskarray(oracle.collapse_remask(plane, "RHOHV")),
                                      [[False, False, False, True, True]])

    @pytest.fixture
    def gradient(self):                          # tests/test_geotiff_generation.py:61-74
        xx, yy = np.meshgrid(np.linspace(0, 1, 100), np.linspace(0, 1, 100))
        data = 50 * (xx + yy) / 2
        data[45:55, 45:55] = np.nan
        return data

    def test_colormap_basic(self, gradient):     # :81-87
        out = oracle.colormap_rgba(gradient, "viridis")
        assert out.shape == (100, 100, 4) and out.dtype == np.uint8

    def test_colormap_limits(self, gradient):    # :89-94
        out = oracle.colormap_rgba(gradient, "viridis", vmin=0, vmax=70)
        assert out.shape == (100, 100, 4) and out.dtype == np.uint8

    def test_colormap_nan_transparent(self, gradient):   # :96-106
        out = oracle.colormap_rgba(gradient, "viridis")
        assert np.all(out[45:55, 45:55, 3] == 0) and np.all(out[0:10, 0:10, 3] == 255)

    def test_colormap_object(self, gradient):    # :108-114
        import matplotlib.pyplot as plt
        out = oracle.colormap_rgba(gradient, plt.get_cmap("jet"))
        assert out.shape == (100, 100, 4) and out.dtype == np.uint8

    def test_colormap_fill_value(self, gradient):        # :116-127
        data = gradient.copy()
        data[45:55, 45:55] = -9999.0
        out = oracle.colormap_rgba(data, "viridis", fill_value=-9999.0)
        assert np.all(out[45:55, 45:55, 3] == 0)
