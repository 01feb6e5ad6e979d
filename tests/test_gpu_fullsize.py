"""BASELINE.json's full sizes on the GPU, checked through size-independent properties (the oracle cannot run at
these sizes in seconds):

 * config 2 (12x360x1000 gates -> 20x1000x1000, ~1 G CSR pairs): CSR structure invariants, constant-field
   reproduction, linearity, fused no-CSR gridder == CSR path, mask monotonicity, product consistency, and an
   oracle spot-check on sampled voxel rows;
 * config 4 (14x720x2000 gates -> 40x2000x2000, CSR not materialised): fused gridder invariants.
"""
import numpy as np
import pytest

from conftest import ATOL_FRAC, assert_same_to_rounding
from oracle import radar_grid_oracle as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2(tmp_path_factory):
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    rg.load_library()
    cfg = synthetic.CONFIGS["C2"]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH", "ZDR", "RHOHV"))
    geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"],
                                    str(tmp_path_factory.mktemp("geom")))
    dev = torch.device("cuda", 0)
    to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    fields = {k: to(np.ma.getdata(v), torch.float32) for k, v in vol.fields.items()}
    masks = {k: to(np.ma.getmaskarray(v), torch.uint8) for k, v in vol.fields.items()}
    return dict(rg=rg, torch=torch, cfg=cfg, vol=vol, geom=geom, dev=dev, fields=fields, masks=masks)


def test_c2_csr_structure(c2):
    torch, geom = c2["torch"], c2["geom"]
    csr = geom.device_csr(c2["dev"])
    n_vox = int(np.prod(c2["cfg"]["grid_shape"]))
    assert csr.n_vox == n_vox and csr.n_pairs > 8e8            # ~1.0 G pairs, ~51 per voxel
    ip = csr.indptr.to(torch.int64)
    assert int(ip[0]) == 0 and int(ip[-1]) == csr.n_pairs
    lengths = ip[1:] - ip[:-1]
    assert int(lengths.min()) >= 0
    empty = float((lengths == 0).float().mean())
    assert 0.2 < empty < 0.45                                   # grid corners / top levels are out of reach
    assert int(csr.gate_indices.min()) >= 0 and int(csr.gate_indices.max()) < c2["vol"].n_total_gates
    w = csr.weights
    assert bool(torch.isfinite(w).all()) and float(w.min()) > 0 and float(w.max()) <= 1.00002   # barnes2 in (1e-5, 1+1e-5]
    # every weight is at least exp(-4)+1e-5 (d2 < r2) -- the rim value of compute.py:83
    assert float(w.min()) >= np.float32(np.exp(-4.0) + 1e-5) * (1 - 1e-6)


def test_c2_constant_field_and_linearity(c2):
    rg, torch, geom, dev = c2["rg"], c2["torch"], c2["geom"], c2["dev"]
    g = c2["vol"].n_total_gates
    const = torch.full((g,), 7.25, dtype=torch.float32, device=dev)
    a, b = c2["fields"]["DBZH"].clone(), c2["fields"]["ZDR"].clone()
    a[torch.isnan(a)] = 0.0
    b[torch.isnan(b)] = 0.0
    combo = 2.0 * a - 0.5 * b
    out = rg.grid_fields_device(geom, [const, a, b, combo])
    filled = torch.isfinite(out[0])
    # weighted mean of a constant is the constant up to the float32 rounding of products and tile partial sums
    assert float((out[0][filled] - 7.25).abs().max()) <= 7.25 * 2e-6
    lengths = geom.device_csr(dev).indptr.to(torch.int64).diff()
    assert bool((filled.view(-1) == (lengths > 0)).all())      # no mask: filled <=> row non-empty
    lin = 2.0 * out[1] - 0.5 * out[2]
    err = (out[3] - lin)[filled].abs().max()
    assert float(err) <= 1e-5 * float(combo.abs().max())


def test_c2_fused_gridder_equals_csr_path(c2):
    rg, torch, geom, dev, vol = c2["rg"], c2["torch"], c2["geom"], c2["dev"], c2["vol"]
    names = ["DBZH", "ZDR", "RHOHV"]
    f = [c2["fields"][n] for n in names]
    m = [c2["masks"][n] for n in names]
    qc = rg.device_gate_mask(c2["fields"]["RHOHV"], "below", 0.8)
    k1 = rg.grid_fields_device(geom, f, m, shared_mask=qc)
    search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, c2["cfg"]["grid_shape"], c2["cfg"]["grid_limits"])
    k2 = rg.roi_grid_fields_device(search, f, m, shared_mask=qc)
    assert bool((torch.isnan(k1) == torch.isnan(k2)).all())
    for i, n in enumerate(names):
        scale = float(torch.nan_to_num(f[i], nan=0.0).abs().max())
        d = torch.nan_to_num(k1[i] - k2[i], nan=0.0).abs().max()
        assert float(d) <= 1e-5 * scale, n
    # masking more gates can only remove voxels, never add them
    plain = rg.grid_fields_device(geom, f[:1], m[:1])
    assert bool((torch.isfinite(k1[0]) <= torch.isfinite(plain[0])).all())
    assert int(torch.isfinite(k1[0]).sum()) < int(torch.isfinite(plain[0]).sum())


def test_c2_products_consistency_and_oracle_rows(c2):
    rg, torch, geom, dev, vol = c2["rg"], c2["torch"], c2["geom"], c2["dev"], c2["vol"]
    grid = rg.grid_fields_device(geom, [c2["fields"]["DBZH"]], [c2["masks"]["DBZH"]])[0]
    cmax, arg = rg.column_argmax(grid)
    cmin = rg.column_min(grid)
    cmean = rg.column_mean(grid)
    has = arg >= 0
    assert bool((has == torch.isfinite(cmax)).all())
    picked = torch.gather(grid, 0, arg.clamp(min=0).long().unsqueeze(0))[0]
    assert bool((picked[has] == cmax[has]).all())                       # grid[argmax] is the max, bitwise
    assert bool((torch.nan_to_num(grid, nan=-1e30) <= torch.nan_to_num(cmax, nan=1e30).unsqueeze(0)).all())
    assert bool((cmin[has] <= cmean[has] + 1e-4).all()) and bool((cmean[has] <= cmax[has] + 1e-4).all())
    # first-index tie rule: no lower level holds the same value
    lower = torch.arange(grid.shape[0], device=dev).view(-1, 1, 1) < arg.unsqueeze(0)
    assert not bool(((grid == cmax.unsqueeze(0)) & lower).any())
    cap = rg.constant_altitude_ppi(grid, geom, 4000.0)                  # z 0..15 km / 20 levels: a lerp
    plan = oracle.cappi_plan(c2["cfg"]["grid_limits"][0], 20, 4000.0)
    assert plan[0] == "lerp"
    lo, hi = grid[plan[1]], grid[plan[1] + 1]
    both = torch.isfinite(lo) & torch.isfinite(hi)
    assert bool((torch.isfinite(cap) == both).all())
    assert bool((cap[both] >= torch.minimum(lo, hi)[both] - 1e-4).all())
    assert bool((cap[both] <= torch.maximum(lo, hi)[both] + 1e-4).all())
    # oracle spot-check: a few whole y-rows of the full-size CSR, pulled to the host
    csr = geom.device_csr(dev)
    nz, ny, nx = c2["cfg"]["grid_shape"]
    data, mask = oracle.merge_masks(vol.fields["DBZH"])
    for iz, iy in ((0, 500), (3, 517), (7, 40), (12, 950), (19, 499)):
        v0 = (iz * ny + iy) * nx
        ip = csr.indptr[v0:v0 + nx + 1].cpu().numpy().astype(np.int64)
        idx = csr.gate_indices[ip[0]:ip[-1]].cpu().numpy()
        w = csr.weights[ip[0]:ip[-1]].cpu().numpy()
        want = oracle.csr_apply(ip - ip[0], idx, w, data, mask, (1, 1, nx))[0, 0]
        got = grid[iz, iy].cpu().numpy()
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=ATOL_FRAC * 75.0, equal_nan=True)
        # and the builder's rows against the brute-force oracle on a 4-voxel window of this row
        limits = c2["cfg"]["grid_limits"]
        xc = np.linspace(limits[2][0], limits[2][1], nx, dtype="float32")
        yc = np.linspace(limits[1][0], limits[1][1], ny, dtype="float32")
        zc = np.linspace(limits[0][0], limits[0][1], nz, dtype="float32")
        ix0 = 611
        sub_limits = ((float(zc[iz]), float(zc[iz])), (float(yc[iy]), float(yc[iy])), (float(xc[ix0]), float(xc[ix0 + 3])))
        o_ip, o_idx, o_w = oracle.build_geometry(vol.gate_x, vol.gate_y, vol.gate_z, (1, 1, 4), sub_limits)
        # linspace over a 4-point sub-range reproduces the full grid's float32 x values only approximately, so
        # compare neighbour COUNTS loosely and the end points exactly (they are the same float32 numbers)
        seg = ip[ix0:ix0 + 5] - ip[ix0]
        for k in (0, 3):
            s_ref = o_idx[o_ip[k]:o_ip[k + 1]]
            s_gpu = np.sort(idx[(ip[ix0 + k] - ip[0]):(ip[ix0 + k + 1] - ip[0])])
            np.testing.assert_array_equal(s_gpu, s_ref)
        assert seg[-1] > 0 or o_ip[-1] == 0


def test_c4_fused_gridder_invariants():
    """Config 4: 14x720x2000 gates -> 40x2000x2000 through the no-CSR kernel (the reference cannot represent this
    geometry)."""
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    cfg = synthetic.CONFIGS["C4"]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=4, fields=("DBZH",))
    search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"])
    dev = search.dev
    g = vol.n_total_gates
    const = torch.full((g,), -3.5, dtype=torch.float32, device=dev)
    dbz = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
    msk = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
    out = rg.roi_grid_fields_device(search, [const, dbz], [None, msk])
    filled = torch.isfinite(out[0])
    frac = float(filled.float().mean())
    assert 0.6 < frac < 0.9
    # a constant field comes back as the constant up to the rounding of two float32 sums of up to ~10^4 positive terms per lane
    # (measured worst case 2.04e-6 relative with the per-level gate lists' order, 1.9e-6 with the single list's; parity bar: 1e-5)
    assert float((out[0][filled] + 3.5).abs().max()) <= 3.5 * 4e-6
    assert bool((torch.isfinite(out[1]) <= filled).all())
    valid = dbz[(msk == 0) & torch.isfinite(dbz)]
    seen = out[1][torch.isfinite(out[1])]
    assert float(seen.min()) >= float(valid.min()) - 1e-3 and float(seen.max()) <= float(valid.max()) + 1e-3
    # 8-fold symmetry of the geometry is NOT assumed; but the centre column must see the radar's own gates
    assert bool(torch.isfinite(out[0][0, 1000, 1000]))


def test_c2_compact_csr_is_bit_identical(c2):
    """rg_csr_compact_apply_f32 (16-bit dictionary positions, LDS field window) against rg_csr_apply_f32 on the full
    config-2 geometry: the dictionaries reproduce the gate indices exactly and the grids agree bit for bit -- with the
    geometry's own window size, with a window too small for most chunks, and with no window at all (every chunk on
    the per-pair fallback); one field and the fused three-field pass of config 3.  The packed records: the tile kernel
    agrees bit for bit too, the row-wise kernel (the default) to float32 rounding."""
    from radar_processor_amd import _native
    from radar_processor_amd.grid_geometry import CompactCSR
    from radar_processor_amd.gridding import CsrGridder
    rg, torch, geom, dev = c2["rg"], c2["torch"], c2["geom"], c2["dev"]
    nz, ny, nx = c2["cfg"]["grid_shape"]
    f, m = c2["fields"]["DBZH"], c2["masks"]["DBZH"]
    g_c = CsrGridder(geom, f.numel(), 1, device=dev, compact=True, packed=False)
    g_s = CsrGridder(geom, f.numel(), 1, device=dev)
    assert g_c.compact is not None and not g_c.packed_stream and g_s.compact is None
    csr, c = g_c.csr, g_c.compact
    assert c.local_idx.numel() == csr.n_pairs and int(c.dict_ptr[-1]) == c.n_dict
    assert c.dict_ptr.numel() == CompactCSR.layout((nz, ny, nx))[2] + 1
    assert g_c.compact_bytes() < 0.82 * g_c.algorithmic_bytes()
    assert c.window_cap <= 2048 and c.fallback_fraction(c.window_cap) <= 1e-3    # 4 x 64 patches: small dictionaries
    # decode a sample of pairs: dict[dict_ptr[chunk(row)] + position] == gate index
    ip = csr.indptr.to(torch.int64)
    rows = torch.randint(0, csr.n_vox, (20000,), device=dev)
    rows = rows[(ip[rows + 1] - ip[rows]) > 0]
    pairs = ip[rows] + (torch.rand(rows.numel(), device=dev) * (ip[rows + 1] - ip[rows]).float()).long()
    pos = c.local_idx[pairs].to(torch.int64) & 0xFFFF
    decoded = c.dict[c.dict_ptr[CompactCSR.chunk_of_rows(rows, (nz, ny, nx))] + pos]
    assert bool((decoded == csr.gate_indices[pairs]).all())
    # the dictionaries hold every chunk's distinct gates exactly once (checked against torch.unique on 40 whole lines)
    r_lo, r_hi = (7 * ny + 480) * nx, (7 * ny + 520) * nx
    ipw = ip[r_lo:r_hi + 1]
    chunk_of_row = CompactCSR.chunk_of_rows(torch.arange(r_lo, r_hi, device=dev), (nz, ny, nx))
    chunk_of_pair = torch.repeat_interleave(chunk_of_row, ipw[1:] - ipw[:-1])
    keys = (chunk_of_pair << 32) | csr.gate_indices[int(ipw[0]):int(ipw[-1])].to(torch.int64)
    want_keys = torch.unique(keys)
    chunks = torch.unique(chunk_of_row)
    got_keys = torch.cat([(ch << 32) | c.dict[int(c.dict_ptr[ch]):int(c.dict_ptr[ch + 1])].to(torch.int64)
                          for ch in chunks[::7]])
    want_sub = want_keys[torch.isin(want_keys >> 32, chunks[::7])]
    assert bool(torch.equal(torch.sort(got_keys).values, want_sub))
    # grids
    g_c.pack([f], [m]); g_s.pack([f], [m])
    want = torch.empty((1, g_s.n_vox), dtype=torch.float32, device=dev)
    g_s.apply(want)
    lib = _native.load_library()

    def compact_apply(gr, out, window):
        _native.check(lib.rg_csr_compact_apply_f32(
            _native.ptr(csr.indptr), int(csr.is_i64), _native.ptr(c.local_idx), _native.ptr(csr.weights),
            _native.ptr(c.dict_ptr), _native.ptr(c.dict), gr.n_vox, csr.n_pairs, nx, ny, _native.ptr(gr.packed),
            gr.n_fields, gr.stride, gr.n_gates, float("nan"), _native.ptr(out), window, 0, _native.stream_ptr()),
            "rg_csr_compact_apply_f32")

    for cap in (c.window_cap, 512, 0):
        got = torch.full_like(want, -7.0)
        compact_apply(g_c, got, cap)
        assert bool(torch.equal(got.view(torch.int32), want.view(torch.int32))), f"window_cap={cap}"
    # and through the gridder's own apply()
    got = torch.empty_like(want)
    g_c.apply(got, fill_value=-9999.0)
    g_s.apply(want, fill_value=-9999.0)
    assert bool(torch.equal(got.view(torch.int32), want.view(torch.int32)))
    # config 3: DBZH + ZDR + RHOHV with the RHOHV >= 0.8 mask, one fused pass through the compact copy
    names = ["DBZH", "ZDR", "RHOHV"]
    qc = rg.device_gate_mask(c2["fields"]["RHOHV"], "below", 0.8)
    m_c = CsrGridder(geom, f.numel(), 3, device=dev, compact=True, packed=False)
    m_s = CsrGridder(geom, f.numel(), 3, device=dev)
    assert m_c.compact is c and not m_c.packed_stream and m_s.compact is None   # 3 fields: 21 KiB of window
    # policy: 4 fused fields (28 KiB window) run the standard kernel unless the row-wise kernel (no tiles in LDS) can
    assert CsrGridder(geom, f.numel(), 4, device=dev, compact=True, packed=False).compact is None
    four = CsrGridder(geom, f.numel(), 4, device=dev, compact=True)
    assert four.compact is c and four.packed_stream
    for gr in (m_c, m_s):
        gr.pack([c2["fields"][n] for n in names], [c2["masks"][n] for n in names], qc)
    want3 = torch.empty((3, m_s.n_vox), dtype=torch.float32, device=dev)
    got3 = torch.full_like(want3, -7.0)
    m_s.apply(want3)
    m_c.apply(got3)
    assert bool(torch.equal(got3.view(torch.int32), want3.view(torch.int32)))
    got3.fill_(-7.0)
    compact_apply(m_c, got3, 256)
    assert bool(torch.equal(got3.view(torch.int32), want3.view(torch.int32)))
    # the packed records (1 and 3 fields): tile kernel bit for bit; row-wise kernel to rounding, reproducible, and the
    # same bits with a window too small for most chunks
    scale = max(float(c2["fields"][n][torch.isfinite(c2["fields"][n])].abs().max()) for n in names)
    for gr_s, want_s in ((g_s, want), (m_s, want3)):
        g_r = CsrGridder(geom, f.numel(), gr_s.n_fields, device=dev, compact=True)
        assert g_r.compact is c and g_r.packed_stream
        g_r.packed = gr_s.packed
        gr_s.apply(want_s)
        got_r = torch.full_like(want_s, -7.0)
        g_r.tile = 384
        g_r.apply(got_r)
        assert bool(torch.equal(got_r.view(torch.int32), want_s.view(torch.int32)))
        g_r.tile = 0
        got_r.fill_(-7.0)
        g_r.apply(got_r)
        assert_same_to_rounding(got_r, want_s, scale)
        again = torch.full_like(want_s, -7.0)
        g_r.apply(again)
        assert bool(torch.equal(again.view(torch.int32), got_r.view(torch.int32)))
        g_r.window = 256
        again.fill_(-7.0)
        g_r.apply(again)
        assert bool(torch.equal(again.view(torch.int32), got_r.view(torch.int32)))


def test_c3_fused_pass_against_oracle_rows_with_the_error_tail(c2):
    """Config 3 as bench.py runs it -- DBZH + ZDR + RHOHV, the RHOHV >= 0.8 QC mask OR-ed into every field's own mask, ONE
    fused three-field row-wise pass -- against ``oracle.csr_apply`` (interpolate.py:69-104 with the merged masks of
    :59-64) on 60 whole (z,y) rows of the full config-2 grid, per field.  Beyond the pass/fail bar (same voxels filled,
    rtol 1e-5 + 1e-5 * max|field|) it MEASURES the tail the bar hides: the worst purely relative error where |want| >
    1e-3 * max|field| (asserted <= 1e-5 with no absolute term) and how many voxels pass only thanks to the floor --
    written to gpurun_out/parity_relerr_fullsize.json next to the fixture-level parity_relerr.json."""
    import json
    import os
    rg, torch, geom, dev, vol = c2["rg"], c2["torch"], c2["geom"], c2["dev"], c2["vol"]
    from radar_processor_amd.gridding import CsrGridder
    names = ["DBZH", "ZDR", "RHOHV"]
    f = [c2["fields"][n] for n in names]
    m = [c2["masks"][n] for n in names]
    qc = rg.device_gate_mask(c2["fields"]["RHOHV"], "below", 0.8)
    gr = CsrGridder(geom, f[0].numel(), 3, device=dev, compact=True)
    assert gr.compact is not None and gr.packed_stream and gr.tile == 0          # the fused row-wise pass
    gr.pack(f, m, qc)
    grid = torch.empty((3, gr.n_vox), dtype=torch.float32, device=dev)
    gr.apply(grid)
    nz, ny, nx = c2["cfg"]["grid_shape"]
    grid = grid.view(3, nz, ny, nx)
    csr = geom.device_csr(dev)
    qc_host = qc.cpu().numpy().astype(bool)
    rng = np.random.default_rng(33)
    rows = {(0, 500), (3, 499), (7, 501), (12, 500), (19, 499), (0, 0), (19, 999)}          # radar column, corners
    while len(rows) < 60:
        rows.add((int(rng.integers(0, nz)), int(rng.integers(0, ny))))
    report = {}
    for k, name in enumerate(names):
        data, own = oracle.merge_masks(vol.fields[name])
        mask = own | qc_host                                                     # interpolate.py:59-64
        scale = float(np.abs(data[np.isfinite(data) & ~mask]).max())
        rec = dict(rows=len(rows), voxels=0, significant=0, needed_floor=0, max_rel_significant=0.0, max_abs_over_scale=0.0)
        for iz, iy in sorted(rows):
            v0 = (iz * ny + iy) * nx
            ip = csr.indptr[v0:v0 + nx + 1].cpu().numpy().astype(np.int64)
            idx = csr.gate_indices[int(ip[0]):int(ip[-1])].cpu().numpy()
            w = csr.weights[int(ip[0]):int(ip[-1])].cpu().numpy()
            want = oracle.csr_apply(ip - ip[0], idx, w, data, mask, (1, 1, nx))[0, 0]
            got = grid[k, iz, iy].cpu().numpy()
            np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=ATOL_FRAC * scale, equal_nan=True)
            filled = np.isfinite(want)
            err = np.abs(got[filled].astype(np.float64) - want[filled].astype(np.float64))
            mag = np.abs(want[filled].astype(np.float64))
            sig = mag > 1e-3 * scale
            rec["voxels"] += int(filled.sum())
            rec["significant"] += int(sig.sum())
            rec["needed_floor"] += int((err > 1e-5 * mag).sum())
            if sig.any():
                rec["max_rel_significant"] = max(rec["max_rel_significant"], float((err[sig] / mag[sig]).max()))
            if err.size:
                rec["max_abs_over_scale"] = max(rec["max_abs_over_scale"], float(err.max() / scale))
        assert rec["voxels"] > 20_000 and rec["max_rel_significant"] <= 1e-5, (name, rec)
        report[name] = rec
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "parity_relerr_fullsize.json"), "w") as fh:
        json.dump({"config": "C3: fused DBZH+ZDR+RHOHV row-wise pass, RHOHV>=0.8 mask, 60 whole rows of 20x1000x1000 vs "
                             "oracle.csr_apply", "per_field": report}, fh, indent=1)
    print("parity_relerr_fullsize", json.dumps(report))


def test_c2_passes_use_the_compact_copy_from_the_first(c2):
    """gridding._use_compact: a geometry of config 2's size grids through the compact copy from its first pass on --
    single- or multi-field, row-wise kernel over the packed records -- so every pass of the geometry returns the same bits;
    they equal the standard kernel's to float32 rounding."""
    rg, torch, dev = c2["rg"], c2["torch"], c2["dev"]
    from radar_processor_amd.grid_geometry import GridGeometry
    from radar_processor_amd.gridding import CsrGridder
    geom = GridGeometry.from_device(c2["geom"].grid_shape, c2["geom"].grid_limits, c2["geom"].device_csr(dev), 17000.0)
    f, m = c2["fields"]["ZDR"], c2["masks"]["ZDR"]
    assert getattr(geom, "_compact", None) is None
    first = rg.grid_fields_device(geom, [f], [m]).clone()
    assert geom._compact is not None and geom._compact[1] is not None and geom._compact[1].rec is not None
    two_a = rg.grid_fields_device(geom, [f, c2["fields"]["DBZH"]], [m, None]).clone()
    second = rg.grid_fields_device(geom, [f], [m])
    assert torch.equal(first.view(torch.int32), second.view(torch.int32))
    assert len(geom._gridders) >= 2                                 # gridders (and staging buffers) are reused
    g_s = CsrGridder(geom, f.numel(), 2, device=dev)                # the standard kernel on the same inputs
    g_s.pack([f, c2["fields"]["DBZH"]], [m, None])
    std = torch.empty((2, g_s.n_vox), dtype=torch.float32, device=dev)
    g_s.apply(std)
    scale = float(c2["fields"]["DBZH"][torch.isfinite(c2["fields"]["DBZH"])].abs().max())
    assert_same_to_rounding(two_a.view(2, -1), std, scale)
    assert_same_to_rounding(first.view(1, -1), std[:1], scale)
    # a 1792-entry window does not hold the 40-byte entries of an eight-field pass: groups of four on this geometry
    from radar_processor_amd import batch, gridding
    assert gridding.volumes_per_pass_cap(geom, dev) == 4 and batch.VolumeBatch(geom, ["DBZH"], device=dev).volumes_per_pass == 4


def test_c2_compact_only_layout(c2, tmp_path):
    """compute_grid_geometry(layout="compact"): the int32 index array is never materialised for the whole grid; row
    pointers and weights equal the standard build bit for bit, the decoded indices equal its gate_indices, gridding
    gives the same values (to float32 rounding: its passes run the row-wise kernel); layout="auto" builds the packed
    layout alone from 50 M codable pairs on and keeps the reference's arrays below."""
    rg, torch, dev, vol, cfg = c2["rg"], c2["torch"], c2["dev"], c2["vol"], c2["cfg"]
    std = c2["geom"].device_csr(dev)
    geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"],
                                    str(tmp_path), layout="compact")
    csr = geom.device_csr(dev)
    assert csr.gate_indices is None and csr.n_pairs == std.n_pairs and csr.max_gate == std.max_gate
    assert torch.equal(csr.indptr.to(torch.int64), std.indptr.to(torch.int64))
    assert torch.equal(csr.weights.view(torch.int32), std.weights.view(torch.int32))
    compact = geom.device_compact(dev)
    assert torch.equal(compact.decode(csr), std.gate_indices)
    nx = cfg["grid_shape"][2]
    r0, r1 = 7 * nx + 13, 1234 * nx + 5          # a row range that cuts through chunks (4 lines x 64 rows)
    assert torch.equal(compact.decode(csr, r0, r1), std.gate_indices[int(std.indptr[r0]):int(std.indptr[r1])])
    f, m = c2["fields"]["DBZH"], c2["masks"]["DBZH"]
    from radar_processor_amd.gridding import CsrGridder
    scale = float(f[torch.isfinite(f)].abs().max())
    g_s = CsrGridder(c2["geom"], f.numel(), 2, device=dev)                    # the standard kernel
    g_s.pack([f, c2["fields"]["ZDR"]], [m, None])
    want2 = torch.empty((2, g_s.n_vox), dtype=torch.float32, device=dev)
    g_s.apply(want2)
    got = rg.grid_fields_device(geom, [f], [m])
    assert_same_to_rounding(got.view(1, -1), want2[:1], scale)
    got2 = rg.grid_fields_device(geom, [f, c2["fields"]["ZDR"]], [m, None])      # multi-field works on the compact copy too
    assert_same_to_rounding(got2.view(2, -1), want2, scale)
    # and bit for bit through the tile kernel of the compact copy (plain position / weight arrays)
    g_t = CsrGridder(geom, f.numel(), 2, device=dev, packed=False)
    assert g_t.compact is not None and not g_t.packed_stream
    g_t.packed = g_s.packed
    got_t = torch.empty_like(want2)
    g_t.apply(got_t)
    assert torch.equal(got_t.view(torch.int32), want2.view(torch.int32))
    auto = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"],
                                    str(tmp_path), layout="auto")
    # 0.87 G pairs of Barnes weights: 'auto' builds the packed layout alone (>= 50 M pairs, codable) ...
    a_csr = auto.device_csr(dev)
    assert a_csr.gate_indices is None and a_csr.weights is None and auto.device_compact(dev).rec is not None
    # ... from which the reference's arrays come back bit for bit on demand
    assert torch.equal(auto.device_compact(dev).decode(a_csr, 0, 5000), std.gate_indices[:int(std.indptr[5000])])
    small = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, (2, 40, 40), cfg["grid_limits"], str(tmp_path), layout="auto")
    assert small.device_csr(dev).gate_indices is not None and small.n_pairs() < 50_000_000      # small: the reference's arrays
    with pytest.raises(ValueError):
        rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"],
                                 str(tmp_path), layout="coo")


def test_c2_packed_only_layout(c2, tmp_path):
    """compute_grid_geometry(layout="packed"): only the row pointers, the dictionaries and the packed pair stream exist
    (positions + losslessly coded weights, three pairs per 16-byte record).  The decoded indices AND weights equal the
    standard build bit for bit, 1-4 fused fields grid to the same bits as the standard kernel through the tile kernel
    and to the same values (float32 rounding) through the default row-wise kernel, larger groups are split, and a
    weighting whose weights do not fit the 26-bit code (Cressman) falls back to the compact layout."""
    rg, torch, dev, vol, cfg = c2["rg"], c2["torch"], c2["dev"], c2["vol"], c2["cfg"]
    std = c2["geom"].device_csr(dev)
    geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"],
                                    str(tmp_path), layout="packed")
    csr = geom.device_csr(dev)
    assert csr.gate_indices is None and csr.weights is None and csr.n_pairs == std.n_pairs and csr.max_gate == std.max_gate
    assert torch.equal(csr.indptr.to(torch.int64), std.indptr.to(torch.int64))
    compact = geom.device_compact(dev)
    assert compact.local_idx is None and compact.rec is not None
    assert compact.rec.shape[0] * 16 < 5.45 * csr.n_pairs
    nx = cfg["grid_shape"][2]
    for r0, r1 in ((0, 40 * nx), (7 * nx + 13, 1234 * nx + 5), (csr.n_vox - 999, csr.n_vox)):
        q0, q1 = int(std.indptr[r0]), int(std.indptr[r1])
        assert torch.equal(compact.decode(csr, r0, r1), std.gate_indices[q0:q1])
        assert torch.equal(compact.decode_weights(csr, r0, r1).view(torch.int32), std.weights[q0:q1].view(torch.int32))
    names = ["DBZH", "ZDR", "RHOHV"]
    f = [c2["fields"][n] for n in names]
    m = [c2["masks"][n] for n in names]
    from radar_processor_amd.gridding import CsrGridder
    scale = max(float(t[torch.isfinite(t)].abs().max()) for t in f)
    for nf in (1, 2, 3):
        g_s = CsrGridder(c2["geom"], f[0].numel(), nf, device=dev)            # the standard kernel
        g_s.pack(f[:nf], m[:nf])
        want = torch.empty((nf, g_s.n_vox), dtype=torch.float32, device=dev)
        g_s.apply(want)
        got = rg.grid_fields_device(geom, f[:nf], m[:nf])
        assert_same_to_rounding(got.view(nf, -1), want, scale)
        g_t = CsrGridder(geom, f[0].numel(), nf, device=dev, tile=384)        # tile kernel over the same records
        assert g_t.packed_stream
        g_t.packed = g_s.packed
        got_t = torch.empty_like(want)
        g_t.apply(got_t)
        assert torch.equal(got_t.view(torch.int32), want.view(torch.int32)), nf
        del want, got, got_t
    five = rg.grid_fields_device(geom, f + f[:2], m + m[:2])                # 5 fields: a pass of 4 and a pass of 1
    one = rg.grid_fields_device(geom, f[1:2], m[1:2])
    assert torch.equal(five[4].view(torch.int32), one[0].view(torch.int32))
    four = rg.grid_fields_device(geom, f + f[:1], m + m[:1])
    assert torch.equal(five[:4].view(torch.int32), four.view(torch.int32))
    del five, one, four
    assert geom.n_pairs() == std.n_pairs and geom.memory_usage_mb() < 0.75 * c2["geom"].memory_usage_mb()
    cress = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, (2, 64, 64),
                                     ((1000.0, 2000.0), (-20e3, 20e3), (-20e3, 20e3)), str(tmp_path),
                                     weighting="cressman", layout="packed")
    assert cress.device_csr(dev).weights is not None and cress.device_csr(dev).gate_indices is None


@pytest.mark.parametrize("n_fields", [1, 3])
def test_c2_pipeline_graph_with_compact_copy(c2, n_fields):
    """VolumePipeline(compact=True), one field and config 3's three: the captured hipGraph (pack ->
    rg_csr_compact_apply_packed_f32 -> COLMAX/argmax -> CAPPI) replays to the same bits every time and to the standard
    pipeline's values (float32 rounding: the compact pass runs the row-wise kernel; the argmax level may differ where two
    levels tie to rounding)."""
    from radar_processor_amd.pipeline import VolumePipeline
    torch, geom, dev = c2["torch"], c2["geom"], c2["dev"]
    names = ["DBZH", "ZDR", "RHOHV"][:n_fields]
    f, m = c2["fields"]["DBZH"], c2["masks"]["DBZH"]
    fl, ml = [c2["fields"][n] for n in names], [c2["masks"][n] for n in names]
    outs = []
    for compact in (False, True):
        pipe = VolumePipeline(geom, f.numel(), n_fields, compact=compact, device=dev)
        assert (pipe.gridder.compact is not None) == compact and pipe.gridder.packed_stream == compact
        pipe.run(fl, ml)
        res = pipe.run(fl, ml)                      # second call replays the graph
        torch.cuda.synchronize()
        outs.append([res[k].clone() for k in ("grid", "colmax", "argmax", "cappi")])
        again = pipe.run(fl, ml)
        torch.cuda.synchronize()
        for k, t in zip(("grid", "colmax", "argmax", "cappi"), outs[-1]):
            assert torch.equal(torch.nan_to_num(again[k].float(), nan=-7e9), torch.nan_to_num(t.float(), nan=-7e9)), k
    assert len(outs[0]) == len(outs[1]) and len(outs[0]) > 0
    scale = float(f[torch.isfinite(f)].abs().max())
    for k, a, b in zip(("grid", "colmax", "argmax", "cappi"), outs[0], outs[1]):
        if k == "argmax":       # a different level only where the two levels tie to rounding
            differ = a != b
            assert float(differ.float().mean()) < 2e-3
            grid_a, cmax_a = outs[0][0], outs[0][1]
            at_b = grid_a[0].gather(0, b[0].clamp_min(0).long().unsqueeze(0))[0]
            assert float((at_b - cmax_a[0]).abs()[differ[0]].max()) <= 2e-5 * scale
        else:
            assert_same_to_rounding(b, a, scale)


# ---------------------------------------------------------------------------------------------------------------
# The METRIC workload itself (BASELINE.json's metric; bench.py's default): 12x360x1000 gates -> 40x2000x2000,
# 8.3 G CSR pairs -- the only configuration with a REAL int64 indptr and pair offsets beyond 2^31.
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def metric(tmp_path_factory):
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    rg.load_library()
    cfg = synthetic.CONFIGS["METRIC"]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"],
                                    str(tmp_path_factory.mktemp("geom_metric")))
    to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    f = to(np.ma.getdata(vol.fields["DBZH"]), torch.float32)
    m = to(np.ma.getmaskarray(vol.fields["DBZH"]), torch.uint8)
    ctx = dict(rg=rg, torch=torch, cfg=cfg, vol=vol, geom=geom, dev=dev, f=f, m=m)
    yield ctx
    ctx.clear()
    del geom, f, m
    torch.cuda.empty_cache()


def _metric_grids(metric):
    """K1 (reference-format CSR) and K1c (compact copy, row-wise kernel over the packed records: what bench.py times)
    grids of DBZH, computed once per module; also whether the tile kernel over the same records reproduced K1's bits."""
    if "k1" not in metric:
        from radar_processor_amd.gridding import CsrGridder
        torch, geom, dev, f, m = metric["torch"], metric["geom"], metric["dev"], metric["f"], metric["m"]
        g_s = CsrGridder(geom, f.numel(), 1, device=dev)
        g_c = CsrGridder(geom, f.numel(), 1, device=dev, compact=True)
        assert g_s.compact is None and g_c.compact is not None
        g_s.pack([f], [m]); g_c.pack([f], [m])
        assert g_c.packed_stream
        k1 = torch.empty((1, g_s.n_vox), dtype=torch.float32, device=dev)
        k1c = torch.full_like(k1, -7.0)
        g_s.apply(k1); g_c.apply(k1c)
        g_c.tile = 384                                 # the tile kernel over the same packed records
        k1t = torch.full_like(k1, -7.0)
        g_c.apply(k1t)
        g_c.tile = 0
        torch.cuda.synchronize()
        metric["tile_kernel_equals_k1"] = bool(torch.equal(k1.view(torch.int32), k1t.view(torch.int32)))
        del k1t
        metric["k1"], metric["k1c"], metric["g_c"] = k1, k1c, g_c
    return metric["k1"], metric["k1c"]


def test_metric_int64_csr_structure(metric):
    torch, geom, dev, cfg = metric["torch"], metric["geom"], metric["dev"], metric["cfg"]
    csr = geom.device_csr(dev)
    n_vox = int(np.prod(cfg["grid_shape"]))
    assert csr.n_vox == n_vox == 160_000_000
    assert csr.is_i64 and csr.indptr.dtype == torch.int64        # the reference's int32 indptr overflows here (F6)
    assert csr.n_pairs > 2 ** 32 and csr.n_pairs == csr.weights.numel() == csr.gate_indices.numel()
    ip = csr.indptr
    assert int(ip[0]) == 0 and int(ip[-1]) == csr.n_pairs
    lengths = ip[1:] - ip[:-1]
    assert int(lengths.min()) >= 0 and int(lengths.max()) < 4096          # monotone, sane row lengths
    first_big = int(torch.searchsorted(ip, torch.tensor([2 ** 31], device=dev, dtype=torch.int64))[0])
    assert 0 < first_big < n_vox // 2                                      # most of the grid lies beyond offset 2^31
    del lengths
    assert int(csr.gate_indices.min()) >= 0 and csr.max_gate < metric["vol"].n_total_gates
    assert int(csr.gate_indices.max()) == csr.max_gate
    w = csr.weights
    assert float(w.min()) >= np.float32(np.exp(-4.0) + 1e-5) * (1 - 1e-6) and float(w.max()) <= 1.00002


def test_metric_compact_kernels_match_reference_format(metric):
    """On the full 8.3 G-pair geometry: the tile kernel over the packed records == K1 (the reference's CSR format) bit
    for bit; K1c's row-wise kernel (what bench.py times) == K1 to float32 rounding on every voxel and the same bits run
    to run; a constant field grids to the constant; the decoded compact copy reproduces gate_indices beyond offset
    2^31."""
    torch, geom, dev = metric["torch"], metric["geom"], metric["dev"]
    k1, k1c = _metric_grids(metric)
    assert metric["tile_kernel_equals_k1"]
    f = metric["f"]
    scale = float(f[torch.isfinite(f) & (metric["m"] == 0)].abs().max())
    assert bool(torch.equal(torch.isnan(k1), torch.isnan(k1c)))
    err = (k1c - k1).abs() - 1e-5 * k1.abs()
    assert float(torch.nan_to_num(err, nan=0.0).max()) <= ATOL_FRAC * scale
    del err
    again = torch.full_like(k1c, -7.0)
    metric["g_c"].pack([f], [metric["m"]])
    metric["g_c"].apply(again)
    assert bool(torch.equal(again.view(torch.int32), k1c.view(torch.int32)))
    del again
    csr = geom.device_csr(dev)
    filled = torch.isfinite(k1[0])
    assert 0.55 < float(filled.float().mean()) < 0.8
    assert bool((filled <= ((csr.indptr[1:] - csr.indptr[:-1]) > 0)).all())     # filled => row non-empty
    g_c = metric["g_c"]
    const = torch.full_like(metric["f"], -12.5)
    g_c.pack([const], [None])
    out = torch.empty_like(k1c)
    g_c.apply(out)
    nonempty = (csr.indptr[1:] - csr.indptr[:-1]) > 0
    assert bool((torch.isfinite(out[0]) == nonempty).all())
    assert float((out[0][nonempty] + 12.5).abs().max()) <= 12.5 * 2e-6
    del out, nonempty, filled
    compact = geom.device_compact(dev)
    nz, ny, nx = metric["cfg"]["grid_shape"]
    for r0, r1 in (((25 * ny + 1000) * nx + 3, (25 * ny + 1003) * nx + 1777), (csr.n_vox - 3 * nx - 5, csr.n_vox)):
        p0, p1 = int(csr.indptr[r0]), int(csr.indptr[r1])
        assert p0 > 2 ** 31
        assert torch.equal(compact.decode(csr, r0, r1), csr.gate_indices[p0:p1])


def test_metric_compact_only_layout_with_int64_offsets(metric, tmp_path):
    """compute_grid_geometry(layout="compact") on the metric workload: the builder fills one slab of grid levels at a
    time into a scratch index buffer addressed by ABSOLUTE pair offsets (pointer shifts of up to 8.3e9 * 4 bytes), so
    this is the 64-bit path of the slab builder.  Row pointers and weights equal the standard build bit for bit, the
    decoded indices equal its gate_indices beyond offset 2^31, and gridding (row-wise kernel over records packed from
    the slab-built copy) gives the bits of the standard build's row-wise pass."""
    rg, torch, dev, vol, cfg = metric["rg"], metric["torch"], metric["dev"], metric["vol"], metric["cfg"]
    std = metric["geom"].device_csr(dev)
    _, k1c = _metric_grids(metric)
    geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"],
                                    str(tmp_path), layout="compact")
    csr = geom.device_csr(dev)
    assert csr.gate_indices is None and csr.is_i64 and csr.n_pairs == std.n_pairs and csr.max_gate == std.max_gate
    assert torch.equal(csr.indptr, std.indptr)
    assert torch.equal(csr.weights.view(torch.int32), std.weights.view(torch.int32))
    compact = geom.device_compact(dev)
    nz, ny, nx = cfg["grid_shape"]
    for r0, r1 in (((21 * ny + 7) * nx + 3, (21 * ny + 12) * nx + 1999), ((39 * ny + 1990) * nx, csr.n_vox), (0, 3 * nx)):
        p0, p1 = int(std.indptr[r0]), int(std.indptr[r1])
        assert torch.equal(compact.decode(csr, r0, r1), std.gate_indices[p0:p1])
    got = rg.grid_fields_device(geom, [metric["f"]], [metric["m"]])
    assert torch.equal(got.view(-1).view(torch.int32), k1c.view(-1).view(torch.int32))
    del got, geom, compact, csr
    torch.cuda.empty_cache()


def test_metric_packed_only_layout_with_int64_offsets(metric, tmp_path):
    """layout="packed" on the metric workload: records packed slab by slab through shifted 64-bit pointers; decoded
    indices and weights beyond pair offset 2^31 equal the standard build, the grid equals the standard build's row-wise
    pass bit for bit."""
    rg, torch, dev, vol, cfg = metric["rg"], metric["torch"], metric["dev"], metric["vol"], metric["cfg"]
    std = metric["geom"].device_csr(dev)
    _, k1c = _metric_grids(metric)
    geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"],
                                    str(tmp_path), layout="packed")
    csr = geom.device_csr(dev)
    assert csr.gate_indices is None and csr.weights is None and csr.is_i64 and csr.n_pairs == std.n_pairs
    assert torch.equal(csr.indptr, std.indptr)
    compact = geom.device_compact(dev)
    nz, ny, nx = cfg["grid_shape"]
    for r0, r1 in (((21 * ny + 7) * nx + 3, (21 * ny + 12) * nx + 1999), ((39 * ny + 1990) * nx, csr.n_vox), (0, 3 * nx)):
        p0, p1 = int(std.indptr[r0]), int(std.indptr[r1])
        assert torch.equal(compact.decode(csr, r0, r1), std.gate_indices[p0:p1])
        assert torch.equal(compact.decode_weights(csr, r0, r1).view(torch.int32), std.weights[p0:p1].view(torch.int32))
    got = rg.grid_fields_device(geom, [metric["f"]], [metric["m"]])
    assert torch.equal(got.view(-1).view(torch.int32), k1c.view(-1).view(torch.int32))
    del got, geom, compact, csr
    torch.cuda.empty_cache()


def test_metric_oracle_rows_beyond_2_31(metric):
    """oracle.csr_apply (the restatement of interpolate.py:69-104) on whole y-rows of the full-size CSR whose pair
    offsets lie beyond 2^31 -- including the first row past 2^31, the last row of the grid (last chunk) -- and the
    builder's neighbour sets there against the brute-force oracle (compute.py:46-91)."""
    torch, geom, dev, vol, cfg = metric["torch"], metric["geom"], metric["dev"], metric["vol"], metric["cfg"]
    k1, k1c = _metric_grids(metric)
    csr = geom.device_csr(dev)
    nz, ny, nx = cfg["grid_shape"]
    limits = cfg["grid_limits"]
    data, mask = oracle.merge_masks(vol.fields["DBZH"])
    ip_rows = csr.indptr[::nx].cpu().numpy()
    first = int(np.searchsorted(ip_rows, 2 ** 31))                 # first (z,y) row that starts beyond 2^31
    rows = [first, first + 1, (20 * ny + 1000), (30 * ny + 777), (37 * ny + 1500), nz * ny - 1]
    scale = float(np.abs(data[np.isfinite(data) & ~mask]).max())
    xc = np.linspace(limits[2][0], limits[2][1], nx, dtype="float32")
    yc = np.linspace(limits[1][0], limits[1][1], ny, dtype="float32")
    zc = np.linspace(limits[0][0], limits[0][1], nz, dtype="float32")
    checked_pairs = 0
    for r in rows:
        iz, iy = divmod(r, ny)
        v0 = r * nx
        ip = csr.indptr[v0:v0 + nx + 1].cpu().numpy()
        assert ip[0] >= 2 ** 31
        idx = csr.gate_indices[int(ip[0]):int(ip[-1])].cpu().numpy()
        w = csr.weights[int(ip[0]):int(ip[-1])].cpu().numpy()
        want = oracle.csr_apply(ip - ip[0], idx, w, data, mask, (1, 1, nx))[0, 0]
        for got_t in (k1, k1c):
            got = got_t[0, v0:v0 + nx].cpu().numpy()
            np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=ATOL_FRAC * scale, equal_nan=True)
        checked_pairs += int(ip[-1] - ip[0])
        for ix0 in (37, 1000, 1960):
            sub = ((float(zc[iz]), float(zc[iz])), (float(yc[iy]), float(yc[iy])), (float(xc[ix0]), float(xc[ix0 + 3])))
            o_ip, o_idx, o_w = oracle.build_geometry(vol.gate_x, vol.gate_y, vol.gate_z, (1, 1, 4), sub)
            for k in (0, 3):   # the end points of a 4-point linspace are the grid's own float32 coordinates
                s_ref = o_idx[o_ip[k]:o_ip[k + 1]]
                lo, hi = int(ip[ix0 + k] - ip[0]), int(ip[ix0 + k + 1] - ip[0])
                order = np.argsort(idx[lo:hi], kind="stable")
                np.testing.assert_array_equal(idx[lo:hi][order], s_ref)
                w_ref = o_w[o_ip[k]:o_ip[k + 1]]
                ulp = np.abs(w[lo:hi][order].view(np.int32).astype(np.int64) - w_ref.view(np.int32).astype(np.int64))
                assert ulp.max(initial=0) <= 1
    assert checked_pairs > 100_000


def test_metric_fused_gridder_and_products(metric):
    """K2 (no CSR) against K1 on the full metric grid: identical NaN pattern, values within 1e-5 * max|field|;
    COLMAX / argmax / CAPPI consistency on the 40-level grid."""
    rg, torch, geom, dev, vol, cfg = (metric[k] for k in ("rg", "torch", "geom", "dev", "vol", "cfg"))
    k1, _ = _metric_grids(metric)
    nz, ny, nx = cfg["grid_shape"]
    search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"])
    k2 = rg.roi_grid_fields_device(search, [metric["f"]], [metric["m"]])
    del search
    grid = k1.view(1, nz, ny, nx)
    assert bool((torch.isnan(grid) == torch.isnan(k2)).all())
    scale = float(torch.nan_to_num(metric["f"], nan=0.0).abs().max())
    assert float(torch.nan_to_num(grid - k2, nan=0.0).abs().max()) <= 1e-5 * scale
    del k2
    grid = grid[0]
    cmax, arg = rg.column_argmax(grid)
    has = arg >= 0
    assert bool((has == torch.isfinite(cmax)).all())
    picked = torch.gather(grid, 0, arg.clamp(min=0).long().unsqueeze(0))[0]
    assert bool((picked[has] == cmax[has]).all())
    for iz in range(nz):                                # level by level: no 160 M-element temporaries
        lvl = grid[iz]
        fin = torch.isfinite(lvl)
        assert bool((lvl[fin] <= cmax[fin]).all())
        assert not bool(((lvl == cmax) & (arg > iz)).any())          # first-index tie rule
    cap = rg.constant_altitude_ppi(grid, geom, 4000.0)               # z 0..15 km over 40 levels: 4000 m is a lerp
    plan = oracle.cappi_plan(cfg["grid_limits"][0], nz, 4000.0)
    assert plan[0] == "lerp"
    lo, hi = grid[plan[1]], grid[plan[1] + 1]
    both = torch.isfinite(lo) & torch.isfinite(hi)
    assert bool((torch.isfinite(cap) == both).all())
    for iy in (0, 1000, 1999):                                       # oracle on whole y-rows, bit for bit
        rows = grid[:, iy, :].cpu().numpy()[:, None, :]
        np.testing.assert_array_equal(cap[iy].cpu().numpy(), oracle.cappi(rows, cfg["grid_limits"][0], 4000.0)[0])
        np.testing.assert_array_equal(cmax[iy].cpu().numpy(), oracle.column_max(rows, 0, nz - 1)[0])
        np.testing.assert_array_equal(arg[iy].cpu().numpy(), oracle.column_argmax(rows, 0, nz - 1)[0])
    assert bool((cap[both] >= torch.minimum(lo, hi)[both] - 1e-4).all())
    assert bool((cap[both] <= torch.maximum(lo, hi)[both] + 1e-4).all())


def test_metric_column_mode_grid_and_products_only(metric):
    """rg_csr_compact_apply_columns_f32 on the metric workload (640 000 chunks, int64 row pointers, offsets past 2^31): with
    the 3-D store its grid is the row-wise kernel's, bit for bit, on every voxel; products only -- nothing but planes is
    written -- COLMAX / first argmax over an altitude window and the two levels of the 4000 m CAPPI equal
    rg_column_reduce_f32 / the grid's own levels on the whole 2000 x 2000 plane, and the oracle (products.py:361-412,
    462-490 restated) on whole rows of the 40 x 2000 x 2000 grid; the fused API returns the same planes."""
    from radar_processor_amd.gridding import CsrGridder
    rg, torch, geom, dev, cfg = (metric[k] for k in ("rg", "torch", "geom", "dev", "cfg"))
    _, k1c = _metric_grids(metric)
    nz, ny, nx = cfg["grid_shape"]
    f, m = metric["f"], metric["m"]
    g = CsrGridder(geom, f.numel(), 1, device=dev, compact=True)
    assert g.has_columns_kernel
    g.pack([f], [m])
    col = torch.full_like(k1c, -7.0)
    g.apply_columns(out=col)
    assert bool(torch.equal(col.view(torch.int32), k1c.view(torch.int32)))
    del col
    grid = k1c.view(nz, ny, nx)
    plan = oracle.cappi_plan(cfg["grid_limits"][0], nz, 4000.0)
    assert plan[0] == "lerp"
    lo, hi = oracle.column_range(nz, z_min_alt=1000.0, z_max_alt=12000.0, z_limits=cfg["grid_limits"][0])
    cmax = torch.full((1, ny, nx), -7.0, dtype=torch.float32, device=dev)
    carg = torch.full((1, ny, nx), -7, dtype=torch.int32, device=dev)
    planes = torch.full((1, 2, ny, nx), -7.0, dtype=torch.float32, device=dev)
    for pieces in (1, 3):
        g.apply_columns(out=None, level_planes=planes, keep_lo=plan[1], col_max=cmax, col_arg=carg, col_window=(lo, hi),
                        z_pieces=pieces)
        want_max, want_arg = rg.column_argmax(grid, z_min_idx=lo, z_max_idx=hi)
        assert bool(torch.equal(cmax[0].view(torch.int32), want_max.view(torch.int32))) and bool(torch.equal(carg[0], want_arg))
        assert bool(torch.equal(planes[0].view(torch.int32), grid[plan[1]:plan[1] + 2].contiguous().view(torch.int32)))
    for iy in (0, 1000, 1999):                                       # the oracle on whole y-rows, bit for bit
        rows = grid[:, iy, :].cpu().numpy()[:, None, :]
        np.testing.assert_array_equal(cmax[0, iy].cpu().numpy(), oracle.column_max(rows, lo, hi)[0])
        np.testing.assert_array_equal(carg[0, iy].cpu().numpy(), oracle.column_argmax(rows, lo, hi)[0])
    rec = rg.grid_products_device(geom, [f], [m], products=rg.PlaneProducts(cappi=(4000.0,), z_min_alt=1000.0, z_max_alt=12000.0),
                                  fused=True)[0]
    assert bool(torch.equal(rec["colmax"].view(torch.int32), cmax[0].view(torch.int32))) and bool(torch.equal(rec["argmax"], carg[0]))
    cap = rg.constant_altitude_ppi(grid, geom, 4000.0)
    assert bool(torch.equal(rec["cappi"][4000.0].view(torch.int32), cap.view(torch.int32)))
    for iy in (3, 1234):
        rows = grid[:, iy, :].cpu().numpy()[:, None, :]
        np.testing.assert_array_equal(rec["cappi"][4000.0][iy].cpu().numpy(), oracle.cappi(rows, cfg["grid_limits"][0], 4000.0)[0])


def test_c2_settling_the_record_placement_changes_nothing_but_where_the_records_live(c2):
    """CsrGridder.settle_records: the records copied into further allocations, each probed with the gridder's own kernel, the
    fastest kept, the others freed -- the same records, hence the same grid bit for bit; the report lists one probe per try."""
    from radar_processor_amd.gridding import CsrGridder
    rg, torch, geom, dev = c2["rg"], c2["torch"], c2["geom"], c2["dev"]
    f, m = c2["fields"]["DBZH"], c2["masks"]["DBZH"]
    g = CsrGridder(geom, f.numel(), 1, device=dev, compact=True)
    assert g.has_columns_kernel
    g.pack([f], [m])
    before = torch.empty((1, g.n_vox), dtype=torch.float32, device=dev)
    g.apply(before)
    rec0 = g.compact.rec
    snapshot = rec0.clone()
    report = g.settle_records(tries=3)
    assert report is not None and report["tries"] == 3 and len(report["probe_ms"]) == 3 and 0 <= report["kept"] < 3
    assert all(0.3 < t < 10 for t in report["probe_ms"])
    assert torch.equal(g.compact.rec, snapshot) and (report["kept"] == 0) == (g.compact.rec.data_ptr() == rec0.data_ptr())
    after = torch.full_like(before, -7.0)
    g.apply(after)
    assert torch.equal(after.view(torch.int32), before.view(torch.int32))
    assert g.settle_records(tries=1) is None


def test_metric_eight_volumes_in_one_pass(metric):
    """Eight field-volumes -- eight seeded DBZH volumes of a batch, what batch.VolumeBatch fuses on this geometry -- in ONE
    row-wise pass over the records (5-8 fields: byte-mask window entries, one chain for the weight sums): against two passes
    of four on every voxel (the same products, another add order of the weights: float32 rounding) and against
    ``oracle.csr_apply`` (interpolate.py:69-104) on whole rows; the pass size is the geometry's choice (8 here: a 768-entry
    window holds 40-byte entries; config 2's 1792-entry window does not: 4)."""
    from radar_processor_amd import batch, gridding, synthetic
    from radar_processor_amd.gridding import CsrGridder
    rg, torch, geom, dev, cfg = (metric[k] for k in ("rg", "torch", "geom", "dev", "cfg"))
    vols = [metric["vol"]] + [synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=s, fields=("DBZH",))
                             for s in range(1, 8)]
    host = [oracle.merge_masks(v.fields["DBZH"]) for v in vols]
    f = [torch.from_numpy(np.ascontiguousarray(d)).to(dev) for d, _ in host]
    m = [torch.from_numpy(k.astype(np.uint8)).to(dev) for _, k in host]
    assert gridding.volumes_per_pass_cap(geom, dev) == 8 and batch.VolumeBatch(geom, ["DBZH"], device=dev).volumes_per_pass == 8
    g8 = CsrGridder(geom, f[0].numel(), 8, device=dev, compact=True)
    assert g8.packed_stream and g8.tile == 0 and g8.window == 768
    g8.pack(f, m)
    out8 = torch.full((8, g8.n_vox), -7.0, dtype=torch.float32, device=dev)
    g8.apply(out8)
    g4 = CsrGridder(geom, f[0].numel(), 4, device=dev, compact=True)
    out4 = torch.empty((4, g8.n_vox), dtype=torch.float32, device=dev)
    for lo in (0, 4):
        g4.pack(f[lo:lo + 4], m[lo:lo + 4])
        g4.apply(out4)
        for k in range(4):
            d, msk = host[lo + k]
            assert_same_to_rounding(out8[lo + k], out4[k], scale=float(np.abs(d[np.isfinite(d) & ~msk]).max()))
    del out4
    nz, ny, nx = cfg["grid_shape"]
    csr = geom.device_csr(dev)
    for r in (0, 20 * ny + 1000, 33 * ny + 17, nz * ny - 1):
        v0 = r * nx
        ip = csr.indptr[v0:v0 + nx + 1].cpu().numpy().astype(np.int64)
        idx = csr.gate_indices[int(ip[0]):int(ip[-1])].cpu().numpy()
        w = csr.weights[int(ip[0]):int(ip[-1])].cpu().numpy()
        for k in (0, 3, 4, 7):
            d, msk = host[k]
            want = oracle.csr_apply(ip - ip[0], idx, w, d, msk, (1, 1, nx))[0, 0]
            assert_same_to_rounding(out8[k, v0:v0 + nx], want, scale=float(np.abs(d[np.isfinite(d) & ~msk]).max()))
    # the batch driver takes the same pass: 8 volumes -> one launch, the same bits
    vb = batch.VolumeBatch(geom, ["DBZH"], device=dev)
    events = []
    got = vb.grid_shard([{"DBZH": (f[b], m[b])} for b in range(8)], events=events)
    assert len(events) == 1
    for b in (0, 5, 7):
        assert torch.equal(got[b][0].reshape(-1).view(torch.int32), out8[b].view(torch.int32))
