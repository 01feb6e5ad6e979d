"""The column mode of the row-wise kernel (rg_csr_compact_apply_columns_f32, csrc/rg_csr_columns.hip) through the C ABI:

 * its 3-D grids are the same BITS as the one-chunk-per-workgroup row-wise kernel's (and therefore as
   oracle.csr_apply_rowwise_order's) -- on the reference's CSRs (tests/golden), on hand-made CSRs with ragged shapes, empty
   and over-long rows, windows that do not fit (per-pair path) and split chunks, for 1-4 fused fields, any number of level
   pieces, both record orders, the heaviest-first and the plain workgroup order;
 * its products-only mode -- no 3-D store -- returns the planes the separate kernels compute from the stored grid, bit for
   bit: COLMAX / first argmax (radar_grid/products.py:462-490 on the grid of interpolate.py:69-104; np.nanargmax semantics
   for the index) against rg_column_reduce_f32 AND against oracle.column_max / column_argmax of the oracle's
   order-exact grid; kept levels and the CAPPI blended from them (products.py:361-412) against the grid's own levels and
   constant_altitude_ppi; and against the reference's own product fixtures to the float32 tolerance of the grids.
"""
import numpy as np
import pytest

from conftest import ATOL_FRAC, golden_names, load_golden, reference_indices, volume_for
from oracle import radar_grid_oracle as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rg():
    import radar_processor_amd as pkg
    pkg.load_library()
    return pkg


def _gridder(geom, compact, n_gates, nf, dev, window=None):
    from radar_processor_amd.gridding import CsrGridder
    g = CsrGridder(geom, n_gates, nf, device=dev)
    g.compact, g.window, g.packed_stream = compact, (compact.window_for(nf) if window is None else window), True
    return g


def _same_bits(a, b):
    import torch
    return torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))


@pytest.mark.parametrize("name", golden_names("g2_") + golden_names("g3_") + golden_names("g6_"))
def test_columns_kernel_on_the_reference_geometries(rg, name):
    """HIP fed the REFERENCE's CSR: grids bit for bit with the row-wise kernel and with the oracle's restatement of its
    order, within tolerance of the reference's grids; products bit for bit with the separate kernels and the oracle."""
    import torch
    from radar_processor_amd.grid_geometry import GridGeometry
    meta, ref = load_golden(name)
    if meta["weighting"] == "cressman":
        pytest.skip("weights down to 0 are not codable: no packed records")
    vol = volume_for(meta)
    dev = torch.device("cuda", 0)
    shape = tuple(meta["grid_shape"])
    limits = tuple(tuple(float(x) for x in lim) for lim in meta["grid_limits"])
    gidx = reference_indices(name, meta, ref)
    geom = GridGeometry(shape, limits, ref["indptr"], gidx, ref["weights"], toa=meta["toa"])
    if geom.device_csr(dev).n_pairs == 0:
        pytest.skip("fixture without pairs")
    compact = geom.device_compact(dev)
    assert compact is not None and compact.ensure_packed(geom.device_csr(dev))
    names = list(meta["fields"])
    data_mask = [oracle.merge_masks(vol.fields[f]) for f in names]
    f_t = [torch.from_numpy(np.ascontiguousarray(d)).to(dev) for d, _ in data_mask]
    m_t = [torch.from_numpy(m.astype(np.uint8)).to(dev) for _, m in data_mask]
    nz, ny, nx = shape
    n_vox = nz * ny * nx
    groups = [[0]] + ([list(range(len(names)))] if len(names) > 1 else [])
    for group in groups:
        nf = len(group)
        g = _gridder(geom, compact, f_t[0].numel(), nf, dev)
        g.pack([f_t[i] for i in group], [m_t[i] for i in group])
        row = torch.empty((nf, n_vox), dtype=torch.float32, device=dev)
        g.apply(row)
        emu = oracle.csr_apply_rowwise_order(ref["indptr"], gidx, ref["weights"], [data_mask[i][0] for i in group],
                                             [data_mask[i][1] for i in group], shape).reshape(nf, nz, ny, nx)
        want_max = np.stack([oracle.column_max(emu[k], 0, nz - 1) for k in range(nf)])
        want_arg = np.stack([oracle.column_argmax(emu[k], 0, nz - 1) for k in range(nf)])
        for pieces in sorted({1, min(2, nz), nz}):
            for ordered in (True, False):
                col = torch.full_like(row, -5.0)
                cmax = torch.full((nf, ny, nx), -7.0, dtype=torch.float32, device=dev)
                carg = torch.full((nf, ny, nx), -7, dtype=torch.int32, device=dev)
                g.apply_columns(out=col, col_max=cmax, col_arg=carg, z_pieces=pieces, ordered=ordered)
                assert _same_bits(col, row), (name, nf, pieces, ordered)
                got_np = col.cpu().numpy().reshape(nf, nz, ny, nx)
                live = ~np.isnan(emu)
                np.testing.assert_array_equal(np.isnan(got_np), ~live)
                assert np.array_equal(got_np.view(np.int32)[live], emu.view(np.int32)[live])
                # products, bit for bit: the separate kernel on the stored grid, and the oracle on the order-exact grid
                for k in range(nf):
                    k3_max, k3_arg = rg.column_argmax(row[k].view(nz, ny, nx))
                    assert _same_bits(cmax[k], k3_max) and torch.equal(carg[k], k3_arg), (name, nf, pieces, k)
                    got_max = cmax[k].cpu().numpy()
                    np.testing.assert_array_equal(np.isnan(got_max), np.isnan(want_max[k]))
                    ok = ~np.isnan(want_max[k])
                    assert np.array_equal(got_max.view(np.int32)[ok], want_max[k].view(np.int32)[ok])
                    np.testing.assert_array_equal(carg[k].cpu().numpy(), want_arg[k])
        # products only (no 3-D store): the same planes; a level window; kept levels
        lo, hi = (1, nz - 2) if nz >= 3 else (0, nz - 1)
        keep_lo, n_keep = (nz // 2 - 1, 2) if nz >= 2 else (0, 1)
        keep_lo = max(keep_lo, 0)
        cmax = torch.full((nf, ny, nx), -7.0, dtype=torch.float32, device=dev)
        carg = torch.full((nf, ny, nx), -7, dtype=torch.int32, device=dev)
        planes = torch.full((nf, n_keep, ny, nx), -7.0, dtype=torch.float32, device=dev)
        g.apply_columns(out=None, level_planes=planes, keep_lo=keep_lo, col_max=cmax, col_arg=carg, col_window=(lo, hi))
        for k in range(nf):
            grid = row[k].view(nz, ny, nx)
            k3_max, k3_arg = rg.column_argmax(grid, z_min_idx=lo, z_max_idx=hi)
            assert _same_bits(cmax[k], k3_max) and torch.equal(carg[k], k3_arg)
            assert _same_bits(planes[k], grid[keep_lo:keep_lo + n_keep])
        # the reference's own grids: within the float32 tolerance of every gridding comparison
        for k, i in enumerate(group):
            key = f"grid_{names[i]}"
            if key in ref:
                data, mask = data_mask[i]
                good = np.isfinite(data) & ~mask
                atol = ATOL_FRAC * float(np.abs(data[good]).max()) if good.any() else 0.0
                got = col[k].cpu().numpy().reshape(shape)
                np.testing.assert_array_equal(np.isnan(got), np.isnan(ref[key]))
                np.testing.assert_allclose(got, ref[key], rtol=1e-5, atol=atol, equal_nan=True)


@pytest.mark.parametrize("name", golden_names("g5_"))
def test_fused_products_bit_for_bit_with_the_reference_product_fixtures(rg, name):
    """g5: CAPPI / COLMAX planes the REFERENCE computed (radar_grid/products.py) from a dense 3-D grid with all-NaN columns,
    ties and -0.0.  An identity geometry -- voxel v's only neighbour is gate v, weight 1.0 -- grids the fixture's own
    values back exactly ((1.0 * v) / 1.0 == v; NaN voxels are masked gates), so the products-only pass must return the
    reference's planes BIT FOR BIT although no 3-D grid is ever stored: column maximum over all levels, over an altitude
    window and over an index window, the first-argmax contract, linear / nearest / out-of-range CAPPI."""
    import torch
    meta, ref = load_golden(name)
    grid = ref["grid"]
    nz, ny, nx = grid.shape
    limits = (tuple(meta["z_limits"]), (-1e4, 1e4), (-1.4e4, 1.4e4))
    n = grid.size
    geom = rg.GridGeometry(grid.shape, limits, np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32),
                           np.ones(n, dtype=np.float32), toa=17000.0)
    dev = torch.device("cuda", 0)
    compact = geom.device_compact(dev)                 # cached: grid_products_device then runs through the packed records
    assert compact is not None and compact.ensure_packed(geom.device_csr(dev))
    flat = grid.reshape(-1)
    f_t = torch.from_numpy(np.nan_to_num(flat, nan=123.0)).to(dev)
    m_t = torch.from_numpy(np.isnan(flat).astype(np.uint8)).to(dev)
    eq = np.testing.assert_array_equal

    def run(**kw):
        recs = rg.grid_products_device(geom, [f_t, f_t], [m_t, m_t], products=rg.PlaneProducts(**kw), fused=True)
        assert len(recs) == 2
        for key in recs[0]:                            # two fused fields: the same planes twice
            if key == "cappi":
                for alt in recs[0][key]:
                    assert _same_bits(recs[0][key][alt], recs[1][key][alt])
            else:
                assert _same_bits(recs[0][key], recs[1][key])          # (NaN == NaN bitwise; torch.equal would say no)
        return recs[0]
    rec = run(cappi=(4000.0, 2500.0, 99000.0))
    eq(rec["colmax"].cpu().numpy(), ref["P_colmax"])
    eq(rec["argmax"].cpu().numpy(), oracle.column_argmax(grid, 0, nz - 1))
    eq(rec["cappi"][4000.0].cpu().numpy(), ref["P_cappi4000_linear"])
    eq(rec["cappi"][2500.0].cpu().numpy(), ref["P_cappi2500_linear"])
    eq(rec["cappi"][99000.0].cpu().numpy(), ref["P_cappi_above"])
    eq(run(cappi=(4000.0,), interpolation="nearest", colmax=False, argmax=False)["cappi"][4000.0].cpu().numpy(),
       ref["P_cappi4000_nearest"])
    eq(run(z_min_alt=1000, z_max_alt=8000, argmax=False)["colmax"].cpu().numpy(), ref["P_colmax_alt"])
    rec = run(z_min_idx=2, z_max_idx=6)
    eq(rec["colmax"].cpu().numpy(), ref["P_colmax_idx"])
    eq(rec["argmax"].cpu().numpy(), oracle.column_argmax(grid, 2, 6))
    # and the unfused route of the same API (grid stored, separate kernels): the same planes
    plain = rg.grid_products_device(geom, [f_t], [m_t], products=rg.PlaneProducts(cappi=(4000.0,)), fused=False)[0]
    eq(plain["colmax"].cpu().numpy(), ref["P_colmax"])
    eq(plain["cappi"][4000.0].cpu().numpy(), ref["P_cappi4000_linear"])


@pytest.mark.parametrize("seed", range(12))
def test_columns_kernel_fuzz(rg, seed):
    """Random hand-made CSRs (the generator of test_compact_and_packed_kernels_fuzz): several planes, lines that are not
    multiples of 64 rows, planes that are not multiples of 4 lines, empty rows, rows of a thousand pairs, tiny or huge
    dictionaries, int32 / int64 row pointers; 1-4 fields with masks, unmasked NaN / Inf, a finite fill value; windows
    that fit and windows that do not (per-pair path); every number of level pieces; both record orders; lane-split hints.
    Always the same bits as the row-wise kernel -- grid and products."""
    import torch
    from radar_processor_amd import _native
    from radar_processor_amd.grid_geometry import CompactCSR, DeviceCSR, GridGeometry
    dev = torch.device("cuda")
    rng = np.random.default_rng(7000 + seed)
    nz, ny, nx = int(rng.integers(1, 7)), int(rng.integers(1, 11)), int(rng.integers(1, 300))
    n_vox = nz * ny * nx
    n_gates = int(rng.integers(50, 100_000))
    lengths = rng.integers(0, int(rng.choice([3, 40, 130])), size=n_vox)
    lengths[rng.random(n_vox) < rng.choice([0.0, 0.3, 0.9])] = 0
    for r in rng.integers(0, n_vox, size=3):
        lengths[r] = int(rng.integers(400, 1500))
    indptr = np.zeros(n_vox + 1, dtype=np.int64)
    np.cumsum(lengths, out=indptr[1:])
    n_pairs = int(indptr[-1])
    if n_pairs == 0:
        pytest.skip("empty case")
    spread = int(rng.choice([30, 2000, n_gates]))
    base = rng.integers(0, n_gates, size=n_vox)
    gidx = ((base[np.repeat(np.arange(n_vox), lengths)] + rng.integers(0, spread, size=n_pairs)) % n_gates).astype(np.int32)
    wts = np.exp(-4.0 * rng.random(n_pairs)).astype(np.float32) + np.float32(1e-5)
    ip_t = torch.from_numpy(indptr if seed % 2 else indptr.astype(np.int32)).to(dev)
    csr = DeviceCSR(ip_t, torch.from_numpy(gidx).to(dev), torch.from_numpy(wts).to(dev), int(gidx.max()))
    shape = (nz, ny, nx)
    geom = GridGeometry.from_device(shape, ((0.0, 1.0),) * 3, csr, 17000.0)
    compact = geom.device_compact(dev)
    assert compact is not None and compact.ensure_packed(csr)
    other = CompactCSR(compact.local_idx, compact.dict_ptr, compact.dict, compact.max_dict, compact.window_cap,
                       compact.grid_shape, compact.chunk_pairs, compact.chunk_counts)
    other.rec_order = _native.RG_REC_ORDER_SEGMENT
    assert other.ensure_packed(csr)
    fields = [torch.from_numpy(rng.normal(10, 20, n_gates).astype(np.float32)).to(dev) for _ in range(4)]
    masks = [torch.from_numpy((rng.random(n_gates) < 0.2).astype(np.uint8)).to(dev) if k % 2 == 0 else None for k in range(4)]
    fields[1][::7] = float("nan")
    fields[1][3::11] = float("inf")
    fields[2][5::13] = float("-inf")
    fill = -3.0 if seed % 3 == 0 else float("nan")
    for nf in (1, 2, 3, 4):
        g = _gridder(geom, compact, n_gates, nf, dev)
        g.pack(fields[:nf], masks[:nf])
        row = torch.empty((nf, n_vox), dtype=torch.float32, device=dev)
        g.apply(row, fill_value=fill)
        k3 = [rg.column_argmax(row[k].view(shape)) for k in range(nf)]
        for window in (None, 0):
            g.window = compact.window_for(nf) if window is None else window
            for pieces in sorted({1, min(3, nz), nz}):
                col = torch.full_like(row, 9.0)
                cmax = torch.full((nf, ny, nx), 9.0, dtype=torch.float32, device=dev)
                carg = torch.full((nf, ny, nx), 9, dtype=torch.int32, device=dev)
                g.apply_columns(out=col, fill_value=fill, col_max=cmax, col_arg=carg, z_pieces=pieces, ordered=bool(pieces % 2))
                assert _same_bits(col, row), ("grid", nf, window, pieces, shape)
                for k in range(nf):
                    assert _same_bits(cmax[k], k3[k][0]) and torch.equal(carg[k], k3[k][1]), ("colmax", nf, k, pieces, shape)
        g.window = compact.window_for(nf)
        # products only, kept levels
        keep_lo, n_keep = nz // 2, min(2, nz - nz // 2)
        planes = torch.full((nf, n_keep, ny, nx), 9.0, dtype=torch.float32, device=dev)
        cmax = torch.full((nf, ny, nx), 9.0, dtype=torch.float32, device=dev)
        g.apply_columns(out=None, fill_value=fill, level_planes=planes, keep_lo=keep_lo, col_max=cmax, z_pieces=min(2, nz))
        for k in range(nf):
            assert _same_bits(planes[k], row[k].view(shape)[keep_lo:keep_lo + n_keep]) and _same_bits(cmax[k], k3[k][0])
        # the other record order; diagnostic lane splits (each is another order of the adds: compare like with like)
        g_o = _gridder(geom, other, n_gates, nf, dev)
        g_o.packed = g.packed
        col = torch.full_like(row, 9.0)
        g_o.apply_columns(out=col, fill_value=fill)
        assert _same_bits(col, row), ("segment order", nf, shape)
        if nf in (1, 3) and seed < 4:
            for hint in (1, 8, 64, 71, 99):
                g.tile = 2000 + hint
                g.apply(row, fill_value=fill)
                g.apply_columns(out=col, fill_value=fill, lanes_hint=hint)
                assert _same_bits(col, row), ("lanes hint", hint, nf, shape)
            g.tile = 0


def test_columns_kernel_split_chunk_and_argument_checks(rg):
    """A chunk with more than 65536 distinct gates (one dictionary per wavefront behind a header) takes the per-pair path in
    every level; bad arguments are refused before anything is launched."""
    import torch
    from radar_processor_amd import _native
    from radar_processor_amd.grid_geometry import CompactCSR, DeviceCSR, GridGeometry
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(5)
    rich = torch.cat([torch.randperm(256 * 300, device=dev, generator=gen).to(torch.int32) for _ in range(2)]).contiguous()
    ip = torch.arange(0, 2 * 256 * 300 + 1, 300, device=dev, dtype=torch.int64)
    wts = torch.exp(-4.0 * torch.rand(rich.numel(), device=dev, generator=gen)) + 1e-5
    csr = DeviceCSR(ip.to(torch.int32), rich, wts.float(), int(rich.max()))
    shape = (2, 4, 64)
    split = CompactCSR.build(csr, shape)
    assert split is not None and split.max_dict > 65536 and split.ensure_packed(csr)
    geom = GridGeometry.from_device(shape, ((0.0, 1.0),) * 3, csr, 17000.0, compact=split)
    vals = torch.randn(76800, device=dev, generator=gen)
    mask = (torch.rand(76800, device=dev, generator=gen) < 0.3).to(torch.uint8)
    for nf in (1, 3):
        g = _gridder(geom, split, 76800, nf, dev)
        g.pack([vals, vals * 2.0, -vals][:nf], [mask] * nf)
        row = torch.empty((nf, 512), dtype=torch.float32, device=dev)
        col = torch.full_like(row, 5.0)
        g.apply(row)
        for pieces in (1, 2):
            g.apply_columns(out=col, z_pieces=pieces)
            assert _same_bits(col, row), (nf, pieces)
    lib = rg.load_library()
    c = split
    P = _native.ptr

    def call(**kw):
        a = dict(indptr=P(csr.indptr), is64=0, rec=P(c.rec), rec_ptr=P(c.rec_ptr), order=c.rec_order, w_base=c.w_base,
                 dict_ptr=P(c.dict_ptr), dict=P(c.dict), n_vox=512, n_pairs=csr.n_pairs, nx=64, ny=4, packed=P(g.packed), nf=3,
                 stride=4, n_gates=76800, fill=0.0, out=P(col), planes=0, keep_lo=0, n_keep=0, cmax=0, carg=0, lo=0, hi=1,
                 window=256, pieces=1, wg_order=0, ws=0, ws_bytes=0, hint=0)
        a.update(kw)
        return lib.rg_csr_compact_apply_columns_f32(*a.values(), 0)
    assert call() == _native.RG_OK
    assert call(out=0) == _native.RG_EINVAL                              # nothing to produce
    assert call(pieces=3) == _native.RG_EINVAL and call(pieces=0) == _native.RG_EINVAL
    assert call(planes=P(col), keep_lo=1, n_keep=2) == _native.RG_EINVAL  # kept levels outside the grid
    assert call(carg=P(col)) == _native.RG_EINVAL                        # arg without max
    assert call(cmax=P(col), lo=1, hi=0) == _native.RG_EINVAL
    assert call(cmax=P(col), pieces=2) == _native.RG_EWORKSPACE          # level pieces need the workspace
    assert call(hint=3) == _native.RG_EINVAL and call(nf=5, stride=8) == _native.RG_EUNSUPPORTED
    assert call(w_base=1) == _native.RG_EINVAL and call(order=7) == _native.RG_EINVAL
    assert lib.rg_csr_columns_workspace_bytes(4, 64, 3, 2) == 2 * 3 * 256 * 8 and lib.rg_csr_columns_workspace_bytes(4, 64, 3, 1) == 0
    torch.cuda.synchronize()
