"""CPU oracle for the radar_grid hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy restatement of the reference's algorithm for the path named by BASELINE.json
(`radar_grid` geometry build -> CSR apply -> CAPPI / COLMAX).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
package (``radar_processor_amd``) never does and fails loudly when its HIP library is missing.

Pinning: every function here is checked in ``tests/test_oracle_golden.py`` against
 (1) the literal known answers of the reference's own tests
     (``tests/test_radar_grid_interpolate.py:75-93,116-153,218-317``,
      ``tests/test_radar_grid_products.py:190-226,299-357``), and
 (2) golden vectors produced by importing the reference's modules in the build container
     (``tests/golden/make_golden.py``; NumPy/SciPy versions are recorded inside each fixture because the
     reference's float32-vs-float64 intermediates depend on NEP-50).

Each function cites the reference lines it restates (paths relative to /root/reference/).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np

EARTH_RADIUS = 6371000.0            # src/radar_grid/products.py:19
EFFECTIVE_RADIUS_FACTOR = 4.0 / 3.0  # src/radar_grid/products.py:20


# --------------------------------------------------------------------------------------------------
# grid coordinates
# --------------------------------------------------------------------------------------------------
def axis_coords_f32(lo: float, hi: float, n: int) -> np.ndarray:
    """Voxel-centre coordinates along one axis, exactly as src/radar_grid/compute.py:184-186:
    float64 ``lo + i*step`` (NumPy's linspace, endpoint pinned to ``hi``) rounded once to float32."""
    return np.linspace(lo, hi, n, dtype="float32")


# --------------------------------------------------------------------------------------------------
# geometry builder (src/radar_grid/compute.py:18-103 and :179-193)
# --------------------------------------------------------------------------------------------------
def gate_validity(gate_z: np.ndarray, radar_altitude: float, toa: float) -> Tuple[np.ndarray, np.ndarray]:
    """compute.py:182 and :193 -- ``gate_z - radar_altitude`` is evaluated in the dtype of ``gate_z``
    (float32 with a weak Python scalar under NumPy >= 2) and compared ``<= toa``."""
    z_rel = gate_z - radar_altitude
    return z_rel, z_rel <= toa


def build_geometry(gate_x, gate_y, gate_z, grid_shape, grid_limits, radar_altitude=0.0,
                   min_radius=250.0, beam_factor=0.01746, weighting="barnes2", toa=17000.0,
                   chunk_pairs: int = 8_000_000):
    """Brute-force restatement of ``compute_grid_geometry`` (compute.py:106-284).

    Membership is the reference's: gate valid (``z_rel <= toa``) and ``d2 < r2`` with everything in
    float64 computed from float32-rounded gate and voxel coordinates, ``d2 = dx*dx + dy*dy + dz*dz``
    (compute.py:69-74), ``r = max(min_radius, sqrt(x^2+y^2+z^2) * beam_factor)`` (compute.py:46-47).
    The cKDTree closed-ball query (compute.py:60) is a superset of the strict test, so it does not
    change the set.  Weights follow compute.py:82-87.

    Rows are returned sorted by ascending gate index (the reference's order is KD-tree traversal order;
    tests canonicalise both before comparing).  Returns ``(indptr int64[V+1], gate_indices int32[P],
    weights float32[P])``.

    Intended for small grids / windows: gates are pre-filtered to the grid's bounding box grown by the
    largest ROI, then tested densely in chunks.
    """
    if weighting not in ("barnes2", "cressman", "nearest"):
        raise ValueError(f"Unknown weighting function: {weighting}")
    nz, ny, nx = grid_shape
    zc = axis_coords_f32(grid_limits[0][0], grid_limits[0][1], nz).astype(np.float64)
    yc = axis_coords_f32(grid_limits[1][0], grid_limits[1][1], ny).astype(np.float64)
    xc = axis_coords_f32(grid_limits[2][0], grid_limits[2][1], nx).astype(np.float64)

    z_rel, valid = gate_validity(np.asarray(gate_z), radar_altitude, toa)
    gx = np.asarray(gate_x).astype(np.float64)
    gy = np.asarray(gate_y).astype(np.float64)
    gz = np.asarray(z_rel).astype(np.float64)

    # largest ROI over the grid is reached at a corner
    cx = max(abs(xc[0]), abs(xc[-1])); cy = max(abs(yc[0]), abs(yc[-1])); cz = max(abs(zc[0]), abs(zc[-1]))
    rmax = max(min_radius, np.sqrt(cx * cx + cy * cy + cz * cz) * beam_factor) * (1.0 + 1e-9) + 1e-6
    near = (valid
            & (gx >= xc.min() - rmax) & (gx <= xc.max() + rmax)
            & (gy >= yc.min() - rmax) & (gy <= yc.max() + rmax)
            & (gz >= zc.min() - rmax) & (gz <= zc.max() + rmax))
    cand = np.nonzero(near)[0]            # ascending gate index
    cgx, cgy, cgz = gx[cand], gy[cand], gz[cand]

    n_vox = nz * ny * nx
    counts = np.zeros(n_vox, dtype=np.int64)
    idx_parts, w_parts = [], []
    yy, xx = np.meshgrid(yc, xc, indexing="ij")
    vy_level = yy.ravel(); vx_level = xx.ravel()
    rows_per_chunk = max(1, chunk_pairs // max(1, cand.size))
    for iz in range(nz):
        vz = zc[iz]
        dist = np.sqrt(vx_level ** 2 + vy_level ** 2 + np.full(vx_level.shape, vz) ** 2)
        roi = np.maximum(min_radius, dist * beam_factor)
        r2_level = roi * roi
        for s in range(0, ny * nx, rows_per_chunk):
            e = min(ny * nx, s + rows_per_chunk)
            if cand.size == 0:
                continue
            dx = cgx[None, :] - vx_level[s:e, None]
            dy = cgy[None, :] - vy_level[s:e, None]
            dz = cgz[None, :] - vz
            d2 = dx * dx + dy * dy + dz * dz
            r2 = r2_level[s:e, None]
            hit = d2 < r2
            rr, cc = np.nonzero(hit)       # row-major => per voxel, ascending candidate (= gate) order
            if rr.size == 0:
                continue
            counts[iz * ny * nx + s: iz * ny * nx + e] = np.bincount(rr, minlength=e - s)
            d2h = d2[rr, cc]
            r2h = np.broadcast_to(r2, d2.shape)[rr, cc]
            if weighting == "barnes2":
                w = (np.exp(-d2h / (r2h / 4)) + 1e-5).astype("float32")
            elif weighting == "cressman":
                w = ((r2h - d2h) / (r2h + d2h)).astype("float32")
            else:
                w = np.ones(d2h.shape[0], dtype="float32")
            idx_parts.append(cand[cc].astype(np.int32))
            w_parts.append(w)
    indptr = np.zeros(n_vox + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    gate_indices = np.concatenate(idx_parts) if idx_parts else np.zeros(0, dtype=np.int32)
    weights = np.concatenate(w_parts) if w_parts else np.zeros(0, dtype=np.float32)
    return indptr, gate_indices, weights


def canonical_rows(indptr, gate_indices, weights):
    """Sort every CSR row by gate index (stable) so two builders with different in-row order compare."""
    indptr = np.asarray(indptr, dtype=np.int64)
    n_rows = indptr.shape[0] - 1
    row_of = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(indptr))
    order = np.lexsort((np.asarray(gate_indices), row_of))
    return indptr, np.asarray(gate_indices)[order], np.asarray(weights)[order]


# --------------------------------------------------------------------------------------------------
# CSR apply (src/radar_grid/interpolate.py:69-104) -- THE hot loop, also the timed CPU baseline
# --------------------------------------------------------------------------------------------------
def csr_apply(indptr, gate_indices, weights, field_values, field_mask, grid_shape,
              fill_value=np.nan) -> np.ndarray:
    """Masked weighted mean per voxel, in the reference's arithmetic: excluded gates contribute zero to
    both sums (interpolate.py:78-79), products and segment sums are float32 under NumPy >= 2
    (interpolate.py:82,92-93), a voxel is filled when its weight sum is > 0
    (interpolate.py:100-102), otherwise it gets ``fill_value`` (interpolate.py:99)."""
    indptr = np.asarray(indptr)
    vals = np.asarray(field_values)
    excluded = np.asarray(field_mask, dtype=bool)
    n_vox = int(np.prod(grid_shape))
    out = np.full(n_vox, fill_value, dtype="float32")
    if gate_indices.shape[0] == 0:
        return out.reshape(grid_shape)
    drop = excluded[gate_indices]
    w_eff = np.where(drop, 0.0, weights)
    num = w_eff * np.where(drop, 0.0, vals[gate_indices])
    lengths = np.diff(indptr)
    rows = np.nonzero(lengths > 0)[0]
    starts = indptr[:-1][rows]
    num_sum = np.add.reduceat(num, starts)
    den_sum = np.add.reduceat(w_eff, starts)
    ok = den_sum > 0
    out[rows[ok]] = num_sum[ok] / den_sum[ok]
    return out.reshape(grid_shape)


def csr_apply_f64(indptr, gate_indices, weights, field_values, field_mask, grid_shape,
                  fill_value=np.nan) -> np.ndarray:
    """Same contract as :func:`csr_apply` with float64 sums (float32 products kept): the yardstick used to size
    the tolerance between different orders of float32 summation (the reference's pairwise ``reduceat``, the kernels'
    tile / lane orders)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    n_vox = int(np.prod(grid_shape))
    out = np.full(n_vox, fill_value, dtype="float32")
    if gate_indices.shape[0] == 0:
        return out.reshape(grid_shape)
    drop = np.asarray(field_mask, dtype=bool)[gate_indices]
    w32 = np.where(drop, np.float32(0), np.asarray(weights, dtype=np.float32))
    p32 = w32 * np.where(drop, np.float32(0), np.asarray(field_values, dtype=np.float32)[gate_indices])
    row_of = np.repeat(np.arange(n_vox, dtype=np.int64), np.diff(indptr))
    den = np.bincount(row_of, weights=w32.astype(np.float64), minlength=n_vox)
    num = np.bincount(row_of, weights=p32.astype(np.float64), minlength=n_vox)
    ok = den > 0
    out[ok] = (num[ok] / den[ok]).astype(np.float32)
    return out.reshape(grid_shape)

ROWWISE_TARGET = {1: 4, 2: 4, 3: 6, 4: 8, 5: 8, 6: 8, 7: 8, 8: 12}   # records per lane and row the row-wise kernel aims for
ROWWISE_KPRE = {n: 3 for n in range(1, 9)}  # records per lane and step (batch slots), by field count
ROWWISE_CHAINS = 2                          # running sums per lane and value: batch slot k adds into chain k mod 2
ROWWISE_WEIGHT_CHAINS = {1: 2, 2: 2, 3: 2, 4: 2, 5: 1, 6: 1, 7: 1, 8: 1}   # ... of the weight sums (5-8 fields: one chain)


def csr_apply_rowwise_order(indptr, gate_indices, weights, fields, masks, grid_shape, fill_value=np.nan,
                            lanes_hint: int = 0, kpre: int = 0, chains: int = ROWWISE_CHAINS,
                            weight_chains: int = 0) -> np.ndarray:
    """The masked weighted mean of :func:`csr_apply` (interpolate.py:69-104) with the float32 additions performed in
    exactly the order the row-wise kernel of ``rg_csr_compact_apply_packed_f32`` documents
    (radar_processor_amd/csrc/rg_csr_compact.hip), so that the kernel can be checked BIT FOR BIT on small cases:

    * a grid line of nx rows is cut into ceil(nx / 64) balanced segments; a segment's pairs are numbered from 0 and
      grouped in records of three;
    * per segment, L = 2^k lanes per row, k = ceil(log2(ceil(m / T))) capped at 6, m = span // (3 * rows) + 1 the mean
      records per row and T = ROWWISE_TARGET[fields] (``lanes_hint``: 1..64 = that many lanes, 70 + t = target t);
    * lane j of a row owns the row's records q0 + j, q0 + j + L, ... (q0 = first pair // 3); its t-th record sits in
      batch slot t mod K, K = ROWWISE_KPRE[fields] (``kpre``: another K), and belongs to chain (t mod K) mod C, C =
      ``chains`` = 2; per chain one running float32 (sum w*v, sum w) per field over the chain's pairs in ascending order, a
      masked gate adding +0 to both; the lane's chains are then added in ascending order; passes of 5-8 fields keep ONE
      chain for the weight sums (``weight_chains``, default ROWWISE_WEIGHT_CHAINS[fields]): all of the lane's weights in
      ascending pair order;
    * the L lane sums are folded by an xor butterfly (x[i] += x[i ^ 1], then ^ 2, ^ 4, ...);
    * value = float32(float64(sum w*v) / float64(sum w)) where sum w > 0, else ``fill_value``.

    ``fields`` / ``masks``: sequences of F arrays [G] (mask True = excluded).  Returns float32 [F, nz, ny, nx].
    Test infrastructure like the rest of this module; rows are looped in Python: small cases only."""
    indptr = np.asarray(indptr, dtype=np.int64)
    gate_indices = np.asarray(gate_indices)
    w_all = np.asarray(weights, dtype=np.float32)
    nz, ny, nx = (int(v) for v in grid_shape)
    nf = len(fields)
    kpre = int(kpre) if kpre else ROWWISE_KPRE[nf]
    chains = max(1, min(int(chains), kpre))
    wchains = max(1, min(int(weight_chains) if weight_chains else ROWWISE_WEIGHT_CHAINS[nf], chains))
    vals = [np.asarray(f, dtype=np.float32) for f in fields]
    excl = [np.zeros(vals[0].shape, dtype=bool) if m is None else np.asarray(m, dtype=bool) for m in masks]
    out = np.full((nf, nz * ny * nx), fill_value, dtype=np.float32)
    nsx = (nx + 63) // 64
    seg_base, seg_extra = nx // nsx, nx % nsx
    zero = np.float32(0.0)
    for line in range(nz * ny):
        x0 = 0
        for sx in range(nsx):
            nrows = seg_base + (1 if sx < seg_extra else 0)
            r0 = line * nx + x0
            x0 += nrows
            seg_b = int(indptr[r0])
            span = int(indptr[r0 + nrows]) - seg_b
            if span == 0:
                continue
            if 0 < lanes_hint <= 64:
                lgl = int(lanes_hint).bit_length() - 1
            else:
                target = lanes_hint - 70 if lanes_hint > 70 else ROWWISE_TARGET[nf]
                need = (span // (3 * nrows) + 1 + target - 1) // target
                lgl = 0 if need <= 1 else (need - 1).bit_length()
            lanes = 1 << min(lgl, 6)
            for r in range(r0, r0 + nrows):
                ps, pe = int(indptr[r]), int(indptr[r + 1])
                if pe == ps:
                    continue
                rs = ps - seg_b
                o = np.arange(rs, pe - seg_b)                     # pair offsets in the segment
                q = o // 3 - rs // 3                              # record number within the row
                lane, trip = q % lanes, q // lanes
                def place(n_chains):
                    """(chain, column) of every pair: its chain's records in ascending order, three pairs each."""
                    chain = (trip % kpre) % n_chains
                    per_step = [len([k for k in range(kpre) if k % n_chains == c]) for c in range(n_chains)]   # slots per chain
                    before = [[len([k for k in range(s_) if k % n_chains == c]) for s_ in range(kpre)] for c in range(n_chains)]
                    rec_in_chain = (trip // kpre) * np.asarray(per_step)[chain] + np.asarray(before)[chain, trip % kpre]
                    return chain, rec_in_chain * 3 + o % 3
                chain, col = place(chains)
                wchain, wcol = (chain, col) if wchains == chains else place(wchains)
                width = int(max(col.max(), wcol.max())) + 1
                g = gate_indices[ps:pe]
                w = w_all[ps:pe]
                for f in range(nf):
                    good = ~excl[f][g]
                    prod = np.where(good, w * vals[f][g], zero).astype(np.float32)   # float32 product, then the add
                    wgt = np.where(good, w, zero).astype(np.float32)
                    mp = np.zeros((chains, lanes, width), dtype=np.float32)
                    mw = np.zeros((wchains, lanes, width), dtype=np.float32)
                    mp[chain, lane, col] = prod
                    mw[wchain, lane, wcol] = wgt
                    with np.errstate(invalid="ignore", over="ignore"):
                        chain_p = np.add.accumulate(mp, axis=2, dtype=np.float32)[:, :, -1]   # strictly sequential per chain
                        chain_w = np.add.accumulate(mw, axis=2, dtype=np.float32)[:, :, -1]
                        sp, sw = chain_p[0], chain_w[0]
                        for c in range(1, chains):                                            # chains in ascending order
                            sp = (sp + chain_p[c]).astype(np.float32)
                        for c in range(1, wchains):
                            sw = (sw + chain_w[c]).astype(np.float32)
                        m = 1
                        while m < lanes:
                            partner = np.arange(lanes) ^ m
                            sp, sw = (sp + sp[partner]).astype(np.float32), (sw + sw[partner]).astype(np.float32)
                            m <<= 1
                        if sw[0] > 0:
                            out[f, r] = np.float32(np.float64(sp[0]) / np.float64(sw[0]))
    return out.reshape(nf, nz, ny, nx)


def merge_masks(field_data, extra_masks: Sequence[np.ndarray] = ()) -> Tuple[np.ndarray, np.ndarray]:
    """interpolate.py:59-64 -- OR the field's own mask with every filter's ``gate_excluded``."""
    mask = np.ma.getmaskarray(field_data).copy()
    for m in extra_masks:
        mask |= np.asarray(m, dtype=bool)
    return np.ma.getdata(field_data), mask


# --------------------------------------------------------------------------------------------------
# gate filters (src/radar_grid/filters.py:114-258) -- mask producers
# --------------------------------------------------------------------------------------------------
def gate_mask(op: str, data: np.ndarray, a: float = 0.0, b: float = 0.0) -> np.ndarray:
    """Boolean ``True = excluded`` predicates of GateFilter: NaN compares False, so NaN gates are *not*
    excluded by threshold filters (filters.py:134,157,182,207); ``invalid`` is NaN|Inf (filters.py:257)."""
    d = np.asarray(data)
    with np.errstate(invalid="ignore"):
        if op == "below":
            return d < a
        if op == "above":
            return d > a
        if op == "between":
            return (d > a) & (d < b)
        if op == "outside":
            return (d < a) | (d > b)
        if op == "equal":
            return np.abs(d - a) < b
        if op == "invalid":
            return np.isnan(d) | np.isinf(d)
    raise ValueError(op)


# --------------------------------------------------------------------------------------------------
# products (src/radar_grid/products.py:317-580)
# --------------------------------------------------------------------------------------------------
def cappi_plan(z_limits, nz: int, altitude: float, interpolation: str = "linear"):
    """Scalar control flow of ``constant_altitude_ppi`` (products.py:361-412).  Returns one of
    ``("nan",)``, ``("level", k)`` or ``("lerp", k_lo, w_lo, w_hi)`` (weights are Python floats)."""
    z_min, z_max = z_limits
    zc = np.linspace(z_min, z_max, nz, dtype="float32")
    if altitude < z_min or altitude > z_max:                      # products.py:370-372
        return ("nan",)
    if interpolation == "nearest":                                 # products.py:375-378
        return ("level", int(np.argmin(np.abs(zc - altitude))))
    if interpolation != "linear":
        raise ValueError(f"Unknown interpolation method: {interpolation}")
    hit = np.isclose(zc, altitude, rtol=1e-6)                      # products.py:382-386
    if np.any(hit):
        return ("level", int(np.where(hit)[0][0]))
    z_step = (z_max - z_min) / (nz - 1) if nz > 1 else 1.0         # products.py:389
    z_frac = (altitude - z_min) / z_step
    k = int(np.floor(z_frac))
    if k < 0:
        return ("level", 0)
    if k + 1 >= nz:
        return ("level", nz - 1)
    w_hi = z_frac - k
    return ("lerp", k, 1.0 - w_hi, w_hi)


def cappi(grid: np.ndarray, z_limits, altitude: float, interpolation: str = "linear") -> np.ndarray:
    """CAPPI (products.py:317-415).  The lerp is ``w_lo*lo + w_hi*hi`` with weak Python-float weights,
    i.e. float32 multiplies and a float32 add under NumPy >= 2; NaN in either level gives NaN."""
    nz, ny, nx = grid.shape
    plan = cappi_plan(z_limits, nz, altitude, interpolation)
    if plan[0] == "nan":
        return np.full((ny, nx), np.nan, dtype="float32")
    if plan[0] == "level":
        return grid[plan[1]]
    _, k, w_lo, w_hi = plan
    return (w_lo * grid[k] + w_hi * grid[k + 1]).astype("float32")


def column_range(nz: int, z_min_idx=None, z_max_idx=None, z_min_alt=None, z_max_alt=None, z_limits=None):
    """Index window of ``column_max|min|mean`` (products.py:462-485): altitude limits go through
    ``searchsorted`` on the float64 linspace, then defaults and clipping."""
    if z_min_alt is not None or z_max_alt is not None:
        if z_limits is None:
            raise ValueError("geometry is required when using altitude-based limits")
        zc = np.linspace(z_limits[0], z_limits[1], nz)
        if z_min_alt is not None:
            z_min_idx = int(np.searchsorted(zc, z_min_alt))
        if z_max_alt is not None:
            z_max_idx = int(np.searchsorted(zc, z_max_alt, side="right")) - 1
    if z_min_idx is None:
        z_min_idx = 0
    if z_max_idx is None:
        z_max_idx = nz - 1
    return max(0, z_min_idx), min(nz - 1, z_max_idx)


def column_max(grid, lo: int, hi: int) -> np.ndarray:
    """products.py:488-490 -- ``np.nanmax`` over levels ``lo..hi``; an all-NaN column stays NaN."""
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            return np.nanmax(grid[lo:hi + 1], axis=0)


def column_min(grid, lo: int, hi: int) -> np.ndarray:
    """products.py:533-535 -- ``np.nanmin``."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        return np.nanmin(grid[lo:hi + 1], axis=0)


def column_mean(grid, lo: int, hi: int) -> np.ndarray:
    """products.py:578-580 -- ``np.nanmean``."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        return np.nanmean(grid[lo:hi + 1], axis=0)


def column_argmax(grid, lo: int, hi: int) -> np.ndarray:
    """Build-defined contract (the reference has no argmax): level index, *relative to the
    full grid*, of the first level attaining the column's NaN-ignoring maximum (``np.nanargmax`` order),
    ``-1`` for an all-NaN column.  int32."""
    sl = grid[lo:hi + 1]
    all_nan = np.all(np.isnan(sl), axis=0)
    safe = np.where(np.isnan(sl), -np.inf, sl)
    arg = np.argmax(safe, axis=0).astype(np.int32) + np.int32(lo)
    arg[all_nan] = -1
    return arg


# --------------------------------------------------------------------------------------------------
# antenna transform (PyART, not in /root/reference: parity unpinned)
# --------------------------------------------------------------------------------------------------
def antenna_to_cartesian(ranges_m, az_deg, el_deg):
    """4/3-earth model as published for PyART's ``antenna_to_cartesian`` (arm-pyart >= 2.1.1, call sites
    src/radar_grid/utils.py:35-37).  No reference test pins a gate coordinate: parity unpinned."""
    r = np.asarray(ranges_m, dtype=np.float64)
    az = np.asarray(az_deg, dtype=np.float64) * np.pi / 180.0
    el = np.asarray(el_deg, dtype=np.float64) * np.pi / 180.0
    big_r = EARTH_RADIUS * EFFECTIVE_RADIUS_FACTOR
    z = (r ** 2 + big_r ** 2 + 2.0 * r * big_r * np.sin(el)) ** 0.5 - big_r
    s = big_r * np.arcsin(r * np.cos(el) / (big_r + z))
    return s * np.sin(az), s * np.cos(az), z


# --------------------------------------------------------------------------------------------------
# constant-elevation PPI and beam height (src/radar_grid/products.py:23-314)
# --------------------------------------------------------------------------------------------------
def beam_height(ground_range, elevation_deg, radar_altitude=0.0, ke=EFFECTIVE_RADIUS_FACTOR, re=EARTH_RADIUS):
    """products.py:70-89 -- 4/3-earth beam height; slant range from ground range via max(cos, 0.01)."""
    el = np.radians(elevation_deg)
    kr = ke * re
    sr = ground_range / np.maximum(np.cos(el), 0.01)
    return np.sqrt(sr ** 2 + kr ** 2 + 2 * sr * kr * np.sin(el)) - kr + radar_altitude


def beam_height_flat(ground_range, elevation_deg, radar_altitude=0.0):
    """products.py:164-165."""
    return ground_range * np.tan(np.radians(elevation_deg)) + radar_altitude


def elevation_ppi(grid, grid_limits, elevation_deg, interpolation="linear", earth_curvature=True,
                  ke=EFFECTIVE_RADIUS_FACTOR):
    """products.py:224-314 -- per-pixel target altitude from the beam height (float32 ground range, float64
    height), then float64 lerp between bracketing levels (NaN outside the grid) or nearest level (float32)."""
    nz, ny, nx = grid.shape
    (z_min, z_max), (y_min, y_max), (x_min, x_max) = grid_limits
    yy, xx = np.meshgrid(np.linspace(y_min, y_max, ny, dtype="float32"),
                         np.linspace(x_min, x_max, nx, dtype="float32"), indexing="ij")
    dist = np.sqrt(xx ** 2 + yy ** 2)
    tz = beam_height(dist, elevation_deg, 0.0, ke=ke) if earth_curvature else beam_height_flat(dist, elevation_deg, 0.0)
    z_step = (z_max - z_min) / (nz - 1) if nz > 1 else 1.0
    iy, ix = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    if interpolation == "nearest":
        k = np.round((tz - z_min) / z_step).astype(int)
        out = grid[np.clip(k, 0, nz - 1), iy, ix]
        out[~((k >= 0) & (k < nz))] = np.nan
        return out
    if interpolation != "linear":
        raise ValueError(f"Unknown interpolation method: {interpolation}")
    zf = (tz - z_min) / z_step
    lo = np.floor(zf).astype(int)
    w_hi = zf - lo
    out = (1.0 - w_hi) * grid[np.clip(lo, 0, nz - 1), iy, ix] + w_hi * grid[np.clip(lo + 1, 0, nz - 1), iy, ix]
    out[tz < z_min] = np.nan
    out[tz > z_max] = np.nan
    return out


# --------------------------------------------------------------------------------------------------
# closest-gate gridding (PyART map_gates_to_grid 'nearest'; NOT in /root/reference: parity unpinned)
# --------------------------------------------------------------------------------------------------
def closest_gate_grid(gate_x, gate_y, gate_z, values, excluded, grid_shape, grid_limits, roi):
    """Value of the closest non-excluded gate within a constant ROI (ties: lowest gate index); NaN where none.
    Restated from PyART's documented behaviour for the call at src/radar_processor/processor.py:152-163."""
    nz, ny, nx = grid_shape
    zc = axis_coords_f32(grid_limits[0][0], grid_limits[0][1], nz).astype(np.float64)
    yc = axis_coords_f32(grid_limits[1][0], grid_limits[1][1], ny).astype(np.float64)
    xc = axis_coords_f32(grid_limits[2][0], grid_limits[2][1], nx).astype(np.float64)
    keep = ~np.asarray(excluded, dtype=bool)
    gx = np.asarray(gate_x, dtype=np.float64)[keep]
    gy = np.asarray(gate_y, dtype=np.float64)[keep]
    gz = np.asarray(gate_z, dtype=np.float64)[keep]
    val = np.asarray(values, dtype=np.float32)[keep]
    out = np.full(grid_shape, np.nan, dtype=np.float32)
    second = np.full(grid_shape, np.inf)          # gap to the runner-up, for tie-aware comparisons
    for iz in range(nz):
        for iy in range(ny):
            d2 = (gx[None, :] - xc[:, None]) ** 2 + (gy[None, :] - yc[iy]) ** 2 + (gz[None, :] - zc[iz]) ** 2
            d2 = np.where(d2 < roi * roi, d2, np.inf)
            if d2.shape[1] == 0:
                continue
            best = np.argmin(d2, axis=1)
            bd = d2[np.arange(nx), best]
            ok = np.isfinite(bd)
            out[iz, iy, ok] = val[best[ok]]
            if d2.shape[1] > 1:
                part = np.partition(d2, 1, axis=1)
                with np.errstate(invalid="ignore"):          # inf - inf where fewer than two gates are in reach
                    second[iz, iy] = np.where(np.isfinite(part[:, 1]), part[:, 1] - part[:, 0], np.inf)
    return out, second


# --------------------------------------------------------------------------------------------------
# 2-D raster stage behind the 3-D grid cache
#
# Pinning: src/radar_processor does not import here (pyart, rasterio, cachetools are absent) and
# src/radar_grid/geotiff.py needs rasterio at import time, so no golden vectors could be generated from the
# reference for these functions.  They are pinned by the reference's own test expectations, which are restated in
# tests/test_oracle_golden.py: tests/test_utils.py:209-259 (colmax == data3d.max(axis=0), cappi == nearest level,
# ppi shape / type), tests/test_processor_phases.py:265-357 (filter masks) and
# tests/test_geotiff_generation.py:78-127 (RGBA shape / dtype / alpha).  The PPI level choice and the RGBA colour
# values have no known answer in the reference: "parity unpinned" beyond those properties.
# --------------------------------------------------------------------------------------------------
PROCESSOR_EARTH_RADIUS = 8.49e6     # "4/3 Earth radius", src/radar_processor/processor.py:517

REFLECTIVITY_FIELDS = ("filled_DBZH", "DBZH", "DBZV", "DBZHF", "composite_reflectivity")   # processor.py:543
STRICT_FIELDS = ("KDP", "ZDR")                                                              # processor.py:545


def ppi_levels(x_coords, y_coords, z_levels, elevation_deg: float) -> np.ndarray:
    """Level index per pixel for the processor's 'ppi' collapse (src/radar_processor/utils.py:366-371 ==
    processor.py:512-523): beam height ``r sin(e) + r^2 / (2 Re)`` over the ground range ``r``, nearest level."""
    xs, ys = np.meshgrid(np.asarray(x_coords), np.asarray(y_coords), indexing="xy")
    ground = np.sqrt(xs ** 2 + ys ** 2)
    height = ground * np.sin(np.deg2rad(elevation_deg)) + (ground ** 2) / (2.0 * PROCESSOR_EARTH_RADIUS)
    return np.abs(height[..., None] - np.asarray(z_levels)[None, None, :]).argmin(axis=2)


def collapse_3d_to_2d(data3d, product: str, x_coords=None, y_coords=None, z_levels=None, elevation_deg=None,
                      target_height_m=None) -> np.ma.MaskedArray:
    """src/radar_processor/utils.py:336-387: 'ppi' picks ``data3d[level(y, x), y, x]``, 'cappi' the level nearest
    to the target height, 'colmax' the (masked-aware) maximum over levels; float32 masked result."""
    if data3d.ndim == 2:
        plane = data3d
    elif product == "ppi":
        levels = ppi_levels(x_coords, y_coords, z_levels, elevation_deg)
        rows = np.arange(levels.shape[0])[:, None]
        cols = np.arange(levels.shape[1])[None, :]
        plane = data3d[levels, rows, cols]
    elif product == "cappi":
        plane = data3d[int(np.abs(np.asarray(z_levels) - float(target_height_m)).argmin())]
    elif product == "colmax":
        plane = data3d.max(axis=0)
    else:
        raise ValueError("Producto inválido")
    return np.ma.array(plane.astype(np.float32), mask=np.ma.getmaskarray(plane))


def collapse_remask(plane, field: str, vmin: float = -30.0) -> np.ma.MaskedArray:
    """The re-mask at the end of collapse_grid_to_2d (src/radar_processor/processor.py:541-546)."""
    out = np.ma.masked_invalid(plane)
    if field in REFLECTIVITY_FIELDS:
        out = np.ma.masked_less_equal(out, vmin)
    elif field in STRICT_FIELDS:
        out = np.ma.masked_less(out, vmin)
    return out


def filter_masks(plane: np.ma.MaskedArray, visual_filters, qc_filters, field_to_use: str, qc_planes: dict):
    """src/radar_processor/processor.py:802-886.  Filters are objects with ``field`` / ``min`` / ``max``; a visual
    filter on the plotted field tests the plane itself (a minimum <= 0.3 on RHOHV is skipped, :849), any other
    filter tests the cached QC plane of its field when there is one."""
    if not visual_filters and not qc_filters:
        return plane
    out = np.ma.array(plane, copy=True)
    drop = np.zeros(out.shape, dtype=bool)
    for flt in visual_filters:
        name = str(getattr(flt, "field", None) or "").upper()
        if not name:
            continue
        lo, hi = getattr(flt, "min", None), getattr(flt, "max", None)
        if name == str(field_to_use).upper():
            if lo is not None and not (lo <= 0.3 and field_to_use == "RHOHV"):
                drop |= (out < float(lo))
            if hi is not None:
                drop |= (out > float(hi))
        elif qc_planes.get(name) is not None:
            q = qc_planes[name]
            if lo is not None:
                drop |= (q < float(lo))
            if hi is not None:
                drop |= (q > float(hi))
    out.mask = np.ma.getmaskarray(out) | drop
    for flt in qc_filters or ():
        q = qc_planes.get(str(getattr(flt, "field", "") or "").upper())
        if q is None:
            continue
        lo, hi = getattr(flt, "min", None), getattr(flt, "max", None)
        hit = np.zeros(out.shape, dtype=bool)
        if lo is not None:
            hit |= (q < float(lo))
        if hi is not None:
            hit |= (q > float(hi))
        out.mask = np.ma.getmaskarray(out) | hit
    return out


def colormap_rgba(data: np.ndarray, cmap, vmin=None, vmax=None, fill_value=None) -> np.ndarray:
    """src/radar_grid/geotiff.py:70-145: no-data = NaN (or ``== fill_value``), limits default to the valid pixels'
    nanmin / nanmax (0 / 1 when there are none), matplotlib ``Normalize(clip=True)`` then the colormap, ``* 255``
    truncated to uint8, alpha 0 on no-data.  Calls matplotlib -- the same third-party code the reference calls."""
    import matplotlib.pyplot as plt
    from matplotlib.colors import Normalize
    if isinstance(cmap, str):
        cmap = plt.get_cmap(cmap)
    values = data.copy()
    nodata = (values == fill_value) if fill_value is not None else np.isnan(values)
    valid = values[~nodata]
    if vmin is None:
        vmin = np.nanmin(valid) if len(valid) else 0.0
    if vmax is None:
        vmax = np.nanmax(valid) if len(valid) else 1.0
    rgba = (cmap(Normalize(vmin=vmin, vmax=vmax, clip=True)(values)) * 255).astype(np.uint8)
    rgba[nodata, 3] = 0
    return rgba
