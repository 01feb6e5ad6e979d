"""Seeded synthetic radar volumes and the antenna -> Cartesian transform.

The reference never computes gate coordinates itself: ``radar_grid/utils.py:35-38`` pulls
``radar.gate_x/gate_y/gate_z`` from PyART, whose ``antenna_vectors_to_cartesian`` (arm-pyart >= 2.1.1,
not vendored under /root/reference, so **parity unpinned** for this transform) uses the 4/3-earth model

    z = sqrt(r^2 + R^2 + 2 r R sin(el)) - R,   s = R asin(r cos(el) / (R + z)),
    x = s sin(az),  y = s cos(az),             R = 4/3 * 6371 km

The same constants appear in ``radar_grid/products.py:19-20``.  Gate coordinates are only *inputs* to the
gridding path, so both the oracle and the HIP path consume the identical float32 arrays produced here.

Everything in this module is host-side NumPy: it builds benchmark / test inputs, it is not the hot path.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

EARTH_RADIUS = 6371000.0
EFFECTIVE_RADIUS_FACTOR = 4.0 / 3.0

#: Elevation list fixed by SURVEY.md §8(d); a volume with ``n_elev`` sweeps takes the first ``n_elev``.
ELEVATIONS_DEG = (0.5, 0.9, 1.3, 1.9, 2.3, 3.0, 3.5, 5.0, 6.9, 9.1, 11.8, 15.1, 19.5, 25.0)
MAX_RANGE_M = 240000.0


def antenna_to_cartesian(ranges_m, azimuths_deg, elevations_deg):
    """4/3-earth antenna -> Cartesian transform (float64 math), broadcasting its three inputs.

    Returns ``(x, y, z)`` in metres relative to the radar, float64.  Callers round to float32
    exactly like ``radar_grid/utils.py:35-37`` does with PyART's arrays.
    """
    r = np.asarray(ranges_m, dtype=np.float64)
    az = np.deg2rad(np.asarray(azimuths_deg, dtype=np.float64))
    el = np.deg2rad(np.asarray(elevations_deg, dtype=np.float64))
    big_r = EARTH_RADIUS * EFFECTIVE_RADIUS_FACTOR
    z = np.sqrt(r * r + big_r * big_r + 2.0 * r * big_r * np.sin(el)) - big_r
    s = big_r * np.arcsin(r * np.cos(el) / (big_r + z))
    return s * np.sin(az), s * np.cos(az), z


@dataclass
class SyntheticVolume:
    """One synthetic volume scan: ``n_sweeps`` PPI sweeps of ``n_az`` rays x ``n_gates`` gates.

    Gate order is PyART's: ray-major, rays grouped by sweep (``ray = sweep * n_az + iaz``), so the flat
    gate index is ``(sweep * n_az + iaz) * n_gates + k`` -- the layout ``ravel()`` produces in
    ``radar_grid/utils.py:35-37``.
    """

    n_sweeps: int
    n_az: int
    n_gates: int
    seed: int
    elevations_deg: np.ndarray            # [n_sweeps]
    azimuths_deg: np.ndarray              # [n_az]
    ranges_m: np.ndarray                  # [n_gates]
    gate_x: np.ndarray                    # float32 [G]
    gate_y: np.ndarray                    # float32 [G]
    gate_z: np.ndarray                    # float32 [G]
    fields: Dict[str, np.ma.MaskedArray] = field(default_factory=dict)  # float32 masked [G]
    radar_altitude: float = 0.0

    @property
    def n_rays(self) -> int:
        return self.n_sweeps * self.n_az

    @property
    def n_total_gates(self) -> int:
        return self.n_rays * self.n_gates

    def digest(self) -> str:
        """sha256 over coordinates and fields; golden fixtures store it to detect generator drift."""
        h = hashlib.sha256()
        for a in (self.gate_x, self.gate_y, self.gate_z):
            h.update(np.ascontiguousarray(a).tobytes())
        for name in sorted(self.fields):
            f = self.fields[name]
            h.update(name.encode())
            # NaN payloads are canonical (np.nan) so raw bytes are stable
            h.update(np.ascontiguousarray(np.ma.getdata(f)).tobytes())
            h.update(np.ascontiguousarray(np.ma.getmaskarray(f)).tobytes())
        return h.hexdigest()

    def as_radar(self):
        """Duck-typed stand-in for ``pyart.core.Radar`` (what ``radar_grid/utils.py`` and
        ``radar_grid/filters.py:40-52`` touch): 2-D ``[n_rays, n_gates]`` field arrays, gate coordinates,
        range / elevation / azimuth tables, altitude and metadata."""
        shape2d = (self.n_rays, self.n_gates)
        radar = SimpleNamespace()
        radar.nrays = self.n_rays
        radar.ngates = self.n_gates
        radar.nsweeps = self.n_sweeps
        radar.fields = {k: {"data": v.reshape(shape2d)} for k, v in self.fields.items()}
        radar.gate_x = {"data": self.gate_x.reshape(shape2d)}
        radar.gate_y = {"data": self.gate_y.reshape(shape2d)}
        radar.gate_z = {"data": self.gate_z.reshape(shape2d)}
        radar.gate_altitude = {"data": (self.gate_z + np.float32(self.radar_altitude)).reshape(shape2d)}
        radar.range = {"data": self.ranges_m.astype(np.float32)}
        radar.elevation = {"data": np.repeat(self.elevations_deg, self.n_az).astype(np.float32)}
        radar.azimuth = {"data": np.tile(self.azimuths_deg, self.n_sweeps).astype(np.float32)}
        radar.altitude = {"data": np.array([self.radar_altitude])}
        radar.latitude = {"data": np.array([-31.44])}
        radar.longitude = {"data": np.array([-64.19])}
        radar.metadata = {"instrument_name": "SYNTH", "scan_id": f"{self.n_sweeps}x{self.n_az}x{self.n_gates}",
                          "volume_number": self.seed}
        return radar


def sweep_geometry(n_elev: int, n_az: int, n_gates: int,
                   max_range_m: float = MAX_RANGE_M) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Elevation / azimuth / range tables of SURVEY.md §8(d): fixed elevation list, ``az = i*360/n_az``,
    gate centres ``(k + 0.5) * dr`` with ``dr = max_range / n_gates``."""
    if not 1 <= n_elev <= len(ELEVATIONS_DEG):
        raise ValueError(f"n_elev must be in 1..{len(ELEVATIONS_DEG)}")
    elev = np.asarray(ELEVATIONS_DEG[:n_elev], dtype=np.float64)
    az = np.arange(n_az, dtype=np.float64) * (360.0 / n_az)
    dr = max_range_m / n_gates
    rng = (np.arange(n_gates, dtype=np.float64) + 0.5) * dr
    return elev, az, rng


def gate_coordinates(elev_deg, az_deg, ranges_m) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Flat float32 gate coordinates in PyART ray-major order."""
    e = np.asarray(elev_deg)[:, None, None]
    a = np.asarray(az_deg)[None, :, None]
    r = np.asarray(ranges_m)[None, None, :]
    x, y, z = antenna_to_cartesian(r, a, e)
    shape = (e.shape[0], a.shape[1], r.shape[2])
    return tuple(np.broadcast_to(c, shape).astype(np.float32).ravel() for c in (x, y, z))


def _storm_cells(rng: np.random.Generator, gx, gy, gz, n_cells: int = 12) -> np.ndarray:
    """Reflectivity background: a stratiform layer plus ``n_cells`` 3-D Gaussian storm cells (dBZ)."""
    g = gx.shape[0]
    out = np.full(g, -12.0, dtype=np.float32)
    # stratiform rain below ~5 km, fading with height
    out += (18.0 * np.exp(-np.square(gz / 4500.0))).astype(np.float32)
    for _ in range(n_cells):
        cx, cy = rng.uniform(-200e3, 200e3, size=2)
        cz = rng.uniform(1500.0, 7000.0)
        sx, sy = rng.uniform(8e3, 40e3, size=2)
        sz = rng.uniform(1500.0, 5000.0)
        amp = rng.uniform(20.0, 48.0)
        d = (np.square((gx - np.float32(cx)) / np.float32(sx))
             + np.square((gy - np.float32(cy)) / np.float32(sy))
             + np.square((gz - np.float32(cz)) / np.float32(sz)))
        out += np.float32(amp) * np.exp(-0.5 * d, dtype=np.float32)
    return out


def make_volume(n_elev: int = 12, n_az: int = 360, n_gates: int = 1000, seed: int = 0,
                fields: Sequence[str] = ("DBZH",), max_range_m: float = MAX_RANGE_M,
                radar_altitude: float = 0.0) -> SyntheticVolume:
    """Build the seeded synthetic volume of SURVEY.md §8(d).

    * DBZH: storm cells (-10..60 dBZ) + N(0, 2) noise; gates below a range-dependent noise floor are
      masked (about 30 %), 1 % of the remaining gates are NaN.
    * ZDR: N(1, 1.5), 0.5 % NaN.
    * RHOHV: ``clip(0.98 - 0.3*Beta(2, 8) - clutter, 0, 1)`` with near-range clutter patches so that about
      15 % of gates fail ``>= 0.8``.

    Every field is returned the way ``radar_grid/utils.py:64-66`` hands it on: a float32 masked array with
    NaN/Inf masked (``np.ma.masked_invalid``) on top of the explicit below-noise mask.
    """
    elev, az, rng_m = sweep_geometry(n_elev, n_az, n_gates, max_range_m)
    gx, gy, gz = gate_coordinates(elev, az, rng_m)
    vol = SyntheticVolume(n_sweeps=n_elev, n_az=n_az, n_gates=n_gates, seed=seed,
                          elevations_deg=elev, azimuths_deg=az, ranges_m=rng_m,
                          gate_x=gx, gate_y=gy, gate_z=gz, radar_altitude=float(radar_altitude))
    g = gx.shape[0]
    for name in fields:
        # one independent, reproducible stream per (seed, field)
        tag = int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little")
        rng = np.random.default_rng([seed, tag])
        if name == "DBZH":
            data = _storm_cells(rng, gx, gy, gz)
            data += rng.normal(0.0, 2.0, size=g).astype(np.float32)
            np.clip(data, -32.0, 75.0, out=data)
            slant = np.sqrt(np.square(gx) + np.square(gy) + np.square(gz))
            noise_floor = (-14.0 + 20.0 * np.log10(np.maximum(slant, 1.0) / 100e3)).astype(np.float32)
            below = data < noise_floor + np.float32(rng.uniform(-0.5, 0.5))
            data[rng.random(g) < 0.01] = np.nan
            arr = np.ma.masked_invalid(np.ma.array(data, mask=below))
        elif name == "ZDR":
            data = rng.normal(1.0, 1.5, size=g).astype(np.float32)
            data[rng.random(g) < 0.005] = np.nan
            arr = np.ma.masked_invalid(data)
        elif name == "RHOHV":
            data = (0.98 - 0.3 * rng.beta(2.0, 8.0, size=g)).astype(np.float32)
            ground = np.sqrt(np.square(gx) + np.square(gy))
            clutter = (ground < 35e3) & (gz < 1200.0) & (rng.random(g) < 0.55)
            speckle = rng.random(g) < 0.10
            data[clutter | speckle] -= rng.uniform(0.2, 0.6, size=int((clutter | speckle).sum())).astype(np.float32)
            np.clip(data, 0.0, 1.0, out=data)
            arr = np.ma.masked_invalid(data)
        else:
            data = rng.normal(0.0, 1.0, size=g).astype(np.float32)
            arr = np.ma.masked_invalid(data)
        # masked_invalid on an unmasked array may leave mask == nomask; the reference (F9) needs a full mask
        arr = np.ma.array(np.ma.getdata(arr).astype(np.float32), mask=np.ma.getmaskarray(arr))
        vol.fields[name] = arr
    return vol


#: BASELINE.json configs -> (n_elev, n_az, n_gates, grid_shape, grid_limits); z in [0, 15 km], x/y in +-240 km.
CONFIGS = {
    "C1": dict(n_elev=1, n_az=360, n_gates=500, grid_shape=(1, 500, 500),
               grid_limits=((0.0, 0.0), (-240e3, 240e3), (-240e3, 240e3))),
    "C2": dict(n_elev=12, n_az=360, n_gates=1000, grid_shape=(20, 1000, 1000),
               grid_limits=((0.0, 15e3), (-240e3, 240e3), (-240e3, 240e3))),
    "C4": dict(n_elev=14, n_az=720, n_gates=2000, grid_shape=(40, 2000, 2000),
               grid_limits=((0.0, 15e3), (-240e3, 240e3), (-240e3, 240e3))),
    # the workload BASELINE.json's `metric` is quoted on: the 12-elevation volume onto the 40x2000x2000 grid
    "METRIC": dict(n_elev=12, n_az=360, n_gates=1000, grid_shape=(40, 2000, 2000),
                   grid_limits=((0.0, 15e3), (-240e3, 240e3), (-240e3, 240e3))),
}


def gate_coordinates_device(elev_deg, az_deg, ranges_m, device=None):
    """Same as :func:`gate_coordinates` but computed in HBM by ``rg_antenna_to_cartesian_f32``
    (csrc/rg_core.hip): returns three cuda float32 tensors ``[G]`` in ray-major order."""
    from . import _native
    torch = _native.torch_mod()
    lib = _native.load_library()
    dev = _native.canonical_device(device)
    elev = np.asarray(elev_deg, dtype=np.float64)
    az = np.asarray(az_deg, dtype=np.float64)
    rng_m = np.ascontiguousarray(ranges_m, dtype=np.float64)
    n_rays, n_gates = elev.shape[0] * az.shape[0], rng_m.shape[0]
    el_ray = torch.from_numpy(np.repeat(elev, az.shape[0])).to(dev)
    az_ray = torch.from_numpy(np.tile(az, elev.shape[0])).to(dev)
    r_t = torch.from_numpy(rng_m).to(dev)
    x, y, z = (torch.empty(n_rays * n_gates, dtype=torch.float32, device=dev) for _ in range(3))
    with torch.cuda.device(dev):
        _native.check(lib.rg_antenna_to_cartesian_f32(_native.ptr(r_t), n_gates, _native.ptr(az_ray), _native.ptr(el_ray),
                                                      n_rays, _native.ptr(x), _native.ptr(y), _native.ptr(z),
                                                      _native.stream_ptr()), "rg_antenna_to_cartesian_f32")
    return x, y, z
