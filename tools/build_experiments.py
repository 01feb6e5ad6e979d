#!/usr/bin/env python3
"""Experiment build of the library: the product sources compiled with -DRG_EXPERIMENTS (plus any -D given on the command
line) into tools/_exp/libradargrid_hip_exp[_<tag>].so.

The product library (radar_processor_amd/csrc/libradargrid_hip.so, built by radar_processor_amd.build) computes right
answers or refuses.  What measurement scripts need beyond that lives ONLY in this build:
  * timing-only ablations whose results are wrong by construction (rg_csr_compact_apply_packed_f32 tile = 2100 + bits: no
    output store, no record loads, ...; rg_csr_compact_apply_f32 tile = 901..909; rg_csr_apply_f32_ex variant 28),
  * several chunks per workgroup (tile = 2200 + n), block-rotation overrides (tile + 1000 * rotation), K1 tuning variants,
  * the A/B knobs of the row-wise kernel (-DRG_ROWWISE_KPRE3= / _TARGET3= / _REGS3= / _WAVES1= / _WAVES3= / _SLOTS= /
    _TWO_SELECTS, -DRG_FILL_BATCH=; round 4: _BYTEMASK= (smallest field count with byte-mask entries), _MASK_UBYTE,
    _KPRE8= / _TARGET8= / _REGS8= / _WAVES8= / _WSLOTS8= (five to eight fields), _SCATTER_MIN_NF= + _FENCE_MIN_NF=,
    _NO_STAGE, _NO_FALLBACK (register-count probe), _PREFETCH= + _PREFETCH_MIN_NF= + _PREFETCH_HEAD= (touch loads)),
  * timing-only ablations of the eight-field pass (tile = 2100 + 1 / 2 / 16 / 19 / 40 / 42 / 43 / 59) and the overlapping-streams
    proxy of a denser record (tile = 2173).

    python tools/build_experiments.py [--tag NAME] [-DFLAG[=V] ...]      -> prints the path of the library

Scripts use it by assigning the path to ``radar_processor_amd._native.LIB_PATH`` before the first load (``ensure()``
below); the package itself has no override (the ABI version check of the loader applies to this build as well).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)
OUT_DIR = os.path.join(HERE, "_exp")


def ensure(tag: str = "", defines=(), verbose: bool = True) -> str:
    """Build (if older than the sources) and return the experiment library for this set of defines."""
    from radar_processor_amd import build as rg_build
    os.makedirs(OUT_DIR, exist_ok=True)
    name = "libradargrid_hip_exp" + (f"_{tag}" if tag else "") + ".so"
    lib = os.path.join(OUT_DIR, name)
    # stamp = digest of the sources' CONTENTS + the defines (time stamps do not survive a snapshot copy)
    import hashlib
    stamp = hashlib.sha256((rg_build.source_digest() + repr(sorted(defines))).encode()).hexdigest()
    if os.path.exists(lib) and os.path.exists(lib + ".stamp") and open(lib + ".stamp").read().strip() == stamp:
        return lib
    hipcc = rg_build._hipcc()
    flags = ["-std=c++17", "-O3", f"--offload-arch={rg_build.ARCH}", "-fPIC", f"-I{rg_build.INCLUDE}", f"-I{rg_build.CSRC}",
             "-Wno-unused-result", *rg_build.COMMON_FLAGS, "-DRG_EXPERIMENTS", *defines]
    objs = []
    procs = []
    for src, extra in rg_build.SOURCES:
        obj = os.path.join(OUT_DIR, (tag + "_" if tag else "") + src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *flags, *extra, "-c", os.path.join(rg_build.CSRC, src), "-o", obj]
        if verbose:
            print("[exp-build]", " ".join(cmd), file=sys.stderr, flush=True)
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("experiment build failed")
    subprocess.run([hipcc, f"--offload-arch={rg_build.ARCH}", "-shared", "-fPIC", *objs, "-o", lib], check=True)
    with open(lib + ".stamp", "w") as f:
        f.write(stamp + "\n")
    return lib


def use(tag: str = "", defines=()) -> str:
    """Point the package's loader at the experiment build (call before anything loads the library)."""
    from radar_processor_amd import _native
    if _native._lib is not None:
        raise RuntimeError("the product library is already loaded in this process")
    _native.LIB_PATH = ensure(tag, defines)
    return _native.LIB_PATH


if __name__ == "__main__":
    argv = sys.argv[1:]
    tag = ""
    if "--tag" in argv:
        i = argv.index("--tag")
        tag = argv[i + 1]
        del argv[i:i + 2]
    print(ensure(tag, [a for a in argv if a.startswith("-D")]))
