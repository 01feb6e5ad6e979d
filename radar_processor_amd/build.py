"""Build recipe for libradargrid_hip.so (gfx950 only, in-tree so the .so travels with the snapshot).

    python -m radar_processor_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  All translation units are compiled with FP contraction
disabled (they reproduce NumPy's unfused arithmetic).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_PATH = os.path.join(CSRC, "libradargrid_hip.so")
ARCH = "gfx950"

# (source, extra flags)
SOURCES = [
    ("rg_core.hip", []),
    ("rg_csr_apply.hip", []),
    ("rg_csr_compact.hip", []),
    ("rg_csr_columns.hip", []),
    ("rg_products.hip", []),
    ("rg_geometry.hip", []),
    ("rg_roi_grid.hip", []),
    ("rg_raster.hip", []),
]

# Every translation unit is built with FP contraction OFF: the parity contract is NumPy's arithmetic, which
# never fuses a multiply with an add (on AMD, HIP's __fmul_rn/__fadd_rn are plain operators and do not stop the
# compiler from contracting).  None of the kernels is FMA-throughput bound, so this costs nothing measurable.
COMMON_FLAGS = ["-ffp-contract=off"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libradargrid_hip.so cannot be built")
    return exe


def _stale(target: str, deps: List[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source for gfx950 and link the shared library. Returns its path."""
    hipcc = _hipcc()
    if not force and os.path.exists(STAMP_PATH):
        with open(STAMP_PATH) as f:             # contents changed behind unchanged time stamps (a restored file): rebuild all
            force = f.read().strip() != source_digest() and all(
                not _stale(os.path.join(CSRC, src.replace(".hip", ".o")), sources_and_headers()) for src, _ in SOURCES)
    headers = ([os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".hpp")]
               + [os.path.join(INCLUDE, "radargrid_hip.h")])
    common = ["-std=c++17", "-O3", f"--offload-arch={ARCH}", "-fPIC",
              f"-I{INCLUDE}", f"-I{CSRC}", "-Wall", "-Wno-unused-result", *COMMON_FLAGS]
    objs, jobs = [], []
    for src, extra in SOURCES:
        src_path = os.path.join(CSRC, src)
        if not os.path.exists(src_path):
            continue
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src_path] + headers):
            jobs.append([hipcc, *common, *extra, "-c", src_path, "-o", obj])
    if jobs:      # the translation units are independent: compile them side by side (hipcc is a separate process each)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as pool:
            list(pool.map(run, jobs))
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with open(STAMP_PATH, "w") as f:       # what this library was compiled from (ensure_built compares it)
        f.write(source_digest() + "\n")
    return LIB_PATH


def sources_and_headers() -> List[str]:
    """Every file the library is compiled from."""
    return ([os.path.join(CSRC, src) for src, _ in SOURCES if os.path.exists(os.path.join(CSRC, src))]
            + [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".hpp")]
            + [os.path.join(INCLUDE, "radargrid_hip.h")])


STAMP_PATH = LIB_PATH + ".stamp"


def source_digest() -> str:
    """sha256 over the names and CONTENTS of every source and header plus the compile flags: what the library was built
    from.  Contents, not time stamps -- a snapshot copy or a checkout may reorder mtimes without changing a byte."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((ARCH, COMMON_FLAGS, [(s, e) for s, e in SOURCES])).encode())
    for path in sources_and_headers():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale() -> bool:
    """The library is missing, or was built from other sources / headers than the ones in the tree now (its stamp file
    holds the digest of what it was compiled from)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    with open(STAMP_PATH) as f:
        return f.read().strip() != source_digest()


def ensure_built(verbose: bool = True) -> str:
    """Build the library when it is missing OR was compiled from other sources / headers than the tree holds now (a
    git-ignored ``.so`` left over from an earlier revision would otherwise be used silently).  Called by the entry points
    that own a process -- tests, ``bench.py``, ``smoke()`` -- never by the product path, which keeps failing loudly when
    the library is missing and refuses a library of another ABI version (``_native.load_library``)."""
    if is_stale():
        import fcntl
        # several ranks of one node may get here together: one compiles, the others wait on the lock and find it built
        with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if is_stale():
                    if verbose:
                        what = "is missing" if not os.path.exists(LIB_PATH) else "was built from other sources"
                        print(f"[build] {LIB_PATH} {what}: compiling it now (hipcc, {ARCH})", flush=True)
                    build(force=os.path.exists(LIB_PATH), verbose=verbose)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
