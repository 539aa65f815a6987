"""Build the gfx950 kernel library in-tree: ``ops/libmitune_hip.so``.

Plain ``hipcc --offload-arch=gfx950 -shared -fPIC`` over ``ops/csrc/*.hip``: the sources use only
the HIP runtime (no torch headers), so a full rebuild takes seconds and the resulting library has a
C ABI that ``ops/lib.py`` binds with ctypes.  hipcc cross-compiles, so this also runs on a machine
without a GPU.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmitune_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def sources() -> List[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


_MARK = b"mt-src-digest:"


def library_digest(path: str = LIB) -> str:
    """Digest of the sources a built library was compiled from, "" when the file is missing or predates the marker.
    The build passes ``-DMT_SRC_DIGEST=...`` and runtime.hip keeps it as the string ``mt-src-digest:<hex>`` (also returned
    by ``mt_source_digest()``); it is read here from the file's bytes, without loading the library, so a stale ``.so``
    next to newer sources can neither pass for current nor stay mapped while it is rebuilt."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return ""
    i = blob.find(_MARK)
    if i < 0:
        return ""
    tail = blob[i + len(_MARK):i + len(_MARK) + 64]
    return tail.decode("ascii", "replace") if len(tail) == 64 else ""


def is_current() -> bool:
    return library_digest() == _digest()


def build(force: bool = False, verbose: bool = True, extra_flags: List[str] | None = None) -> str:
    if not force and is_current():
        return LIB
    tmp = f"{LIB}.tmp.{os.getpid()}"          # built aside and renamed: a concurrent reader never maps a half-written file
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", f'-DMT_SRC_DIGEST="{_digest()}"', "-o", tmp] + (extra_flags or []) + sources()
    if verbose:
        print("[mitune] " + " ".join(cmd).replace(tmp, LIB), flush=True)
    try:
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


def build_sanitized(verbose: bool = True) -> str:
    """Host-side AddressSanitizer + UBSan build of the launchers and the step driver (``libmitune_hip_asan.so``).

    Only the host half is instrumented (``-Xarch_host -fsanitize=...``); device code is compiled as usual.  Load it
    with ``LD_PRELOAD=$(clang -print-file-name=libclang_rt.asan-x86_64.so)`` to screen descriptor handling and
    launch-argument code on a CPU-only machine; it is never the library the framework loads by default.
    """
    out = os.path.join(HERE, "libmitune_hip_asan.so")
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host",
           "-fno-omit-frame-pointer", f'-DMT_SRC_DIGEST="{_digest()}"', "-o", out] + sources()
    if verbose:
        print("[mitune] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
