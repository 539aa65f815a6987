"""Build the gfx950 kernel library in-tree: ``ops/libmitune_hip.so``.

Plain ``hipcc --offload-arch=gfx950`` over ``ops/csrc/*.hip``: the sources use only the HIP runtime (no torch
headers) and the resulting library has a C ABI that ``ops/lib.py`` binds with ctypes.  Every source is compiled to
its own object file, several at once, and the objects are kept under ``ops/.build`` keyed by a digest of the source,
the headers and the flags, so that after an edit only the edited file is compiled again before the link.  hipcc
cross-compiles, so this also runs on a machine without a GPU.
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
OBJ_DIR = os.path.join(HERE, ".build")
LIB = os.path.join(HERE, "libmitune_hip.so")
ARCH = "gfx950"
DIGEST_SOURCE = "runtime.hip"        # the one source that embeds the digest of all sources (mt_source_digest)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def sources() -> List[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest_of(suffixes) -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(suffixes):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def _digest() -> str:
    return _digest_of((".hip", ".h"))


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


def _jobs() -> int:
    env = int(os.environ.get("MAX_JOBS", "0") or 0)
    return max(1, min(env or (os.cpu_count() or 4), 16))


def _compile_objects(flags: List[str], tag: str, verbose: bool) -> List[str]:
    """One object per source under ``ops/.build``, named ``<source>.<tag>.<digest of source + headers + flags>.o``;
    objects that exist are reused, the others are compiled by up to ``MAX_JOBS`` hipcc processes at once."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers, everything = _digest_of((".h",)), _digest()
    todo, objs = [], []
    for src in sources():
        name = os.path.basename(src)
        own = list(flags)
        if name == DIGEST_SOURCE:
            own.append(f'-DMT_SRC_DIGEST="{everything}"')
        with open(src, "rb") as fh:
            key = hashlib.sha256(fh.read() + headers.encode() + " ".join(own).encode()).hexdigest()[:20]
        stem = os.path.splitext(name)[0]
        obj = os.path.join(OBJ_DIR, f"{stem}.{tag}.{key}.o")
        objs.append(obj)
        if not os.path.exists(obj):
            todo.append((src, obj, stem, own))

    def compile_one(job) -> None:
        src, obj, stem, own = job
        for old in os.listdir(OBJ_DIR):              # objects of earlier versions of this source
            if old.startswith(f"{stem}.{tag}.") and old.endswith(".o"):
                try:
                    os.remove(os.path.join(OBJ_DIR, old))
                except OSError:
                    pass
        tmp = f"{obj}.tmp.{os.getpid()}"
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-Wno-unused-result"] + own + ["-c", src, "-o", tmp]
        if verbose:
            print("[mitune] " + " ".join(cmd).replace(tmp, obj), flush=True)
        try:
            subprocess.run(cmd, check=True, cwd=CSRC)
            os.replace(tmp, obj)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    if todo:
        with ThreadPoolExecutor(min(len(todo), _jobs())) as pool:
            list(pool.map(compile_one, todo))
    return objs


def _link(objs: List[str], out: str, flags: List[str], verbose: bool) -> None:
    tmp = f"{out}.tmp.{os.getpid()}"          # linked aside and renamed: a concurrent reader never maps a half-written file
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared"] + flags + ["-o", tmp] + objs
    if verbose:
        print("[mitune] " + " ".join(cmd).replace(tmp, out), flush=True)
    try:
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def build(force: bool = False, verbose: bool = True, extra_flags: List[str] | None = None) -> str:
    """Compile what changed and link ``libmitune_hip.so``.  ``force`` also discards the object cache."""
    if not force and is_current():
        return LIB
    if force and os.path.isdir(OBJ_DIR):
        shutil.rmtree(OBJ_DIR, ignore_errors=True)
    objs = _compile_objects(["-O3"] + (extra_flags or []), "rel", verbose)
    _link(objs, LIB, [], verbose)
    return LIB


def build_sanitized(verbose: bool = True) -> str:
    """Host-side AddressSanitizer + UBSan build of the launchers and the step driver (``libmitune_hip_asan.so``).

    Only the host half is instrumented (``-Xarch_host -fsanitize=...``); device code is compiled as usual.  Load it
    with ``LD_PRELOAD=$(clang -print-file-name=libclang_rt.asan-x86_64.so)`` to screen descriptor handling and
    launch-argument code on a CPU-only machine; it is never the library the framework loads by default.
    """
    out = os.path.join(HERE, "libmitune_hip_asan.so")
    san = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host",
           "-fno-omit-frame-pointer"]
    objs = _compile_objects(["-O1", "-g"] + san, "asan", verbose)
    _link(objs, out, san, verbose)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
