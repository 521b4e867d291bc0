"""Builds libacmi.so (host builder + C ABI + gfx950 HIP kernels) in-tree with hipcc.

    python -m php_aho_corasick_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the
GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libacmi.so")
SOURCES = ["automaton.cpp", "engine.cpp", "engine_pipeline.cpp", "engine_choice.cpp", "engine_comm.cpp", "engine_replace.cpp", "k_anchor.hip", "k_anchor_dense.hip", "k_lds.hip", "k_walk.hip", "k_walk_items.hip", "k_walk_items_emit.hip", "k_tiles.hip", "k_win.hip", "k_chain.hip", "k_drain.hip", "k_pipeline.hip", "k_small.hip", "k_misc.hip"]
HEADERS = ["automaton.hpp", "device_types.hpp", "kernels.hpp", "engine.hpp", "kdev.hpp", "k_anchor_impl.hpp", "k_walk_items_impl.hpp", "scan_asm.inc", os.path.join("..", "..", "include", "acmi.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libacmi.so cannot be built (there is no CPU-only build)")


OBJ_DIR = os.path.join(CSRC, "build")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per source (compiled in parallel, only when it or a header changed), then the link."""
    if not force and not stale():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, src + ".o")
        objs.append(obj)
        path = os.path.join(CSRC, src)
        # (the device-side helpers are only seen by the kernel files: a change there leaves the host objects alone)
        deps = [h for h in HEADERS if h != "engine.hpp"] if src.endswith(".hip") else [h for h in HEADERS if h not in ("kdev.hpp", "k_anchor_impl.hpp", "k_walk_items_impl.hpp", "scan_asm.inc")]
        hdr_time = max(os.path.getmtime(os.path.join(CSRC, f)) for f in deps)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
               "-c", path, "-o", obj]
        if src.endswith(".hip"):
            cmd.insert(1, "-save-temps=obj")   # keeps the device assembly next to the object: tools/check_asm.py, tests/test_asm.py
        jobs.append(cmd)
    # the longest translation units first, as many at a time as the machine has CPUs
    order = {"k_anchor.hip": 0, "k_walk_items.hip": 1, "k_walk_items_emit.hip": 2, "k_anchor_dense.hip": 3, "k_lds.hip": 4, "k_tiles.hip": 5, "k_walk.hip": 6, "engine_pipeline.cpp": 7, "engine.cpp": 8}
    jobs.sort(key=lambda c: order.get(os.path.basename(c[-3]), 9))
    running, limit = [], max(2, os.cpu_count() or 4)
    while jobs or running:
        while jobs and len(running) < limit:
            cmd = jobs.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd, cwd=OBJ_DIR)))
        cmd, p = running.pop(0)
        if p.wait() != 0:
            for _c, q in running:
                q.kill()
            raise subprocess.CalledProcessError(p.returncode, cmd)
    # (of what -save-temps leaves behind only the device assembly is kept)
    for f in os.listdir(OBJ_DIR):
        if f.endswith((".bc", ".hipi", ".hipfb", ".out", ".resolution.txt")) or (f.endswith(".s") and "gfx950" not in f) or f.endswith("gfx950.o"):
            os.remove(os.path.join(OBJ_DIR, f))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
