"""Builds libsventt_hip.so (gfx950) in-tree with hipcc.

    python -m sve_ntt_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the
GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsventt_hip.so")
SOURCES = ["kernels.hip", "kernels_gold.hip", "kernels_shoup.hip", "plan.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))) + [
    os.path.join(ROOT, "include", "sventt_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra",
         "-Wno-unused-parameter", "-fno-gpu-rdc"]


def _newest_input() -> float:
    paths = [os.path.join(CSRC, s) for s in SOURCES]
    paths += [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return max(os.path.getmtime(p) for p in paths)


def needs_build() -> bool:
    return not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest_input()


def regenerate_stage_asm() -> None:
    """stage_asm.inc is generated (and committed); refresh it when its generator is newer.  Only
    called on the way to a compilation; the file is replaced atomically, and left alone when the
    generator's output is what it already holds (several ranks or test workers may import the
    package at once, and an installed tree may be read-only)."""
    gen = os.path.join(CSRC, "gen_stage_asm.py")
    inc = os.path.join(CSRC, "stage_asm.inc")
    if os.path.exists(inc) and os.path.getmtime(inc) >= os.path.getmtime(gen):
        return
    text = subprocess.run([sys.executable, gen], check=True, capture_output=True, text=True).stdout
    try:
        with open(inc) as f:
            if f.read() == text:
                os.utime(inc)  # same content: only remember that it was checked
                return
    except OSError:
        pass
    tmp = f"{inc}.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, inc)


def _stage_asm_is_stale() -> bool:
    gen = os.path.join(CSRC, "gen_stage_asm.py")
    inc = os.path.join(CSRC, "stage_asm.inc")
    return not os.path.exists(inc) or os.path.getmtime(inc) < os.path.getmtime(gen)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build() and not _stage_asm_is_stale():
        return LIB
    regenerate_stage_asm()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
