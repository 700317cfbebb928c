"""Build libfreepose_hip.so (gfx950 only) in-tree with hipcc, and the oracle's C library with gcc.

    python -m freepose_amd.build            # build everything that is stale
    python -m freepose_amd.build --force
    python -m freepose_amd.build --lab      # additionally libfreepose_hip_lab.so: the same sources with -DFP_LAB (measurement
                                            # variants, hooks and FP_* environment toggles; loaded only by tools/ via _lib.use_lab())

hipcc cross-compiles without a GPU.  -ffp-contract=off: only explicit fmaf() calls become FMAs, so the
kernels that promise bit-exact agreement with oracle/fp_oracle.c really execute the documented op sequence.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIBDIR = ROOT / "lib"
OBJDIR = ROOT / "lib" / "obj"
LIB = LIBDIR / "libfreepose_hip.so"
LAB_LIB = LIBDIR / "libfreepose_hip_lab.so"
REPO = ROOT.parent

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
HIP_FLAGS = [
    "--offload-arch=gfx950:sramecc+", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    # gfx950 has one unified VGPR/AGPR file: keep MFMA accumulators in VGPRs so VALU epilogue / softmax-rescale work on
    # them needs no v_accvgpr_read/write round trips (attention: -128 moves per KV tile, occupancy 2 -> 3 waves/SIMD)
    "-mllvm", "-amdgpu-mfma-vgpr-form=1",
    "-Wno-unused-result", "-DNDEBUG",
]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_hip(force: bool = False, verbose: bool = True, lab: bool = False) -> Path:
    """the product library, or with lab=True the lab build (-DFP_LAB) next to it"""
    srcs = sorted(CSRC.glob("*.hip"))
    hdrs = sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.inc")) + [REPO / "include" / "freepose_hip.h"]
    objdir = LIBDIR / "obj_lab" if lab else OBJDIR
    lib_out = LAB_LIB if lab else LIB
    extra = ["-DFP_LAB"] if lab else []
    objdir.mkdir(parents=True, exist_ok=True)
    jobs = []
    for s in srcs:
        o = objdir / (s.stem + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC, *HIP_FLAGS, *extra, "-c", str(s), "-o", str(o)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s.name}:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[build] hipcc {s.name}", flush=True)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [objdir / (s.stem + ".o") for s in srcs]
    if force or jobs or _stale(lib_out, objs):
        cmd = [HIPCC, "--offload-arch=gfx950:sramecc+", "-shared", "-fPIC", "-o", str(lib_out), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[build] linked {lib_out.relative_to(REPO)}", flush=True)
    return lib_out


def build_oracle(force: bool = False, verbose: bool = True) -> Path:
    odir = REPO / "oracle"
    src = odir / "fp_oracle.c"
    lib = odir / "libfp_oracle.so"
    if src.exists() and (force or _stale(lib, [src])):
        cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
               "-o", str(lib), str(src), "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"gcc failed for oracle:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[build] gcc {src.name}", flush=True)
    return lib


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    force = "--force" in argv
    build_hip(force)
    if "--lab" in argv:
        build_hip(force, lab=True)
    build_oracle(force)


if __name__ == "__main__":
    main()
