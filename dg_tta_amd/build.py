"""Builds libdgtta_hip.so (gfx950) in-tree with hipcc. No torch dependency; cross-compiles without a GPU."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = Path(__file__).resolve().parent / "libdgtta_hip.so"
SOURCES = ["lib.hip", "mind3d.hip", "gin.hip", "warp.hip", "softdice.hip", "dice_ce.hip", "adamw.hip", "resample.hip", "window_features.hip", "unet_ref.hip", "conv_mfma.hip", "conv_rows.hip", "conv_ring.hip",
           "conv_wgrad.hip", "conv_wgrad_ring.hip", "convt_gemm.hip", "conv_s2.hip"]
# conv_ring.hip: the 64-input-channel step body (432 MFMAs, 192 fragment reads, fully unrolled) is above hipcc's default
# pragma-unroll threshold; partially unrolled its register arrays are indexed dynamically and land in scratch
EXTRA_FLAGS = {"conv_ring.hip": ["-mllvm", "-pragma-unroll-threshold=262144"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def source_sha16():
    """sha256 (16 hex) over the kernel sources and the C ABI header: the stamp a profile summary carries, so that a bench line
    can tell whether the profile it quotes was measured on the kernels it is timing."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [CSRC.parents[1] / "include" / "dgtta.h"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _stale(out, deps):
    return (not out.exists()) or any(d.stat().st_mtime > out.stat().st_mtime for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = list(CSRC.glob("*.h")) + [CSRC.parents[1] / "include" / "dgtta.h"]
    objs, jobs = [], []
    for s in SOURCES:
        src, obj = CSRC / s, CSRC / (s + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(s, []), "-c", str(src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)])
    return LIB


ASAN_DIR = CSRC.parents[1] / "build" / "asan"
ASAN_FLAGS = ["--offload-arch=gfx950", "-O1", "-g", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",
              "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-Wno-unused-function"]


def build_asan(verbose=False):
    """Host-side sanitizer build of the C-ABI shim: the HOST code of every translation unit (argument
    checks, size queries, launch plans, dispatch) instrumented with AddressSanitizer + UndefinedBehaviorSanitizer, the
    device code left as it is (GPU sanitizers are not available on the pool).  Goes to build/asan/ (git-ignored), never
    into the package: the product always loads dg_tta_amd/libdgtta_hip.so.  tests/test_host_logic.py drives it."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    ASAN_DIR.mkdir(parents=True, exist_ok=True)
    hdrs = list(CSRC.glob("*.h")) + [CSRC.parents[1] / "include" / "dgtta.h"]
    lib = ASAN_DIR / "libdgtta_hip_asan.so"
    objs, jobs = [], []
    for s in SOURCES:
        src, obj = CSRC / s, ASAN_DIR / (s + ".o")
        objs.append(obj)
        if _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *ASAN_FLAGS, *EXTRA_FLAGS.get(s, []), "-c", str(src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, capture_output=not verbose)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    if jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-o", str(lib), *map(str, objs)])
    return lib


DIAG_DIR = CSRC.parents[1] / "build" / "diag"
DIAG_LIB = CSRC.parents[1] / "profiles" / "tools" / "libdgtta_hip_diag.so"


def build_diag(verbose=False):
    """The LABORATORY build (-DDGTTA_DIAG): the same sources plus the timing-model / cycle-stamp instantiations and the
    DGTTA_*_ABL, DGTTA_ROWS_VAR, DGTTA_RING_NT, DGTTA_WGRAD_RING_CLK switches that select them (results wrong by
    construction, stamps written behind the caller's buffers).  It lands beside the scripts that use it
    (profiles/tools/, loaded through DGTTA_LIB) and never in the package: the product library has none of this."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    DIAG_DIR.mkdir(parents=True, exist_ok=True)
    hdrs = list(CSRC.glob("*.h")) + [CSRC.parents[1] / "include" / "dgtta.h"]
    objs, jobs = [], []
    for s in SOURCES:
        src, obj = CSRC / s, DIAG_DIR / (s + ".o")
        objs.append(obj)
        if _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *FLAGS, "-DDGTTA_DIAG", *EXTRA_FLAGS.get(s, []), "-c", str(src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    if jobs or _stale(DIAG_LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(DIAG_LIB), *map(str, objs)])
    return DIAG_LIB


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan(verbose=True))
    elif "--diag" in sys.argv:
        print(build_diag(verbose=True))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
