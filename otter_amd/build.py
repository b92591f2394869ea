"""Build libotter_hip.so (gfx950 only) in-tree with hipcc.  `python -m otter_amd.build [--force]`.

The .so is git-ignored but travels with the gpurun snapshot; the product refuses to run without it (no fallback)."""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libotter_hip.so")
SOURCES = ["gemm.hip", "norm.hip", "attn.hip", "elementwise.hip", "flash.hip", "optim.hip", "attn_mfma.hip", "loss.hip", "fuyu.hip", "decode.hip", "probe.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", "-Wno-unused-value"]
# gemm.hip: the K-major instantiations of variant 26 keep their accumulators in explicit AGPRs behind asm MFMAs; hipcc must not use the
# AGPR half as spill space of its own there (it would, between the K loop and the tail's read-back: measured as wrong blocks)
EXTRA = {"gemm.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_diag(mask: int, verbose: bool = True) -> str:
    """Ablation build of the GEMM (-DOTTER_DIAG=mask, see csrc/gemm.hip) -> lib/libotter_hip_diag<mask>.so.  Only
    tools/gemm_ablate.py loads these (through OTTER_LIB_PATH); their results are wrong by construction."""
    os.makedirs(LIBDIR, exist_ok=True)
    build(verbose=verbose)  # the other objects
    cc = hipcc()
    obj = os.path.join(LIBDIR, "gemm_diag%d.o" % mask)
    out = os.path.join(LIBDIR, "libotter_hip_diag%d.so" % mask)
    subprocess.check_call([cc, *FLAGS, *EXTRA["gemm.hip"], "-DOTTER_DIAG=%d" % mask, "-c", os.path.join(CSRC, "gemm.hip"), "-o", obj])
    others = [os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in SOURCES if s != "gemm.hip"]
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj, *others])
    return out


def build_experimental(verbose: bool = True) -> str:
    """Tools-only library with every GEMM schedule of rounds 1-2 (-DOTTER_EXPERIMENTAL: variants 4-12, 14-23, 27-29) ->
    lib/libotter_hip_experimental.so.  tools/gemm_*.py load it through OTTER_LIB_PATH; the product never does."""
    build(verbose=verbose)
    cc = hipcc()
    obj = os.path.join(LIBDIR, "gemm_experimental.o")
    out = os.path.join(LIBDIR, "libotter_hip_experimental.so")
    subprocess.check_call([cc, *FLAGS, *EXTRA["gemm.hip"], "-DOTTER_EXPERIMENTAL", "-c", os.path.join(CSRC, "gemm.hip"), "-o", obj])
    others = [os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in SOURCES if s != "gemm.hip"]
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj, *others])
    return out


def build_gemm_define(define: str, suffix: str, verbose: bool = True) -> str:
    """Debug build of gemm.hip with one extra -D (e.g. OTTER_KMDBG=1) -> lib/libotter_hip_<suffix>.so (tools only, via OTTER_LIB_PATH)."""
    build(verbose=verbose)
    cc = hipcc()
    obj = os.path.join(LIBDIR, "gemm_%s.o" % suffix)
    out = os.path.join(LIBDIR, "libotter_hip_%s.so" % suffix)
    subprocess.check_call([cc, *FLAGS, *EXTRA["gemm.hip"], "-D" + define, "-c", os.path.join(CSRC, "gemm.hip"), "-o", obj])
    others = [os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in SOURCES if s != "gemm.hip"]
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj, *others])
    return out


def build_flash_define(define: str, suffix: str, verbose: bool = True) -> str:
    """A/B build of flash.hip with one extra -D (e.g. OTTER_FLASH_SAFE_DMA) -> lib/libotter_hip_<suffix>.so (tools only, via OTTER_LIB_PATH)."""
    build(verbose=verbose)
    cc = hipcc()
    obj = os.path.join(LIBDIR, "flash_%s.o" % suffix)
    out = os.path.join(LIBDIR, "libotter_hip_%s.so" % suffix)
    subprocess.check_call([cc, *FLAGS, "-D" + define, "-c", os.path.join(CSRC, "flash.hip"), "-o", obj])
    others = [os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in SOURCES if s != "flash.hip"]
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj, *others])
    return out


def build_source_define(source: str, define: str, suffix: str, verbose: bool = True) -> str:
    """A/B build of one source file with one extra -D -> lib/libotter_hip_<suffix>.so (tools only, via OTTER_LIB_PATH)."""
    build(verbose=verbose)
    cc = hipcc()
    obj = os.path.join(LIBDIR, "%s_%s.o" % (source.replace(".hip", ""), suffix))
    out = os.path.join(LIBDIR, "libotter_hip_%s.so" % suffix)
    subprocess.check_call([cc, *FLAGS, *EXTRA.get(source, []), "-D" + define, "-c", os.path.join(CSRC, source), "-o", obj])
    others = [os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in SOURCES if s != source]
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj, *others])
    return out


def build_flash_timing(verbose: bool = True) -> str:
    """Diagnostics build with the in-kernel timeline of the flash forward (-DOTTER_FLASH_TIMING) ->
    lib/libotter_hip_flashtiming.so; only tools/flash_timeline.py loads it."""
    build(verbose=verbose)
    cc = hipcc()
    obj = os.path.join(LIBDIR, "flash_timing.o")
    out = os.path.join(LIBDIR, "libotter_hip_flashtiming.so")
    subprocess.check_call([cc, *FLAGS, "-DOTTER_FLASH_TIMING", "-c", os.path.join(CSRC, "flash.hip"), "-o", obj])
    others = [os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in SOURCES if s != "flash.hip"]
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj, *others])
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(os.path.dirname(HERE), "include", "otter_hip.h"),
            os.path.join(CSRC, "gemm_t4_ktile.inc")]   # (generated K-tile schedules of gemm.hip: tools/gen/gemm_t4_schedule.py inc)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and _newer(LIB, srcs + hdrs):
        return LIB
    cc = hipcc()
    objs = [os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in SOURCES]

    def one(pair):
        src, obj = pair
        if not force and _newer(obj, [src] + hdrs):
            return obj
        cmd = [cc, *FLAGS, *EXTRA.get(os.path.basename(src), []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(one, zip(srcs, objs)))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--define" in sys.argv:
        i = sys.argv.index("--define")
        print(build_gemm_define(sys.argv[i + 1], sys.argv[i + 2]))
    elif "--flash-define" in sys.argv:
        i = sys.argv.index("--flash-define")
        print(build_flash_define(sys.argv[i + 1], sys.argv[i + 2]))
    elif "--src-define" in sys.argv:      # --src-define norm.hip OTTER_NORM_NT=1 normnt
        i = sys.argv.index("--src-define")
        print(build_source_define(sys.argv[i + 1], sys.argv[i + 2], sys.argv[i + 3]))
    elif "--experimental" in sys.argv:
        print(build_experimental())
    elif "--flash-timing" in sys.argv:
        print(build_flash_timing())
    elif "--diag" in sys.argv:
        print(build_diag(int(sys.argv[sys.argv.index("--diag") + 1])))
    else:
        print(build(force="--force" in sys.argv))
