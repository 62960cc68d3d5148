"""Build libramp_hip.so (gfx950) in-tree with hipcc.  `python -m ramp_amd.build` or build()."""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libramp_hip.so")
TOOLS_LIB = os.path.join(LIBDIR, "libramp_hip_tools.so")
# the product library: kernels + engine.hip (context, schedule, sampler, graphs) + ops.hip (context-free kernel-level entry points)
SOURCES = ["gemm.hip", "ffx.hip", "ffx16.hip", "tkl.hip", "tkl16.hip", "atk.hip", "atb.hip", "atl.hip", "tkc.hip", "tkw.hip", "rowops.hip", "attention.hip", "sampler.hip", "scene.hip", "metrics.hip", "engine.hip", "ops.hip"]
# the tools library = the same objects + the micro-benchmark / stress harness (ramp_bench_gemm, ramp_stress_gemm): tests/ and ramp_amd/tools/ only
TOOLS_SOURCES = ["bench.hip"]
# default GEMM mode 2 = fp16x3 split with delayed operand scaling, 1 = bf16x6 split (both fp32-accurate, see gemm.hip);
# 0 = exact fp32 MFMA
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-DRAMP_DEFAULT_GEMM_MODE=2"]
# sampler.hip mirrors the reference's elementwise fp32 expressions rounding for rounding: hipcc's default
# -ffp-contract=fast would fuse a*b - c*d into an FMA (HIP's __fmul_rn is a plain multiply), so it is off there.
# ffx.hip: hipcc's SLP vectoriser pairs the GEGLU elements into v_pk_* instructions, which issue slower beside MFMAs and
# bunch the elementwise work in front of a slab's first MFMA
EXTRA_FLAGS = {"sampler.hip": ["-ffp-contract=off"], "ffx.hip": ["-fno-slp-vectorize"], "ffx16.hip": ["-fno-slp-vectorize"], "tkl.hip": ["-fno-slp-vectorize"], "tkl16.hip": ["-fno-slp-vectorize"], "atk.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"], "atb.hip": ["-fno-slp-vectorize"], "atl.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"], "tkc.hip": ["-fno-slp-vectorize"], "tkw.hip": ["-fno-slp-vectorize"]}


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def header_deps():
    """Every shared header / include file a translation unit may pull in (csrc/*.h, csrc/*.inc, the public headers) plus this
    script (its flags)."""
    import glob
    hs = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")))
    return hs + [os.path.join(HERE, "..", "include", "ramp_hip.h"), os.path.join(HERE, "..", "include", "ramp_hip_tools.h"), os.path.abspath(__file__)]


_INCLUDE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def source_deps(path: str):
    """The files ONE translation unit depends on: itself, every file its `#include "..."` lines reach (followed recursively, resolved
    against the including file's directory, as hipcc does) and this script.  A kernel file includes core.h + its family's argument
    header, so an edit to one family's arguments recompiles that family, engine.hip, ops.hip and bench.hip -- not every object."""
    seen, todo = [], [os.path.normpath(path)]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        with open(f, encoding="utf-8") as fh:
            for inc in _INCLUDE.findall(fh.read()):
                todo.append(os.path.normpath(os.path.join(os.path.dirname(f), inc)))
    return seen + [os.path.abspath(__file__)]


def stale_sources(force: bool = False):
    """(source, object) pairs that build() would recompile now."""
    objdir = os.path.join(LIBDIR, "obj")
    jobs = []
    for src in SOURCES + TOOLS_SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(op, source_deps(sp)):
            jobs.append((sp, op))
    return jobs


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = stale_sources(force)

    def compile_one(job):
        sp, op = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(sp), []) + ["-c", sp, "-o", op]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp}:\n{r.stderr[-4000:]}")
        return sp

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            for sp in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[ramp_amd.build] compiled {os.path.basename(sp)}", file=sys.stderr)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    tobjs = objs + [os.path.join(objdir, s.replace(".hip", ".o")) for s in TOOLS_SOURCES]
    for lib, oo in ((LIB, objs), (TOOLS_LIB, tobjs)):
        if force or jobs or _stale(lib, oo):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + oo
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
            if verbose:
                print(f"[ramp_amd.build] linked {lib}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
