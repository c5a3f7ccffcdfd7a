"""Build libtsp2opt_hip.so in-tree with hipcc for gfx950 (MI355X).

    python -m ml4tspbench_amd.build [--force]

-ffp-contract=off is REQUIRED: the exact delta evaluation must round the two squares and their sum
separately, like the reference's separate torch kernels (search/two_opt.py:27).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRCS = [os.path.join(HERE, "csrc", f) for f in ("tsp2opt.hip", "greedy.hip", "greedy_sparse.hip", "ls_ops.hip")]
DEPS = [os.path.join(HERE, "csrc", f) for f in ("wave_reduce.hpp", "host_util.hpp", "two_opt_common.hpp",
                                                 "two_opt_kernel.hpp", "two_opt_band.hpp", "two_opt_aux.hpp")]
HDR = os.path.join(ROOT, "include", "tsp2opt.h")
LIB = os.path.join(HERE, "libtsp2opt_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC",
    "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"),
]
OBJ_DIR = os.path.join(ROOT, "build")
# headers each translation unit includes (an object is rebuilt when its source or one of these is newer)
UNIT_DEPS = {
    "tsp2opt.hip": ("wave_reduce.hpp", "host_util.hpp", "two_opt_common.hpp", "two_opt_kernel.hpp",
                    "two_opt_band.hpp", "two_opt_aux.hpp"),
    "greedy.hip": ("wave_reduce.hpp", "host_util.hpp", "greedy_walk.hpp"),
    "greedy_sparse.hip": ("wave_reduce.hpp", "host_util.hpp", "greedy_walk.hpp"),
    "ls_ops.hip": ("wave_reduce.hpp", "host_util.hpp"),
}


def build(force: bool = False, verbose: bool = False, fine: bool = False) -> str:
    """fine=True: the diagnostic variant libtsp2opt_hip_fine.so (-DTSP2OPT_FINE: sub-phase cycle marks,
    tools/phase_breakdown.py --fine); never the library that is benchmarked or shipped."""
    lib = LIB.replace(".so", "_fine.so") if fine else LIB
    newest = max(os.path.getmtime(f) for f in SRCS + DEPS + [HDR, __file__])
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= newest:
        return lib
    # one object per translation unit, compiled side by side (the sources share no device code), then linked
    os.makedirs(OBJ_DIR, exist_ok=True)
    defs = ["-DTSP2OPT_FINE"] if fine else []
    jobs, objs = [], []
    for src in SRCS:
        base = os.path.basename(src)
        obj = os.path.join(OBJ_DIR, base.replace(".hip", "_fine.o" if fine else ".o"))
        objs.append(obj)
        deps = [src, HDR, __file__] + [os.path.join(HERE, "csrc", h) for h in UNIT_DEPS[base]]
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(f) for f in deps):
            continue
        cmd = [HIPCC, *FLAGS, *defs, "-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, job in jobs:
        if job.wait() != 0:
            for _, other in jobs:
                if other.poll() is None:
                    other.wait()
            raise subprocess.CalledProcessError(job.returncode, cmd)
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, fine="--fine" in sys.argv))
