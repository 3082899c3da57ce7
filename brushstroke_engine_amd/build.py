"""Ahead-of-time build of the gfx950 kernels into one C-ABI shared library (no JIT, no torch headers).

    python -m brushstroke_engine_amd.build            # build if stale
    python -m brushstroke_engine_amd.build --force

The reference JIT-compiles its plugins at first use (torch_utils/custom_ops.py:46-124); here the
library is built in-tree with ``hipcc --offload-arch=gfx950`` so that it travels with the source
snapshot and is visible as a loaded ``.so`` to whoever audits the process.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libneube_hip.so")
SOURCES = ["nb_ops.hip", "nb_modconv.hip", "nb_modconv_h3.hip", "nb_modconv_small.hip", "nb_grad.hip", "nb_canvas.hip", "nb_encoder.hip"]
HEADERS = ["nb_common.h", "nb_torgb.h", os.path.join("..", "..", "include", "neube_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"]


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
