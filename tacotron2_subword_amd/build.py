"""Builds libt2amd.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libt2amd.so")
SOURCES = ["gemm.hip", "lstm.hip", "chain.hip", "chain_bwd.hip", "chain_enc.hip", "attention.hip", "conv.hip", "elementwise.hip", "infer.hip", "optim.hip", "c_api.hip"]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "t2amd.h"))
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest(deps):
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the HIP extension")
    objs = []
    procs = []
    os.makedirs(os.path.join(_HERE, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(_HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) >= _newest(deps):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"] + os.environ.get("T2_EXTRA_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + out.decode(errors="replace"))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed: " + r.stdout.decode(errors="replace"))
    return OUT


if __name__ == "__main__":
    print(build(verbose=True))
