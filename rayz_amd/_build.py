"""In-tree build of the gfx950 library (hipcc cross-compiles without a GPU).

`build()` produces rayz_amd/csrc/librayz_hip.so = HIP kernels + the C ABI of include/rayz_hip.h +
the host mirror's C view (include/rayz_host.h), and the `rayz` command-line driver next to it.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(CSRC, "librayz_hip.so")
CLI = os.path.join(HOST, "rayz")

# -ffp-contract=off: FMAs appear only where the source writes them (bit parity with the oracle).
# f32 divide and sqrt stay correctly rounded, denormals are kept.
HIPFLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fno-gpu-flush-denormals-to-zero", "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wextra",
]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force: bool = False, verbose: bool = False) -> str:
    lib_src = [os.path.join(CSRC, "rayz_hip.hip"), os.path.join(HOST, "rayz_host.cpp")]
    deps = lib_src + [
        os.path.join(CSRC, "rayz_device.hpp"), os.path.join(CSRC, "bvh_build.hpp"), os.path.join(HOST, "rayz.hpp"),
        os.path.join(ROOT, "include", "rayz_hip.h"), os.path.join(ROOT, "include", "rayz_host.h"),
        os.path.abspath(__file__),
    ]
    extra = os.environ.get("RAYZ_EXTRA_HIPFLAGS", "").split()  # experiments only (e.g. -DRAYZ_GROUP=8)
    if force or extra or _stale(LIB, deps):
        cmd = [_hipcc(), *HIPFLAGS, *extra, "-shared", "-o", LIB, *lib_src]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    cli_src = os.path.join(HOST, "rayz_main.cpp")
    if os.path.exists(cli_src) and (force or _stale(CLI, [cli_src, LIB] + deps)):
        cmd = [_hipcc(), "-O2", "-std=c++17", "-ffp-contract=off", "-o", CLI, cli_src, "-L" + CSRC, "-lrayz_hip",
               "-Wl,-rpath,$ORIGIN/../csrc"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
