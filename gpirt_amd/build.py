"""Build libgpirt_hip.so in-tree with hipcc (gfx950).  No torch, no cmake: `make -C gpirt_amd/csrc`."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgpirt_hip.so")


def build(force: bool = False, jobs: int = 8) -> str:
    cmd = ["make", "-C", CSRC, f"-j{jobs}", "-s"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    if not os.path.exists(LIB):
        raise RuntimeError("hipcc build did not produce " + LIB)
    return LIB


def build_fences(jobs: int = 8) -> str:
    """The fenced variant of the panel kernel's hand-off (test-only second library, csrc/Makefile `fences`)."""
    subprocess.check_call(["make", "-C", CSRC, f"-j{jobs}", "-s", "fences"])
    return os.path.join(HERE, "libgpirt_hip_fences.so")


if __name__ == "__main__":
    print(build())
