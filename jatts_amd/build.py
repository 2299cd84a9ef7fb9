"""In-tree build of libjatts_hip.so (hipcc --offload-arch=gfx950; cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(jobs=4, verbose=False):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    path = os.path.join(_HERE, "lib", "libjatts_hip.so")
    if not os.path.exists(path):
        raise RuntimeError("build did not produce " + path)
    return path
