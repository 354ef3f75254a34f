"""Build the engine's shared library in-tree with hipcc for gfx950 (csrc/Makefile)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_engine(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/engine.hip -> libfluid_engine.so (cross-compiles without a GPU).
    Returns the library path; raises CalledProcessError with the compiler output on failure."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc]
    if force:
        cmd.append("-B")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise subprocess.CalledProcessError(res.returncode, cmd, output=res.stdout)
    lib = os.path.join(_HERE, "libfluid_engine.so")
    if not os.path.exists(lib):
        raise FileNotFoundError(lib)
    return lib
