"""In-tree build of libbhr_hip.so (hipcc --offload-arch=gfx950; cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")


def library_path() -> str:
    """In-tree library; BHR_HIP_LIBRARY selects another build of the same ABI (A/B experiments)."""
    return os.environ.get("BHR_HIP_LIBRARY") or os.path.join(LIB_DIR, "libbhr_hip.so")


def sources():
    inc = os.path.join(os.path.dirname(_HERE), "include")
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", "Makefile"))) + \
        sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h"))


def is_stale() -> bool:
    lib = library_path()
    if not os.path.isfile(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(s) > t for s in sources())


def build_library(force: bool = False, jobs: int = 4, extra: str = "") -> str:
    """Compile every HIP translation unit for gfx950 and link the shared library."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    cmd = ["make", "-C", CSRC, f"-j{jobs}", "all"]
    if extra:
        cmd.append(f"EXTRA={extra}")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    if not os.path.isfile(library_path()):
        raise RuntimeError("libbhr_hip.so was not produced")
    return library_path()
