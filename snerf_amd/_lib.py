"""ctypes binding of libsnerf_hip.so (the C-ABI declared in include/snerf_hip.h).

The prototypes are parsed from the header itself, so the Python side cannot
drift from the ABI.  There is NO fallback: if the shared library is missing
``load()`` raises, and every op in ``snerf_amd`` goes through ``load()``.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("SNERF_HIP_LIB") or os.path.join(_HERE, "lib", "libsnerf_hip.so")   # (the override: A/B builds of tools/probes)
HEADER_PATH = os.path.join(_REPO, "include", "snerf_hip.h")
IO_LIB_PATH = os.path.join(_HERE, "lib", "libsnerf_io.so")
IO_HEADER_PATH = os.path.join(_REPO, "include", "snerf_io.h")

_CTYPE = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double}
RESTYPE = {}          # name -> ctypes return type (int status for the entry points, long for the size queries)
_lib = None
_iolib = None


def parse_header(path: str = HEADER_PATH):
    """-> {name: [(ctype, argname), ...]} for every ``int snerf_*(...)`` declaration."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long)\s+(snerf_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(2), m.group(3).strip()
        RESTYPE[name] = _CTYPE[m.group(1)]
        sig = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    sig.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, an = a.rsplit(" ", 1)
                    sig.append((_CTYPE[ty.replace("const ", "").strip()], an))
        protos[name] = sig
    return protos


def load():
    """Load (once) and return the ctypes library with argtypes/restype set."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own libamdhip64: import it FIRST so that the dynamic linker binds libsnerf_hip.so to that already
    # loaded runtime (same SONAME).  Loading this library before torch would bring the system runtime into the process as a
    # second HIP instance, and kernels registered with one runtime cannot be launched on the other's streams.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libsnerf_hip.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C snerf_amd/csrc`).  snerf_amd has no CPU/PyTorch fallback for its kernels.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, sig in parse_header().items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = [t for t, _ in sig]
        fn.restype = RESTYPE[name]
    _lib = lib
    return lib


def load_io():
    """libsnerf_io.so (include/snerf_io.h): the host-side PNG writer of the frame writer.  Host-only, no GPU runtime."""
    global _iolib
    if _iolib is not None:
        return _iolib
    if not os.path.exists(IO_LIB_PATH):
        raise RuntimeError(f"libsnerf_io.so not found at {IO_LIB_PATH}: build it with `make -C snerf_amd/csrc`")
    lib = ctypes.CDLL(IO_LIB_PATH)
    for name, sig in parse_header(IO_HEADER_PATH).items():
        fn = getattr(lib, name)
        fn.argtypes = [t for t, _ in sig]
        fn.restype = ctypes.c_long if name == "snerf_png_encode" else ctypes.c_int
    _iolib = lib
    return lib


class SnerfHipError(RuntimeError):
    pass


_STATUS = {1: "bad argument", 2: "kernel launch failure"}


def query(name: str, *args):
    """value-returning entries (workspace size queries): no status to check"""
    return getattr(load(), name)(*args)


def call(name: str, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        extra = f", hipError_t {load().snerf_last_hip_error()}" if rc == 2 else ""
        raise SnerfHipError(f"{name} failed: status {rc} ({_STATUS.get(rc, 'unknown')}{extra})")
