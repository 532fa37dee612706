"""ctypes binding of libedrl_hip.so (the C-ABI declared in include/edrl_hip.h).

The prototypes are parsed from the header itself so the Python side cannot drift from the
C declarations.  There is NO fallback: if the library is missing or a call fails, this
module raises — the product path never routes around the HIP kernels.
"""
import ctypes
import os
import re

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_REPO_DIR = os.path.dirname(_PKG_DIR)
LIB_PATH = os.environ.get("EDRL_LIB_PATH") or os.path.join(_PKG_DIR, "libedrl_hip.so")   # override: A/B builds
HEADER_PATH = os.path.join(_REPO_DIR, "include", "edrl_hip.h")

_CTYPE = {
    "int": ctypes.c_int,
    "long": ctypes.c_long,
    "size_t": ctypes.c_size_t,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "hipStream_t": ctypes.c_void_p,
}


def parse_header(path=HEADER_PATH):
    """Return {name: (restype, [argtypes])} for every `edrl_*` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|long)\s+(edrl_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        for a in args.split(","):
            a = a.strip()
            if a in ("", "void"):
                continue
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
            else:
                argtypes.append(_CTYPE[a.split()[0]])
        protos[name] = (_CTYPE[ret], argtypes)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). The EDRL hot path has no non-HIP fallback."
            )
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.fn = {}
        for name, (ret, argtypes) in self.protos.items():
            f = getattr(self.cdll, name)  # AttributeError if the .so lacks a declared symbol
            f.restype = ret
            f.argtypes = argtypes
            self.fn[name] = f


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an `int edrl_*` launcher on torch's current stream; raise on any non-zero status."""
    rc = lib().fn[name](*args, stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc}")


def query(name, *args):
    """Invoke a `size_t edrl_*_bytes` helper."""
    return lib().fn[name](*args)


def set_switches(**kw):
    """Change library switches of this process (csrc/edrl_config.h): EDRL_<NAME>=value in the environment, then
    edrl_config_reload() -- the library reads its environment once, not per launch.  value None removes the variable.
    Call between launches only.  -> 1 if the loaded library holds the diagnostic kernels (a -DEDRL_DIAG build)."""
    for k, v in kw.items():
        if not k.startswith("EDRL_"):
            raise ValueError(f"not a library switch: {k}")
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    return lib().fn["edrl_config_reload"]()
