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


def parse_header(path=HEADER_PATH, names=False):
    """Return {name: (restype, [argtypes])} for every `edrl_*` prototype in the header (names=True: {name: [parameter names]})."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|long)\s+(edrl_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes, argnames = [], []
        for a in args.split(","):
            a = a.strip()
            if a in ("", "void"):
                continue
            argnames.append(re.split(r"[\s*]+", a)[-1])
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
            else:
                argtypes.append(_CTYPE[a.split()[0]])
        protos[name] = argnames if names else (_CTYPE[ret], argtypes)
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
        self.argnames = None          # parameter names, parsed on first use by the call tracer
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


# ---- call tracer (measurement aid, off by default): scripts/step_trace.py cuts a rocprofv3 kernel trace / --pmc pass of a whole
# training step into library calls.  While tracing, every call() is preceded by an empty marker dispatch (edrl_trace_mark) and
# logged with its scalar arguments (pointers as 0 / 1 = NULL / given) plus the annotation ops._launch_timed / call_timed_bytes
# left for it (kind, algorithmic flops and bytes).  The k-th marker row of the trace precedes the kernels of the k-th record.
_trace = None
_trace_note = None


def trace_begin():
    global _trace, _trace_note
    _trace, _trace_note = [], None


def trace_end():
    """-> the list of call records since trace_begin(); tracing is off afterwards."""
    global _trace, _trace_note
    t, _trace, _trace_note = _trace, None, None
    return t


def tracing():
    return _trace is not None


def trace_note(**kw):
    """Annotation for the NEXT traced call (consumed by it)."""
    global _trace_note
    if _trace is not None:
        _trace_note = kw


def _trace_call(name, args):
    global _trace_note
    l = lib()
    l.fn["edrl_trace_mark"](stream())
    argtypes = l.protos[name][1]
    if l.argnames is None:
        l.argnames = parse_header(names=True)
    rec = {"name": name, "args": {n: ((0 if a is None else 1) if t is ctypes.c_void_p else a)
                                  for n, a, t in zip(l.argnames[name], args, argtypes)}}
    if _trace_note is not None:
        rec.update(_trace_note)
        _trace_note = None
    _trace.append(rec)


# K-split slabs of the fp32 gather family (include/edrl_hip.h edrl_gather_ksplit_set_workspace): the library allocates nothing,
# so the first library call a (device, stream) pair issues registers a 16 MiB slab for it, taken from the torch caching allocator
# and kept for the life of the process.  Without one the family runs unsplit on that stream.
_ksplit_slabs = {}


def _register_ksplit_slab(key, st):
    l = lib()
    nbytes = l.fn["edrl_gather_ksplit_workspace_bytes"]()
    slab = torch.empty(nbytes // 4, device=torch.device("cuda", key[0]), dtype=torch.float32)
    rc = l.fn["edrl_gather_ksplit_set_workspace"](slab.data_ptr(), nbytes, st)
    if rc != 0:
        raise RuntimeError(f"edrl_gather_ksplit_set_workspace failed with status {rc}")
    _ksplit_slabs[key] = slab


def call(name, *args):
    """Invoke an `int edrl_*` launcher on torch's current stream; raise on any non-zero status."""
    st = stream()
    key = (torch.cuda.current_device(), st)
    if key not in _ksplit_slabs:
        _register_ksplit_slab(key, st)
    if _trace is not None:
        _trace_call(name, args)
    rc = lib().fn[name](*args, st)
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc}")


def query(name, *args):
    """Invoke a `size_t edrl_*_bytes` helper."""
    return lib().fn[name](*args)


def library_switches():
    """Names of the EDRL_* switches csrc/config.hip reads (the ones edrl_config_reload() re-reads), parsed from its source."""
    src = open(os.path.join(_PKG_DIR, "csrc", "config.hip")).read()
    return set(re.findall(r'env_(?:int|long)\("(EDRL_\w+)"', src))


def set_switches(**kw):
    """Change library switches of this process (csrc/edrl_config.h): EDRL_<NAME>=value in the environment, then
    edrl_config_reload() -- the library reads its environment once, not per launch.  value None removes the variable.
    Call between launches only.  -> 1 if the loaded library holds the diagnostic kernels (a -DEDRL_DIAG build)."""
    known = library_switches()
    for k, v in kw.items():
        if k not in known:
            raise ValueError(f"not a library switch: {k} (switches the library re-reads: {sorted(known)}; the Python-level "
                             "EDRL_* switches of encoders.py / ops.py / train.py are read at import and cannot be flipped here)")
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    return lib().fn["edrl_config_reload"]()
