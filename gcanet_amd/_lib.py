"""ctypes binding of libgcanet_hip.so (the C ABI declared in include/gcanet_hip.h).

The argument types of every entry point are derived from the header itself, so the
header is the single source of truth for the boundary.  There is NO fallback: if the
library is missing or a call fails, a RuntimeError is raised (the reference raises
RuntimeError via AT_ASSERT, P2/_ext-src/include/utils.h:5-25; it never silently degrades).
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "gcanet_hip.h")
SO_PATH = os.path.join(_HERE, "lib", "libgcanet_hip.so")

_PROTO = re.compile(r"^\s*(const\s+char\s*\*|int|long)\s*(gcn_\w+)\s*\(([^;]*?)\)\s*;", re.S | re.M)


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    protos = {}
    for ret, name, args in _PROTO.findall(text):
        argtypes = []
        for a in [x.strip() for x in args.replace("\n", " ").split(",")]:
            if a in ("void", ""):
                continue
            if "*" in a:
                argtypes.append(C.c_void_p)
            elif re.match(r"(const\s+)?float\b", a):
                argtypes.append(C.c_float)
            elif re.match(r"(const\s+)?double\b", a):
                argtypes.append(C.c_double)
            elif re.match(r"(const\s+)?(long|int64_t)\b", a):
                argtypes.append(C.c_int64)
            elif re.match(r"(const\s+)?(int|int32_t)\b", a):
                argtypes.append(C.c_int)
            else:
                raise ValueError("unhandled parameter %r in %s" % (a, name))
        protos[name] = (C.c_char_p if "char" in ret else (C.c_int64 if ret == "long" else C.c_int), argtypes)
    return protos


_lib = None


def lib():
    """Load the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        # torch must load ITS HIP runtime first: libgcanet_hip.so then binds to the same
        # libamdhip64 instead of pulling a second copy (two runtimes -> "no ROCm-capable device")
        import torch  # noqa: F401
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "gcanet_amd: %s not found -- build it with `python -m gcanet_amd.build` "
                "(there is no CPU fallback)" % SO_PATH)
        dll = C.CDLL(SO_PATH)
        for name, (res, args) in parse_header().items():
            fn = getattr(dll, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = dll
    return _lib


# optional per-entry-point device timing (bench.py): {key: [(start_event, end_event), ...]}
_TIMING = None


def enable_timing(flag=True):
    """Record a HIP event pair (on torch's current stream == the launch stream) around every call."""
    global _TIMING
    _TIMING = {} if flag else None


def timing_results():
    """-> {key: (n_calls, total_ms)}; call after torch.cuda.synchronize()."""
    out = {}
    for key, evs in (_TIMING or {}).items():
        out[key] = (len(evs), sum(a.elapsed_time(b) for a, b in evs))
    return out


def call(name, *args, tag=None):
    """Invoke an entry point; non-zero status -> RuntimeError with gcn_last_error()."""
    dll = lib()
    if _TIMING is not None and tag is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(dll, name)(*args)
        e1.record()
        _TIMING.setdefault(tag, []).append((e0, e1))
    else:
        rc = getattr(dll, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (status %d): %s" % (name, rc, dll.gcn_last_error().decode()))


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    if t is None:
        return None
    return t.data_ptr()


def stream_of(t):
    """Raw hipStream_t of torch's current stream on the tensor's device."""
    import torch
    idx = t.device.index
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice() if idx is None else idx)


class on_device:
    """`with on_device(t):` -- make t's GPU the current one for the call, like torch.cuda.device_of(t) but without its
    per-use Python cost when the device already is current (one process per GPU: always; an eager step makes ~200 such
    calls, and the host, not the GPU, bounds that mode)."""
    __slots__ = ("idx", "prev")

    def __init__(self, t):
        self.idx = t.device.index if t.is_cuda else None
        self.prev = -1

    def __enter__(self):
        if self.idx is not None:
            import torch
            cur = torch._C._cuda_getDevice()
            if cur != self.idx:
                self.prev = cur
                torch._C._cuda_setDevice(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            import torch
            torch._C._cuda_setDevice(self.prev)
            self.prev = -1
        return False


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("gcanet_amd: expected a GPU tensor (the reference asserts 'CPU not supported', "
                               "P2/_ext-src/src/group_points.cpp:31-33); got device %s" % t.device)
