"""ctypes binding of libcxrmate_hip.so. The prototypes are parsed from include/cxrmate_hip.h, so the header is the single
source of truth for the C ABI. There is NO fallback: if the library is missing the product path raises."""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "cxrmate_hip.h")
LIB_PATH = os.environ.get("CXR_LIB") or os.path.join(HERE, "lib", "libcxrmate_hip.so")      # CXR_LIB: a differently BUILT library (A/B of build-time switches)

_CTYPES = {"long": ctypes.c_long, "int": ctypes.c_int, "float": ctypes.c_float, "unsigned int": ctypes.c_uint,
           "hipStream_t": ctypes.c_void_p}


def parse_header(path: str = HEADER):
    """-> {name: [(ctype, argname), ...]} for every `int cxr_*(...)` prototype."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(cxr_\w+)\s*\(([^)]*)\)\s*;", text):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            if not a or a == "void":
                continue
            if "*" in a:
                args.append((ctypes.c_void_p, a.split("*")[-1].strip()))
            else:
                ty, name = a.rsplit(" ", 1)
                args.append((_CTYPES[ty.replace("const ", "").strip()], name))
        protos[m.group(1)] = args
    return protos


class CxrError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        self._dll = None
        self._fns = {}
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise CxrError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(cxrmate_amd has no CPU / PyTorch fallback)")
            if os.environ.get("CXR_LIB"):
                import warnings
                warnings.warn(f"CXR_LIB is set: loading {LIB_PATH} instead of the in-tree libcxrmate_hip.so (A/B of build-time switches)", RuntimeWarning, stacklevel=2)
            import torch  # noqa: F401  -- torch's HIP runtime must be resident first so that the library binds to the same one
            self._dll = ctypes.CDLL(LIB_PATH)
            for name, args in self.protos.items():
                fn = getattr(self._dll, name)
                fn.restype = ctypes.c_int
                fn.argtypes = [t for t, _ in args]
        return self._dll

    def call(self, name: str, *args):
        fn = self._fns.get(name)
        if fn is None:
            fn = self._fns[name] = getattr(self.load(), name)
        rc = fn(*args)
        if rc != 0:
            detail = ""
            if rc == -2:
                f = self._dll.cxr_last_hip_error_string
                f.restype = ctypes.c_char_p
                detail = f" [hip error {self._dll.cxr_last_hip_error()}: {f().decode()}]"
            raise CxrError(f"{name} failed with code {rc}{detail} (args: {args})")


LIB = _Lib()
