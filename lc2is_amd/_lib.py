"""ctypes binding of ``liblc2is_hip.so`` (the C ABI declared in ``include/lc2is_hip.h``).

The product path has no fallback: if the shared library is missing, or a kernel launcher refuses its
arguments, a ``RuntimeError`` is raised (the reference's convention is that every failure is a Python
exception).  Nothing in this module imports ``oracle``.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_SO = Path(os.environ.get("LC2IS_LIB", _PKG / "liblc2is_hip.so"))   # LC2IS_LIB: A/B an alternative build of the same ABI
_HEADER = _PKG.parent / "include" / "lc2is_hip.h"

_ERR = {
    -1: "bad shape / leading dimension / divisibility",
    -2: "required pointer is NULL",
    -3: "unsupported configuration",
    -4: "workspace too small",
    -5: "HIP launch error",
}

_lib = None


def lib_path() -> Path:
    return _SO


def header_symbols() -> list[str]:
    """Every function name declared in include/lc2is_hip.h (used by the CPU-side export test)."""
    text = _HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lc2is_[a-z0-9_]+)\s*\(", text)))


def load() -> C.CDLL:
    """Load the HIP library once; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not _SO.exists():
            raise RuntimeError(
                f"{_SO} is missing: build it with `make -C {_PKG / 'csrc'}` or "
                "`python -c 'import __graft_entry__ as g; g.build()'` — lc2is_amd has no CPU/eager fallback."
            )
        _lib = C.CDLL(str(_SO))
        _lib.lc2is_version.restype = C.c_char_p
        for name in header_symbols():
            fn = getattr(_lib, name)  # AttributeError here = header/library mismatch
            if name.endswith("_bytes"):
                fn.restype = C.c_size_t
            elif name != "lc2is_version":
                fn.restype = C.c_int
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"lc2is_hip: {what} refused: {_ERR.get(rc, rc)} (rc={rc})")


def version() -> str:
    return load().lc2is_version().decode()
