"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/sea_current_hip.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

import sea_current_amd as sc


@pytest.fixture(scope="module")
def built():
    sc.build()
    return ctypes.CDLL(sc.LIB_PATH)


def _declared():
    src = open(sc.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built):
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(built, n), f"{n} declared in include/sea_current_hip.h but not exported"
    assert sorted(sc.EXPORTS) == names  # the Python binding covers the whole ABI


def test_abi_version_and_strings(built):
    built.sc_abi_version.restype = ctypes.c_int
    assert built.sc_abi_version() == 1
    built.sc_status_string.restype = ctypes.c_char_p
    assert built.sc_status_string(0) == b"ok"
    assert built.sc_status_string(1) == b"invalid argument"


def test_no_gpu_fails_loudly(built):
    """Without a GPU the product path must fail, not fall back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    built.sc_ctx_create.restype = ctypes.c_int
    st = built.sc_ctx_create(0, ctypes.byref(h))
    assert st != 0 and not h.value


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under sea-current_amd/ may reference it."""
    bad = []
    for root, _, files in os.walk(sc.NATIVE_DIR):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"(import\s+oracle|from\s+oracle|libsc_oracle|sc_oracle\.h\"|sco_)", txt) and "sc_oracle.h)" not in txt:
                    for line in txt.splitlines():
                        if re.search(r"(import\s+oracle|from\s+oracle|libsc_oracle|#include.*sc_oracle|sco_[a-z])", line):
                            bad.append((f, line.strip()))
    assert not bad, bad
