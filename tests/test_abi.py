"""The C-ABI library loads on a CPU-only box and exports every symbol include/tcmi.h declares."""
import ctypes as C
import os
import re

import pytest

from trueconsense_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="tcmi.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tcmi_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _ffi.lib()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
        assert n in _ffi._SIGS, "no ctypes signature for " + n
    assert lib.tcmi_abi_version() == 5


def test_every_header_under_include_is_covered():
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["tcmi.h", "tcmi_rccl.h"]


def test_rccl_hook_library_exports_what_its_header_declares():
    """include/tcmi_rccl.h -> trueconsense_amd/lib/libtcmi_rccl.so, built by build(): tcmi_split_step's reduce hook over RCCL
    (no collective is called here: that needs a GPU, tests/test_rccl.py).  The hook is optional (csrc/Makefile builds it where RCCL's
    header and library are installed): a box without RCCL has no libtcmi_rccl.so, and libtcmi.so does not need it."""
    if not os.path.exists(_ffi.RCCL_LIB_PATH) and not os.path.exists(os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include", "rccl", "rccl.h")):
        pytest.skip("no RCCL on this box: the hook library is not built")
    lib = _ffi.rccl_lib()
    names = declared_symbols("tcmi_rccl.h")
    assert len(names) == 6 and "tcmi_rccl_reduce" in names, names
    for n in names:
        assert hasattr(lib, n), n
        assert n in _ffi._RCCL_SIGS, "no ctypes signature for " + n
    assert lib.tcmi_rccl_reduce(None, None, 0, None) != 0 and b"tcmi_rccl_reduce" in lib.tcmi_rccl_last_error()
    src = open(os.path.join(ROOT, "trueconsense_amd", "csrc", "rccl_hook.cpp")).read()
    assert "TCMI_RCCL_ID_BYTES" in src and _ffi.RCCL_ID_BYTES == 128


def test_no_device_is_reported_not_hidden():
    n = _ffi.device_count()
    if n > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = _ffi.lib().tcmi_ctx_create(0, C.byref(h))
    assert rc == _ffi.E_NODEVICE
    assert b"no CPU path" in _ffi.lib().tcmi_last_error(None)
    from trueconsense_amd.engine import Context
    with pytest.raises(_ffi.TcmiError):
        Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "trueconsense_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "tc_oracle" not in src and "libtcoracle" not in src, f
