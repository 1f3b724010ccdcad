"""CPU-side check of the drop-in boundary: libntg_amd.so loads and exports every symbol that
include/ntg.h and include/ntg_amd.h declare (no compute calls without a GPU)."""
import ctypes as C
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//.*", "", txt)
    txt = re.sub(r"#.*", "", txt)
    names = set()
    for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\((?!\s*\*)", txt):
        n = m.group(1)
        if n in ("void", "int", "double", "if", "sizeof", "defined") or n.isupper():
            continue
        names.add(n)
    # function-pointer parameter names are not functions
    return {n for n in names if n not in ("nlicf", "nltcf", "nlfcf", "icf", "ucf", "fcf")}


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    return C.CDLL(os.path.join(ROOT, "ntg_amd", "libntg_amd.so"))


@pytest.mark.parametrize("header", ["ntg.h", "ntg_amd.h"])
def test_every_declared_symbol_is_exported(lib, header):
    names = declared_functions(header)
    assert len(names) >= 8
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, f"{header}: not exported: {missing}"


def test_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ntg_amd import api, configs
    with pytest.raises(api.NtgError):
        api.Plan(configs.config_K0())
