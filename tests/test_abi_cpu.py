"""CPU: libttvdm.so loads and exports every symbol include/ttvdm.h declares (no compute calls)."""
import os
import re

import pytest

from this_and_that_vdm_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_header_and_binding_agree(lib):
    text = open(os.path.join(REPO, "include", "ttvdm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(tt_[a-z0-9_]+)\s*\(", text))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ttvdm.h but not exported"


def test_version_and_arch(lib):
    assert lib.tt_abi_version() == 5
    assert lib.tt_target_arch() == b"gfx950"


def test_struct_layout_matches_header():
    """ctypes mirrors of TtGemmArgs / TtAttnArgs must have the C compiler's size."""
    import subprocess, tempfile, ctypes
    src = '#include <stdio.h>\n#include "ttvdm.h"\nint main(){printf("%zu %zu\\n", sizeof(TtGemmArgs), sizeof(TtAttnArgs));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe])
        a, b = map(int, subprocess.check_output([exe]).split())
    assert a == ctypes.sizeof(_lib.TtGemmArgs) and b == ctypes.sizeof(_lib.TtAttnArgs)


def test_ops_refuse_cpu_tensors(lib):
    import torch
    from this_and_that_vdm_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(8, 16, dtype=torch.float16), torch.zeros(8, 16, dtype=torch.float16))
