"""CPU tests of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/hallo_amd.h declares; argument validation works without touching a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from hallo_amd import build, lib as hl
    build.build()
    return hl.load()


def test_header_symbols_are_exported(lib):
    from hallo_amd import lib as hl
    hdr = open(os.path.join(ROOT, "include", "hallo_amd.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t)\s+(hallo_\w+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed from include/hallo_amd.h"
    assert declared == set(hl.SYMBOLS), (declared ^ set(hl.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hallo_abi_version() == 3


def test_struct_sizes_match_header(lib):
    """The ctypes mirrors must have the layout a C compiler gives the header's structs."""
    import subprocess
    import tempfile
    from hallo_amd import lib as hl
    src = '#include "hallo_amd.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu\\n", sizeof(hallo_gemm_desc),' \
          ' sizeof(hallo_conv_desc), sizeof(hallo_attn_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    assert [int(v) for v in out] == [C.sizeof(hl.GemmDesc), C.sizeof(hl.ConvDesc), C.sizeof(hl.AttnDesc)]


def test_argument_validation_returns_einval(lib):
    from hallo_amd import lib as hl
    d = hl.GemmDesc()
    assert lib.hallo_gemm(C.byref(d), None) == -22          # null pointers
    a = hl.AttnDesc()
    assert lib.hallo_attention(C.byref(a), None) == -22
    c = hl.ConvDesc()
    assert lib.hallo_conv3x3_nhwc(C.byref(c), None) == -22
    assert lib.hallo_temporal_attention(None, None, 1, 18, 64, 320, 8, 0.1, 1, None) == -22
    assert lib.hallo_groupnorm_chunks(4096) == 64 and lib.hallo_groupnorm_chunks(4) == 1
    assert lib.hallo_set_option(b"no_such_option", 1) == -22 and lib.hallo_set_option(b"gemm_variant", 2) == 0
    assert lib.hallo_set_option(b"gemm_variant", 9) == -22 and lib.hallo_set_option(b"gemm_variant", 6) == 0   # back to auto
    assert lib.hallo_groupnorm_chunks(64 * 64 * 64) == 64


def test_missing_library_fails_loudly(tmp_path):
    from hallo_amd import lib as hl
    with pytest.raises(hl.HalloLibraryError):
        hl.load(str(tmp_path / "nope.so"))


def test_cpu_tensors_are_rejected():
    """There is no CPU fallback: operators refuse host tensors."""
    import torch
    from hallo_amd import ops, lib as hl
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(hl.HalloLibraryError):
        ops.gemm(a, a)
