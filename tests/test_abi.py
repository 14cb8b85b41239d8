"""The C-ABI shared library builds in-tree, loads, and exports every symbol that
include/chromosight_hip.h declares.  No compute calls (no GPU here)."""
import ctypes
import pathlib
import re

import pytest

from chromosight_amd import _lib

ROOT = pathlib.Path(__file__).resolve().parents[1]


def header_functions():
    text = (ROOT / "include" / "chromosight_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_functions() == list(_lib.ABI_SYMBOLS)


def test_library_exports_every_symbol():
    lib = _lib.load_library()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.cs_version()


def test_no_cpu_fallback_without_gpu():
    """Creating a device context without a GPU must fail loudly, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.HipLibraryError):
        _lib.Device(0)


def test_struct_layouts_match_header():
    # sizes implied by the field lists of include/chromosight_hip.h on LP64
    assert ctypes.sizeof(_lib.CsMatrix) == 40
    assert ctypes.sizeof(_lib.CsKernel) == 32
    assert ctypes.sizeof(_lib.CsNormxcorr2Params) == 80
    assert ctypes.sizeof(_lib.CsCsr) == 72
    assert ctypes.sizeof(_lib.CsFociParams) == 48
    assert ctypes.sizeof(_lib.CsStageBlock) == 80
    assert ctypes.sizeof(_lib.CsCall) == 176
    assert ctypes.sizeof(_lib.CsFocus) == 48 == _lib.FOCUS_DTYPE.itemsize


def test_struct_sizes_against_the_header_compiled_as_c(tmp_path):
    """include/chromosight_hip.h compiles as plain C, and every struct the ctypes binding mirrors has the size the C
    compiler gives it (a field added on one side only would shift everything behind it)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    pairs = [("cs_matrix", _lib.CsMatrix), ("cs_kernel", _lib.CsKernel), ("cs_normxcorr2_params", _lib.CsNormxcorr2Params),
             ("cs_csr", _lib.CsCsr), ("cs_foci_params", _lib.CsFociParams), ("cs_focus", _lib.CsFocus),
             ("cs_stage_block", _lib.CsStageBlock), ("cs_call", _lib.CsCall)]
    header = ROOT / "include" / "chromosight_hip.h"
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(void) {\n%s    return 0;\n}\n' % (
        header, "".join(f'    printf("%zu\\n", sizeof({name}));\n' for name, _ in pairs)))
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(src)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    for (name, struct), size in zip(pairs, sizes):
        assert ctypes.sizeof(struct) == size, name


def test_counts_band_constants_of_the_header(tmp_path):
    """CS_LAYOUT_BAND_COUNTS (include/chromosight_hip.h): the layout codes and header bytes the binding uses, and
    CS_COUNTS_LAW_BYTES -- what pipeline.DeviceCool._stage_fast allocates behind d_law for a block of raw counts (the law, its
    reciprocals with a slot on either side, float32 copies of the reciprocals and of the block's weights) -- against the C macro."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    cases = [(50_000, 251), (7, 3), (1000, 1000), (33, 18), (200_000, 1018), (1, 1)]
    header = ROOT / "include" / "chromosight_hip.h"
    src = tmp_path / "counts.c"
    body = '    printf("%d %d %d\\n", (int)CS_LAYOUT_BAND_COUNTS, (int)CS_LAYOUT_BAND_COUNTS_VIEW, (int)CS_COUNTS_HEADER_BYTES);\n'
    body += "".join(f'    printf("%lld\\n", (long long)CS_COUNTS_LAW_BYTES({n}, {nd}));\n' for n, nd in cases)
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(void) {\n%s    return 0;\n}\n' % (header, body))
    exe = tmp_path / "counts"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    assert [int(x) for x in out[0].split()] == [_lib.LAYOUT_BAND_COUNTS, _lib.LAYOUT_BAND_COUNTS_VIEW, _lib.COUNTS_HEADER_BYTES]
    for (n, nd), line in zip(cases, out[1:]):
        need = int(line)
        have = 8 * (2 * nd + 2 + (nd + 3) // 2 + (n + 1) // 2)          # pipeline.py: law_len of a counts block, in float64
        assert need <= have < need + 16, (n, nd)
        assert need == 8 * (2 * nd + 2) + 4 * ((nd + 3) // 2 * 2) + 4 * n
