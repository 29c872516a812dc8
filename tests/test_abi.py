"""CPU: the C-ABI library loads and exports every symbol include/mser.h declares; the ctypes table mirrors the header.
No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "mser.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mser_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    fns = _header_functions()
    for must in ("mser_gemm", "mser_marn_cell_fwd", "mser_marn_cell_bwd", "mser_marn_cell_run", "mser_softmax_rows",
                 "mser_add_layernorm_fwd", "mser_layernorm_bwd", "mser_build_slot_tables", "mser_reverse_by_length",
                 "mser_masked_nll_fwd", "mser_adam_flat", "mser_adam_flat_dev", "mser_dp_pack", "mser_version", "mser_last_error",
                 "mser_gemm_grouped", "mser_encoder_layer_fwd", "mser_encoder_layer_bwd", "mser_head_tail_fwd", "mser_head_tail_bwd",
                 "mser_ingest_features", "mser_confusion_update"):
        assert must in fns, must


def test_library_exports_every_declared_symbol():
    from mser import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libmser.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [f for f in _header_functions() if not hasattr(lib, f)]
    assert not missing, missing
    lib.mser_version.restype = ctypes.c_int
    assert lib.mser_version() >= 110


def test_binding_table_matches_header():
    from mser import _lib
    assert sorted(_lib.SIGNATURES) == _header_functions()


def test_struct_layouts_match_the_header_sizes():
    """sizeof of the ctypes mirrors must equal what the C compiler lays out (compiled here with the host compiler)."""
    import subprocess
    import tempfile
    from mser import _lib
    src = '#include <stdio.h>\n#include "mser.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mser_gemm_desc), sizeof(mser_cell_params), sizeof(mser_cell_dir), sizeof(mser_cell_desc), sizeof(mser_encoder_desc), sizeof(mser_head_tail_desc), sizeof(mser_xattn_desc), sizeof(mser_drnn_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_lib.GemmDesc), ctypes.sizeof(_lib.CellParams), ctypes.sizeof(_lib.CellDir), ctypes.sizeof(_lib.CellDesc),
                     ctypes.sizeof(_lib.EncoderDesc), ctypes.sizeof(_lib.HeadTailDesc), ctypes.sizeof(_lib.XAttnDesc), ctypes.sizeof(_lib.DrnnDesc)]
