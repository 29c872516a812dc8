"""ctypes binding of libmser.so (the C-ABI declared in include/mser.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol is absent this module raises, and
every op raises ``RuntimeError`` with ``mser_last_error()`` when a call returns non-zero.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSER_LIB") or os.path.join(_HERE, "libmser.so")      # MSER_LIB: a diagnostic build (scratch/diag_stamps.py)

c_float_p = C.c_void_p   # raw device pointers travel as integers


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("sAm", C.c_int64), ("sAk", C.c_int64), ("sBk", C.c_int64), ("sBn", C.c_int64), ("ldc", C.c_int64),
        ("batch1", C.c_int32), ("batch2", C.c_int32),
        ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64), ("sC1", C.c_int64), ("sC2", C.c_int64),
        ("bias", C.c_void_p), ("alpha_dev", C.c_void_p), ("alpha", C.c_float), ("flags", C.c_int32), ("splitk", C.c_int32),
        ("R1", C.c_void_p), ("R2", C.c_void_p), ("ldr1", C.c_int64), ("ldr2", C.c_int64),
        ("sR1_1", C.c_int64), ("sR1_2", C.c_int64),
    ]


_P2 = C.c_void_p * 2


class CellParams(C.Structure):
    _fields_ = [
        ("lsthm_W", _P2), ("lsthm_Wb", _P2), ("lsthm_U", _P2), ("lsthm_Ub", _P2),
        ("lsthm_V", _P2), ("lsthm_Vb", _P2), ("lsthm_S", _P2), ("lsthm_Sb", _P2),
        ("q_Wih", _P2), ("q_Whh", _P2), ("q_bih", _P2), ("q_bhh", _P2),
        ("att_Wq", C.c_void_p), ("att_Wk", C.c_void_p),
    ]


class CellDir(C.Structure):
    _fields_ = [("p", CellParams), ("g", CellParams), ("qmask", C.c_void_p), ("rev", C.c_void_p),
                ("out", C.c_void_p), ("dout", C.c_void_p)]


class CellDesc(C.Structure):
    _fields_ = [
        ("T", C.c_int32), ("B", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("ndir", C.c_int32),
        ("x_l", C.c_void_p), ("ldxl", C.c_int64), ("x_a", C.c_void_p), ("ldxa", C.c_int64),
        ("dx_l", C.c_void_p), ("dx_a", C.c_void_p), ("ldo", C.c_int64),
        ("dir", CellDir * 2), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("dx_l_add", _P2), ("dx_a_add", _P2),
        ("rng", C.c_void_p), ("drop_site", C.c_uint32 * 2), ("p_state", C.c_float * 2), ("p_attn", C.c_float * 2),
        ("ext_hq", _P2), ("ext_dhq", _P2), ("ext_linked", C.c_int32), ("fault", C.c_void_p),
    ]


class EncoderDesc(C.Structure):
    """mser_encoder_desc (include/mser.h): one EncoderLayer, forward inputs / parameters / saved tensors / backward buffers."""
    _fields_ = [
        ("nb", C.c_int32), ("nl", C.c_int32), ("sb", C.c_int64), ("sl", C.c_int64),
        ("D", C.c_int32), ("nh", C.c_int32), ("dk", C.c_int32), ("dv", C.c_int32), ("dff", C.c_int32), ("eps", C.c_float),
        ("x", C.c_void_p), ("mask", C.c_void_p),
        ("w_qs", C.c_void_p), ("w_ks", C.c_void_p), ("w_vs", C.c_void_p), ("fc", C.c_void_p),
        ("ln1_g", C.c_void_p), ("ln1_b", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
        ("w2", C.c_void_p), ("b2", C.c_void_p), ("ln2_g", C.c_void_p), ("ln2_b", C.c_void_p),
        ("qkv", C.c_void_p), ("P", C.c_void_p), ("O", C.c_void_p),
        ("y1", C.c_void_p), ("mean1", C.c_void_p), ("rstd1", C.c_void_p), ("e1", C.c_void_p), ("hdn", C.c_void_p),
        ("y2", C.c_void_p), ("mean2", C.c_void_p), ("rstd2", C.c_void_p), ("out", C.c_void_p),
        ("dout", C.c_void_p), ("dy2", C.c_void_p), ("dh", C.c_void_p), ("dy1", C.c_void_p), ("dO", C.c_void_p),
        ("dqkv", C.c_void_p), ("dx", C.c_void_p),
        ("g_w_qs", C.c_void_p), ("g_w_ks", C.c_void_p), ("g_w_vs", C.c_void_p), ("g_fc", C.c_void_p),
        ("g_ln1_g", C.c_void_p), ("g_ln1_b", C.c_void_p), ("g_w1", C.c_void_p), ("g_b1", C.c_void_p),
        ("g_w2", C.c_void_p), ("g_b2", C.c_void_p), ("g_ln2_g", C.c_void_p), ("g_ln2_b", C.c_void_p),
        ("rng", C.c_void_p), ("drop_site", C.c_uint32), ("p_attn", C.c_float), ("p_fc", C.c_float), ("p_ffn", C.c_float),
        ("dt1", C.c_void_p),
    ]


class GruSpeakerDesc(C.Structure):
    """mser_gru_speaker_desc (include/mser.h)."""
    _fields_ = [("T", C.c_int32), ("B", C.c_int32), ("H", C.c_int32),
                ("gi", C.c_void_p), ("w_hh", C.c_void_p), ("b_hh", C.c_void_p), ("qmask", C.c_void_p),
                ("hs", C.c_void_p), ("out", C.c_void_p), ("ldo", C.c_int64), ("rev", C.c_void_p), ("save", C.c_void_p),
                ("dhs", C.c_void_p), ("dhs_add", C.c_void_p * 2), ("dgi", C.c_void_p), ("dgh", C.c_void_p),
                ("rng", C.c_void_p), ("drop_site", C.c_uint32), ("p", C.c_float),
                ("pub_counter", C.c_void_p), ("pub_per_step", C.c_uint32), ("pub_replicas", C.c_int32),
                ("pub_replica_stride", C.c_int32), ("pub_progress", C.c_void_p),
                ("sub_counter", C.c_void_p), ("sub_per_step", C.c_uint32), ("sub_parts", C.c_void_p), ("sub_nparts", C.c_int32),
                ("sub_part_stride", C.c_int64), ("status", C.c_void_p),
                ("listener_blend", C.c_int32), ("hli", C.c_void_p), ("dhli", C.c_void_p)]


class DrnnParams(C.Structure):
    """mser_drnn_params (include/mser.h): one DialogueRNN.dialogue_cell."""
    _fields_ = [(n, C.c_void_p) for n in ("g_wih", "g_whh", "g_bih", "g_bhh", "p_wih", "p_whh", "p_bih", "p_bhh",
                                           "e_wih", "e_whh", "e_bih", "e_bhh", "l_wih", "l_whh", "l_bih", "l_bhh", "att_w")]


class DrnnDesc(C.Structure):
    """mser_drnn_desc (include/mser.h)."""
    _fields_ = [("T", C.c_int32), ("B", C.c_int32), ("Dm", C.c_int32), ("Dg", C.c_int32), ("Dp", C.c_int32), ("De", C.c_int32),
                ("U", C.c_void_p), ("ldu", C.c_int64), ("qmask", C.c_void_p), ("rev", C.c_void_p),
                ("p", DrnnParams * 2), ("g", DrnnParams * 2), ("out", C.c_void_p), ("ldo", C.c_int64), ("dout", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("rng", C.c_void_p), ("drop_site", C.c_uint32 * 2), ("p_drop", C.c_float), ("fault", C.c_void_p)]


class XAttnDesc(C.Structure):
    """mser_xattn_desc (include/mser.h)."""
    _fields_ = [("nb", C.c_int32), ("nh", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32), ("dk", C.c_int32),
                ("q", C.c_void_p), ("ldq", C.c_int64), ("k", C.c_void_p), ("ldk", C.c_int64), ("v", C.c_void_p), ("ldv", C.c_int64),
                ("sbq", C.c_int64), ("slq", C.c_int64), ("sbk", C.c_int64), ("slk", C.c_int64),
                ("o", C.c_void_p), ("ldo", C.c_int64), ("stats", C.c_void_p), ("scale", C.c_float),
                ("rng", C.c_void_p), ("site", C.c_uint32), ("p", C.c_float),
                ("dO", C.c_void_p), ("lddo", C.c_int64), ("dq", C.c_void_p), ("lddq", C.c_int64), ("dk_", C.c_void_p), ("lddk", C.c_int64),
                ("dv", C.c_void_p), ("lddv", C.c_int64), ("part_stride", C.c_int64)]


class HeadTailDesc(C.Structure):
    """mser_head_tail_desc (include/mser.h)."""
    _fields_ = [("L", C.c_int32), ("B", C.c_int32), ("D", C.c_int32), ("F", C.c_int32), ("C", C.c_int32),
                ("y1", C.c_void_p), ("x_l", C.c_void_p), ("x_a", C.c_void_p),
                ("w0", C.c_void_p), ("b0", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p),
                ("y1r", C.c_void_p), ("y2", C.c_void_p), ("lp", C.c_void_p),
                ("dlp", C.c_void_p), ("dx_l_in", C.c_void_p), ("dx_a_in", C.c_void_p),
                ("dy3", C.c_void_p), ("dy2", C.c_void_p), ("dy1", C.c_void_p), ("dx_l", C.c_void_p), ("dx_a", C.c_void_p),
                ("g_b0", C.c_void_p), ("g_b3", C.c_void_p), ("g_bfc", C.c_void_p),
                ("rng", C.c_void_p), ("site_out", C.c_uint32), ("p_out", C.c_float), ("p_fc", C.c_float)]


MSER_GEMM_RELU = 1
MSER_GEMM_ACCUM = 2
MSER_FAULT_CHAIN_TIMEOUT, MSER_FAULT_LINK_TIMEOUT, MSER_FAULT_BAD_LABEL = 1, 2, 4

_i32, _i64, _f32, _vp, _sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); mirrors include/mser.h one to one (tests/test_abi.py checks the header against this table)
SIGNATURES = {
    "mser_version": (C.c_int, []),
    "mser_last_error": (C.c_char_p, []),
    "mser_gemm": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "mser_gemm_grouped": (C.c_int, [C.POINTER(GemmDesc), _i32, _vp]),
    "mser_softmax_rows": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _f32, _vp]),
    "mser_softmax_bwd_rows": (C.c_int, [_vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "mser_add_layernorm_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "mser_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "mser_colsum_acc": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _vp]),
    "mser_relu_bwd": (C.c_int, [_vp, _vp, _i64, _vp]),
    "mser_add_rows": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "mser_scale_acc_dot": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _vp]),
    "mser_encoder_layer_supported": (C.c_int, [C.POINTER(EncoderDesc)]),
    "mser_encoder_layer_fwd": (C.c_int, [C.POINTER(EncoderDesc), _vp]),
    "mser_encoder_layer_bwd": (C.c_int, [C.POINTER(EncoderDesc), _i32, _vp]),
    "mser_encoder_layer_wgrad_descs": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(GemmDesc), _i32]),
    "mser_head_tail_fwd": (C.c_int, [C.POINTER(HeadTailDesc), _vp]),
    "mser_head_tail_bwd": (C.c_int, [C.POINTER(HeadTailDesc), _vp]),
    "mser_build_reverse_index": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "mser_reverse_by_length": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "mser_build_slot_tables": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mser_marn_cell_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "mser_marn_cell_fwd": (C.c_int, [C.POINTER(CellDesc), _vp]),
    "mser_marn_cell_bwd": (C.c_int, [C.POINTER(CellDesc), _vp]),
    "mser_marn_cell_run": (C.c_int, [C.POINTER(CellDesc), _i32, _vp]),
    "mser_marn_cell_pipelined": (C.c_int, [_i32, _i32, _i32]),
    "mser_lsthm_step_fwd": (C.c_int, [_vp] * 16 + [_i32] * 5 + [_vp]),
    "mser_lsthm_step_bwd": (C.c_int, [_vp] * 7 + [_i32, _i32, _vp]),
    "mser_rank1_attention_bwd": (C.c_int, [_vp] * 9 + [_i32, _i32, _vp, C.c_uint32, _f32, _vp]),
    "mser_rank1_attention_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, C.c_uint32, _f32, _vp]),
    "mser_logsoftmax_tb_fwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "mser_logsoftmax_tb_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mser_masked_nll_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "mser_masked_nll_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "mser_marn_cell_ext_link": (C.c_int, [C.POINTER(CellDesc), _i32, _i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]),
    "mser_marn_cell_ext_link_bwd": (C.c_int, [C.POINTER(CellDesc), _i32, _i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                              C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]),
    "mser_gru_speaker_save_bytes": (C.c_size_t, [_i32, _i32, _i32]),
    "mser_gru_speaker_fwd": (C.c_int, [C.POINTER(GruSpeakerDesc), _i32, _vp]),
    "mser_gru_speaker_bwd": (C.c_int, [C.POINTER(GruSpeakerDesc), _i32, _vp]),
    "mser_dropout_apply": (C.c_int, [_vp, _i64, _i32, _i64, _vp, C.c_uint32, _f32, C.c_uint32, _vp]),
    "mser_dropout_scale": (C.c_int, [_vp, _i64, _vp, C.c_uint32, _f32, C.c_uint32, _i32, _vp]),
    "mser_rng_advance": (C.c_int, [_vp, _vp]),
    "mser_ingest_features": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "mser_confusion_update": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "mser_masked_loss_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    "mser_masked_loss_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp]),
    "mser_adam_flat": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _f32, _f32, _f32, _f32, _vp]),
    "mser_adam_flat_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _f32, _f32, _vp, _f32, _vp, _vp, _vp]),
    "mser_dp_pack": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "mser_xattn_seq_supported": (C.c_int, [C.POINTER(XAttnDesc)]),
    "mser_xattn_seq_fwd": (C.c_int, [C.POINTER(XAttnDesc), _vp]),
    "mser_xattn_seq_bwd": (C.c_int, [C.POINTER(XAttnDesc), _vp]),
    "mser_drnn_workspace_bytes": (C.c_size_t, [_i32] * 6),
    "mser_drnn_fwd": (C.c_int, [C.POINTER(DrnnDesc), _vp]),
    "mser_drnn_bwd": (C.c_int, [C.POINTER(DrnnDesc), _vp]),
    "mser_drnn_alpha": (C.c_int, [C.POINTER(DrnnDesc), _i32, C.POINTER(C.c_void_p)]),
    "mser_general2_rows_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "mser_general2_rows_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "mser_set_option": (C.c_int, [_i32, _i32]),
    "mser_marn_cell_status": (C.c_int, [C.POINTER(CellDesc), _vp]),
    "mser_prof_enable": (C.c_int, [_i32, _i32]),
    "mser_prof_collect": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
}

_lib = None


def load():
    """dlopen libmser.so and bind every symbol of include/mser.h; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libmser.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback on the product path.")
    # torch first: its wheel carries the HIP runtime of the process (libamdhip64 of its own ROCm).  libmser.so dlopen-ed BEFORE torch
    # would pull in /opt/rocm's copy instead, the process would hold two runtimes and libmser's launches fail with "no ROCm-capable
    # device is detected" (seen when build() and smoke() ran in one process).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.mser_version() < 120:
        raise RuntimeError("libmser.so is older than this binding")
    _lib = lib
    # A/B switch for measurements: MSER_OPTIONS="6=0,3=1" applies mser_set_option(key, value) pairs once at load
    for kv in filter(None, os.environ.get("MSER_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        if lib.mser_set_option(int(k), int(v)) != 0:
            raise RuntimeError(f"MSER_OPTIONS: bad pair {kv}")
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mser_last_error().decode(errors="replace")
        raise RuntimeError(f"libmser {what} failed (code {rc}): {msg}")
