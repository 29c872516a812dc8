"""Tensor-level wrappers over the C-ABI: every function takes torch tensors that already live on the GPU, passes raw
pointers / strides and enqueues on PyTorch's current HIP stream.  PyTorch is plumbing here (device memory, streams);
all arithmetic happens in libmser.so.

2-D operands are torch views whose last stride is 1 (``stride(0)`` is the leading dimension), so slices such as
``x.view(L*B, d_r+100)[:, d_r:]`` are passed without a copy.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import fault

Tensor = torch.Tensor


# ---- deferred parameter-gradient work -----------------------------------------------------------------------------------------
# Weight/bias gradients are leaves of the backward graph: nothing in the same step reads them before the optimiser.  Inside a
# ``wgrad_scope`` every grad_weight / colsum_acc is enqueued on a dedicated side stream behind an event recorded at the call
# site, so the activation-gradient chain (the critical path) never waits for them.  Operands are kept alive until the scope
# joins; callers must not modify an operand in place after handing it over (functional.py is written accordingly).
# With ``batch=True`` the weight-gradient GEMMs issued through ``grad_weight`` are not launched one by one: their descriptors
# are collected and go out as ONE grouped launch (mser_gemm_grouped) when the scope closes, on the stream that is current
# there -- the caller closes the scope after every producer stream has been joined.
_WG = {"stream": None, "keep": [], "batch": None}


class wgrad_scope:
    def __init__(self, stream, batch: bool = False):
        self.stream = stream
        self.batch = batch

    def __enter__(self):
        _WG["stream"] = self.stream
        _WG["keep"] = []
        _WG["batch"] = [] if self.batch else None
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
        return self

    @staticmethod
    def flush():
        """Launch the weight-gradient products collected so far as one group on the CURRENT stream (which must already be
        ordered behind every producer of their operands); later grad_weight calls start a new batch."""
        descs = _WG["batch"]
        if descs:
            arr = (L.GemmDesc * len(descs))(*descs)
            L.check(_lib().mser_gemm_grouped(arr, len(descs), _stream()), "mser_gemm_grouped")
            _WG["batch"] = []

    def __exit__(self, *exc):
        try:
            if self.stream is not None:
                torch.cuda.current_stream().wait_stream(self.stream)
            if exc[0] is None:
                self.flush()
        finally:
            _WG["stream"] = None
            _WG["keep"] = []
            _WG["batch"] = None
        return False


def wgrad_stream():
    """The active wgrad_scope's side stream (None outside a scope or in eager mode)."""
    return _WG.get("stream")


class _deferred:
    """Context: run the enclosed launches on the wgrad stream (if a scope is active) after the work issued so far."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.ctx = None

    def __enter__(self):
        st = _WG["stream"]
        if st is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            st.wait_event(ev)
            _WG["keep"].append(self.tensors)
            self.ctx = torch.cuda.stream(st)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


_raw_stream = torch._C._cuda_getCurrentRawStream      # ~0.3 us; torch.cuda.current_stream().cuda_stream costs ~2.8 us per call
_DEV = {"idx": None}


def _stream() -> int:
    idx = _DEV["idx"]
    if idx is None:
        idx = _DEV["idx"] = torch.cuda.current_device()
    return _raw_stream(idx)


def _p(t: Optional[Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mser ops need GPU tensors (the product path has no CPU fallback)")
    return t.data_ptr()


def _f32(t: Tensor, what: str) -> None:
    if t.dtype != torch.float32:
        raise RuntimeError(f"{what}: expected float32, got {t.dtype}")


def _ld(t: Tensor) -> int:
    """leading dimension of a 2-D row view"""
    st = t.stride()
    if len(st) != 2 or (st[1] != 1 and t.shape[1] > 1):
        raise RuntimeError(f"expected a 2-D view with unit inner stride, got shape {tuple(t.shape)} stride {st}")
    return st[0] if t.shape[0] > 1 else max(t.shape[1], st[0])


_GD = L.GemmDesc()          # one reusable descriptor: the C call reads it synchronously, so it can be refilled per launch
_GD_REF = C.byref(_GD)
_LIBH = {"lib": None}


def _lib():
    lib = _LIBH["lib"]
    if lib is None:
        lib = _LIBH["lib"] = L.load()
    return lib


def gemm_raw(A: Tensor, B: Tensor, Cm: Tensor, M: int, N: int, K: int, sAm: int, sAk: int, sBk: int, sBn: int, ldc: int,
             batch: Tuple[int, int] = (1, 1), sA=(0, 0), sB=(0, 0), sC=(0, 0), bias: Optional[Tensor] = None,
             alpha: float = 1.0, alpha_dev: Optional[Tensor] = None, relu: bool = False, accum: bool = False,
             splitk: int = 1, R1: Optional[Tensor] = None, ldr1: int = 0, R2: Optional[Tensor] = None, ldr2: int = 0,
             sR=(0, 0)) -> None:
    if not (A.is_cuda and B.is_cuda and Cm.is_cuda):
        raise RuntimeError("mser ops need GPU tensors (the product path has no CPU fallback)")
    d = _GD
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cm.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.sAm, d.sAk, d.sBk, d.sBn, d.ldc = sAm, sAk, sBk, sBn, ldc
    d.batch1, d.batch2 = batch
    d.sA1, d.sA2 = sA
    d.sB1, d.sB2 = sB
    d.sC1, d.sC2 = sC
    d.bias = bias.data_ptr() if bias is not None else None
    d.alpha_dev = alpha_dev.data_ptr() if alpha_dev is not None else None
    d.alpha = alpha
    d.flags = (1 if relu else 0) | (2 if accum else 0)
    d.splitk = splitk
    d.R1 = R1.data_ptr() if R1 is not None else None
    d.R2 = R2.data_ptr() if R2 is not None else None
    d.ldr1, d.ldr2 = ldr1, ldr2
    d.sR1_1, d.sR1_2 = sR
    rc = _lib().mser_gemm(_GD_REF, _stream())
    if rc:
        L.check(rc, "mser_gemm")


def linear(x: Tensor, W: Tensor, out: Tensor, bias: Optional[Tensor] = None, relu: bool = False, accum: bool = False,
           alpha_dev: Optional[Tensor] = None, R1: Optional[Tensor] = None, R2: Optional[Tensor] = None) -> None:
    """out[rows,N] (+)= x[rows,K] @ W[N,K]^T (+bias, relu, +R1 +R2)   -- nn.Linear layout"""
    rows, K = x.shape
    N = W.shape[0]
    gemm_raw(x, W, out, rows, N, K, _ld(x), 1, 1, W.stride(0), _ld(out), (1, 1), (0, 0), (0, 0), (0, 0), bias, 1.0, alpha_dev,
             relu, accum, 1, R1, _ld(R1) if R1 is not None else 0, R2, _ld(R2) if R2 is not None else 0)


def matmul(x: Tensor, Wkn: Tensor, out: Tensor, accum: bool = False, alpha_dev: Optional[Tensor] = None,
           R1: Optional[Tensor] = None) -> None:
    """out[rows,N] (+)= alpha * x[rows,K] @ Wkn[K,N] (+ R1)   -- torch.matmul(x, W) layout (CrossAttention2/3)"""
    rows, K = x.shape
    N = Wkn.shape[1]
    gemm_raw(x, Wkn, out, rows, N, K, _ld(x), 1, Wkn.stride(0), 1, _ld(out), accum=accum, alpha_dev=alpha_dev, R1=R1,
             ldr1=_ld(R1) if R1 is not None else 0)


def matmul_group(items) -> None:
    """Several independent products out_i[rows,N] = alpha_i * x_i[rows,K] @ W_i[K,N] (torch.matmul layout) as ONE grouped launch
    (mser_gemm_grouped): items = [(x, Wkn, out, alpha_dev or None), ...]."""
    descs = []
    for x, Wkn, out, alpha_dev in items:
        if not (x.is_cuda and Wkn.is_cuda and out.is_cuda):
            raise RuntimeError("mser ops need GPU tensors (the product path has no CPU fallback)")
        d = L.GemmDesc()
        rows, K = x.shape
        d.A, d.B, d.C = x.data_ptr(), Wkn.data_ptr(), out.data_ptr()
        d.M, d.N, d.K = rows, Wkn.shape[1], K
        d.sAm, d.sAk, d.sBk, d.sBn, d.ldc = _ld(x), 1, Wkn.stride(0), 1, _ld(out)
        d.batch1 = d.batch2 = 1
        d.alpha = 1.0
        d.alpha_dev = alpha_dev.data_ptr() if alpha_dev is not None else None
        d.splitk = 1
        descs.append(d)
    arr = (L.GemmDesc * len(descs))(*descs)
    L.check(_lib().mser_gemm_grouped(arr, len(descs), _stream()), "mser_gemm_grouped")


def matmul_nt(dy: Tensor, Wkn: Tensor, out: Tensor, accum: bool = False, alpha_dev: Optional[Tensor] = None) -> None:
    """out[rows,K] (+)= alpha * dy[rows,N] @ Wkn[K,N]^T"""
    rows, N = dy.shape
    K = Wkn.shape[0]
    gemm_raw(dy, Wkn, out, rows, K, N, _ld(dy), 1, 1, Wkn.stride(0), _ld(out), accum=accum, alpha_dev=alpha_dev)


def grad_weight(dy: Tensor, x: Tensor, dW: Tensor, transposed: bool = False, alpha_dev: Optional[Tensor] = None,
                splitk: int = 16) -> None:
    """dW += dy^T x (nn.Linear weight [N,K]) or, transposed, dW += x^T dy (matmul weight [K,N]); split-K float atomics."""
    rows = dy.shape[0]
    batch = _WG["batch"]
    if batch is not None:
        d = L.GemmDesc()
        if not transposed:
            N, K = dy.shape[1], x.shape[1]
            d.A, d.B, d.M, d.N, d.sAm, d.sAk, d.sBk, d.sBn = dy.data_ptr(), x.data_ptr(), N, K, 1, _ld(dy), _ld(x), 1
        else:
            K, N = x.shape[1], dy.shape[1]
            d.A, d.B, d.M, d.N, d.sAm, d.sAk, d.sBk, d.sBn = x.data_ptr(), dy.data_ptr(), K, N, 1, _ld(x), _ld(dy), 1
        d.C, d.K, d.ldc = dW.data_ptr(), rows, dW.stride(0)
        d.batch1 = d.batch2 = 1
        d.alpha = 1.0
        d.alpha_dev = alpha_dev.data_ptr() if alpha_dev is not None else None
        d.splitk = splitk
        batch.append(d)
        _WG["keep"].append((dy, x, dW, alpha_dev))
        return
    with _deferred(dy, x):
        if not transposed:
            N, K = dy.shape[1], x.shape[1]
            gemm_raw(dy, x, dW, N, K, rows, 1, _ld(dy), _ld(x), 1, dW.stride(0), splitk=splitk, alpha_dev=alpha_dev)
        else:
            K, N = x.shape[1], dy.shape[1]
            gemm_raw(x, dy, dW, K, N, rows, 1, _ld(x), _ld(dy), 1, dW.stride(0), splitk=splitk, alpha_dev=alpha_dev)


def softmax_rows_(S: Tensor, rows: int, n: int, ld: int, mul: Optional[Tensor] = None, mask: Optional[Tensor] = None,
                  mask_on: int = 1, fill: float = float("-inf")) -> None:
    L.check(_lib().mser_softmax_rows(_p(S), rows, n, ld, _p(mul), _p(mask), mask_on, fill, _stream()), "softmax_rows")


def softmax_bwd_rows_(P: Tensor, dP: Tensor, rows: int, n: int, ld: int, mul: Optional[Tensor] = None) -> None:
    L.check(_lib().mser_softmax_bwd_rows(_p(P), _p(dP), rows, n, ld, _p(mul), _stream()), "softmax_bwd_rows")


def add_layernorm_fwd(x: Tensor, res: Optional[Tensor], gamma: Tensor, beta: Tensor, y: Tensor, sum_out: Optional[Tensor],
                      mean: Tensor, rstd: Tensor, eps: float) -> None:
    rows, D = x.shape
    L.check(_lib().mser_add_layernorm_fwd(_p(x), _ld(x), _p(res), _ld(res) if res is not None else 0, _p(gamma), _p(beta),
                                            _p(y), _p(sum_out), _p(mean), _p(rstd), rows, D, eps, _stream()), "add_layernorm_fwd")


def layernorm_bwd(dy: Tensor, xsum: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor, dx: Tensor, dgamma: Tensor,
                  dbeta: Tensor) -> None:
    rows, D = dy.shape
    L.check(_lib().mser_layernorm_bwd(_p(dy), _p(xsum), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dgamma), _p(dbeta), rows, D,
                                        _stream()), "layernorm_bwd")


def colsum_acc(X: Tensor, out: Tensor) -> None:
    rows, n = X.shape
    with _deferred(X):
        L.check(_lib().mser_colsum_acc(_p(X), rows, n, _ld(X), _p(out), _stream()), "colsum_acc")


def relu_bwd_(dY: Tensor, Y: Tensor) -> None:
    L.check(_lib().mser_relu_bwd(_p(dY), _p(Y), dY.numel(), _stream()), "relu_bwd")


def add_rows(out: Tensor, a: Tensor, b: Optional[Tensor] = None) -> None:
    rows, D = a.shape
    L.check(_lib().mser_add_rows(_p(out), _ld(out), _p(a), _ld(a), _p(b), _ld(b) if b is not None else 0, rows, D, _stream()),
            "add_rows")


def scale_acc_dot(acc: Tensor, t: Tensor, x: Optional[Tensor], s_dev: Optional[Tensor], ds: Optional[Tensor]) -> None:
    rows, D = t.shape
    L.check(_lib().mser_scale_acc_dot(_p(acc), _ld(acc), _p(t), _ld(t), _p(x), _ld(x) if x is not None else 0, _p(s_dev), _p(ds),
                                        rows, D, _stream()), "scale_acc_dot")


ENC_BWD_ACT, ENC_BWD_WGRAD = 1, 2


def encoder_layer_supported(desc: L.EncoderDesc) -> bool:
    return bool(_lib().mser_encoder_layer_supported(C.byref(desc)))


def encoder_layer_fwd(desc: L.EncoderDesc) -> None:
    L.check(_lib().mser_encoder_layer_fwd(C.byref(desc), _stream()), "encoder_layer_fwd")


def encoder_layer_bwd(desc: L.EncoderDesc, phases: int, deferred: Optional[tuple] = None) -> None:
    """phases = ENC_BWD_ACT (activation-gradient chain) and / or ENC_BWD_WGRAD (weight-gradient GEMMs).  With ``deferred``
    (the tensors the weight-gradient phase reads) the call goes to the wgrad side stream when a wgrad_scope is active."""
    batch = _WG["batch"]
    if batch is not None and phases == ENC_BWD_WGRAD:
        # hand the layer's weight-gradient products to the step's single grouped launch (wgrad_scope(batch=True))
        arr = (L.GemmDesc * 6)()
        n = _lib().mser_encoder_layer_wgrad_descs(C.byref(desc), arr, 6)
        if n < 0:
            L.check(n, "encoder_layer_wgrad_descs")
        for i in range(n):
            d = L.GemmDesc()
            C.memmove(C.byref(d), C.byref(arr[i]), C.sizeof(L.GemmDesc))
            batch.append(d)
        _WG["keep"].append((desc,) + tuple(deferred or ()))
        return
    if deferred is not None:
        with _deferred(*deferred):
            L.check(_lib().mser_encoder_layer_bwd(C.byref(desc), phases, _stream()), "encoder_layer_bwd")
    else:
        L.check(_lib().mser_encoder_layer_bwd(C.byref(desc), phases, _stream()), "encoder_layer_bwd")


def head_tail_desc(Ln: int, B: int, y1: Tensor, x_l: Tensor, x_a: Tensor, W0: Tensor, b0: Tensor, W3: Tensor, b3: Tensor,
                   y1r: Tensor, y2: Tensor, lp: Tensor) -> L.HeadTailDesc:
    d = L.HeadTailDesc()
    d.L, d.B, d.D, d.F, d.C = Ln, B, y1.shape[1], W0.shape[0], W3.shape[0]
    for t in (y1, x_l, x_a, W0, b0, W3, b3, y1r, y2, lp):
        if not t.is_contiguous():
            raise RuntimeError("head_tail: every operand must be contiguous")
    d.y1, d.x_l, d.x_a, d.w0, d.b0, d.w3, d.b3 = (_p(t) for t in (y1, x_l, x_a, W0, b0, W3, b3))
    d.y1r, d.y2, d.lp = _p(y1r), _p(y2), _p(lp)
    return d


def head_tail_fwd(desc: L.HeadTailDesc) -> None:
    L.check(_lib().mser_head_tail_fwd(C.byref(desc), _stream()), "head_tail_fwd")


def head_tail_bwd(desc: L.HeadTailDesc) -> None:
    L.check(_lib().mser_head_tail_bwd(C.byref(desc), _stream()), "head_tail_bwd")


def build_reverse_index(umask: Tensor, lens: Tensor, rev: Tensor) -> None:
    B, Ln = umask.shape
    L.check(_lib().mser_build_reverse_index(_p(umask), B, Ln, _p(lens), _p(rev), _stream()), "build_reverse_index")


def reverse_by_length(X: Tensor, rev: Tensor, out: Tensor, Ln: int, B: int) -> None:
    D = X.shape[1]
    L.check(_lib().mser_reverse_by_length(_p(X), _ld(X), _p(rev), _p(out), _ld(out), Ln, B, D, _stream()), "reverse_by_length")


def build_slot_tables(qmask: Tensor, rev: Optional[Tensor], party: Tensor, perm: Tensor, n0: Tensor, qm_out: Tensor) -> None:
    T, B = qmask.shape[0], qmask.shape[1]
    L.check(_lib().mser_build_slot_tables(_p(qmask), _p(rev), T, B, _p(party), _p(perm), _p(n0), _p(qm_out), _stream()),
            "build_slot_tables")


def logsoftmax_tb_fwd(y: Tensor, lp: Tensor, Ln: int, B: int) -> None:
    L.check(_lib().mser_logsoftmax_tb_fwd(_p(y), _p(lp), Ln, B, y.shape[-1], _stream()), "logsoftmax_tb_fwd")


def logsoftmax_tb_bwd(dlp: Tensor, lp: Tensor, dy: Tensor, Ln: int, B: int) -> None:
    L.check(_lib().mser_logsoftmax_tb_bwd(_p(dlp), _p(lp), _p(dy), Ln, B, lp.shape[-1], _stream()), "logsoftmax_tb_bwd")


def masked_nll_fwd(pred: Tensor, target: Tensor, mask: Tensor, loss_out: Tensor) -> None:
    rows, Cn = pred.shape
    L.check(_lib().mser_masked_nll_fwd(_p(pred), _p(target), _p(mask), rows, Cn, _p(loss_out), _stream()), "masked_nll_fwd")


def masked_nll_bwd(target: Tensor, mask: Tensor, loss_out: Tensor, gscale: Optional[Tensor], dpred: Tensor) -> None:
    rows, Cn = dpred.shape
    L.check(_lib().mser_masked_nll_bwd(_p(target), _p(mask), _p(loss_out), _p(gscale), _p(dpred), rows, Cn, _stream()),
            "masked_nll_bwd")


# ---- GRU speaker state (include/mser.h mser_gru_speaker_desc)
def gru_speaker_desc(T: int, B: int, H: int, gi: Tensor, w_hh: Tensor, b_hh: Tensor, qmask: Tensor, hs: Tensor, save: Tensor,
                     out: Optional[Tensor] = None, rev: Optional[Tensor] = None, drop=None, lblend: bool = False,
                     hli: Optional[Tensor] = None) -> L.GruSpeakerDesc:
    for t in (gi, w_hh, b_hh, qmask, hs, save):
        _f32(t, "gru_speaker")
        if not t.is_contiguous():
            raise RuntimeError("gru_speaker: gi, w_hh, b_hh, qmask, hs and save must be contiguous")
    if save.numel() * 4 < _lib().mser_gru_speaker_save_bytes(T, B, H):
        raise RuntimeError("gru_speaker: save buffer too small")
    d = L.GruSpeakerDesc()
    d.T, d.B, d.H = T, B, H
    d.gi, d.w_hh, d.b_hh, d.qmask, d.hs, d.save = (_p(t) for t in (gi, w_hh, b_hh, qmask, hs, save))
    if out is not None:
        d.out, d.ldo = _p(out), _ld(out)
    d.rev = _p(rev)
    if drop is not None:
        d.rng, d.drop_site, d.p = _p(drop.rng), drop.site, float(drop.p)
    d.listener_blend = 1 if lblend else 0            # model/lsthm_nsps.py:188-191 (see include/mser.h)
    d.hli = _p(hli)
    # the descriptor holds raw pointers: keep the operands referenced for as long as it lives
    d._keep = (gi, w_hh, b_hh, qmask, hs, save, out, rev, drop.rng if drop is not None else None, hli)
    return d


def _gru_array(descs):
    descs = list(descs) if isinstance(descs, (list, tuple)) else [descs]
    if len(descs) not in (1, 2):
        raise RuntimeError("gru_speaker: one or two chains per launch")
    arr = (L.GruSpeakerDesc * len(descs))(*descs)      # (copies the structs: the tensors stay referenced by the originals' _keep)
    return arr, len(descs)


def gru_speaker_fwd(descs) -> None:
    """descs: one descriptor or a list of two (same T, B): the chains share one launch."""
    arr, n = _gru_array(descs)
    L.check(_lib().mser_gru_speaker_fwd(arr, n, _stream()), "gru_speaker_fwd")


def gru_speaker_set_grads(desc: L.GruSpeakerDesc, dhs: Tensor, dgi: Tensor, dgh: Tensor, dhs_add: Sequence[Tensor] = ()) -> None:
    for t in (dhs, dgi, dgh, *dhs_add):
        if not t.is_contiguous():
            raise RuntimeError("gru_speaker_bwd: gradient buffers must be contiguous")
    desc.dhs, desc.dgi, desc.dgh = _p(dhs), _p(dgi), _p(dgh)
    desc.sub_counter = None                     # (not a linked consumer)
    desc._keep_bwd = (dhs, dgi, dgh, tuple(dhs_add))
    for i in range(2):
        desc.dhs_add[i] = _p(dhs_add[i]) if i < len(dhs_add) else None


def gru_speaker_link_bwd(desc: L.GruSpeakerDesc, link, dgi: Tensor, dgh: Tensor, status: Optional[Tensor] = None) -> None:
    """Prepare a chain's BPTT as a linked consumer of the cell's BPTT launch (``link`` = cell_ext_link_bwd(...))."""
    dhq, parts, n_parts, part_stride, cnt, per_step, _ = link
    desc.dhs, desc.dgi, desc.dgh = dhq, _p(dgi), _p(dgh)
    desc.dhs_add[0] = desc.dhs_add[1] = None
    desc.sub_counter, desc.sub_per_step, desc.sub_parts, desc.sub_nparts, desc.sub_part_stride = cnt, per_step, parts, n_parts, part_stride
    if status is None:
        status = fault.word(dgi.device)
    desc.status = _p(status)
    desc._keep_bwd = (dgi, dgh, status)


def gru_speaker_bwd(descs, dhs: Optional[Tensor] = None, dgi: Optional[Tensor] = None, dgh: Optional[Tensor] = None,
                    dhs_add: Sequence[Tensor] = ()) -> None:
    """One descriptor with its gradient buffers given here, or a list of (one or two) descriptors prepared with
    gru_speaker_set_grads."""
    if dhs is not None:
        gru_speaker_set_grads(descs, dhs, dgi, dgh, dhs_add)
    arr, n = _gru_array(descs)
    L.check(_lib().mser_gru_speaker_bwd(arr, n, _stream()), "gru_speaker_bwd")


# ---- fused sequence-level cross-modal attention core (include/mser.h mser_xattn_seq_*)
def xattn_seq_supported(desc: L.XAttnDesc) -> bool:
    return bool(_lib().mser_xattn_seq_supported(C.byref(desc)))


def xattn_seq_fwd(desc: L.XAttnDesc) -> None:
    L.check(_lib().mser_xattn_seq_fwd(C.byref(desc), _stream()), "xattn_seq_fwd")


def xattn_seq_bwd(desc: L.XAttnDesc) -> None:
    L.check(_lib().mser_xattn_seq_bwd(C.byref(desc), _stream()), "xattn_seq_bwd")


# ---- DialogueRNN (include/mser.h mser_drnn_*)
_DRNN_FIELDS = (("g_wih", "g_cell.weight_ih"), ("g_whh", "g_cell.weight_hh"), ("g_bih", "g_cell.bias_ih"), ("g_bhh", "g_cell.bias_hh"),
                ("p_wih", "p_cell.weight_ih"), ("p_whh", "p_cell.weight_hh"), ("p_bih", "p_cell.bias_ih"), ("p_bhh", "p_cell.bias_hh"),
                ("e_wih", "e_cell.weight_ih"), ("e_whh", "e_cell.weight_hh"), ("e_bih", "e_cell.bias_ih"), ("e_bhh", "e_cell.bias_hh"),
                ("l_wih", "l_cell.weight_ih"), ("l_whh", "l_cell.weight_hh"), ("l_bih", "l_cell.bias_ih"), ("l_bhh", "l_cell.bias_hh"),
                ("att_w", "attention.transform.weight"))


def drnn_param_struct(get) -> L.DrnnParams:
    """``get(name)``: tensor for a parameter name relative to a DialogueRNN's ``dialogue_cell`` (contiguous)."""
    dp = L.DrnnParams()
    for field, name in _DRNN_FIELDS:
        t = get(name)
        if not t.is_contiguous():
            raise RuntimeError(f"drnn: parameter {name} must be contiguous")
        setattr(dp, field, _p(t))
    return dp


def drnn_workspace_bytes(T: int, B: int, Dm: int, Dg: int, Dp: int, De: int) -> int:
    return int(L.load().mser_drnn_workspace_bytes(T, B, Dm, Dg, Dp, De))


def make_drnn_desc(T: int, B: int, dims, U: Tensor, qmask: Tensor, rev: Tensor, params, out: Tensor, workspace: Tensor, grads=None,
                   dout: Optional[Tensor] = None, drop=None) -> L.DrnnDesc:
    """dims = (Dm, Dg, Dp, De); params / grads: two DrnnParams (dialog_rnn_f, dialog_rnn_r); drop: None or (rng, [site_f, site_r], p)."""
    d = L.DrnnDesc()
    d.T, d.B = T, B
    d.Dm, d.Dg, d.Dp, d.De = dims
    d.U, d.ldu = _p(U), _ld(U)
    d.qmask, d.rev = _p(qmask), _p(rev)
    for i in range(2):
        d.p[i] = params[i]
        if grads is not None:
            d.g[i] = grads[i]
    d.out, d.ldo = _p(out), _ld(out)
    d.dout = _p(dout)
    d.workspace, d.workspace_bytes = _p(workspace), workspace.numel() * workspace.element_size()
    if drop is not None:
        rng, sites, p = drop
        d.rng, d.p_drop = _p(rng), float(p)
        d.drop_site[0], d.drop_site[1] = int(sites[0]), int(sites[1])
    d.fault = _p(fault.word(workspace.device))
    d._keep = (U, qmask, rev, out, workspace, dout)
    return d


def drnn_fwd(desc: L.DrnnDesc) -> None:
    L.check(_lib().mser_drnn_fwd(C.byref(desc), _stream()), "drnn_fwd")


def drnn_bwd(desc: L.DrnnDesc) -> None:
    L.check(_lib().mser_drnn_bwd(C.byref(desc), _stream()), "drnn_bwd")


def drnn_alpha_ptr(desc: L.DrnnDesc, direction: int) -> int:
    ptr = C.c_void_p()
    L.check(_lib().mser_drnn_alpha(C.byref(desc), direction, C.byref(ptr)), "drnn_alpha")
    return ptr.value


def general2_rows_fwd(S0: Tensor, alpha: Tensor, mask: Tensor, rows: int, n: int, Ln: int) -> None:
    L.check(_lib().mser_general2_rows_fwd(_p(S0), _p(alpha), _p(mask), rows, n, Ln, _stream()), "general2_rows_fwd")


def general2_rows_bwd(S0: Tensor, mask: Tensor, dA: Tensor, rows: int, n: int, Ln: int) -> None:
    L.check(_lib().mser_general2_rows_bwd(_p(S0), _p(mask), _p(dA), rows, n, Ln, _stream()), "general2_rows_bwd")


# ---- dropout (include/mser.h "Dropout"): rng = int32 tensor {seed, step} on the device
def dropout_apply_(x: Tensor, rng: Tensor, site: int, p: float, idx0: int = 0) -> None:
    """In place x[r, c] *= keep ? 1/(1-p) : 0 over a 2-D row view (unit column stride) or any contiguous tensor."""
    _f32(x, "dropout_apply")
    if x.dim() == 2 and x.stride(1) == 1:
        rows, cols, ld = x.shape[0], x.shape[1], x.stride(0)
    elif x.is_contiguous():
        rows, cols, ld = 1, x.numel(), x.numel()
    else:
        raise RuntimeError("dropout_apply: need a contiguous tensor or a 2-D row view")
    if rows * cols >= 1 << 32:
        raise RuntimeError("dropout_apply: more than 2^32 elements in one site")
    L.check(_lib().mser_dropout_apply(_p(x), rows, cols, ld, _p(rng), site, float(p), idx0, _stream()), "dropout_apply")


def dropout_scale(n: int, rng: Tensor, site: int, p: float, idx0: int = 0, draw_bits: int = 32) -> Tensor:
    """The factors keep ? 1/(1-p) : 0 of elements idx0 .. idx0+n-1 of a site as data (for the checker); draw_bits = 16 for the
    rank-1 attention sites."""
    out = torch.empty(n, device=rng.device)
    L.check(_lib().mser_dropout_scale(_p(out), n, _p(rng), site, float(p), idx0, draw_bits, _stream()), "dropout_scale")
    return out


def rng_advance_(rng: Tensor) -> None:
    L.check(_lib().mser_rng_advance(_p(rng), _stream()), "rng_advance")


def ingest_features(r1: Tensor, r2: Tensor, r3: Tensor, r4: Tensor, acouf: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """x = cat((r1+r2+r3+r4)/4, acouf) along the last dim (reference model_trainer.py:104-105) in one launch."""
    for t in (r1, r2, r3, r4, acouf):
        _f32(t, "ingest_features")
        if not t.is_cuda:
            raise RuntimeError("mser ops need GPU tensors (the product path has no CPU fallback)")
    if not (r1.shape == r2.shape == r3.shape == r4.shape) or r1.shape[:-1] != acouf.shape[:-1]:
        raise RuntimeError(f"ingest_features: shape mismatch {tuple(r1.shape)} / {tuple(acouf.shape)}")
    r1, r2, r3, r4, acouf = (t.contiguous() for t in (r1, r2, r3, r4, acouf))
    d_r, d_a = r1.shape[-1], acouf.shape[-1]
    rows = r1.numel() // d_r
    if out is None:
        out = torch.empty(*r1.shape[:-1], d_r + d_a, device=r1.device)
    L.check(_lib().mser_ingest_features(_p(r1), _p(r2), _p(r3), _p(r4), _p(acouf), _p(out), rows, d_r, d_a, _stream()),
            "ingest_features")
    return out


def confusion_update(lp: Tensor, label: Tensor, mask: Tensor, conf: Tensor, pred_out: Optional[Tensor] = None) -> None:
    """conf[C, C] (float64, accumulated) += mask-weighted counts of (label, argmax lp); pred_out int64 [rows] optional."""
    rows, Cn = lp.shape
    if conf.dtype != torch.float64 or label.dtype != torch.int64:
        raise RuntimeError("confusion_update: conf must be float64 and label int64")
    L.check(_lib().mser_confusion_update(_p(lp.contiguous()), _p(label.contiguous()), _p(mask.contiguous()), rows, Cn, _p(conf),
                                           _p(pred_out), _stream()), "confusion_update")


def masked_loss_fwd(pred: Tensor, target: Tensor, mask: Tensor, weight: Optional[Tensor], is_ce: bool, loss_out: Tensor) -> None:
    rows, Cn = pred.shape
    L.check(_lib().mser_masked_loss_fwd(_p(pred), _p(target), _p(mask), _p(weight), int(is_ce), rows, Cn, _p(loss_out),
                                          _p(fault.word(pred.device)), _stream()), "masked_loss_fwd")


def masked_loss_bwd(pred: Tensor, target: Tensor, mask: Tensor, weight: Optional[Tensor], is_ce: bool, loss_out: Tensor,
                    gscale: Optional[Tensor], dpred: Tensor) -> None:
    rows, Cn = dpred.shape
    L.check(_lib().mser_masked_loss_bwd(_p(pred), _p(target), _p(mask), _p(weight), int(is_ce), _p(loss_out), _p(gscale), _p(dpred),
                                          rows, Cn, _stream()), "masked_loss_bwd")


def adam_flat(p: Tensor, g: Tensor, m: Tensor, v: Tensor, live: Optional[Tensor], step: int, lr: float, beta1: float = 0.9,
              beta2: float = 0.999, eps: float = 1e-8, wd: float = 0.0, gscale: float = 1.0) -> None:
    L.check(_lib().mser_adam_flat(_p(p), _p(g), _p(m), _p(v), _p(live), p.numel(), step, lr, beta1, beta2, eps, wd, gscale,
                                    _stream()), "adam_flat")


def adam_flat_dev(p: Tensor, g: Tensor, m: Tensor, v: Tensor, live: Optional[Tensor], step_dev: Tensor, hp_dev: Tensor,
                  sched_dev: Tensor, eps: float = 1e-8, wd: float = 0.0, gscale_div_dev: Optional[Tensor] = None,
                  gscale: float = 1.0, gfault: Optional[Tensor] = None) -> None:
    """The update is skipped on the device while this device's sticky fault word (mser.fault) is set, or while ``gfault`` (float
    [1]: the ranks' fault flags summed by the data-parallel all-reduce) is non-zero."""
    L.check(_lib().mser_adam_flat_dev(_p(p), _p(g), _p(m), _p(v), _p(live), p.numel(), _p(step_dev), _p(hp_dev), _p(sched_dev),
                                        eps, wd, _p(gscale_div_dev), gscale, _p(fault.word(p.device)), _p(gfault), _stream()),
            "adam_flat_dev")


def dp_pack(buf: Tensor, g: Tensor, cnt_dev: Tensor) -> None:
    """buf [n + 2] = g * cnt | cnt | (this device's fault word != 0)."""
    L.check(_lib().mser_dp_pack(_p(buf), _p(g), _p(cnt_dev), g.numel(), _p(fault.word(g.device)), _stream()), "dp_pack")


def lsthm_step_fwd(x, c, h, z, s, W, Wb, U, Ub, V, Vb, S, Sb, c_out, h_out, gates=None) -> None:
    B, D = x.shape
    H, Hz, Hs = c.shape[1], z.shape[1], s.shape[1]
    L.check(_lib().mser_lsthm_step_fwd(_p(x), _p(c), _p(h), _p(z), _p(s), _p(W), _p(Wb), _p(U), _p(Ub), _p(V), _p(Vb), _p(S),
                                         _p(Sb), _p(c_out), _p(h_out), _p(gates), B, D, H, Hz, Hs, _stream()), "lsthm_step_fwd")


def lsthm_step_bwd(gates: Tensor, c_prev: Tensor, c_new: Tensor, dc_new: Optional[Tensor], dh_new: Optional[Tensor],
                   dgates: Tensor, dc_prev: Tensor) -> None:
    B, H = c_prev.shape
    L.check(_lib().mser_lsthm_step_bwd(_p(gates), _p(c_prev), _p(c_new), _p(dc_new), _p(dh_new), _p(dgates), _p(dc_prev), B, H,
                                         _stream()), "lsthm_step_bwd")


def rank1_attention_bwd(x1: Tensor, x2: Tensor, Wq: Tensor, Wk: Tensor, dout: Tensor, dx1: Tensor, dx2: Tensor, gWq: Tensor,
                        gWk: Tensor, drop=None) -> None:
    """drop: None or an object with .rng (int32[2] tensor), .site, .p (mser.functional.DropSite)."""
    B, H = x1.shape
    rng, site, p = (_p(drop.rng), drop.site, float(drop.p)) if drop is not None else (None, 0, 0.0)
    L.check(_lib().mser_rank1_attention_bwd(_p(x1), _p(x2), _p(Wq), _p(Wk), _p(dout), _p(dx1), _p(dx2), _p(gWq), _p(gWk), B, H,
                                              rng, site, p, _stream()), "rank1_attention_bwd")


def rank1_attention_fwd(x1: Tensor, x2: Tensor, Wq: Tensor, Wk: Tensor, out: Tensor, drop=None) -> None:
    B, H = x1.shape
    rng, site, p = (_p(drop.rng), drop.site, float(drop.p)) if drop is not None else (None, 0, 0.0)
    L.check(_lib().mser_rank1_attention_fwd(_p(x1), _p(x2), _p(Wq), _p(Wk), _p(out), B, H, rng, site, p, _stream()),
            "rank1_attention_fwd")


# ---------------------------------------------------------------------------------------------- MARN cell
CELL_KEYS = ("lsthm_W", "lsthm_Wb", "lsthm_U", "lsthm_Ub", "lsthm_V", "lsthm_Vb", "lsthm_S", "lsthm_Sb",
             "q_Wih", "q_Whh", "q_bih", "q_bhh")


def cell_param_struct(get) -> L.CellParams:
    """``get(name)`` returns the tensor for a reference parameter name relative to the MARN_cell (or None)."""
    cp = L.CellParams()
    names = {
        "lsthm_W": ("lsthm_l.W.weight", "lsthm_a.W.weight"), "lsthm_Wb": ("lsthm_l.W.bias", "lsthm_a.W.bias"),
        "lsthm_U": ("lsthm_l.U.weight", "lsthm_a.U.weight"), "lsthm_Ub": ("lsthm_l.U.bias", "lsthm_a.U.bias"),
        "lsthm_V": ("lsthm_l.V.weight", "lsthm_a.V.weight"), "lsthm_Vb": ("lsthm_l.V.bias", "lsthm_a.V.bias"),
        "lsthm_S": ("lsthm_l.S.weight", "lsthm_a.S.weight"), "lsthm_Sb": ("lsthm_l.S.bias", "lsthm_a.S.bias"),
        "q_Wih": ("lstm_q0.weight_ih", "lstm_q1.weight_ih"), "q_Whh": ("lstm_q0.weight_hh", "lstm_q1.weight_hh"),
        "q_bih": ("lstm_q0.bias_ih", "lstm_q1.bias_ih"), "q_bhh": ("lstm_q0.bias_hh", "lstm_q1.bias_hh"),
    }
    def opt(n):           # the speaker LSTM cells are absent from the GRU-speaker variants' cells (ext_hq mode: may be NULL)
        try:
            return get(n)
        except KeyError:
            if n.startswith("lstm_q"):
                return None
            raise

    for field, (n0, n1) in names.items():
        arr = getattr(cp, field)
        arr[0] = _p(opt(n0))
        arr[1] = _p(opt(n1))
    cp.att_Wq = _p(get("crossatt_l2a.Wq"))
    cp.att_Wk = _p(get("crossatt_l2a.Wk"))
    return cp


def cell_workspace_bytes(T: int, B: int, D: int, H: int, ndir: int) -> int:
    return int(L.load().mser_marn_cell_workspace_bytes(T, B, D, H, ndir))


def make_cell_desc(T: int, B: int, D: int, H: int, x_l: Tensor, x_a: Tensor, dirs: Sequence[dict], ldo: int,
                   workspace: Tensor, dx_l: Optional[Tensor] = None, dx_a: Optional[Tensor] = None,
                   dx_l_add: Sequence[Tensor] = (), dx_a_add: Sequence[Tensor] = (), drop=None,
                   ext_hq: Sequence[Tensor] = (), ext_dhq: Sequence[Tensor] = (), ext_linked: bool = False) -> L.CellDesc:
    """dirs: list of dicts with keys p (CellParams), g (CellParams or None), qmask, rev (or None), out, dout (or None).
    drop: None or (rng int32[2] tensor, [site per direction], [p_state per direction], [p_attn per direction])."""
    d = L.CellDesc()
    d.T, d.B, d.D, d.H, d.ndir = T, B, D, H, len(dirs)
    d.x_l, d.ldxl = _p(x_l), _ld(x_l)
    d.x_a, d.ldxa = _p(x_a), _ld(x_a)
    d.dx_l, d.dx_a = _p(dx_l), _p(dx_a)
    d.ldo = ldo
    for i, r in enumerate(dirs):
        d.dir[i].p = r["p"]
        if r.get("g") is not None:
            d.dir[i].g = r["g"]
        d.dir[i].qmask = _p(r["qmask"])
        d.dir[i].rev = _p(r.get("rev"))
        d.dir[i].out = _p(r["out"])
        d.dir[i].dout = _p(r.get("dout"))
    d.workspace = _p(workspace)
    d.workspace_bytes = workspace.numel() * workspace.element_size()
    for i, t in enumerate(dx_l_add):       # contiguous [T*B, D] partial sums folded into dx_l / dx_a by PHASE_LSTHM_BWD_DX
        if not t.is_contiguous():
            raise RuntimeError("dx_l_add entries must be contiguous")
        d.dx_l_add[i] = _p(t)
    for i, t in enumerate(dx_a_add):
        if not t.is_contiguous():
            raise RuntimeError("dx_a_add entries must be contiguous")
        d.dx_a_add[i] = _p(t)
    if drop is not None:
        rng, sites, p_state, p_attn = drop
        d.rng = _p(rng)
        for i in range(len(dirs)):
            d.drop_site[i], d.p_state[i], d.p_attn[i] = int(sites[i]), float(p_state[i]), float(p_attn[i])
    d.ext_linked = 1 if ext_linked else 0
    d.fault = _p(fault.word(workspace.device))
    for i, t in enumerate(ext_hq):          # external speaker state per direction ([T*B, H] contiguous) and its gradient buffer
        if not t.is_contiguous() or (i < len(ext_dhq) and not ext_dhq[i].is_contiguous()):
            raise RuntimeError("ext_hq / ext_dhq must be contiguous")
        d.ext_hq[i] = _p(t)
        if i < len(ext_dhq):
            d.ext_dhq[i] = _p(ext_dhq[i])
    return d


def cell_ext_link(desc: L.CellDesc, direction: int, partner_wgs: int):
    """(hq_rows pointer, counter pointer, replicas, replica stride, per-step increment, link possible?) for a linked producer of
    direction ``direction``'s speaker rows whose launch has ``partner_wgs`` workgroups (include/mser.h mser_marn_cell_ext_link)."""
    hq, cnt = C.c_void_p(), C.c_void_p()
    rep, stride, inc = C.c_int32(), C.c_int32(), C.c_uint32()
    rc = _lib().mser_marn_cell_ext_link(C.byref(desc), direction, int(partner_wgs), C.byref(hq), C.byref(cnt), C.byref(rep),
                                        C.byref(stride), C.byref(inc))
    if rc < 0:
        L.check(rc, "marn_cell_ext_link")
    return hq.value, cnt.value, rep.value, stride.value, inc.value, rc == 1


def cell_ext_link_bwd(desc: L.CellDesc, direction: int, partner_wgs: int):
    """(dhq pointer, parts pointer, n_parts, part stride, counter pointer, per-step count, link possible?) for a linked consumer
    (``partner_wgs`` workgroups) of direction ``direction``'s speaker-state gradient (include/mser.h mser_marn_cell_ext_link_bwd)."""
    dhq, parts, cnt = C.c_void_p(), C.c_void_p(), C.c_void_p()
    n, rep, stride, inc, ps = C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint32(), C.c_int64()
    rc = _lib().mser_marn_cell_ext_link_bwd(C.byref(desc), direction, int(partner_wgs), C.byref(dhq), C.byref(parts), C.byref(n), C.byref(ps),
                                            C.byref(cnt), C.byref(rep), C.byref(stride), C.byref(inc))
    if rc < 0:
        L.check(rc, "marn_cell_ext_link_bwd")
    return dhq.value, parts.value, n.value, ps.value, cnt.value, inc.value, rc == 1


def marn_cell_fwd(desc: L.CellDesc) -> None:
    L.check(_lib().mser_marn_cell_fwd(C.byref(desc), _stream()), "marn_cell_fwd")


def marn_cell_bwd(desc: L.CellDesc) -> None:
    L.check(_lib().mser_marn_cell_bwd(C.byref(desc), _stream()), "marn_cell_bwd")


def marn_cell_status(desc: L.CellDesc) -> None:
    """Synchronising diagnostic: raises if a persistent recurrent kernel gave up at one of its bounded barriers."""
    L.check(_lib().mser_marn_cell_status(C.byref(desc), _stream()), "marn_cell_status")


MSER_OPT_PERSISTENT = 1
MSER_OPT_WGRAD_INKERNEL = 2
MSER_OPT_BPTT_KSPLIT = 3
MSER_OPT_XCD_PLACEMENT = 4
MSER_OPT_FWD_STATS_ROLES = 5
MSER_OPT_FWD_SENTINEL = 6
MSER_OPT_BWD_SENTINEL = 7
MSER_OPT_H256_SPLIT = 8
MSER_OPT_SPK_BWD_KSPLIT = 9
MSER_OPT_BWD_POLL_DELAY = 10
MSER_OPT_WIDE_PERSISTENT = 11
MSER_OPT_DRNN_PERSISTENT = 12


def set_option(key: int, value: int) -> None:
    L.check(_lib().mser_set_option(key, value), "set_option")


PHASE_SPEAKER_FWD, PHASE_LSTHM_FWD, PHASE_LSTHM_BWD, PHASE_SPEAKER_BWD, PHASE_LSTHM_BWD_DX, PHASE_LSTHM_WGRAD = 1, 2, 4, 8, 16, 32
PHASE_FWD_PREP, PHASE_BWD_PREP, PHASE_SEPARATE_SPEAKER, PHASE_PREP_BOTH, PHASE_LSTHM_PRE, PHASE_PRE_DONE = 64, 128, 256, 512, 1024, 2048
PHASE_LSTHM_PRE_L, PHASE_LSTHM_PRE_A = 4096, 8192


def marn_cell_pipelined(B: int, H: int, ndir: int) -> bool:
    return bool(L.load().mser_marn_cell_pipelined(B, H, ndir))


def marn_cell_run(desc: L.CellDesc, phases: int) -> None:
    L.check(_lib().mser_marn_cell_run(C.byref(desc), phases, _stream()), "marn_cell_run")
