"""``MARN_cell.forward`` of the GRU-speaker variants on its own (reference model/lsthm_onlysp.py:158-197 and
model/lsthm_nsps.py:158-216): one direction, the speaker GRU chain followed by the cell's LSTHM launches on the external speaker
state (``mser_cell_desc::ext_hq``).  Inside the models both directions share the launches (mser.onlysp_fn / mser.nsps_fn); this is
the module-level surface of the same kernels."""
from __future__ import annotations

from typing import Optional

import torch

from . import functional as F_
from . import ops
from .onlysp_fn import gru_speaker_dir_bwd, gru_speaker_dir_fwd

Tensor = torch.Tensor

GRU_CELL_LIVE = ["lsthm_l.W.weight", "lsthm_l.W.bias", "lsthm_l.U.weight", "lsthm_l.U.bias", "lsthm_l.V.weight", "lsthm_l.V.bias",
                 "lsthm_l.S.weight", "lsthm_l.S.bias", "lsthm_a.W.weight", "lsthm_a.W.bias", "lsthm_a.U.weight", "lsthm_a.U.bias",
                 "lsthm_a.V.weight", "lsthm_a.V.bias", "lsthm_a.S.weight", "lsthm_a.S.bias",
                 "gru_s.weight_ih", "gru_s.weight_hh", "gru_s.bias_ih", "gru_s.bias_hh", "crossatt_l2a.Wq", "crossatt_l2a.Wk"]


def gru_cell_forward(P, u_l: Tensor, u_a: Tensor, x_l: Tensor, x_a: Tensor, qmask: Tensor, T: int, N: int, D: int, H: int,
                     lblend: bool, ds, da):
    """u_l / u_a [T*N, D]: the two halves of the GRU's input rows; x_l / x_a [T*N, D]: the LSTHM streams' inputs.  ds / da: DropSite
    (or None) of the cell's nn.Dropout (h_s: site, h_l / h_a: site + 1) and of the rank-1 attention (site + 2).
    Returns (out [T*N, 4H] = h_l | h_a | z | h_s, hli [T*N, H] or None, ctx)."""
    dev = x_l.device
    out = torch.empty(T * N, 4 * H, device=dev)
    hli = torch.empty(T * N, H, device=dev) if lblend else None
    cell_drop = None
    hs_site = None
    if ds is not None or da is not None:
        base = (ds or da).site if ds is not None else da.site - 2
        cell_drop = ((ds or da).rng, [base], [ds.p if ds else 0.0], [da.p if da else 0.0])
        hs_site = F_.DropSite(ds.rng, base, ds.p) if ds is not None else None
    g = gru_speaker_dir_fwd(P, u_l, u_a, qmask, None, out[:, 3 * H:], T, N, H, hs_site, launch=False, lblend=lblend, hli=hli)
    ws = torch.empty(ops.cell_workspace_bytes(T, N, D, H, 1), device=dev, dtype=torch.uint8)
    dirs = [dict(p=ops.cell_param_struct(P), qmask=qmask, rev=None, out=out)]
    desc = ops.make_cell_desc(T, N, D, H, x_l, x_a, dirs, 4 * H, ws, drop=cell_drop, ext_hq=[g.hs])
    ops.marn_cell_run(desc, ops.PHASE_FWD_PREP)
    ops.gru_speaker_fwd(g.desc)
    ops.marn_cell_run(desc, ops.PHASE_LSTHM_FWD)
    return out, hli, (g, ws, dirs, cell_drop, x_l, x_a, T, N, D, H)


def gru_cell_backward(ctx, P, G, dout: Tensor, dhli: Optional[Tensor], need_du: bool):
    """dout [T*N, 4H] (gradient of ``out``; the h_s quarter included), dhli [T*N, H] or None.
    Returns (dx_l, dx_a, du_l, du_a): gradients of the LSTHM inputs and (need_du) of the GRU input halves."""
    g, ws, dirs, cell_drop, x_l, x_a, T, N, D, H = ctx
    dev = dout.device
    dirs[0]["g"] = ops.cell_param_struct(G)
    dirs[0]["dout"] = dout
    dx_l, dx_a = torch.zeros(T * N, D, device=dev), torch.zeros(T * N, D, device=dev)
    dhq = torch.empty(T * N, H, device=dev)
    desc = ops.make_cell_desc(T, N, D, H, x_l, x_a, dirs, 4 * H, ws, dx_l=dx_l, dx_a=dx_a, drop=cell_drop, ext_hq=[g.hs], ext_dhq=[dhq])
    ops.marn_cell_run(desc, ops.PHASE_BWD_PREP | ops.PHASE_LSTHM_BWD)
    ops.marn_cell_run(desc, ops.PHASE_LSTHM_BWD_DX | ops.PHASE_LSTHM_WGRAD | ops.PHASE_SPEAKER_BWD)
    dgi, dgh = torch.empty(T * N, 3 * H, device=dev), torch.empty(T * N, 3 * H, device=dev)
    ops.gru_speaker_set_grads(g.desc, dhq, dgi, dgh)
    if dhli is not None:
        g.desc.dhli = dhli.data_ptr()
        g.desc._keep_hli = dhli
    ops.gru_speaker_bwd(g.desc)
    du_l = torch.zeros(T * N, D, device=dev) if need_du else None
    du_a = torch.zeros(T * N, D, device=dev) if need_du else None
    gru_speaker_dir_bwd(g, P, G, dgi, dgh, du_l, du_a, T, N, H)
    return dx_l, dx_a, du_l, du_a
