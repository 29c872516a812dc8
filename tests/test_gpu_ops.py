"""GPU parity of the libmser primitives against the CPU oracle / plain fp32 torch on the host (run with -m gpu).

Tolerances (fp32): primitives 2e-5 relative to the operand scale; they are written next to each assertion.
Every call goes through the C-ABI (ctypes -> libmser.so).
"""
import numpy as np
import pytest
import torch

from gpu_util import maxabs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mser import ops, _lib
    _lib.load()
    return ops


def _rand(*s, seed=0, scale=1.0):
    return torch.tensor(np.random.RandomState(seed).standard_normal(s).astype(np.float32) * scale)


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (64, 64, 16), (100, 37, 53), (257, 130, 100), (33, 512, 768), (4096, 100, 320)])
def test_gemm_linear(env, M, N, K):
    ops = env
    x, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.1), _rand(N, seed=3)
    r1 = _rand(M, N, seed=4)
    out = torch.empty(M, N, device="cuda")
    ops.linear(x.cuda(), W.cuda(), out, bias=b.cuda(), relu=True, R1=r1.cuda())
    ref = torch.relu(x.double() @ W.double().t() + b.double()) + r1.double()
    assert float((out.cpu().double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_gemm_strided_batched_splitk(env):
    ops = env
    # A given as a column slice (ld > K), B in [K,N] layout, transposed-A weight gradient with split-K atomics
    big = _rand(300, 150, seed=5)
    x = big.cuda()[:, 20:120]                        # [300,100] view, ld 150
    Wkn = _rand(100, 70, seed=6, scale=0.1)
    out = torch.zeros(300, 70, device="cuda")
    ops.matmul(x, Wkn.cuda(), out)
    ref = big[:, 20:120].double() @ Wkn.double()
    assert float((out.cpu().double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    dy = _rand(300, 70, seed=7)
    gW = torch.zeros(100, 70, device="cuda")
    ops.grad_weight(dy.cuda(), x, gW, transposed=True, splitk=8)
    refg = big[:, 20:120].double().t() @ dy.double()
    assert float((gW.cpu().double() - refg).abs().max()) < 3e-5 * float(refg.abs().max())
    gW2 = torch.zeros(70, 100, device="cuda")
    ops.grad_weight(dy.cuda(), x, gW2, splitk=4)
    assert float((gW2.cpu().double() - refg.t()).abs().max()) < 3e-5 * float(refg.abs().max())
    # two-level batch: C[b,h] = A[b,h] @ B[b,h]^T with head-interleaved rows
    nb, nh, L, d = 3, 4, 17, 8
    q, k = _rand(nb * L, nh * d, seed=8), _rand(nb * L, nh * d, seed=9)
    S = torch.empty(nb, nh, L, L, device="cuda")
    ops.gemm_raw(q.cuda(), k.cuda(), S, L, L, d, nh * d, 1, 1, nh * d, L, batch=(nb, nh), sA=(L * nh * d, d), sB=(L * nh * d, d),
                 sC=(nh * L * L, L * L), alpha=0.5)
    qh = q.view(nb, L, nh, d).permute(0, 2, 1, 3).double()
    kh = k.view(nb, L, nh, d).permute(0, 2, 1, 3).double()
    ref = 0.5 * qh @ kh.transpose(2, 3)
    assert float((S.cpu().double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("n", [1, 7, 64, 128, 200, 256])
def test_softmax_fwd_bwd(env, n):
    ops = env
    rows = 37
    s = _rand(rows, n, seed=n, scale=3.0)
    S = s.clone().cuda()
    ops.softmax_rows_(S, rows, n, n)
    ref = torch.softmax(s.double(), -1)
    assert float((S.cpu().double() - ref).abs().max()) < 2e-6
    dP = _rand(rows, n, seed=n + 1)
    d = dP.clone().cuda()
    ops.softmax_bwd_rows_(S, d, rows, n, n)
    refd = ref * (dP.double() - (ref * dP.double()).sum(-1, keepdim=True))
    assert float((d.cpu().double() - refd).abs().max()) < 5e-6


def test_softmax_mask_and_weights(env):
    ops = env
    rows, n = 19, 33
    s = _rand(rows, n, seed=3)
    w = torch.tensor(np.random.RandomState(4).rand(rows, n).astype(np.float32))
    m = torch.tensor(np.random.RandomState(5).rand(rows, n) < 0.3)
    m[:, 0] = False
    S = s.clone().cuda()
    ops.softmax_rows_(S, rows, n, n, mul=w.cuda(), mask=m.to(torch.uint8).cuda(), mask_on=1, fill=float("-inf"))
    ref = torch.softmax((s * w).masked_fill(m, float("-inf")).double(), -1)
    assert float((S.cpu().double() - ref).abs().max()) < 2e-6
    keep = (~m).to(torch.uint8)
    S = s.clone().cuda()
    ops.softmax_rows_(S, rows, n, n, mask=keep.cuda(), mask_on=0, fill=-1e9)
    ref = torch.softmax(s.masked_fill(keep == 0, -1e9).double(), -1)
    assert float((S.cpu().double() - ref).abs().max()) < 2e-6


def test_layernorm_fwd_bwd(env):
    ops = env
    rows, D = 203, 100
    x, r = _rand(rows, D, seed=1), _rand(rows, D, seed=2)
    g, b = 1 + 0.1 * _rand(D, seed=3), 0.1 * _rand(D, seed=4)
    y, ssum = torch.empty(rows, D, device="cuda"), torch.empty(rows, D, device="cuda")
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.add_layernorm_fwd(x.cuda(), r.cuda(), g.cuda(), b.cuda(), y, ssum, mean, rstd, 1e-6)
    xs = (x + r).double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xs, (D,), gd, bd, 1e-6)
    assert float((y.cpu().double() - ref).abs().max()) < 5e-6
    dy = _rand(rows, D, seed=5)
    ref.backward(dy.double())
    dx, dg, db = torch.empty(rows, D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ops.layernorm_bwd(dy.cuda(), ssum, mean, rstd, g.cuda(), dx, dg, db)
    assert float((dx.cpu().double() - xs.grad).abs().max()) < 2e-5
    assert float((dg.cpu().double() - gd.grad).abs().max()) < 2e-4
    assert float((db.cpu().double() - bd.grad).abs().max()) < 2e-4


def test_reverse_and_slot_tables(env):
    ops = env
    from oracle import ref_cpu as O
    L, B = 11, 5
    rs = np.random.RandomState(0)
    lens = np.array([11, 7, 1, 11, 4])
    umask = torch.tensor((np.arange(L)[None, :] < lens[:, None]).astype(np.float32))
    X = _rand(L, B, 6, seed=1)
    lens_d, rev = torch.empty(B, dtype=torch.int32, device="cuda"), torch.empty(L, B, dtype=torch.int32, device="cuda")
    ops.build_reverse_index(umask.cuda(), lens_d, rev)
    assert lens_d.cpu().tolist() == lens.tolist()
    out = torch.empty(L * B, 6, device="cuda")
    ops.reverse_by_length(X.cuda().view(L * B, 6), rev, out, L, B)
    assert torch.equal(out.cpu().view(L, B, 6), O.reverse_seq(X, umask))          # bit-exact gather
    # slot tables (forward direction and through the reverse index)
    spk = rs.randint(0, 2, (L, B))
    spk[3] = 0
    spk[6] = 1
    qmask = torch.tensor(np.eye(2, dtype=np.float32)[spk]) * umask.t().unsqueeze(2)
    for use_rev in (False, True):
        party, perm = torch.empty(L, B, dtype=torch.int32, device="cuda"), torch.empty(L, B, dtype=torch.int32, device="cuda")
        n0, qm = torch.empty(L, dtype=torch.int32, device="cuda"), torch.empty(L, B, 2, device="cuda")
        ops.build_slot_tables(qmask.cuda(), rev if use_rev else None, party, perm, n0, qm)
        q_ref = O.reverse_seq(qmask, umask) if use_rev else qmask
        p_ref, perm_ref, n0_ref = O.slot_tables(q_ref)
        assert torch.equal(party.cpu().long(), p_ref)
        assert torch.equal(perm.cpu().long(), perm_ref)
        assert torch.equal(n0.cpu().long(), n0_ref)
        assert torch.equal(qm.cpu(), q_ref)


def test_loss_and_logsoftmax(env):
    ops = env
    from oracle import ref_cpu as O
    L, B, C = 9, 4, 6
    y = _rand(L * B, C, seed=1)
    lp = torch.empty(B * L, C, device="cuda")
    ops.logsoftmax_tb_fwd(y.cuda(), lp, L, B)
    ref = torch.log_softmax(y.double().view(L, B, C), 2).permute(1, 0, 2).reshape(B * L, C)
    assert float((lp.cpu().double() - ref).abs().max()) < 2e-6
    target = torch.tensor(np.random.RandomState(2).randint(0, C, B * L))
    mask = torch.tensor((np.random.RandomState(3).rand(B, L) < 0.7).astype(np.float32))
    out = torch.empty(2, device="cuda")
    ops.masked_nll_fwd(lp, target.cuda(), mask.cuda().view(-1), out)
    refl = O.masked_nll(ref.float(), target, mask)
    assert abs(float(out[0]) - float(refl)) < 2e-6
    dp = torch.empty(B * L, C, device="cuda")
    ops.masked_nll_bwd(target.cuda(), mask.cuda().view(-1), out, None, dp)
    pr = ref.float().clone().requires_grad_(True)
    O.masked_nll(pr, target, mask).backward()
    assert float((dp.cpu() - pr.grad).abs().max()) < 1e-7
    dy = torch.empty(L * B, C, device="cuda")
    ops.logsoftmax_tb_bwd(dp, lp, dy, L, B)
    yy = y.double().clone().requires_grad_(True)
    l2 = torch.log_softmax(yy.view(L, B, C), 2).permute(1, 0, 2).reshape(B * L, C)
    l2.backward(pr.grad.double())
    assert float((dy.cpu().double() - yy.grad).abs().max()) < 1e-6


def test_adam_flat(env):
    ops = env
    from oracle import ref_cpu as O
    n = 1000
    p, g = _rand(n, seed=1), _rand(n, seed=2)
    live = torch.ones(n, dtype=torch.uint8)
    live[100:200] = 0
    pd, m, v = p.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pr, mr, vr = p.clone(), torch.zeros(n), torch.zeros(n)
    for step in (1, 2, 3):
        ops.adam_flat(pd, g.cuda(), m, v, live.cuda(), step, 1e-3, wd=2e-5)
        O.adam_step(pr, g, mr, vr, step, 1e-3, wd=2e-5)
    pr[100:200] = p[100:200]
    assert float((pd.cpu() - pr).abs().max()) < 1e-6


def test_rank1_attention_and_lsthm_step(env, golden_dir):
    import os
    ops = env
    g = np.load(os.path.join(golden_dir, "modules.npz"))
    from oracle import ref_cpu as O
    P = O.seeded_params(seed=4)
    out = torch.empty(7, 128, device="cuda")
    ops.rank1_attention_fwd(torch.tensor(g["ca_x1"]).cuda(), torch.tensor(g["ca_x2"]).cuda(),
                            P["marn_cell_f.crossatt_l2a.Wq"].cuda(), P["marn_cell_f.crossatt_l2a.Wk"].cuda(), out)
    assert float(np.abs(out.cpu().numpy() - g["ca_out"]).max()) < 2e-5           # rank-1 form vs the reference's materialised form
    pre = "marn_cell_f.lsthm_l."
    c2, h2 = torch.empty(5, 128, device="cuda"), torch.empty(5, 128, device="cuda")
    ops.lsthm_step_fwd(*[torch.tensor(g[k]).cuda() for k in ("lsthm_x", "lsthm_c", "lsthm_h", "lsthm_z", "lsthm_s")],
                       *[P[pre + n].cuda() for n in ("W.weight", "W.bias", "U.weight", "U.bias", "V.weight", "V.bias", "S.weight", "S.bias")],
                       c2, h2)
    assert float(np.abs(c2.cpu().numpy() - g["lsthm_c2"]).max()) < 5e-6
    assert float(np.abs(h2.cpu().numpy() - g["lsthm_h2"]).max()) < 5e-6


def _enc_ref64(x, P, nb, L, nh, dk, mask=None):
    """float64 restatement of EncoderLayer (model/encoder.py:27-60, :71-86, :101-113, :130-133), batch-major rows."""
    D = x.shape[1]
    X = x.double().view(nb, L, D)
    g = lambda n: P[n].double()
    q = (X @ g("slf_attn.w_qs.weight").t()).view(nb, L, nh, dk).transpose(1, 2)
    k = (X @ g("slf_attn.w_ks.weight").t()).view(nb, L, nh, dk).transpose(1, 2)
    v = (X @ g("slf_attn.w_vs.weight").t()).view(nb, L, nh, dk).transpose(1, 2)
    S = (q / dk ** 0.5) @ k.transpose(2, 3)
    if mask is not None:
        S = S.masked_fill(mask.view(nb, 1, L, L) == 0, -1e9)
    A = torch.softmax(S, -1)
    O = (A @ v).transpose(1, 2).reshape(nb, L, nh * dk)
    y1 = O @ g("slf_attn.fc.weight").t() + X
    e1 = torch.nn.functional.layer_norm(y1, (D,), g("slf_attn.layer_norm.weight"), g("slf_attn.layer_norm.bias"), 1e-6)
    h = torch.relu(e1 @ g("pos_ffn.w_1.weight").t() + g("pos_ffn.w_1.bias"))
    y2 = h @ g("pos_ffn.w_2.weight").t() + g("pos_ffn.w_2.bias") + e1
    out = torch.nn.functional.layer_norm(y2, (D,), g("pos_ffn.layer_norm.weight"), g("pos_ffn.layer_norm.bias"), 1e-6)
    return out.reshape(nb * L, D), A


@pytest.mark.parametrize("nb,L,time_major,use_mask", [(2, 16, False, False), (3, 50, True, False), (4, 128, True, False),
                                                     (2, 33, False, True), (1, 1, False, False)])
def test_fused_encoder_layer_fwd_bwd(env, nb, L, time_major, use_mask):
    """Fused EncoderLayer (csrc/encoder.hip) vs a float64 torch reference (outputs 2e-5, gradients 2e-4 of the tensor's
    scale) and vs the composed path (generic GEMM + row kernels) on the same inputs."""
    from mser import functional as F_
    from mser.functional import Layout
    D, nh, dk, dff = 100, 8, 40, 40
    rs = np.random.RandomState(11)
    shapes = {"slf_attn.w_qs.weight": (nh * dk, D), "slf_attn.w_ks.weight": (nh * dk, D), "slf_attn.w_vs.weight": (nh * dk, D),
              "slf_attn.fc.weight": (D, nh * dk), "slf_attn.layer_norm.weight": (D,), "slf_attn.layer_norm.bias": (D,),
              "pos_ffn.w_1.weight": (dff, D), "pos_ffn.w_1.bias": (dff,), "pos_ffn.w_2.weight": (D, dff), "pos_ffn.w_2.bias": (D,),
              "pos_ffn.layer_norm.weight": (D,), "pos_ffn.layer_norm.bias": (D,)}
    P = {}
    for n, s in shapes.items():
        w = rs.standard_normal(s).astype(np.float32) * (0.15 if len(s) == 2 else 0.3)
        if n.endswith("layer_norm.weight"):
            w = w + 1.0
        P[n] = torch.tensor(w)
    # w_qs | w_ks | w_vs back to back in one storage, like the flat parameter buffer (enables the single N = 960 projection)
    flat = torch.cat([P[n].reshape(-1) for n in ("slf_attn.w_qs.weight", "slf_attn.w_ks.weight", "slf_attn.w_vs.weight")]).cuda()
    Pg = {n: t.cuda() for n, t in P.items()}
    for i, n in enumerate(("slf_attn.w_qs.weight", "slf_attn.w_ks.weight", "slf_attn.w_vs.weight")):
        Pg[n] = flat[i * nh * dk * D:(i + 1) * nh * dk * D].view(nh * dk, D)
    xb = torch.tensor(rs.standard_normal((nb * L, D)).astype(np.float32))          # batch-major rows b*L + l
    dob = torch.tensor(rs.standard_normal((nb * L, D)).astype(np.float32))
    mask = None
    if use_mask:
        mask = torch.tensor((rs.rand(nb, L, L) > 0.3).astype(np.uint8))
        mask[:, :, 0] = 1
    # reference
    Pr = {n: t.clone().double().requires_grad_(True) for n, t in P.items()}
    xr = xb.clone().double().requires_grad_(True)
    out_r, A_r = _enc_ref64(xr, Pr, nb, L, nh, dk, mask)
    (out_r * dob.double()).sum().backward()

    def to_layout(t):      # batch-major rows -> the layout under test
        return t.view(nb, L, -1).transpose(0, 1).reshape(nb * L, -1).contiguous() if time_major else t

    def from_layout(t):
        return t.view(L, nb, -1).transpose(0, 1).reshape(nb * L, -1) if time_major else t

    lay = Layout.time_major(L, nb) if time_major else Layout.batch_major(nb, L)
    m8 = mask.view(nb, 1, L, L).expand(nb, nh, L, L).contiguous().cuda() if mask is not None else None
    res = {}
    for fused in (True, False):
        F_.FUSED_ENCODER = fused
        try:
            G = {n: torch.zeros_like(t) for n, t in Pg.items()}
            out, c = F_.encoder_layer_fwd(to_layout(xb).cuda(), None, Pg.__getitem__, lay, nh, dk, dk, mask=m8)
            assert isinstance(c, F_.EncFusedCtx) == fused
            dx = F_.encoder_layer_bwd(c, to_layout(dob).cuda(), Pg.__getitem__, G.__getitem__)
            torch.cuda.synchronize()
            res[fused] = (from_layout(out.cpu()), F_.encoder_attention(c).cpu(), from_layout(dx.cpu()), {n: g.cpu() for n, g in G.items()})
        finally:
            F_.FUSED_ENCODER = True
    for fused in (True, False):
        out, A, dx, G = res[fused]
        assert float((out.double() - out_r.detach()).abs().max()) < 2e-5, fused
        assert float((A.double() - A_r.detach()).abs().max()) < 5e-6, fused
        assert float((dx.double() - xr.grad).abs().max()) < 2e-4 * max(1.0, float(xr.grad.abs().max())), fused
        for n in P:
            ref = Pr[n].grad
            assert float((G[n].double() - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max())), (fused, n)


def test_gemm_grouped_matches_single_launches(env):
    """mser_gemm_grouped: a mix of weight-gradient shaped products (split-K atomics, different load-mode classes, an empty and a
    tiny member) against float64 references; members land in as few launches as their classes allow."""
    import ctypes as C
    from mser import _lib
    lib = _lib.load()
    rs = np.random.RandomState(3)
    rows = 517
    specs = [(512, 128), (512, 100), (40, 100), (100, 320), (6, 32), (0, 16), (130, 7)] + [(96, 64)] * 14   # > 16 members of one class
    descs, refs, outs, keep = [], [], [], []
    for N, K in specs:
        dy = torch.tensor(rs.standard_normal((rows, max(N, 1))).astype(np.float32))[:, :N].contiguous()
        x = torch.tensor(rs.standard_normal((rows, K)).astype(np.float32))
        init = torch.tensor(rs.standard_normal((N, K)).astype(np.float32))
        dyg, xg, out = dy.cuda(), x.cuda(), init.clone().cuda()
        d = _lib.GemmDesc()
        d.A, d.B, d.C = dyg.data_ptr() if N else xg.data_ptr(), xg.data_ptr(), out.data_ptr() if N else xg.data_ptr()
        d.M, d.N, d.K = N, K, rows
        d.sAm, d.sAk, d.sBk, d.sBn, d.ldc = 1, max(N, 1), K, 1, K
        d.batch1 = d.batch2 = 1
        d.alpha, d.splitk = 1.0, 16
        descs.append(d)
        keep.append((dyg, xg))
        outs.append(out)
        refs.append(init.double() + dy.double().t() @ x.double())
    arr = (_lib.GemmDesc * len(descs))(*descs)
    _lib.check(lib.mser_gemm_grouped(arr, len(descs), torch.cuda.current_stream().cuda_stream), "gemm_grouped")
    torch.cuda.synchronize()
    for (N, K), out, ref in zip(specs, outs, refs):
        if N:
            assert float((out.cpu().double() - ref).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max())), (N, K)


@pytest.mark.parametrize("L,B,d_r,d_a", [(5, 3, 1024, 100), (7, 2, 30, 5), (128, 32, 768, 100)])
def test_ingest_features_bit_exact(env, L, B, d_r, d_a):
    """x = cat((r1+r2+r3+r4)/4, acouf) (reference model_trainer.py:104-105): bit-identical to the torch CPU expression."""
    ops = env
    rs = np.random.RandomState(2)
    r = [torch.tensor(rs.standard_normal((L, B, d_r)).astype(np.float32)) for _ in range(4)]
    ac = torch.tensor(rs.standard_normal((L, B, d_a)).astype(np.float32))
    ref = torch.cat(((r[0] + r[1] + r[2] + r[3]) / 4, ac), dim=-1)
    out = ops.ingest_features(*[t.cuda() for t in r], ac.cuda())
    assert out.shape == ref.shape and torch.equal(out.cpu(), ref)


def test_confusion_update_matches_numpy(env):
    ops = env
    from mser.metrics import confusion_matrix
    rs = np.random.RandomState(4)
    rows, C = 4099, 6
    lp = torch.tensor(rs.standard_normal((rows, C)).astype(np.float32))
    lp[5, 1] = lp[5, 3] = 9.0                            # tie -> first maximum, like torch.argmax on the host
    label = torch.tensor(rs.randint(0, C, rows).astype(np.int64))
    mask = torch.tensor((rs.rand(rows) > 0.25).astype(np.float32))
    conf = torch.zeros(C, C, dtype=torch.float64, device="cuda")
    pred = torch.empty(rows, dtype=torch.int64, device="cuda")
    ops.confusion_update(lp.cuda(), label.cuda(), mask.cuda(), conf, pred)
    ops.confusion_update(lp.cuda(), label.cuda(), mask.cuda(), conf, None)        # accumulates across calls
    p_ref = lp.argmax(1)
    assert torch.equal(pred.cpu(), p_ref) and int(pred[5]) == 1
    assert np.array_equal(conf.cpu().numpy(), 2 * confusion_matrix(label.numpy(), p_ref.numpy(), mask.numpy(), C))


@pytest.mark.parametrize("T,B,use_drop,use_rev", [(7, 37, False, False), (5, 32, True, True), (3, 1, True, False)])
def test_gru_speaker_chain_fwd_bwd(env, T, B, use_drop, use_rev):
    """Speaker state of the GRU-speaker variants (reference model/lsthm_onlysp.py:170-181): one workgroup per
    32-dialogue block, W_hh register-resident.  Against the same recurrence in torch fp32 on the CPU (autograd for the backward):
    h_s for every step, the gradients at gi and -- through dgh and the saved qs0 -- at W_hh / b_hh; padded (all-zero qmask) rows,
    a partial second block, dropout on the carried state, and the optional scatter into the cell's output rows."""
    ops = env
    from mser import functional as F_
    H = 128
    rs = np.random.RandomState(T * 100 + B)
    gi = torch.tensor(rs.standard_normal((T * B, 3 * H)).astype(np.float32) * 0.7)
    w_hh = torch.tensor(rs.uniform(-0.09, 0.09, (3 * H, H)).astype(np.float32))
    b_hh = torch.tensor(rs.uniform(-0.09, 0.09, (3 * H,)).astype(np.float32))
    spk = rs.randint(0, 2, (T, B))
    qmask = np.eye(2, dtype=np.float32)[spk]
    if T > 2 and B > 2:
        qmask[T - 2:, 1] = 0                              # padded tail of dialogue 1
        qmask[:, 2] = np.eye(2, dtype=np.float32)[0]      # one party only
    qmask = torch.tensor(qmask)
    rev = None
    if use_rev:
        rv = np.tile(np.arange(T - 1, -1, -1, dtype=np.int32)[:, None], (1, B))
        rv[0, 0] = -1                                     # an output row that must be skipped
        rev = torch.tensor(rv)
    drop = None
    if use_drop:
        rng = torch.tensor([1234, 5], dtype=torch.int32, device="cuda")
        drop = F_.DropSite(rng, 77, 0.5)
    hs = torch.empty(T * B, H, device="cuda")
    save = torch.empty(T * B, 5 * H, device="cuda")
    out = torch.zeros(T * B, 4 * H, device="cuda")
    # (the descriptor holds raw pointers: the device operands must stay referenced for as long as it is used)
    gi_d, w_d, b_d, qm_d = gi.cuda(), w_hh.cuda(), b_hh.cuda(), qmask.cuda()
    rev_d = rev.cuda() if rev is not None else None
    d = ops.gru_speaker_desc(T, B, H, gi_d, w_d, b_d, qm_d, hs, save, out=out[:, 3 * H:], rev=rev_d, drop=drop)
    ops.gru_speaker_fwd(d)
    # ---- the same recurrence on the CPU
    f = drop.scale(T * B * H).cpu().view(T, B, H) if drop is not None else None
    gi_r, w_r, b_r = gi.clone().requires_grad_(True), w_hh.clone().requires_grad_(True), b_hh.clone().requires_grad_(True)
    q = torch.zeros(B, 2, H)
    rows = torch.arange(B)
    ref = []
    for t in range(T):
        idx = torch.argmax(qmask[t], 1)
        h0 = q[rows, idx]
        g = gi_r[t * B:(t + 1) * B]
        gh = h0 @ w_r.t() + b_r
        r = torch.sigmoid(g[:, :H] + gh[:, :H])
        z = torch.sigmoid(g[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(g[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h0
        if f is not None:
            h = h * f[t]
        m = qmask[t].unsqueeze(2)
        q = q * (1 - m) + h.unsqueeze(1) * m
        ref.append(h)
    ref = torch.cat(ref, 0)
    assert maxabs(hs, ref) < 2e-6
    exp_out = torch.zeros(T * B, H)
    for t in range(T):
        for b in range(B):
            tau = int(rev[t, b]) if rev is not None else t
            if tau >= 0:
                exp_out[tau * B + b] = ref[t * B + b].detach()
    assert maxabs(out[:, 3 * H:], exp_out) < 2e-6 and float(out[:, :3 * H].abs().max()) == 0.0
    # ---- backward
    w = torch.tensor(rs.standard_normal((T * B, H)).astype(np.float32))
    w2 = torch.tensor(rs.standard_normal((T * B, H)).astype(np.float32)) * 0.3
    (ref * (w + w2)).sum().backward()
    dgi, dgh = torch.empty(T * B, 3 * H, device="cuda"), torch.empty(T * B, 3 * H, device="cuda")
    w_dev, w2_dev = w.cuda(), w2.cuda()
    ops.gru_speaker_bwd(d, w_dev, dgi, dgh, dhs_add=[w2_dev])
    assert maxabs(dgi, gi_r.grad) < 3e-5 * max(1.0, float(gi_r.grad.abs().max()))
    dW = dgh.cpu().t() @ save[:, :H].cpu()
    assert maxabs(dW, w_r.grad) < 3e-4 * max(1e-3, float(w_r.grad.norm()))
    assert maxabs(dgh.cpu().sum(0), b_r.grad) < 3e-4 * max(1e-3, float(b_r.grad.norm()))


@pytest.mark.parametrize("nb,Lq,Lk,nh,dk,p", [(3, 128, 128, 1, 128, 0.0), (2, 37, 50, 1, 128, 0.0), (2, 96, 70, 8, 16, 0.0), (2, 33, 128, 1, 128, 0.3),
                                               (1, 1, 1, 1, 8, 0.0), (2, 64, 64, 2, 64, 0.25)])
def test_fused_sequence_cross_attention_core(env, nb, Lq, Lk, nh, dk, p):
    """mser_xattn_seq_fwd / bwd (csrc/xattn.hip: QK^T -> softmax -> .V per 32-query tile, no [B,L,L] in HBM) against a float64
    restatement of the core of CrossAttention2/3 (model/lsthm_sps.py:93-99), time-major rows, with and without the dropout of :98
    (mask for mask: the factors are read back through mser_dropout_scale), ragged tile edges, multi-head."""
    ops = env
    from mser import _lib as L_
    from mser.functional import DropSite
    rs = np.random.RandomState(nb * 131 + Lq + Lk + dk)
    D = nh * dk
    q = torch.tensor(rs.standard_normal((Lq * nb, D)).astype(np.float32)).cuda()
    kv = torch.tensor(rs.standard_normal((Lk * nb, 2 * D)).astype(np.float32)).cuda()
    dO = torch.tensor(rs.standard_normal((Lq * nb, D)).astype(np.float32)).cuda()
    out = torch.zeros(Lq * nb, D, device="cuda")
    stats = torch.empty(nb, nh, Lq, 2, device="cuda")
    d = L_.XAttnDesc()
    d.nb, d.nh, d.Lq, d.Lk, d.dk = nb, nh, Lq, Lk, dk
    k_, v_ = kv[:, :D], kv[:, D:]
    d.q, d.ldq, d.k, d.ldk, d.v, d.ldv = q.data_ptr(), D, k_.data_ptr(), 2 * D, v_.data_ptr(), 2 * D
    d.sbq, d.slq, d.sbk, d.slk = 1, nb, 1, nb                    # time-major rows: row(b, l) = l*nb + b
    d.o, d.ldo, d.stats, d.scale = out.data_ptr(), D, stats.data_ptr(), 1.0 / np.sqrt(dk)
    drop = None
    if p > 0:
        rng = torch.tensor([1234, 7], dtype=torch.int32, device="cuda")
        drop = DropSite(rng, 9, p)
        d.rng, d.site, d.p = rng.data_ptr(), 9, p
    assert ops.xattn_seq_supported(d)
    ops.xattn_seq_fwd(d)
    dq = torch.empty_like(q)
    # round 3: every 32-query tile stores its dK / dV contribution into its own slab (no pre-zeroing: they start as NaN here), the
    # launch's second kernel adds the slabs in a fixed order -> a second run is bit-identical (round 2: float atomics)
    nslab = (Lq + 31) // 32
    slabs = torch.full((nslab,) + tuple(kv.shape), float("nan"), device="cuda")
    dkv = slabs[0]
    if nslab == 1:
        dkv.zero_()
    d.dO, d.lddo, d.dq, d.lddq = dO.data_ptr(), D, dq.data_ptr(), D
    d.dk_, d.lddk, d.dv, d.lddv = dkv.data_ptr(), 2 * D, dkv[:, D:].data_ptr(), 2 * D
    d.part_stride = slabs.stride(0) if nslab > 1 else 0
    ops.xattn_seq_bwd(d)
    dq1, dkv1 = dq.clone(), dkv.clone()
    dq.fill_(float("nan")); slabs.fill_(float("nan"))
    if nslab == 1:
        dkv.zero_()
    ops.xattn_seq_bwd(d)
    assert torch.equal(dq, dq1) and torch.equal(dkv, dkv1)
    # and the atomic form (part_stride = 0, zeroed targets) still gives the same values to rounding
    dkv_a = torch.zeros_like(kv)
    d.dk_, d.dv, d.part_stride = dkv_a.data_ptr(), dkv_a[:, D:].data_ptr(), 0
    ops.xattn_seq_bwd(d)
    assert float((dkv_a - dkv).abs().max()) < 1e-5 * max(1.0, float(dkv.abs().max()))
    # float64 reference with autograd
    Q = q.cpu().double().view(Lq, nb, nh, dk).permute(1, 2, 0, 3).requires_grad_(True)          # [nb, nh, Lq, dk]
    K = kv.cpu().double()[:, :D].reshape(Lk, nb, nh, dk).permute(1, 2, 0, 3).requires_grad_(True)
    V = kv.cpu().double()[:, D:].reshape(Lk, nb, nh, dk).permute(1, 2, 0, 3).requires_grad_(True)
    P = torch.softmax(Q @ K.transpose(2, 3) / np.sqrt(dk), -1)
    if drop is not None:
        P = P * drop.scale(nb * nh * Lq * Lk).cpu().double().view(nb, nh, Lq, Lk)
    O = (P @ V).permute(2, 0, 1, 3).reshape(Lq * nb, D)
    (O * dO.cpu().double()).sum().backward()
    assert float((out.cpu().double() - O.detach()).abs().max()) < 2e-5
    for got, ref in ((dq, Q.grad.permute(2, 0, 1, 3).reshape(Lq * nb, D)), (dkv[:, :D], K.grad.permute(2, 0, 1, 3).reshape(Lk * nb, D)),
                     (dkv[:, D:], V.grad.permute(2, 0, 1, 3).reshape(Lk * nb, D))):
        assert float((got.cpu().double() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
