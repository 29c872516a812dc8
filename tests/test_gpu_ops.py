"""GPU parity of the libmser primitives against the CPU oracle / plain fp32 torch on the host (run with -m gpu).

Tolerances (fp32): primitives 2e-5 relative to the operand scale; they are written next to each assertion.
Every call goes through the C-ABI (ctypes -> libmser.so).
"""
import numpy as np
import pytest
import torch

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
