"""Pin the CPU oracle (oracle/ref_cpu.py) against golden vectors produced by the reference itself
(tests/golden/make_golden.py, run in the build container with /root/reference importable).

CPU only.  Tolerances: the oracle is a re-ordering of the same fp32 torch ops, so agreement is ~1e-6;
the stated gate is 2e-5 on log-probs (the reference's own fp32-vs-fp64 noise is up to 4.7e-5).
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

torch.set_num_threads(min(8, os.cpu_count() or 1))


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _t(a):
    return torch.tensor(np.asarray(a))


def test_modules(golden_dir):
    g = _g(golden_dir, "modules.npz")
    P = O.seeded_params(seed=4)
    c2, h2 = O.lsthm1(P, "marn_cell_f.lsthm_l.", _t(g["lsthm_x"]), _t(g["lsthm_c"]), _t(g["lsthm_h"]),
                      _t(g["lsthm_z"]), _t(g["lsthm_s"]))
    np.testing.assert_allclose(c2.numpy(), g["lsthm_c2"], atol=2e-6)
    np.testing.assert_allclose(h2.numpy(), g["lsthm_h2"], atol=2e-6)
    o = O.cross_attention(P, "marn_cell_f.crossatt_l2a.", _t(g["ca_x1"]), _t(g["ca_x2"]))
    np.testing.assert_allclose(o.numpy(), g["ca_out"], atol=2e-6)
    o = O.cross_attention_seq(P, "crossatt_l2a.", _t(g["ca2_x1"]), _t(g["ca2_x2"]))
    np.testing.assert_allclose(o.numpy(), g["ca2_out"], atol=5e-6)
    o = O.cross_attention_seq(P, "crossatt_l2a_1.", _t(g["ca2_x1"]), _t(g["ca3_x2"]))
    np.testing.assert_allclose(o.numpy(), g["ca3_out"], atol=5e-6)
    eo, ea = O.encoder_layer(P, "encoder_l.", _t(g["enc_x"]))
    np.testing.assert_allclose(eo.numpy(), g["enc_out"], atol=5e-6)
    np.testing.assert_allclose(ea.numpy(), g["enc_attn"], atol=2e-6)
    PS = {k[len("sa_p/"):]: _t(g[k]) for k in g.files if k.startswith("sa_p/")}
    o = O.self_attention_lib(PS, "", _t(g["sa_q"]), _t(g["sa_k"]), _t(g["sa_k"]), 4, 16, 16)
    np.testing.assert_allclose(o.numpy(), g["sa_out"], atol=2e-6)
    o = O.self_attention_lib(PS, "", _t(g["sa_q"]), _t(g["sa_k"]), _t(g["sa_k"]), 4, 16, 16,
                             attention_mask=_t(g["sa_mask"]), attention_weights=_t(g["sa_w"]))
    np.testing.assert_allclose(o.numpy(), g["sa_out_mw"], atol=2e-6)


def _check_grads(g, named, atol_rel=2e-4):
    for name, p in named:
        gn = float(g["gnorm/" + name])
        if gn < 0:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, f"{name} should be dead"
            continue
        assert p.grad is not None, name
        got = p.grad.double().reshape(-1)
        assert abs(float(got.norm()) - gn) <= atol_rel * max(gn, 1e-3), (name, float(got.norm()), gn)
        idx = g["gidx/" + name]
        np.testing.assert_allclose(got[idx].numpy(), g["gval/" + name], atol=atol_rel * max(gn, 1e-3), err_msg=name)


def test_marn_cell(golden_dir):
    g = _g(golden_dir, "cell_T24_N6.npz")
    P = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(seed=3).items() if k.startswith("marn_cell_f.")}
    x_l = _t(g["x_l"]).requires_grad_(True)
    x_a = _t(g["x_a"]).requires_grad_(True)
    h = O.marn_cell(P, "marn_cell_f.", x_l, x_a, _t(g["qmask"]))
    np.testing.assert_allclose(h.detach().numpy(), g["h"], atol=3e-6)
    (h * _t(g["wsum"])).sum().backward()
    np.testing.assert_allclose(x_l.grad.numpy(), g["dx_l"], atol=2e-5)
    np.testing.assert_allclose(x_a.grad.numpy(), g["dx_a"], atol=2e-5)
    _check_grads(g, [(k[len("marn_cell_f."):], v) for k, v in P.items()])


@pytest.mark.parametrize("name", ["model_c1_B2_L16_dr1024.npz", "model_c1r_B3_L12_dr768_ragged.npz",
                                  "model_c2_B32_L128_dr768.npz"])
def test_model(golden_dir, name):
    g = _g(golden_dir, name)
    B, L, d_r, seed, ragged = int(g["B"]), int(g["L"]), int(g["d_r"]), int(g["seed"]), bool(g["ragged"])
    P = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(seed=seed, d_r=d_r).items()}
    x, qmask, umask, label = O.seeded_batch(B, L, d_r=d_r, seed=seed + 1, ragged=ragged)
    lp, x_l, x_a = O.marn1_sps_forward(P, x, qmask, umask, d_r=d_r)
    loss = O.masked_nll(lp, label.view(-1), umask)
    loss.backward()
    lpn = lp.detach().numpy()
    rows = g["rows"] if "rows" in g.files else np.arange(lpn.shape[0])
    err = np.abs(lpn[rows] - g["logits"]).max()
    assert err < 2e-5, err
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    # bit-exact argmax wherever the reference's own top-1/top-2 margin exceeds 2x the tolerance
    safe = g["margin"] > 4e-5
    assert (lpn.argmax(1)[safe] == g["argmax"][safe]).all()
    assert abs(float(x_l.double().sum()) - float(g["x_l_sum"])) < 1e-3 * max(1.0, float(g["x_l_abs"]) * 1e-3)
    _check_grads(g, list(P.items()))


def test_model_train_mode_with_fed_dropout_masks(golden_dir):
    """Train mode: the reference's own forward/backward with its 13 nn.Dropout calls fed from O.seeded_drops
    (tests/golden/make_golden.py::train_mode_case) against the oracle's ``drops`` restatement -- pins where each dropout sits
    (carried states, unnormalised attention rows, before the residuals) and the call order inside MARN_cell (:180-188, :210-215)."""
    g = _g(golden_dir, "model_train_mode.npz")
    B, L, d_r, seed = int(g["B"]), int(g["L"]), int(g["d_r"]), int(g["seed"])
    P = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(seed=seed, d_r=d_r).items()}
    x, qmask, umask, label = O.seeded_batch(B, L, d_r=d_r, seed=seed + 1, ragged=True)
    dr = O.seeded_drops(B, L, seed=seed + 2)
    lp, _, _ = O.marn1_sps_forward(P, x, qmask, umask, d_r=d_r, drops=dr)
    loss = O.masked_nll(lp, label.view(-1), umask)
    loss.backward()
    assert np.abs(lp.detach().numpy() - g["logits"]).max() < 2e-5
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    _check_grads(g, list(P.items()))
    # and the masks matter: eval mode differs by far more than the tolerance
    lp0, _, _ = O.marn1_sps_forward(P, x, qmask, umask, d_r=d_r)
    assert np.abs(lp0.detach().numpy() - g["logits"]).max() > 1e-2


def test_onlysp_variant_vs_reference(golden_dir):
    """SURVEY 8(f) row f1 groundwork: the oracle's restatement of MARN1_onlysp (GRU speaker state per dialogue, no residual on the
    second encoder pass, nn_out head on the concatenation) against the reference's own eval-mode forward/backward."""
    g = _g(golden_dir, "model_onlysp.npz")
    B, L, d_r, seed = int(g["B"]), int(g["L"]), int(g["d_r"]), int(g["seed"])
    P = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(seed=seed, d_r=d_r, variant="onlysp").items()}
    x, qmask, umask, label = O.seeded_batch(B, L, d_r=d_r, seed=seed + 1, ragged=True)
    lp, x_l, _ = O.marn1_onlysp_forward(P, x, qmask, umask, d_r=d_r)
    loss = O.masked_nll(lp, label.view(-1), umask)
    loss.backward()
    assert np.abs(lp.detach().numpy() - g["logits"]).max() < 2e-5
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    assert abs(float(x_l.double().sum()) - float(g["x_l_sum"])) < 1e-2
    _check_grads(g, list(P.items()))


@pytest.mark.parametrize("tag", ["nsps", "no_en"])
def test_nsps_variants_vs_reference(golden_dir, tag):
    """the oracle's restatement of MARN1_nsps / MARN1_no_en (speaker GRU on the pre-encoder features with the
    listener blend, LayerNorm'd CrossAttention2, softmax(p)-weighted fusion + fc residual) against the reference's own eval-mode
    forward/backward (tests/golden/make_golden.py::nsps_cases), dead parameters included (gru_l, fc2, crossatt_a2l of the cells)."""
    g = _g(golden_dir, f"model_{tag}.npz")
    B, L, d_r, seed = int(g["B"]), int(g["L"]), int(g["d_r"]), int(g["seed"])
    P = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(seed=seed, d_r=d_r, variant="nsps").items()}
    x, qmask, umask, label = O.seeded_batch(B, L, d_r=d_r, seed=seed + 1, ragged=True)
    lp, x_l, x_a = O.marn1_nsps_forward(P, x, qmask, umask, d_r=d_r, no_en=(tag == "no_en"))
    loss = O.masked_nll(lp, label.view(-1), umask)
    loss.backward()
    assert np.abs(lp.detach().numpy() - g["logits"]).max() < 2e-5
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    assert abs(float(x_l.double().sum()) - float(g["x_l_sum"])) < 1e-2 and abs(float(x_a.double().sum()) - float(g["x_a_sum"])) < 1e-2
    _check_grads(g, list(P.items()))


def test_gru_variant_cells_vs_reference(golden_dir):
    """MARN_cell.forward of model/lsthm_onlysp.py:158-197 and model/lsthm_nsps.py:158-216 on their own, padded tails included
    (tests/golden/make_golden.py::gru_cell_cases): outputs and input gradients of oracle.marn_cell_onlysp / marn_cell_nsps."""
    g = _g(golden_dir, "cell_gru_variants.npz")
    qmask = _t(g["qmask"])
    for tag, variant in (("onlysp", "onlysp"), ("nsps", "nsps")):
        P = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(seed=52, variant=variant).items() if k.startswith("marn_cell_f.")}
        x_l, x_a, x = (_t(g[f"{tag}/{n}"]).requires_grad_(True) for n in ("x_l", "x_a", "x"))
        if tag == "onlysp":
            outs = (O.marn_cell_onlysp(P, "marn_cell_f.", x_l, x_a, qmask),)
        else:
            outs = O.marn_cell_nsps(P, "marn_cell_f.", x, x_l, x_a, qmask)
        for i, o in enumerate(outs):
            assert np.abs(o.detach().numpy() - g[f"{tag}/out{i}"]).max() < 5e-6, (tag, i)
        sum((o * _t(g[f"{tag}/w{i}"])).sum() for i, o in enumerate(outs)).backward()
        assert np.abs(x_l.grad.numpy() - g[f"{tag}/dx_l"]).max() < 2e-5 and np.abs(x_a.grad.numpy() - g[f"{tag}/dx_a"]).max() < 2e-5
        if tag == "nsps":
            assert np.abs(x.grad.numpy() - g[f"{tag}/dx"]).max() < 2e-5
        gg = {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + "/g")}
        _check_grads(gg, [(k[len("marn_cell_f."):], v) for k, v in P.items()])


@pytest.mark.parametrize("tag", ["small", "ref"])
def test_bimodel_vs_reference(golden_dir, tag):
    """the oracle's restatement of the DialogueRNN BiModel (global / party / listener / emotion GRUs, 'general'
    matching attention over the growing history, 'general2' attention over the bidirectional emotion states) against the
    reference's own eval-mode forward/backward (tests/golden/make_golden.py::bimodel_cases)."""
    g = _g(golden_dir, f"bimodel_{tag}.npz")
    dims = {k: int(g[k]) for k in ("D_m", "D_g", "D_p", "D_e", "D_h")}
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    P = {k: v.clone().requires_grad_(True) for k, v in O.bimodel_seeded_params(seed=seed, **dims).items()}
    U, qmask, umask, label = O.bimodel_seeded_batch(B, L, D_m=dims["D_m"], seed=seed + 1, ragged=True)
    lp, alpha, a_f, a_b = O.bimodel_forward(P, U, qmask, umask)
    lp_ = lp.transpose(0, 1).reshape(-1, lp.shape[2])
    loss = O.masked_nll(lp_, label.view(-1), umask)
    loss.backward()
    assert np.abs(lp.detach().numpy() - g["logits"]).max() < 2e-5
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    if tag == "small":
        assert np.abs(torch.stack(alpha, 0).detach().numpy() - g["alpha"]).max() < 2e-6
        for nm, al in (("alpha_f", a_f), ("alpha_b", a_b)):
            assert len(al) == L - 1
            for t, a in enumerate(al):
                assert np.abs(a.detach().numpy() - g[f"{nm}/{t + 1}"]).max() < 2e-6, (nm, t)
    _check_grads(g, list(P.items()))


@pytest.mark.parametrize("tag", ["long", "small_long"])
def test_bimodel_long_vs_reference(golden_dir, tag):
    """Round 3: the same restatement beyond 64 history rows -- the trainer's widths at B = 4 x L = 200 (BASELINE configs[3]'s length) and
    small widths at B = 33 x L = 150 (tests/golden/make_golden.py::bimodel_long_cases); three attention rows each (t = 70, 129, last)."""
    g = _g(golden_dir, f"bimodel_{tag}.npz")
    dims = {k: int(g[k]) for k in ("D_m", "D_g", "D_p", "D_e", "D_h")}
    B, L, seed = int(g["B"]), int(g["L"]), int(g["seed"])
    P = {k: v.clone().requires_grad_(True) for k, v in O.bimodel_seeded_params(seed=seed, **dims).items()}
    U, qmask, umask, label = O.bimodel_seeded_batch(B, L, D_m=dims["D_m"], seed=seed + 1, ragged=True)
    lp, alpha, a_f, a_b = O.bimodel_forward(P, U, qmask, umask)
    loss = O.masked_nll(lp.transpose(0, 1).reshape(-1, lp.shape[2]), label.view(-1), umask)
    loss.backward()
    assert np.abs(lp.detach().numpy() - g["logits"]).max() < 5e-5
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    ts = [int(t) for t in g["alpha_ts"]]
    assert np.abs(torch.stack([alpha[t] for t in ts], 0).detach().numpy() - g["alpha"]).max() < 2e-6
    for nm, al in (("alpha_f", a_f), ("alpha_b", a_b)):
        for t in ts:
            assert np.abs(al[t].detach().numpy() - g[f"{nm}/{t + 1}"]).max() < 2e-6, (nm, t)
    _check_grads(g, list(P.items()))


def test_library_self_attention_backward_vs_reference(golden_dir):
    """oracle.self_attention_lib forward AND backward (inputs and every parameter) against the reference's own autograd with
    mask and multiplicative weights (tests/golden/make_golden.py::sa_backward_case)."""
    g = _g(golden_dir, "sa_backward.npz")
    P = {k[len("sa_p/"):]: _t(g[k]).requires_grad_(True) for k in g.files if k.startswith("sa_p/")}
    q, k, v = (_t(g[n]).requires_grad_(True) for n in ("sa_q", "sa_k", "sa_v"))
    out = O.self_attention_lib(P, "", q, k, v, 4, 16, 16, attention_mask=torch.tensor(g["sa_mask"]), attention_weights=_t(g["sa_w"]))
    (out * _t(g["sa_wsum"])).sum().backward()
    assert np.abs(out.detach().numpy() - g["sa_out"]).max() < 2e-6
    for n, t in (("sa_dq", q), ("sa_dk", k), ("sa_dv", v)):
        assert np.abs(t.grad.numpy() - g[n]).max() < 1e-5 * max(1.0, np.abs(g[n]).max()), n
    for n, t in P.items():
        assert np.abs(t.grad.numpy() - g["sa_g/" + n]).max() < 1e-5 * max(1.0, np.abs(g["sa_g/" + n]).max()), n


def test_data_parallel_step_vs_reference(golden_dir):
    """the oracle run on two contiguous shards separately, gradients combined with the mask counts, one Adam step (twice)
    against the REFERENCE doing the same (tests/golden/make_golden.py::dp_case).  Pins what the two-rank tests must reproduce."""
    g = _g(golden_dir, "dp_two_shards.npz")
    d_r, Bg, Ln, world = int(g["d_r"]), int(g["B"]), int(g["L"]), int(g["world"])
    P = {k: v.clone() for k, v in O.seeded_params(seed=int(g["seed_params"]), d_r=d_r).items()}
    x, qmask, umask, label = O.seeded_batch(Bg, Ln, d_r=d_r, seed=int(g["seed_batch"]), ragged=True)
    per = Bg // world
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    losses = []
    for step in (1, 2):
        acc, cnt = {}, 0.0
        for r in range(world):
            sl = slice(r * per, (r + 1) * per)
            Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
            lp, _, _ = O.marn1_sps_forward(Pr, x[:, sl].contiguous(), qmask[:, sl].contiguous(), umask[sl].contiguous(), d_r=d_r)
            loss = O.masked_nll(lp, label[sl].reshape(-1), umask[sl])
            loss.backward()
            n = float(umask[sl].sum())
            losses.append(float(loss.detach()))
            for k_, v_ in Pr.items():
                if v_.grad is not None:
                    acc[k_] = acc.get(k_, 0) + n * v_.grad
            cnt += n
        for k_ in acc:
            O.adam_step(P[k_], acc[k_] / cnt, M[k_], V[k_], step, 1e-3, wd=2e-5)
    assert np.abs(np.array(losses) - g["shard_losses"]).max() < 2e-6
    for k_ in P:
        got = P[k_].reshape(-1).numpy()[g["idx/" + k_]]
        assert np.abs(got - g["p/" + k_]).max() < 2e-7, k_


def test_trainer_lr_schedule(golden_dir):
    g = _g(golden_dir, "trainer.npz")
    assert O.step_lr(1e-3, 0.98, 1, 1) == pytest.approx(float(g["lr1"]), rel=1e-12)
    assert O.step_lr(1e-3, 0.98, 1, 2) == pytest.approx(float(g["lr2"]), rel=1e-12)


def test_masked_loss_variants_vs_reference(golden_dir):
    """oracle.masked_loss against the reference's own loss.MaskedLoss (tests/golden/make_golden.py::loss_cases): NLL / CrossEntropy,
    plain / class-weighted, padded mask.  The CrossEntropy numbers include log(C) per masked row (2.63 vs 2.12 plain)."""
    g = np.load(os.path.join(golden_dir, "loss.npz"))
    lp, target, mask, w = _t(g["lp"]), torch.tensor(g["target"]), _t(g["mask"]), _t(g["weight"])
    for lname, is_ce in (("nll", False), ("ce", True)):
        for wname, ww in (("plain", None), ("weighted", w)):
            pr = lp.clone().requires_grad_(True)
            out = O.masked_loss(pr, target, mask, ww, is_ce)
            out.backward()
            assert abs(float(out.detach()) - float(g[f"{lname}_{wname}/loss"])) < 1e-6, (lname, wname)
            assert float((pr.grad - _t(g[f"{lname}_{wname}/dpred"])).abs().max()) < 1e-7, (lname, wname)
    # the weight-less NLL special case used by the model-level checks
    assert abs(float(O.masked_nll(lp, target, mask)) - float(g["nll_plain/loss"])) < 1e-6
