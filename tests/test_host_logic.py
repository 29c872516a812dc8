"""CPU: host-side logic of the drop-in mirrors (no kernels run): parameter names/shapes/initialisers, flat storage layout,
learning-rate schedule, constructor signatures, and the loud failure on CPU tensors (no fallback on the product path)."""
import inspect
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O


def test_state_dict_keys_and_shapes_match_reference():
    from models.lsthm_sps import MARN1_sps
    net = MARN1_sps(6)
    sd = net.state_dict()
    shapes = O.param_shapes(d_r=1024)
    assert list(sd.keys()) == list(shapes.keys())            # the reference's registration order
    for k, v in sd.items():
        assert tuple(v.shape) == shapes[k], k
    assert sum(p.numel() for p in net.parameters()) == 2488786   # SURVEY: 2,488,786 parameters


def test_initialisers():
    from models.lsthm_sps import MARN1_sps, CrossAttention2
    from attention.SelfAttention import ScaledDotProductAttention
    net = MARN1_sps(6)
    for n in ("w", "v", "v1", "v2", "crossatt_l2a.Wq", "crossatt_a2l_1.Wv", "marn_cell_f.crossatt_l2a.Wk"):
        assert torch.all(net.state_dict()[n] == 1.0), n           # ones, model/lsthm_sps.py:53-55,82-84,340-346
    sa = ScaledDotProductAttention(64, 16, 16, 4)
    assert float(sa.fc_q.weight.std()) < 2e-3 and float(sa.fc_q.bias.abs().max()) == 0.0   # N(0, 0.001), zero bias (:35-47)


@pytest.mark.parametrize("cls,args", [
    ("models.lsthm_sps:LSTHM1", ["cell_size", "in_size", "hybrid_in_size", "speaker_dim"]),
    ("models.lsthm_sps:CrossAttention", ["attn_dropout"]),
    ("models.lsthm_sps:CrossAttention2", ["dh", "dk", "dv", "attn_dropout"]),
    ("models.lsthm_sps:CrossAttention3", ["dh", "dk", "dv", "attn_dropout"]),
    ("models.lsthm_sps:MARN_cell", ["dh_l", "dh_a", "d_l", "d_a", "dropout"]),
    ("models.lsthm_sps:MARN1_sps", ["n_classes"]),
    ("models.encoder:MultiHeadAttention", ["n_head", "d_model", "d_model2", "d_k", "d_v", "dropout"]),
    ("models.encoder:ScaledDotProductAttention", ["temperature", "attn_dropout"]),
    ("models.encoder:PositionwiseFeedForward", ["d_in", "d_hid", "dropout"]),
    ("models.encoder:EncoderLayer", ["d_model", "d_inner", "n_head", "d_k", "d_v", "dropout"]),
    ("attention.SelfAttention:ScaledDotProductAttention", ["d_model", "d_k", "d_v", "h", "dropout"]),
    ("loss:MaskedLoss", ["losser", "weight"]),
    ("model_trainer:ModelTrainer", ["device", "lr", "test_step", "lr_decay", "model", "loss", "n_classes", "dataset"]),
])
def test_constructor_signatures(cls, args):
    import importlib
    mod, name = cls.split(":")
    c = getattr(importlib.import_module(mod), name)
    pos = [p.name for p in inspect.signature(c.__init__).parameters.values()
           if p.kind in (p.POSITIONAL_OR_KEYWORD,) and p.name != "self"]
    assert pos == args, (cls, pos)


def test_forward_signatures():
    from models.lsthm_sps import MARN1_sps, MARN_cell, LSTHM1
    from model_trainer import ModelTrainer
    assert list(inspect.signature(MARN1_sps.forward).parameters)[1:] == ["x", "qmask", "umask"]
    assert list(inspect.signature(MARN_cell.forward).parameters)[1:] == ["x", "x_l", "x_a", "qmask"]
    assert list(inspect.signature(LSTHM1.forward).parameters)[1:] == ["x", "ctm", "htm", "ztm", "speaker_affine"]
    for m in ("train_network", "eval_network", "save_parameters", "load_parameters"):
        assert hasattr(ModelTrainer, m)


def test_no_cpu_fallback():
    from models.lsthm_sps import MARN1_sps
    from models.encoder import EncoderLayer
    net = MARN1_sps(6)
    x, qmask, umask, _ = O.seeded_batch(2, 4)
    with pytest.raises(RuntimeError, match="HIP path only"):
        net(x, qmask, umask)
    with pytest.raises(RuntimeError, match="HIP path only"):
        EncoderLayer(100, 40, 8, 40, 40)(torch.zeros(1, 3, 100))


def test_flat_store_layout_and_dead_parameters():
    from models.lsthm_sps import MARN1_sps
    from mser.flat import ALIGN
    net = MARN1_sps(6)
    st = net.flat_store
    assert all(off % ALIGN == 0 for off in st.offsets.values())
    live = sum(int(np.prod(st.shapes[n])) for n in st.names if n not in st.dead)
    assert live == 2203370                                       # SURVEY: 2,203,370 scalars receive a gradient
    assert len(st.dead) == 20                                    # 20 tensors never do (SURVEY 7 "Dead parameters")
    # q/k/v projection weights of an encoder are adjacent in the flat buffer (enables one fused N=960 GEMM later)
    o = st.offsets
    assert o["encoder_l.slf_attn.w_ks.weight"] - o["encoder_l.slf_attn.w_qs.weight"] == 320 * 100
    st.attach(torch.device("cpu"))                               # the storage logic itself is device-agnostic
    sd = net.state_dict()
    assert sd["fc.0.weight"].data_ptr() == st.data.data_ptr() + 4 * o["fc.0.weight"]
    sd["w"].fill_(3.0)
    assert float(st.data[o["w"]]) == 3.0                         # parameters are views of the flat buffer


def test_step_lr_closed_form(golden_dir):
    from mser.optim import StepLR

    class _Opt:
        param_groups = [dict(lr=1e-3, initial_lr=1e-3)]
    sch = StepLR(_Opt(), step_size=1, gamma=0.98)
    g = np.load(os.path.join(golden_dir, "trainer.npz"))
    sch.step(0)
    assert _Opt.param_groups[0]["lr"] == pytest.approx(float(g["lr1"]), rel=1e-12)
    sch.step(1)
    assert _Opt.param_groups[0]["lr"] == pytest.approx(float(g["lr2"]), rel=1e-12)
    sch2 = StepLR(_Opt(), step_size=3, gamma=0.5)
    for ep in range(1, 10):
        sch2.step(ep - 1)
        assert _Opt.param_groups[0]["lr"] == pytest.approx(O.step_lr(1e-3, 0.5, 3, ep))


def test_out_of_scope_models_are_refused_loudly():
    from model_trainer import ModelTrainer
    with pytest.raises(NotImplementedError):
        ModelTrainer("cpu", 1e-3, 1, 0.98, "MARN1_cf", "NLL", 6, "IEMOCAP", quiet=True)


def test_metrics_from_confusion_match_sklearn():
    """accuracy / weighted F1 from the mask-weighted confusion matrix (mser/metrics.py; the device accumulates the matrix)
    vs the library calls of the reference (model_trainer.py:155-156), incl. a class that never occurs and zero-weight rows."""
    from sklearn.metrics import accuracy_score, f1_score
    from mser.metrics import accuracy_and_weighted_f1, confusion_matrix
    rs = np.random.RandomState(5)
    for C, n in ((6, 500), (6, 7), (3, 64)):
        labels = rs.randint(0, C - 1, n)               # class C-1 never true
        preds = rs.randint(0, C, n)
        masks = (rs.rand(n) > 0.3).astype(np.float32)
        masks[0] = 1.0
        acc, wf1 = accuracy_and_weighted_f1(confusion_matrix(labels, preds, masks, C))
        assert abs(acc - accuracy_score(labels, preds, sample_weight=masks)) < 1e-12
        assert abs(wf1 - f1_score(labels, preds, sample_weight=masks, average="weighted")) < 1e-12
    assert accuracy_and_weighted_f1(np.zeros((4, 4))) == (0.0, 0.0)


def test_mirror_state_dict_order_matches_the_oracle_tables():
    """The module mirrors register their parameters in the reference's order (checkpoints interchange, and the same seed draws the
    same initial weights); oracle.param_shapes restates that order for both model variants.  CPU only: constructing the mirrors
    needs no GPU (the flat store attaches on the first forward)."""
    from oracle import ref_cpu as O
    from models.lsthm_onlysp import MARN1_onlysp
    from models.lsthm_sps import MARN1_sps
    from models.lsthm_nsps import MARN1_nsps
    from models.lsthm_no_en import MARN1_no_en
    for cls, variant, n in ((MARN1_sps, "sps", 120), (MARN1_onlysp, "onlysp", 128), (lambda c: MARN1_nsps(c, "IEMOCAP"), "nsps", 109),
                            (lambda c: MARN1_no_en(c, "IEMOCAP"), "nsps", 109)):
        sd = cls(6).state_dict()
        shapes = O.param_shapes(variant=variant)
        assert list(sd.keys()) == list(shapes.keys()) and len(sd) == n
        assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)


def test_gru_variant_mirrors_follow_the_reference_registration_order(golden_dir):
    """The fixtures written by the reference itself list its parameters in ``named_parameters()`` order (one "gnorm/<name>" entry
    each, tests/golden/make_golden.py::_grad_samples): the mirrors of MARN1_onlysp / MARN1_nsps / MARN1_no_en must register the same
    names in the same order, and mark exactly the reference's gradient-less parameters as dead."""
    from models.lsthm_nsps import MARN1_nsps
    from models.lsthm_no_en import MARN1_no_en
    from models.lsthm_onlysp import MARN1_onlysp
    for fix, make in (("model_onlysp.npz", lambda: MARN1_onlysp(6)), ("model_nsps.npz", lambda: MARN1_nsps(6, "IEMOCAP")),
                      ("model_no_en.npz", lambda: MARN1_no_en(6, "IEMOCAP"))):
        g = np.load(os.path.join(golden_dir, fix))
        ref_names = [k[len("gnorm/"):] for k in g.files if k.startswith("gnorm/")]
        net = make()
        assert [n for n, _ in net.named_parameters()] == ref_names, fix
        ref_dead = {n for n in ref_names if float(g["gnorm/" + n]) < 0}
        assert set(net.flat_store.dead) == ref_dead, (fix, set(net.flat_store.dead) ^ ref_dead)


def test_bimodel_mirror_matches_the_oracle_table_and_the_reference_order(golden_dir):
    """models.DialogueRNN.BiModel as model_trainer.py:42-47 constructs it: parameter names / shapes / order equal the reference's own
    named_parameters() order (recorded by the reference-written fixture) and oracle.bimodel_param_shapes."""
    from models.DialogueRNN import BiModel, DialogueRNNCell, MatchingAttention
    m = BiModel(712, 500, 500, 300, 300, n_classes=6, listener_state=True, context_attention='general', dropout_rec=0.1, dropout=0.1)
    sd = m.state_dict()
    shapes = O.bimodel_param_shapes()
    assert list(sd.keys()) == list(shapes.keys()) and all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    g = np.load(os.path.join(golden_dir, "bimodel_ref.npz"))
    assert [n for n, _ in m.named_parameters()] == [k[len("gnorm/"):] for k in g.files if k.startswith("gnorm/")]
    assert all(float(g[k]) >= 0 for k in g.files if k.startswith("gnorm/"))          # no dead parameters in this model
    assert m.dropout_rec.p == pytest.approx(0.25) and m.dialog_rnn_f.dialogue_cell.dropout.p == 0.1    # :215, :216-217
    pos = [p.name for p in inspect.signature(BiModel.__init__).parameters.values() if p.name != "self"]
    assert pos == ["D_m", "D_g", "D_p", "D_e", "D_h", "n_classes", "listener_state", "context_attention", "D_a", "dropout_rec", "dropout"]
    assert list(inspect.signature(BiModel.forward).parameters)[1:] == ["U", "qmask", "umask", "att2"]
    with pytest.raises(NotImplementedError):
        DialogueRNNCell(10, 8, 8, 4, listener_state=False, context_attention='simple')


def test_nsps_constructor_takes_the_dataset_argument():
    """model_trainer.py:67-68,:71-72: MARN1_nsps(n_classes, dataset) / MARN1_no_en(n_classes, dataset)."""
    from models.lsthm_nsps import MARN1_nsps, MARN_cell, CrossAttention2
    from models.lsthm_no_en import MARN1_no_en
    for c in (MARN1_nsps, MARN1_no_en):
        pos = [p.name for p in inspect.signature(c.__init__).parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD and p.name != "self"]
        assert pos == ["n_classes", "dataset"]
    assert list(inspect.signature(MARN_cell.forward).parameters)[1:] == ["x", "x_l", "x_a", "qmask"]
    ca = CrossAttention2(100, 100, 100)
    assert tuple(ca.Wq.shape) == (100, 100) and ca.layer_norm.eps == 1e-6 and torch.all(ca.Wv == 1.0)


def _checkpoint_pattern(j, numel):          # tests/golden/make_golden.py::checkpoint_pattern
    i = np.arange(numel, dtype=np.int64)
    return (((i * 7 + j * 13) % 61) - 30).astype(np.float32) / 64.0


def test_reference_written_checkpoint_loads(golden_dir, tmp_path):
    """Checkpoint interchange, reference -> build (reference model_trainer.py:170-187): a file written by the REFERENCE's own
    ``ModelTrainer.save_parameters`` (tests/golden/make_golden.py::checkpoint_case, gzip-ed) is read by the build's ``load_parameters``
    (``weights_only=True``): same 120 keys in the same order under the ``model.`` prefix, every tensor arrives, nothing is skipped.  A
    ``module.``-prefixed copy (what a ``DataParallel`` run of the reference would have written, :179) loads as well, and a tensor of the
    wrong shape is skipped with the reference's message."""
    import gzip
    import shutil
    from model_trainer import ModelTrainer
    path = str(tmp_path / "model_0001.model")
    with gzip.open(os.path.join(golden_dir, "ref_checkpoint_model_0001.model.gz"), "rb") as f_in, open(path, "wb") as f_out:
        shutil.copyfileobj(f_in, f_out)
    meta = np.load(os.path.join(golden_dir, "ref_checkpoint_keys.npz"), allow_pickle=False)
    loaded = torch.load(path, map_location="cpu", weights_only=True)
    assert list(loaded.keys()) == [str(k) for k in meta["keys"]]
    tr = ModelTrainer("cpu", 1e-3, 1, 0.98, "MARN1_sps", "NLL", 6, "IEMOCAP", quiet=True)
    assert list(tr.state_dict().keys()) == list(loaded.keys())           # key names AND order interchange
    with torch.no_grad():
        for v in tr.state_dict().values():
            v.fill_(123.0)
    tr.load_parameters(path)
    for j, (k, v) in enumerate(tr.state_dict().items()):
        assert np.array_equal(v.detach().numpy().reshape(-1), _checkpoint_pattern(j, v.numel())), k
    # DataParallel-style prefix (:179) and a shape mismatch (:183-185: printed and skipped)
    pre = {"module." + k: v for k, v in loaded.items()}
    pre["module.model.w"] = torch.zeros(3)
    p2 = str(tmp_path / "model_dp.model")
    torch.save(pre, p2)
    tr2 = ModelTrainer("cpu", 1e-3, 1, 0.98, "MARN1_sps", "NLL", 6, "IEMOCAP", quiet=True)
    w_before = tr2.state_dict()["model.w"].clone()
    tr2.load_parameters(p2)
    sd2 = tr2.state_dict()
    assert torch.equal(sd2["model.w"], w_before)
    for j, (k, v) in enumerate(sd2.items()):
        if k != "model.w":
            assert np.array_equal(v.detach().numpy().reshape(-1), _checkpoint_pattern(j, v.numel())), k
