"""Host side of the evaluation metrics (reference model_trainer.py:152-156): accuracy_score and the weighted f1_score with
sample_weight = umask are functions of the mask-weighted confusion matrix that ``mser_confusion_update`` accumulates on the
device, so nothing per-utterance has to leave the GPU.  Definitions follow scikit-learn (the library the reference calls):

    accuracy    = sum_c conf[c, c] / sum(conf)
    F1_c        = 2 tp_c / (2 tp_c + fp_c + fn_c)           (0 when the denominator is 0: sklearn's zero_division default)
    weighted F1 = sum_c support_c * F1_c / sum_c support_c  over the labels present in y_true or y_pred,
                  support_c = sum of the weights of the true class c
"""
import numpy as np


def confusion_matrix(labels, preds, weights, n_classes: int) -> np.ndarray:
    """numpy restatement of the device kernel (tests compare the two)."""
    conf = np.zeros((n_classes, n_classes), dtype=np.float64)
    np.add.at(conf, (np.asarray(labels, dtype=np.int64), np.asarray(preds, dtype=np.int64)), np.asarray(weights, dtype=np.float64))
    return conf


def accuracy_and_weighted_f1(conf) -> tuple:
    conf = np.asarray(conf, dtype=np.float64)
    total = conf.sum()
    if total <= 0:
        return 0.0, 0.0
    tp = np.diag(conf)
    support = conf.sum(axis=1)           # true class weights
    predicted = conf.sum(axis=0)
    denom = support + predicted          # 2 tp + fp + fn
    f1 = np.where(denom > 0, 2.0 * tp / np.where(denom > 0, denom, 1.0), 0.0)
    acc = float(tp.sum() / total)
    wf1 = float((support * f1).sum() / support.sum())
    return acc, wf1
