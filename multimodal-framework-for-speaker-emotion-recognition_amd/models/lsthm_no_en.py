"""MI355X-native mirror of the reference's ``models.lsthm_no_en`` (reference file model/lsthm_no_en.py):
``MARN1_no_en`` is ``MARN1_nsps`` without the text encoder -- ``encoder_l`` is constructed (same ``state_dict``) and never called
(lsthm_no_en.py:306,:309), so ``linear_in``'s output feeds the cells and the attention modules directly.  Every other class of the
reference file is identical to its ``models.lsthm_nsps`` namesake."""
from models.lsthm_nsps import LSTHM1, CrossAttention, CrossAttention2, CrossAttention3, MARN_cell, MARN1_nsps  # noqa: F401


class MARN1_no_en(MARN1_nsps):
    """Reference model/lsthm_no_en.py:242-360."""

    _no_en = True

    def __init__(self, n_classes, dataset, *, d_r=1024):
        super(MARN1_no_en, self).__init__(n_classes, dataset, d_r=d_r)
