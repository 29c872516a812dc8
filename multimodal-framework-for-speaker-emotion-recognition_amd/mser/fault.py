"""Sticky per-device fault word (include/mser.h MSER_FAULT_*).

Kernels cannot raise: a persistent chain that gives up at a bounded wait, a linked consumer whose producer never came, or a
label outside [0, C) (where the reference's NLLLoss / CrossEntropyLoss raise, loss.py:19-24) OR a bit into ONE uint32 in device
memory per device.  The fused Adam launch reads it ON THE DEVICE and skips its update while it is set, so a faulted step never
reaches the weights; the host reads it wherever it synchronises anyway (``ModelTrainer.train_network`` / ``eval_network`` once
per call, ``bench.py`` after the timed region) and raises.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import _lib as L

_WORDS: Dict[torch.device, torch.Tensor] = {}

_NAMES = {
    L.MSER_FAULT_CHAIN_TIMEOUT: "a persistent recurrent launch gave up at a bounded inter-workgroup wait (workgroups not "
                                "co-resident, or a serialising profiler); its outputs and gradients are invalid",
    L.MSER_FAULT_LINK_TIMEOUT: "a counter-linked consumer launch timed out waiting for its producer; its gradients are invalid",
    L.MSER_FAULT_BAD_LABEL: "a target label is outside [0, n_classes) (torch's NLLLoss / CrossEntropyLoss raise here)",
}


def word(device) -> torch.Tensor:
    """The int32 [1] fault word of ``device`` (created zeroed on first use; outside any stream capture)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("mser.fault: the product path runs on the GPU only")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    w = _WORDS.get(device)
    if w is None:
        w = torch.zeros(1, device=device, dtype=torch.int32)
        _WORDS[device] = w
    return w


def peek(device) -> int:
    """Synchronising read (no reset)."""
    return int(word(device).item())


def describe(bits: int) -> str:
    return "; ".join(txt for bit, txt in _NAMES.items() if bits & bit) or f"unknown bits {bits:#x}"


def clear(device) -> None:
    word(device).zero_()


def check(device, where: str = "", agreed_bits=None) -> None:
    """Synchronising read; raises RuntimeError (and clears the word) if any kernel reported a fault since the last check.
    agreed_bits (data-parallel callers): the OR over ALL ranks' words (mser.dist.agree_on_fault) -- every rank raises when any rank
    faulted, naming its own bits and the others'."""
    w = word(device)
    v = int(w.item())
    bits = v if agreed_bits is None else (v | int(agreed_bits))
    if bits:
        w.zero_()
        other = bits & ~v
        txt = describe(v) if v else ""
        if other:
            txt = (txt + "; " if txt else "") + f"another rank reported: {describe(other)}"
        raise RuntimeError(f"libmser device fault{' in ' + where if where else ''} (code {bits:#x}): {txt}. "
                           "The optimiser skipped every update issued while the fault was set.")
