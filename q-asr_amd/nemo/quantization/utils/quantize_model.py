"""Mode switches of the reference's nemo/quantization/utils/quantize_model.py:5-75:
set_percentile / set_dynamic / freeze_model / evaluate / train / calibrate."""
import torch.nn as nn

from nemo.quantization.utils.quant_modules import *  # noqa: F401,F403
from nemo.quantization.utils.quant_modules import QuantAct, QuantConv1d, QuantLinear

list_all = [QuantAct, QuantLinear, QuantConv1d]


def _quant_ops(model):
    """Every QuantAct / QuantConv1d reachable from `model` (module tree walk; the reference walks
    Sequential / ModuleList / dir() attributes, which reaches the same set for these models)."""
    for m in model.modules():
        if type(m) in list_all:
            yield m


def _touch(model):
    bump = getattr(model, '_quant_state_changed', None)
    if callable(bump):
        bump()


def set_percentile(model, percentile: float):
    for m in _quant_ops(model):
        if type(m) == QuantAct:
            m.set_percentile(percentile)
    _touch(model)


def set_dynamic(model, dynamic: bool):
    for m in _quant_ops(model):
        if type(m) == QuantAct:
            m.dynamic = dynamic
    _touch(model)


def freeze_model(model, freeze_list):
    for m in _quant_ops(model):
        if type(m) in freeze_list:
            m.fix()
        else:
            m.unfix()
    _touch(model)


def evaluate(model):
    """Evaluation mode - fix all operations"""
    freeze_model(model, list_all)


def train(model):
    """Train mode - unfix all operations"""
    freeze_model(model, [])


def calibrate(model):
    """Calibration mode - only unfix QuantAct"""
    freeze_model(model, [QuantConv1d, QuantLinear])
