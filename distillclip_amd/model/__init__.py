"""Host-side mirror of the reference's `model` package (same class names, constructor arguments, state_dict keys).

    reference: /model/{distil_model,dual_distill_model,_loss,utils}.py and /model/component/*.py

All arithmetic runs in libdistillclip_hip.so; these modules only own parameters and sequence C-ABI calls.
"""
from .distil_model import DistillModel                 # noqa: F401
from .dual_distill_model import DualDistillModel       # noqa: F401
from ._loss import LossCalculator                      # noqa: F401

__all__ = ['DistillModel', 'DualDistillModel', 'LossCalculator']
