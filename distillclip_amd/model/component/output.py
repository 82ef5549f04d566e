"""Result carriers of the encoders (API of reference model/component/output.py:7-73: field names are the contract
the LossCalculator and the LightningModules bind to)."""
from dataclasses import dataclass, field
from typing import List, Optional

import torch


@dataclass
class ControlOutput:
    # reference output.py:7-13 — which optional activations a loss term needs back
    need_emb: bool = False
    need_attn_score: bool = False
    need_value_map: bool = False
    need_attn_prob: bool = False
    need_rep: bool = False
    # extension (not a reference field; default keeps the reference's behaviour on the hot path): also materialise
    # `last_layer_output` [B, N, E] — the reference always computes it (_common.py:212-215), here only the class / EOT row is
    # projected unless asked ("producible on request")
    need_last_layer_output: bool = False


@dataclass
class _EncoderOutput:
    last_representation: Optional[torch.Tensor] = None      # [B, E]  class token (image) / EOT token (text)
    last_layer_output: Optional[torch.Tensor] = None        # [B, N, E]; only with ControlOutput.need_last_layer_output, or
                                                            # later from `encoder.last_layer_output()` (detached)
    attention_scores: Optional[List[torch.Tensor]] = field(default_factory=list)
    attention_probs: Optional[List[torch.Tensor]] = field(default_factory=list)
    representations: Optional[List[torch.Tensor]] = field(default_factory=list)
    value_map: Optional[torch.Tensor] = None
    embedding: Optional[torch.Tensor] = None


@dataclass
class VisionTransformerOutput(_EncoderOutput):
    pass


@dataclass
class TextTransformerOutput(_EncoderOutput):
    pass


@dataclass
class CLIPOutput:
    # reference output.py:62-68.  The logits are lazy on the HIP path: the fused loss never materialises [B, B].
    visual_output: Optional[VisionTransformerOutput] = None
    text_output: Optional[TextTransformerOutput] = None
    _i2t: Optional[torch.Tensor] = None

    @property
    def i2t_logits(self):
        if self._i2t is None:      # inspection / validation only — not on the training path
            i = self.visual_output.last_representation.detach().float()
            t = self.text_output.last_representation.detach().float()
            self._i2t = (i / i.norm(dim=1, keepdim=True)) @ (t / t.norm(dim=1, keepdim=True)).t()
        return self._i2t

    @property
    def t2i_logits(self):
        return self.i2t_logits.T
