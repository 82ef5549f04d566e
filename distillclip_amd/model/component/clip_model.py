"""Two-tower wrapper (reference model/component/clip_model.py:7-63).  forward(text, image, control_output): note the
argument order.  The cosine logits are not materialised here — the fused loss kernel recomputes them tile by tile."""
from typing import Optional

from torch import nn

from .output import ControlOutput, CLIPOutput


class CLIPModel(nn.Module):
    def __init__(self, is_student: bool, image_encoder: nn.Module, text_encoder: nn.Module, norm=False, only_last_rep=False):
        super().__init__()
        self.image_encoder = image_encoder
        self.text_encoder = text_encoder
        self.is_student = is_student
        self.norm = norm
        self.only_last_rep = only_last_rep

    def encode_image(self, image, control_output: ControlOutput = None):
        out = self.image_encoder(image, control_output or ControlOutput())
        return out.last_representation if self.only_last_rep else out

    def encode_text(self, text, control_output: ControlOutput = None):
        out = self.text_encoder(text, control_output or ControlOutput())
        return out.last_representation if self.only_last_rep else out

    def forward(self, text, image, control_output: Optional[ControlOutput] = None):
        control_output = control_output or ControlOutput()
        image_output = self.encode_image(image, control_output)
        text_output = self.encode_text(text, control_output)
        if self.only_last_rep:
            i = image_output / image_output.norm(dim=1, keepdim=True)
            t = text_output / text_output.norm(dim=1, keepdim=True)
            return i, t, i @ t.t()
        return CLIPOutput(visual_output=image_output, text_output=text_output)

    def hyper_para(self):
        res = {'image_' + k: v for k, v in self.image_encoder.hyper_para().items()}
        res.update({'text_' + k: v for k, v in self.text_encoder.hyper_para().items()})
        return res
