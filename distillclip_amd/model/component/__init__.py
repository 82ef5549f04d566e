from .output import (ControlOutput, VisionTransformerOutput, TextTransformerOutput, CLIPOutput)   # noqa: F401
from .weight_share_model import RepeatVisionTransformer, RepeatTextTransformer                     # noqa: F401
from .image_encoder import ImageEncoder                                                             # noqa: F401
from .text_encoder import TextEncoder                                                               # noqa: F401
from .clip_model import CLIPModel                                                                   # noqa: F401
