"""Teacher construction (reference model/utils.py:81-181).

The reference downloads the OpenAI CLIP TorchScript archive (utils.py:18-61) and copies tensors by key.  There is no
network here, so `teacher_load` accepts, in this order:
  1. `download_root/<name>.pt` if present (a TorchScript archive or a plain state_dict; keys are the OpenAI ones);
  2. deterministic synthetic weights of the named architecture when DCLIP_SYNTHETIC_TEACHER=1 (benchmarks / tests).
Shape sniffing from the state_dict follows utils.py:81-129.
"""
import os

import torch

from .. import synth
from .component.clip_model import CLIPModel
from .component.image_encoder import ImageEncoder
from .component.text_encoder import TextEncoder

_ARCH = {   # name -> (vision width, layers, patch, resolution, text width, text layers, embed dim)
    'ViT-B/32': (768, 12, 32, 224, 512, 12, 512),
    'ViT-B/16': (768, 12, 16, 224, 512, 12, 512),
    'ViT-L/14': (1024, 24, 14, 224, 768, 12, 768),
}


def available_models():
    """the archive names the reference knows (utils.py:18-28, :64-65).  Runnable on the HIP towers: ViT teachers of at most 128 tokens whose patch
    has a multiple of 64 values (ViT-B/32, the teacher of every shipped config); ResNet teachers are out of scope."""
    return ['RN50', 'RN101', 'RN50x4', 'RN50x16', 'RN50x64', 'ViT-B/32', 'ViT-B/16', 'ViT-L/14', 'ViT-L/14@336px']


def get_transformer_para(sd):
    # reference utils.py:81-90
    return {'embed_dim': sd['text_projection'].shape[1], 'context_length': sd['positional_embedding'].shape[0],
            'vocab_size': sd['token_embedding.weight'].shape[0], 'transformer_width': sd['ln_final.weight'].shape[0],
            'transformer_heads': sd['ln_final.weight'].shape[0] // 64,
            'transformer_layers': len(set(k.split('.')[2] for k in sd if k.startswith('transformer.resblocks')))}


def get_visual_para(sd):
    # reference utils.py:93-114 (ViT branch; ResNet teachers are out of scope)
    if 'visual.proj' not in sd:
        raise NotImplementedError('ResNet CLIP teachers are out of scope (SURVEY.md §2 row 12)')
    width = sd['visual.conv1.weight'].shape[0]
    patch = sd['visual.conv1.weight'].shape[-1]
    grid = round((sd['visual.positional_embedding'].shape[0] - 1) ** 0.5)
    return {'layers': len([k for k in sd if k.startswith('visual.') and k.endswith('.attn.in_proj_weight')]),
            'width': width, 'patch_size': patch, 'input_resolution': patch * grid, 'heads': width // 64,
            'output_dim': sd['visual.proj'].shape[1]}


def load(name, download_root=None, resolution=None):
    path = os.path.join(os.path.expanduser(download_root or '~/.cache/clip'), name.replace('/', '-') + '.pt')
    if os.path.isfile(path):
        try:
            return torch.jit.load(path, map_location='cpu').eval().state_dict()
        except RuntimeError:
            return torch.load(path, map_location='cpu')
    if os.environ.get('DCLIP_SYNTHETIC_TEACHER') == '1':
        if name not in _ARCH:
            raise RuntimeError(f'no synthetic architecture table for teacher {name}')
        vw, vl, p, res, tw, tl, e = _ARCH[name]
        res = resolution or res
        sd = synth.teacher_image_state(2022, vw, vl, p, res, e)
        sd.update(synth.teacher_text_state(2022, tw, tl, 77, 49408, e))
        return {k: torch.from_numpy(v) for k, v in sd.items()}
    raise FileNotFoundError(f'{path} not found and there is no network to download {name}; place the OpenAI CLIP archive '
                            f'there or set DCLIP_SYNTHETIC_TEACHER=1 for seeded synthetic weights')


def _copy_by_key(model, sd):
    mine = model.state_dict()
    for k in mine:
        if k in sd:
            mine[k] = sd[k].float()
    model.load_state_dict(mine)
    return model


def load_image(teacher_name, download_root, need_layers, state_dict=None):
    sd = state_dict if state_dict is not None else load(teacher_name, download_root)
    para = get_visual_para(sd)
    para.update(dict(need_layers=need_layers))
    return _copy_by_key(ImageEncoder(is_student=False, vit_paras=para), sd)


def load_text(teacher_name, download_root, need_layers, state_dict=None):
    sd = state_dict if state_dict is not None else load(teacher_name, download_root)
    para = get_transformer_para(sd)
    para.update(dict(need_layers=need_layers))
    return _copy_by_key(TextEncoder(is_student=False, **para), sd)


def teacher_load(teacher_name: str, download_root, model_type, need_layers=None, only_last_rep=False, state_dict=None):
    if model_type == 'text':
        return load_text(teacher_name, download_root, need_layers, state_dict)
    if model_type == 'image':
        return load_image(teacher_name, download_root, need_layers, state_dict)
    if model_type == 'all':
        sd = state_dict if state_dict is not None else load(teacher_name, download_root)
        return CLIPModel(False, load_image(teacher_name, download_root, need_layers, sd),
                         load_text(teacher_name, download_root, need_layers, sd), only_last_rep)
    raise ValueError(f"the model_type should in ['text', 'image', 'all'], but got {model_type}")
