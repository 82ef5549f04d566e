"""CLIP image encoder (reference model/component/image_encoder.py:8-99 wrapping _common.py:170-221).

Parameters live under `visual.` with the OpenAI-CLIP key names so teacher archives load by key (reference
model/utils.py:140-144).  is_student=False: the frozen teacher (tower kind 0: inference only, fp16 residual stream).
is_student=True: the same architecture as a trainable student (tower kind 2: f32 residual stream, backward through the C ABI) with
the reference's `embedding_projection` / `hidden_projection` linears on the exported hidden states (image_encoder.py:23-25,54-59)
and its layer-mapped initialisation from the teacher (:70-99).
"""
import torch
from torch import nn

from .output import ControlOutput, VisionTransformerOutput
from ._tower import EncoderCfg, HipTower, run_tower
from ._proj import HipLinear


class _LN(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Lin(nn.Module):
    def __init__(self, fan_in, fan_out, std):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(fan_out, fan_in) * std)
        self.bias = nn.Parameter(torch.zeros(fan_out))


class _MHA(nn.Module):
    def __init__(self, width, attn_std, proj_std):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * width, width) * attn_std)
        self.in_proj_bias = nn.Parameter(torch.randn(3 * width) * attn_std)
        self.out_proj = _Lin(width, width, proj_std)


class _MLP(nn.Module):
    def __init__(self, width, fc_std, proj_std):
        super().__init__()
        self.c_fc = _Lin(width, 4 * width, fc_std)
        self.c_proj = _Lin(4 * width, width, proj_std)


class _ResBlock(nn.Module):
    def __init__(self, width, layers):
        super().__init__()
        # init stds of reference image_encoder.py:40-48 / text_encoder.py:98-106
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        self.attn = _MHA(width, width ** -0.5, proj_std)
        self.ln_1 = _LN(width)
        self.mlp = _MLP(width, (2 * width) ** -0.5, proj_std)
        self.ln_2 = _LN(width)


class TeacherTransformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.Sequential(*[_ResBlock(width, layers) for _ in range(layers)])


def teacher_block_names(prefix, layers):
    out = []
    for i in range(layers):
        p = f'{prefix}transformer.resblocks.{i}.'
        out += [p + 'ln_1.weight', p + 'ln_1.bias', p + 'attn.in_proj_weight', p + 'attn.in_proj_bias',
                p + 'attn.out_proj.weight', p + 'attn.out_proj.bias', p + 'ln_2.weight', p + 'ln_2.bias',
                p + 'mlp.c_fc.weight', p + 'mlp.c_fc.bias', p + 'mlp.c_proj.weight', p + 'mlp.c_proj.bias']
    return out


class _Visual(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.output_dim = input_resolution, output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)    # parameters only
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(torch.randn(width) * 0.02)
        self.positional_embedding = nn.Parameter(torch.randn((input_resolution // patch_size) ** 2 + 1, width) * 0.01)
        self.ln_pre = _LN(width)
        self.transformer = TeacherTransformer(width, layers, heads)
        self.ln_post = _LN(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


def student_anchor(module, device):
    """a requires-grad scalar that makes the tower's autograd Function part of the graph whatever its other inputs are"""
    a = getattr(module, '_anchor', None)
    if a is None or a.device != device:
        a = torch.zeros(1, device=device, requires_grad=True)
        object.__setattr__(module, '_anchor', a)
    return a


def init_layers_from_teacher(module, own_state, load, pattern, layer_map, teacher_state_dict, init_type):
    """reference image_encoder.py:70-99 / text_encoder.py:124-155: copy every teacher tensor whose key the student also has; keys of
    transformer layer i take teacher layer i ('begin'), tea - stu + i ('end') or i * layer_map.step ('mid')."""
    import re
    if init_type is None:
        return
    stu, tea = layer_map.stu_total_layer_num, layer_map.tea_total_layer_num
    if init_type == 'begin':
        to = lambda i: str(i)
    elif init_type == 'end':
        to = lambda i: str(tea - stu + i)
    elif init_type == 'mid':
        to = lambda i: str(i * layer_map.step)
    else:
        raise ValueError('the init_type should be begin, end, and mid, but got {}'.format(init_type))
    pat, digit = re.compile(pattern), re.compile('\\d')
    for key in list(own_state.keys()):
        found = re.findall(pat, key)
        if key not in teacher_state_dict:
            continue
        if not found:
            own_state[key] = teacher_state_dict[key]
        else:
            own_state[key] = teacher_state_dict[re.sub(digit, to(int(found[0])), string=key, count=1)]
    load(own_state)


class ImageEncoder(nn.Module):
    def __init__(self, is_student, vit_paras, tea_transformer_width=None):
        super().__init__()
        vit_paras = dict(vit_paras)
        self.layers = vit_paras['layers']
        if vit_paras.get('need_layers') is None:
            vit_paras['need_layers'] = tuple(range(self.layers))
        self.vit_paras = vit_paras
        self.is_student = bool(is_student)
        self.visual = _Visual(vit_paras['input_resolution'], vit_paras['patch_size'], vit_paras['width'],
                              vit_paras['layers'], vit_paras['heads'], vit_paras['output_dim'])
        self.embedding_projection = None
        self.hidden_projection = None
        self.no_trans = vit_paras['width'] == tea_transformer_width            # reference :20-22
        if is_student:
            if not tea_transformer_width:
                raise ValueError('ImageEncoder(is_student=True) needs tea_transformer_width (nn.Linear(width, tea_transformer_width))')
            self.embedding_projection = HipLinear(vit_paras['width'], tea_transformer_width)
            self.hidden_projection = HipLinear(vit_paras['width'], tea_transformer_width)
        w, res, patch = vit_paras['width'], vit_paras['input_resolution'], vit_paras['patch_size']
        cfg = EncoderCfg(kind=2 if is_student else 0, modality=0, tokens=(res // patch) ** 2 + 1, width=w, heads=vit_paras['heads'],
                         layers=self.layers, repeats=1, mlp_dim=4 * w, out_dim=vit_paras['output_dim'], patch=patch,
                         resolution=res, in_chans=3, vocab=0, embed_rank=0, head_mix=0, causal=0)
        names = ['visual.conv1.weight', 'visual.class_embedding', 'visual.positional_embedding', 'visual.ln_pre.weight',
                 'visual.ln_pre.bias'] + teacher_block_names('visual.', self.layers) + \
                ['visual.ln_post.weight', 'visual.ln_post.bias', 'visual.proj']
        object.__setattr__(self, '_tower', HipTower(self, cfg, names))
        self.register_load_state_dict_post_hook(lambda m, keys: setattr(m._tower, 'wcache_dirty', True))

    @property
    def need_layers(self):
        return self.vit_paras['need_layers']

    @property
    def output_layer(self):
        return self.visual.proj

    def extra_parameters(self):
        """trainable parameters that are not part of the tower's flat buffers (the optimizer updates them one by one)"""
        return [p for m in (self.embedding_projection, self.hidden_projection) if m is not None for p in m.parameters()]

    def encode_image(self, image, control_output: ControlOutput = None):
        co = control_output or ControlOutput()
        if co.need_attn_score or co.need_attn_prob or co.need_value_map:
            raise NotImplementedError('teacher attention maps are not exported by the HIP tower (SURVEY.md §2.1)')
        if self.is_student:
            if self.need_layers is not None and list(self.need_layers) != list(range(self.layers)):
                raise NotImplementedError('a trainable CLIP tower exports every layer\'s hidden state (need_layers = all)')
            out, reps, emb = run_tower(self._tower, image, student_anchor(self, image.device), co.need_rep, co.need_emb)
            if not self.no_trans:                                                        # reference :54-59
                if co.need_rep:
                    reps = [self.hidden_projection(r) for r in reps]
                if co.need_emb:
                    emb = self.embedding_projection(emb)
            llo = self._tower.last_layer_output() if getattr(co, 'need_last_layer_output', False) else None
            return VisionTransformerOutput(last_representation=out, last_layer_output=llo, representations=reps, embedding=emb)
        with torch.no_grad():   # hidden states only for `need_layers` (reference _common.py:154-158)
            out, _, reps, emb = self._tower.forward(image, training=False, need_rep=co.need_rep, need_emb=co.need_emb,
                                                    rep_layers=list(self.need_layers) if self.need_layers is not None else None)
        llo = self._tower.last_layer_output() if getattr(co, 'need_last_layer_output', False) else None
        return VisionTransformerOutput(last_representation=out, last_layer_output=llo, representations=reps, embedding=emb)

    def last_layer_output(self):
        """[B, N, E] = ln_post(x) @ proj for every token of the most recent forward (reference _common.py:210-215)"""
        return self._tower.last_layer_output()

    def forward(self, image, control_output: ControlOutput = None):
        return self.encode_image(image, control_output)

    def init_layers_with_teacher(self, layer_map, teacher_state_dict=None, init_type=None):
        """reference :70-99 (keys of `self.visual`, i.e. without the `visual.` prefix, looked up in the teacher's state dict as given)"""
        init_layers_from_teacher(self, self.visual.state_dict(), self.visual.load_state_dict, 'visual.transformer.resblocks.(\\d)',
                                 layer_map, teacher_state_dict, init_type)
        self._tower.wcache_dirty = True

    def hyper_para(self):
        return self.vit_paras
