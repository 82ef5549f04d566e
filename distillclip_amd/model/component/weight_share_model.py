"""Weight-shared MiniViT-style students (reference model/component/weight_share_model.py:226-521).

Same constructor signatures, attribute names and state_dict keys as the reference classes; the modules hold parameters
only — forward / backward are the `dclip_encoder_*` calls of libdistillclip_hip.so.  Supported configuration = what the
shipped YAMLs use: repeated_times > 1, rpe_config = None, hybrid_backbone = None, all dropouts 0.
"""
from typing import List, Optional

import torch
from torch import nn

from .output import ControlOutput, VisionTransformerOutput, TextTransformerOutput
from ._tower import EncoderCfg, HipTower, run_tower


def _trunc_normal_(t, std=.02):
    return nn.init.trunc_normal_(t, std=std, a=-2., b=2.)      # timm.trunc_normal_ semantics (weight_share_model.py:297-311)


class _Affine(nn.Module):
    """nn.LayerNorm's parameters (weight = 1, bias = 0 at init: weight_share_model.py:313-315)."""

    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Linear(nn.Module):
    def __init__(self, fan_in, fan_out, bias=True):
        super().__init__()
        self.weight = nn.Parameter(_trunc_normal_(torch.empty(fan_out, fan_in)))
        if bias:
            self.bias = nn.Parameter(torch.zeros(fan_out))
        else:
            self.register_parameter('bias', None)


class _HeadMix(nn.Module):
    """Conv2d(H, H, 1, bias=False) weight of conv_l / conv_w (weight_share_model.py:79-84)."""

    def __init__(self, heads):
        super().__init__()
        self.weight = nn.Parameter(_trunc_normal_(torch.empty(heads, heads, 1, 1)))


class _Repeated(nn.Module):
    """RepeatedModuleList (weight_share_model.py:20-34): one instance per repeat under `.instances.{r}`."""

    def __init__(self, instances):
        super().__init__()
        self.instances = nn.ModuleList(instances)
        self.repeated_times = len(instances)


class _Attn(nn.Module):
    def __init__(self, dim, heads, qkv_bias, repeats, use_transform):
        super().__init__()
        self.num_heads = heads
        self.qkv = _Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = _Linear(dim, dim)
        if use_transform:
            self.conv_l = _Repeated([_HeadMix(heads) for _ in range(repeats)])
            self.conv_w = _Repeated([_HeadMix(heads) for _ in range(repeats)])
        else:
            self.conv_l = self.conv_w = None


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = _Linear(dim, hidden)
        self.fc2 = _Linear(hidden, dim)


class _MiniBlock(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, qkv_bias, repeats, use_transform):
        super().__init__()
        self.norm1 = _Repeated([_Affine(dim) for _ in range(repeats)])
        self.norm2 = _Repeated([_Affine(dim) for _ in range(repeats)])
        self.attn = _Attn(dim, heads, qkv_bias, repeats, use_transform)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _RepeatedMiniBlock(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.repeated_times = kw['repeats']
        self.block = _MiniBlock(**kw)


class _PatchEmbed(nn.Module):
    """timm PatchEmbed's parameters: Conv2d(in, embed, k=p, s=p, bias=True) under `.proj`."""

    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)   # parameters only


def _check_supported(repeated_times, rpe_config, drop_rate, attn_drop_rate, drop_path_rate, hybrid_backbone=None,
                     qk_scale=None):
    if repeated_times < 2:
        # (the reference cannot run it either: with repeated_times = 1 its blocks are plain MiniBlocks, weight_share_model.py:283, whose
        #  forward returns a TransformerLayerOutput, and forward_features reads `.last_layer_output` / `.representations` of it — :353-355 —
        #  which that dataclass does not have, output.py:56-60)
        raise NotImplementedError('repeated_times = 1 is not runnable in the reference (forward_features reads fields its MiniBlock output '
                                  'lacks: weight_share_model.py:353, output.py:56); distillclip_amd implements the RepeatedMiniBlock layout '
                                  '(repeated_times >= 2) every shipped config uses')
    if rpe_config is not None or hybrid_backbone is not None:
        raise NotImplementedError('iRPE / hybrid backbones are out of scope (rpe_config / hybrid_backbone are null in every '
                                  'shipped config)')
    if drop_rate or attn_drop_rate or drop_path_rate:
        raise NotImplementedError('dropout / stochastic depth are 0 in every shipped config and are not implemented')
    if qk_scale is not None:
        raise NotImplementedError('qk_scale override is not implemented (null in every shipped config)')


def _block_param_names(n_blocks, repeats, qkv_bias, use_transform):
    names = []
    for i in range(n_blocks):
        p = f'blocks.{i}.block.'
        names += [p + 'attn.qkv.weight', p + 'attn.qkv.bias' if qkv_bias else None, p + 'attn.proj.weight',
                  p + 'attn.proj.bias', p + 'mlp.fc1.weight', p + 'mlp.fc1.bias', p + 'mlp.fc2.weight', p + 'mlp.fc2.bias']
        for r in range(repeats):
            names += [p + f'norm1.instances.{r}.weight', p + f'norm1.instances.{r}.bias',
                      p + f'norm2.instances.{r}.weight', p + f'norm2.instances.{r}.bias',
                      p + f'attn.conv_l.instances.{r}.weight' if use_transform else None,
                      p + f'attn.conv_w.instances.{r}.weight' if use_transform else None]
    return names


class _StudentBase(nn.Module):
    _tower: Optional[HipTower] = None

    def _anchor_for(self, device):
        a = getattr(self, '_anchor', None)
        if a is None or a.device != device:
            a = torch.zeros(1, device=device, requires_grad=True)
            object.__setattr__(self, '_anchor', a)
        return a

    @property
    def output_layer(self):
        return self.head

    def hyper_para(self):
        return self.hyper

    def last_layer_output(self):
        """[B, N, E] all-token output of norm + head for the most recent forward (weight_share_model.py:363-366, :503-506)"""
        return self._tower.last_layer_output()

    def _check_control(self, co: Optional[ControlOutput]):
        if co is not None and (co.need_attn_score or co.need_attn_prob or co.need_value_map):
            raise NotImplementedError('the HIP towers keep attention scores / probabilities / value maps on chip; the loss terms '
                                      'that need them (attention_*, last_value_map_kl) are outside the hot path (SURVEY.md §2.1)')


class RepeatVisionTransformer(_StudentBase):
    """reference weight_share_model.py:226-381."""

    def __init__(self, need_layers: Optional[List] = None, img_size=224, patch_size=16, in_chans=3, out_dim=1000,
                 embed_dim=768, depth=12, num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., hybrid_backbone=None, rpe_config=None, repeated_times=1,
                 use_transform=False):
        super().__init__()
        _check_supported(repeated_times, rpe_config, drop_rate, attn_drop_rate, drop_path_rate, hybrid_backbone, qk_scale)
        assert depth % repeated_times == 0
        self.need_layers = list(range(depth)) if need_layers is None else need_layers
        self.num_classes = out_dim
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        n_tok = self.patch_embed.num_patches + 1
        self.cls_token = nn.Parameter(_trunc_normal_(torch.zeros(1, 1, embed_dim)))
        self.pos_embed = nn.Parameter(_trunc_normal_(torch.zeros(1, n_tok, embed_dim)))
        block_kwargs = dict(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                            drop=drop_rate, attn_drop=attn_drop_rate, norm_layer=nn.LayerNorm, rpe_config=rpe_config,
                            use_transform=use_transform)
        self.hyper = {'block_kwargs': block_kwargs, 'depth': depth, 'repeated_times': repeated_times}
        n_blocks = depth // repeated_times
        self.blocks = nn.ModuleList([
            _RepeatedMiniBlock(dim=embed_dim, heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                               repeats=repeated_times, use_transform=use_transform) for _ in range(n_blocks)])
        self.norm = _Affine(embed_dim)
        self.head = _Linear(embed_dim, out_dim)
        cfg = EncoderCfg(kind=1, modality=0, tokens=n_tok, width=embed_dim, heads=num_heads, layers=n_blocks,
                         repeats=repeated_times, mlp_dim=int(embed_dim * mlp_ratio), out_dim=out_dim, patch=patch_size,
                         resolution=img_size, in_chans=in_chans, vocab=0, embed_rank=0, head_mix=int(use_transform), causal=0)
        names = ['patch_embed.proj.weight', 'patch_embed.proj.bias', 'cls_token', 'pos_embed']
        names += _block_param_names(n_blocks, repeated_times, qkv_bias, use_transform)
        names += ['norm.weight', 'norm.bias', 'head.weight', 'head.bias']
        object.__setattr__(self, '_tower', HipTower(self, cfg, names))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def get_classifier(self):
        return self.head

    def forward_features(self, x, control_output: ControlOutput = None):
        self._check_control(control_output)
        co = control_output or ControlOutput()
        rep, hidden, emb = run_tower(self._tower, x, self._anchor_for(x.device), co.need_rep, co.need_emb)
        # like the reference, EVERY block execution contributes a hidden state (weight_share_model.py:211, :356-357)
        llo = self._tower.last_layer_output() if getattr(co, 'need_last_layer_output', False) else None
        return VisionTransformerOutput(last_representation=rep, last_layer_output=llo, representations=hidden, embedding=emb)

    def forward(self, x, control_output: ControlOutput = None):
        return self.forward_features(x, control_output)


class RepeatTextTransformer(_StudentBase):
    """reference weight_share_model.py:384-521.  Bidirectional (no attention mask), EOT pooling = argmax of the ids."""

    def __init__(self, need_layers: Optional[List] = None, vocab_size=49408, context_length=77, out_dim=512,
                 embed_dim=768, depth=12, num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., rpe_config=None, repeated_times=1, use_transform=False,
                 compression_embedding=False, embedding_compression_dim=256):
        super().__init__()
        _check_supported(repeated_times, rpe_config, drop_rate, attn_drop_rate, drop_path_rate, None, qk_scale)
        assert depth % repeated_times == 0
        self.need_layers = list(range(depth)) if need_layers is None else need_layers
        self.num_classes = out_dim
        self.num_features = self.embed_dim = embed_dim
        self.context_length = context_length
        if compression_embedding:
            self.patch_embed = nn.Sequential(nn.Embedding(vocab_size, embedding_compression_dim),
                                             nn.Linear(embedding_compression_dim, embed_dim))
            _trunc_normal_(self.patch_embed[1].weight)
            nn.init.constant_(self.patch_embed[1].bias, 0)
        else:
            self.patch_embed = nn.Embedding(vocab_size, embed_dim)
        self.pos_embed = nn.Parameter(_trunc_normal_(torch.empty(context_length, embed_dim)))
        block_kwargs = dict(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                            drop=drop_rate, attn_drop=attn_drop_rate, norm_layer=nn.LayerNorm, rpe_config=rpe_config,
                            use_transform=use_transform)
        self.hyper = {'block_kwargs': block_kwargs, 'depth': depth, 'repeated_times': repeated_times}
        n_blocks = depth // repeated_times
        self.blocks = nn.ModuleList([
            _RepeatedMiniBlock(dim=embed_dim, heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                               repeats=repeated_times, use_transform=use_transform) for _ in range(n_blocks)])
        self.norm = _Affine(embed_dim)
        self.head = _Linear(embed_dim, out_dim)
        cfg = EncoderCfg(kind=1, modality=1, tokens=context_length, width=embed_dim, heads=num_heads, layers=n_blocks,
                         repeats=repeated_times, mlp_dim=int(embed_dim * mlp_ratio), out_dim=out_dim, patch=0, resolution=0,
                         in_chans=0, vocab=vocab_size, embed_rank=embedding_compression_dim if compression_embedding else 0,
                         head_mix=int(use_transform), causal=0)
        if compression_embedding:
            names = ['patch_embed.0.weight', 'patch_embed.1.weight', 'patch_embed.1.bias', 'pos_embed']
        else:
            names = ['patch_embed.weight', 'pos_embed']
        names += _block_param_names(n_blocks, repeated_times, qkv_bias, use_transform)
        names += ['norm.weight', 'norm.bias', 'head.weight', 'head.bias']
        object.__setattr__(self, '_tower', HipTower(self, cfg, names))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def get_classifier(self):
        return self.head

    def forward_features(self, text, control_output: ControlOutput = None):
        self._check_control(control_output)
        co = control_output or ControlOutput()
        rep, hidden, emb = run_tower(self._tower, text, self._anchor_for(text.device), co.need_rep, co.need_emb)
        llo = self._tower.last_layer_output() if getattr(co, 'need_last_layer_output', False) else None
        return TextTransformerOutput(last_representation=rep, last_layer_output=llo, representations=hidden, embedding=emb)

    def forward(self, x, control_output: ControlOutput = None):
        return self.forward_features(x, control_output)
