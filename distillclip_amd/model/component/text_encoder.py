"""Frozen CLIP text teacher (reference model/component/text_encoder.py:8-92): token + positional embedding, causal
residual-attention blocks, ln_final, text_projection, EOT pooling by argmax of the token ids."""
import torch
from torch import nn

from .output import ControlOutput, TextTransformerOutput
from ._tower import EncoderCfg, HipTower
from .image_encoder import TeacherTransformer, _LN, teacher_block_names


class TextEncoder(nn.Module):
    def __init__(self, transformer_width, transformer_layers, transformer_heads, context_length, need_layers, vocab_size,
                 embed_dim, tea_transformer_width=None, is_student=True, drop_out=0., compression_embedding=False,
                 embedding_compression_dim=256):
        super().__init__()
        if is_student or compression_embedding:
            raise NotImplementedError('TextEncoder as a student (is_student=True / compression_embedding) is not used by any '
                                      'shipped config; students are RepeatTextTransformer (SURVEY.md §2 row 6)')
        if drop_out:
            raise NotImplementedError('dropout is 0 for the frozen teacher')
        self.context_length, self.transformer_width, self.transformer_heads = context_length, transformer_width, transformer_heads
        self.vocab_size, self.embed_dim, self.layers = vocab_size, embed_dim, transformer_layers
        self.is_student = False
        self._need_layers = need_layers
        # optional host-side hint: number of leading positions that contain every caption's EOT (None = all context_length).
        # The tower is causal and only the EOT row is consumed, so later positions are dead work (see include/dclip.h).
        self.max_tokens = None
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        self.positional_embedding = nn.Parameter(torch.randn(context_length, transformer_width) * 0.01)
        self.ln_final = _LN(transformer_width)
        self.text_projection = nn.Parameter(torch.randn(transformer_width, embed_dim) * transformer_width ** -0.5)
        self.transformer = TeacherTransformer(transformer_width, transformer_layers, transformer_heads)
        cfg = EncoderCfg(kind=0, modality=1, tokens=context_length, width=transformer_width, heads=transformer_heads,
                         layers=transformer_layers, repeats=1, mlp_dim=4 * transformer_width, out_dim=embed_dim, patch=0,
                         resolution=0, in_chans=0, vocab=vocab_size, embed_rank=0, head_mix=0, causal=1)
        names = ['token_embedding.weight', 'positional_embedding'] + teacher_block_names('', transformer_layers) + \
                ['ln_final.weight', 'ln_final.bias', 'text_projection']
        object.__setattr__(self, '_tower', HipTower(self, cfg, names))
        self.register_load_state_dict_post_hook(lambda m, keys: setattr(m._tower, 'wcache_dirty', True))

    @property
    def need_layers(self):
        return self._need_layers

    def encode_text(self, text, control_output: ControlOutput = None):
        co = control_output or ControlOutput()
        if co.need_attn_score or co.need_attn_prob or co.need_value_map:
            raise NotImplementedError('teacher attention maps are not exported by the HIP tower (SURVEY.md §2.1)')
        with torch.no_grad():   # hidden states only for `need_layers` (reference _common.py:154-158)
            want_all = getattr(co, 'need_last_layer_output', False)
            hint = self.max_tokens if (self.max_tokens and not co.need_rep and not co.need_emb and not want_all) else 0
            out, _, reps, emb = self._tower.forward(text, training=False, need_rep=co.need_rep, need_emb=co.need_emb,
                                                    rep_layers=list(self.need_layers) if self.need_layers is not None else None,
                                                    tokens_eff=min(int(hint), self.context_length))
        llo = self._tower.last_layer_output() if want_all else None
        return TextTransformerOutput(last_representation=out, last_layer_output=llo, representations=reps, embedding=emb)

    def last_layer_output(self):
        """[B, N, E] = ln_final(x) @ text_projection for every token of the most recent forward (text_encoder.py:69-72)"""
        return self._tower.last_layer_output()

    def forward(self, text, control_output: ControlOutput = None):
        return self.encode_text(text, control_output)

    def hyper_para(self):
        return {'context_length': self.context_length, 'transformer_width': self.transformer_width,
                'transformer_layers': self.layers, 'transformer_heads': self.transformer_heads,
                'vocab_size': self.vocab_size, 'embed_dim': self.embed_dim}
