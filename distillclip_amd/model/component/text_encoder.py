"""CLIP text encoder (reference model/component/text_encoder.py:8-155): token + positional embedding, causal
residual-attention blocks, ln_final, text_projection, EOT pooling by argmax of the token ids.

is_student=False: the frozen teacher (tower kind 0: inference only, fp16 residual stream).  is_student=True (the reference's
default): the same architecture as a trainable student (tower kind 2: f32 residual stream, backward through the C ABI) with the
reference's `embedding_projection` / `hidden_projection` linears on the exported hidden states (:45-47, :75-80) and its
layer-mapped initialisation from the teacher (:124-155)."""
import torch
from torch import nn

from .output import ControlOutput, TextTransformerOutput
from ._tower import EncoderCfg, HipTower, run_tower
from ._proj import HipLinear
from .image_encoder import TeacherTransformer, _LN, teacher_block_names, student_anchor, init_layers_from_teacher


class TextEncoder(nn.Module):
    def __init__(self, transformer_width, transformer_layers, transformer_heads, context_length, need_layers, vocab_size,
                 embed_dim, tea_transformer_width=None, is_student=True, drop_out=0., compression_embedding=False,
                 embedding_compression_dim=256):
        super().__init__()
        if drop_out:
            raise NotImplementedError('dropout is 0 in every shipped config and is not implemented')
        if compression_embedding:
            # the reference's own constructor fails for this option: initialize_parameters (:95) takes `.weight` of the nn.Sequential
            # that :20-23 builds.  The compressed embedding exists for RepeatTextTransformer (weight_share_model.py:401-405).
            raise NotImplementedError('TextEncoder(compression_embedding=True) cannot be constructed in the reference either '
                                      '(text_encoder.py:95); use RepeatTextTransformer(compression_embedding=True)')
        self.context_length, self.transformer_width, self.transformer_heads = context_length, transformer_width, transformer_heads
        self.vocab_size, self.embed_dim, self.layers = vocab_size, embed_dim, transformer_layers
        self.is_student = bool(is_student)
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
        self.embedding_projection = None
        self.hidden_projection = None
        self.no_trans = transformer_layers == tea_transformer_width           # (sic) reference :43-44 compares the LAYER count
        if is_student:
            if not tea_transformer_width:
                raise ValueError('TextEncoder(is_student=True) needs tea_transformer_width (nn.Linear(width, tea_transformer_width))')
            self.embedding_projection = HipLinear(transformer_width, tea_transformer_width)
            self.hidden_projection = HipLinear(transformer_width, tea_transformer_width)
        cfg = EncoderCfg(kind=2 if is_student else 0, modality=1, tokens=context_length, width=transformer_width,
                         heads=transformer_heads, layers=transformer_layers, repeats=1, mlp_dim=4 * transformer_width,
                         out_dim=embed_dim, patch=0, resolution=0, in_chans=0, vocab=vocab_size,
                         embed_rank=0, head_mix=0, causal=1)
        names = ['token_embedding.weight', 'positional_embedding']
        names += teacher_block_names('', transformer_layers) + ['ln_final.weight', 'ln_final.bias', 'text_projection']
        object.__setattr__(self, '_tower', HipTower(self, cfg, names))
        self.register_load_state_dict_post_hook(lambda m, keys: setattr(m._tower, 'wcache_dirty', True))

    @property
    def need_layers(self):
        return self._need_layers

    def extra_parameters(self):
        """trainable parameters that are not part of the tower's flat buffers (the optimizer updates them one by one)"""
        return [p for m in (self.embedding_projection, self.hidden_projection) if m is not None for p in m.parameters()]

    def encode_text(self, text, control_output: ControlOutput = None):
        co = control_output or ControlOutput()
        if co.need_attn_score or co.need_attn_prob or co.need_value_map:
            raise NotImplementedError('teacher attention maps are not exported by the HIP tower (SURVEY.md §2.1)')
        want_all = getattr(co, 'need_last_layer_output', False)
        if self.is_student:
            if self.need_layers is not None and list(self.need_layers) != list(range(self.layers)):
                raise NotImplementedError('a trainable CLIP tower exports every layer\'s hidden state (need_layers = all)')
            out, reps, emb = run_tower(self._tower, text, student_anchor(self, text.device), co.need_rep, co.need_emb)
            if not self.no_trans:                                                        # reference :75-80
                if co.need_rep:
                    reps = [self.hidden_projection(r) for r in reps]
                if co.need_emb:
                    emb = self.embedding_projection(emb)
            llo = self._tower.last_layer_output() if want_all else None
            return TextTransformerOutput(last_representation=out, last_layer_output=llo, representations=reps, embedding=emb)
        with torch.no_grad():   # hidden states only for `need_layers` (reference _common.py:154-158)
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

    def init_layers_with_teacher(self, layer_map, teacher_state_dict=None, init_type=None):
        """reference :124-155"""
        init_layers_from_teacher(self, self.state_dict(), self.load_state_dict, 'transformer.resblocks.([\\d])', layer_map,
                                 teacher_state_dict, init_type)

    def hyper_para(self):
        return {'context_length': self.context_length, 'transformer_width': self.transformer_width,
                'transformer_layers': self.layers, 'transformer_heads': self.transformer_heads,
                'vocab_size': self.vocab_size, 'embed_dim': self.embed_dim}
