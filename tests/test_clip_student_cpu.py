"""Host-side checks of the plain CLIP encoders in the student role (tower kind 2; reference image_encoder.py:16-25,70-99 ; text_encoder.py:41-47,
124-155): the plan, the layer-mapped initialisation from the teacher, the parameter table.  No GPU: nothing here launches a kernel."""
import ctypes

import pytest
import torch

import real_cases as rc
from distillclip_amd import synth


def test_layer_mapped_initialisation_from_the_teacher():
    """init_layers_with_teacher (reference image_encoder.py:70-99, text_encoder.py:124-155): begin / end / mid layer maps"""
    from types import SimpleNamespace
    from distillclip_amd.model.component import TextEncoder
    tsd = rc.T(synth.teacher_text_state(5, 128, 6, 13, 97, 64))
    for init, src in (('begin', [0, 1]), ('end', [4, 5]), ('mid', [0, 3])):
        stu = TextEncoder(128, 2, 2, 13, None, 97, 64, tea_transformer_width=128, is_student=True)
        stu.init_layers_with_teacher(SimpleNamespace(stu_total_layer_num=2, tea_total_layer_num=6, step=3), tsd, init)
        sd = stu.state_dict()
        for i, j in enumerate(src):
            assert torch.equal(sd[f'transformer.resblocks.{i}.mlp.c_fc.weight'], tsd[f'transformer.resblocks.{j}.mlp.c_fc.weight']), (init, i)
        assert torch.equal(sd['token_embedding.weight'], tsd['token_embedding.weight'])
    with pytest.raises(ValueError):
        stu.init_layers_with_teacher(SimpleNamespace(stu_total_layer_num=2, tea_total_layer_num=6, step=3), tsd, 'middle')


def test_trainable_clip_tower_plan_is_host_side():
    """kind 2 = the CLIP architecture and parameter order of kind 0 with the training workspace / backward of kind 1"""
    from distillclip_amd._lib import lib
    from distillclip_amd.model.component._tower import EncoderCfg
    l = lib()
    mk = lambda **kw: EncoderCfg(**dict(dict(kind=2, modality=0, tokens=50, width=512, heads=8, layers=4, repeats=1, mlp_dim=2048, out_dim=512,
                                             patch=32, resolution=224, in_chans=3, vocab=0, embed_rank=0, head_mix=0, causal=0), **kw))
    h2, h0 = l.dclip_encoder_create(ctypes.byref(mk())), l.dclip_encoder_create(ctypes.byref(mk(kind=0)))
    assert h2 and h0
    assert l.dclip_encoder_num_params(h2) == l.dclip_encoder_num_params(h0) == 5 + 12 * 4 + 3
    assert l.dclip_encoder_wcache_bytes(h2) > l.dclip_encoder_wcache_bytes(h0)                    # + the transposed weights of the dgrad GEMMs
    assert l.dclip_encoder_workspace_bytes(h2, 8, 1) > 4 * l.dclip_encoder_workspace_bytes(h2, 8, 0)
    assert l.dclip_encoder_num_grad_buckets(h2) == 4 + 2
    first, end = ctypes.c_int32(), ctypes.c_int32()
    l.dclip_encoder_grad_bucket(h2, 1, ctypes.byref(first), ctypes.byref(end))
    assert (first.value, end.value) == (5 + 12 * 3, 5 + 12 * 4)                                  # the last layer completes first
    with pytest.raises(ValueError, match='inference-only'):                                       # the frozen kind still refuses to train
        l.dclip_encoder_forward(h0, 256, 1, (ctypes.c_void_p * 56)(), 256, 256, 1 << 40, 1, 256, None, None, 0, None)
    for h in (h2, h0):
        l.dclip_encoder_destroy(h)
    assert not l.dclip_encoder_create(ctypes.byref(mk(kind=3)))
    assert not l.dclip_encoder_create(ctypes.byref(mk(modality=1, tokens=77, vocab=49408, embed_rank=256, causal=1, patch=0, resolution=0, in_chans=0)))
    assert b'weight-shared student only' in l.dclip_last_error_string()


def test_student_role_constructor_contract():
    from distillclip_amd.model.component import ImageEncoder, TextEncoder
    paras = dict(input_resolution=32, patch_size=8, width=128, layers=2, heads=2, output_dim=64)
    with pytest.raises(ValueError, match='tea_transformer_width'):
        ImageEncoder(True, paras)
    with pytest.raises(NotImplementedError, match='text_encoder.py:95'):
        TextEncoder(128, 2, 2, 13, None, 97, 64, 128, compression_embedding=True)
    s = ImageEncoder(True, paras, 128)
    assert s.no_trans and len(s.extra_parameters()) == 4 and s._tower.cfg.kind == 2
    t = TextEncoder(128, 2, 2, 13, None, 97, 64, 192)                      # is_student defaults to True, as in the reference
    assert t.is_student and not t.no_trans and t.hidden_projection.weight.shape == (192, 128)
    assert ImageEncoder(False, paras).embedding_projection is None
    with pytest.raises(RuntimeError, match='no CPU fallback'):             # the projections are HIP GEMMs like everything else
        t.hidden_projection(torch.zeros(2, 13, 128))
