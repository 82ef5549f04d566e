"""Host-side logic that needs no GPU: LossCalculator bookkeeping / error behaviour (reference model/_loss.py:18-55, :96),
DistillModel argument checks, the cosine schedule, load_weight prefix stripping, freeze rules."""
import math
import os

import pytest
import torch

os.environ['DCLIP_SYNTHETIC_TEACHER'] = '1'


def test_loss_calculator_bookkeeping():
    from distillclip_amd.model import LossCalculator
    lc = LossCalculator(['out_l1', 'out_cos', 'cos_diff'], {'cos_diff': 0.1})
    assert lc.percent == {'out_l1': 1 / 3, 'out_cos': 1 / 3, 'cos_diff': 1 / 3}
    assert lc.loss_scale == {'out_l1': 1, 'out_cos': 1, 'cos_diff': 0.1}
    assert lc._weights(True) == {'out_l1': 1 / 3, 'out_cos': 1 / 3, 'cos_diff': pytest.approx(0.1 / 3)}
    assert 'cos_diff' not in lc._weights(False)                       # cross-modal terms vanish in one-tower mode
    lc = LossCalculator(['out_l1', 'out_cos'], percent={'out_l1': 0.25})     # reference :32-41: remainder spread
    assert lc.percent['out_cos'] == pytest.approx(0.75)
    with pytest.raises(ValueError):
        LossCalculator(['out_l1', 'out_cos'], percent={'out_l1': 1.5})
    with pytest.raises(ValueError, match='Invalid Loss Type'):
        LossCalculator(['nope'])
    with pytest.raises(NotImplementedError):
        LossCalculator(['vit_kd'])
    with pytest.raises(AssertionError):
        LossCalculator(['out_kl'])._weights(False)                     # temperature required (reference :166)
    co = LossCalculator(['out_l1']).get_control_output()
    assert not (co.need_emb or co.need_rep or co.need_attn_score or co.need_attn_prob or co.need_value_map)


def test_cosine_schedule_matches_hf_formula():
    from distillclip_amd.optim import cosine_with_warmup
    assert cosine_with_warmup(0, 15, 300) == 0.0
    assert cosine_with_warmup(15, 15, 300) == 1.0
    assert cosine_with_warmup(7, 15, 300) == pytest.approx(7 / 15)
    assert cosine_with_warmup(300, 15, 300) == pytest.approx(0.0, abs=1e-12)
    mid = 15 + (300 - 15) // 2
    assert cosine_with_warmup(mid, 15, 300) == pytest.approx(0.5 * (1 + math.cos(math.pi * (mid - 15) / 285)))
    try:
        import transformers
    except Exception:
        return
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sch = transformers.get_cosine_schedule_with_warmup(opt, 10, 200)
    for e in range(30):
        assert sch.get_last_lr()[0] == pytest.approx(cosine_with_warmup(e, 10, 200))
        opt.step()
        sch.step()


def _tiny_students():
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    s_img = RepeatVisionTransformer(img_size=32, patch_size=8, out_dim=64, embed_dim=128, depth=4, num_heads=4, qkv_bias=True,
                                    repeated_times=2, use_transform=True)
    s_txt = RepeatTextTransformer(vocab_size=97, context_length=13, out_dim=64, embed_dim=128, depth=2, num_heads=2,
                                  repeated_times=2, use_transform=True)
    return s_img, s_txt


def _tiny_teacher_sd():
    from distillclip_amd import synth
    sd = synth.teacher_image_state(3, 128, 2, 8, 32, 64)
    sd.update(synth.teacher_text_state(3, 128, 2, 13, 97, 64))
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def test_models_construct_and_freeze_rules(tmp_path):
    from distillclip_amd.model import DistillModel, DualDistillModel
    s_img, s_txt = _tiny_students()
    tsd = _tiny_teacher_sd()
    with pytest.raises(ValueError, match='model_type'):
        DistillModel(s_img, dict(loss_name=['out_l1']), '.', model_type='audio', teacher_state_dict=tsd)
    m = DistillModel(s_img, dict(loss_name=['out_l1', 'out_cos']), '.', freeze_embed=True, teacher_state_dict=tsd)
    frozen = sorted(n for n, p in m.student.named_parameters() if not p.requires_grad)
    assert frozen == ['cls_token', 'patch_embed.proj.weight', 'pos_embed']                      # reference distil_model.py:200-213
    assert torch.equal(m.student.cls_token.data.view(-1), tsd['visual.class_embedding'])
    assert torch.equal(m.student.patch_embed.proj.weight.data, tsd['visual.conv1.weight'])
    assert all(not p.requires_grad for p in m.teacher.parameters())
    d = DualDistillModel(*_tiny_students(), dict(loss_name=['out_l1', 'cos_diff']), 1, 10, 1e-3, 1e-4, '.',
                         freeze_prefix=['image_encoder.blocks.0'], teacher_state_dict=tsd)
    assert all(not p.requires_grad for n, p in d.student.named_parameters() if n.startswith('image_encoder.blocks.0'))
    assert any(p.requires_grad for n, p in d.student.named_parameters() if n.startswith('image_encoder.blocks.1'))
    # load_weight: Lightning checkpoint layout {'state_dict': {'student.<key>': ...}} (dual_distill_model.py:22-38)
    ck_i, ck_t = tmp_path / 'i.ckpt', tmp_path / 't.ckpt'
    a, b = _tiny_students()
    torch.save({'state_dict': {'student.' + k: v + 1 for k, v in a.state_dict().items()}}, ck_i)
    torch.save({'state_dict': {'student.' + k: v + 1 for k, v in b.state_dict().items()}}, ck_t)
    d2 = DualDistillModel(a, b, dict(loss_name=['out_l1']), 1, 10, 1e-3, 1e-4, '.', load_path={'image': str(ck_i), 'text': str(ck_t)},
                          teacher_state_dict=tsd)
    assert torch.allclose(d2.student.image_encoder.norm.weight, torch.full((128,), 2.0))
    from distillclip_amd.model.dual_distill_model import load_weight
    with pytest.raises(ValueError, match='cpk is None'):
        load_weight(a, b, {'image': None, 'text': None})


def test_teacher_load_without_network(monkeypatch, tmp_path):
    from distillclip_amd.model import utils
    monkeypatch.delenv('DCLIP_SYNTHETIC_TEACHER', raising=False)
    with pytest.raises(FileNotFoundError):
        utils.teacher_load('ViT-B/32', str(tmp_path), 'image')
    with pytest.raises(ValueError):
        utils.teacher_load('ViT-B/32', str(tmp_path), 'audio', state_dict=_tiny_teacher_sd())
    # a plain state_dict saved under the archive name is accepted and sniffed (utils.py:81-129)
    torch.save(_tiny_teacher_sd(), tmp_path / 'ViT-B-32.pt')
    t = utils.teacher_load('ViT-B/32', str(tmp_path), 'all')
    assert t.image_encoder.vit_paras['width'] == 128 and t.text_encoder.layers == 2


def test_scheduler_state_loads_into_a_lambda_lr_and_back():
    """checkpoint 'lr_schedulers' entry: the keys torch's LambdaLR.load_state_dict consumes (the reference's scheduler is
    transformers.get_cosine_schedule_with_warmup, a LambdaLR stepped per epoch: distil_model.py:164-169)"""
    import torch
    from distillclip_amd.optim import EpochCosineSchedule, cosine_with_warmup

    class _Opt:
        base_lr = lr = 5e-3
    o = _Opt()
    s = EpochCosineSchedule(o, 10, 200)
    for _ in range(7):
        s.step()
    sd = s.state_dict()
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.AdamW([p], lr=5e-3)
    lam = torch.optim.lr_scheduler.LambdaLR(topt, lambda e: cosine_with_warmup(e, 10, 200))
    lam.load_state_dict(sd)                                    # must not raise; LambdaLR keeps its own lambda for lr_lambdas=[None]
    assert lam.last_epoch == 7 and abs(lam.get_last_lr()[0] - o.lr) < 1e-12
    o2 = _Opt()
    s2 = EpochCosineSchedule(o2, 10, 200)
    s2.load_state_dict(lam.state_dict())                       # and the reference-side state loads here
    assert s2.epoch == 7 and abs(o2.lr - o.lr) < 1e-12


def test_checkpoint_version_is_parseable():
    from packaging.version import Version
    from distillclip_amd import checkpoint
    assert Version(checkpoint.LIGHTNING_VERSION) < Version('2.0')


def test_parameter_gradients_travel_through_autograd_only_when_an_outer_wrapper_owns_the_exchange(monkeypatch):
    """DCLIP_DP_MODE=off (torch DistributedDataParallel / Lightning's ddp strategy reduces the gradients) switches the towers to
    returning parameter gradients to autograd, so the reducer's hooks fire; every other mode writes p.grad directly (the fused optimizer
    and the built-in exchange read the flat buffers).  A per-tower override wins over the environment."""
    from distillclip_amd.model.component._tower import autograd_params_mode

    class T:
        pass
    t = T()
    monkeypatch.delenv('DCLIP_DP_MODE', raising=False)
    assert autograd_params_mode(t) is False
    for mode, want in (('off', True), ('allreduce', False), ('reduce_scatter', False)):
        monkeypatch.setenv('DCLIP_DP_MODE', mode)
        assert autograd_params_mode(t) is want, mode
    monkeypatch.setenv('DCLIP_DP_MODE', 'off')
    t.autograd_params = False
    assert autograd_params_mode(t) is False
    monkeypatch.delenv('DCLIP_DP_MODE')
    t.autograd_params = True
    assert autograd_params_mode(t) is True


def test_shared_image_patches_is_a_no_op_without_two_matching_image_towers():
    """shared_image_patches (one im2row for teacher + student) only engages for CUDA f32 images and at least two image towers that cut them the
    same way; everything else — CPU tensors, one tower, text towers, different patch sizes, DCLIP_SHARE_PATCHES=0 — leaves every tower to its
    own conversion (and must not touch the library: this runs without a GPU)."""
    import types
    import torch
    from distillclip_amd.model.component import _tower

    def tw(modality=0, patch=32, chans=3):
        return types.SimpleNamespace(cfg=types.SimpleNamespace(modality=modality, patch=patch, in_chans=chans))
    img = torch.zeros(2, 3, 64, 64)
    for towers in ([tw(), tw()], [tw()], [tw(), tw(modality=1)], [tw(), tw(patch=16)], [None, tw()], []):
        with _tower.shared_image_patches(img, towers) as sh:            # CPU image: never engages
            assert sh.entry is None
            assert _tower._shared_rows_for(img, tw().cfg) is None
    assert getattr(_tower._SHARE, 'entry', None) is None
