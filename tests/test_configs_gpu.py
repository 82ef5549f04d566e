"""The remaining BASELINE.json configurations as parity cases (real shapes, small batch), HIP path vs the oracle on the same
seeded inputs:  image.yaml (one tower, frozen teacher embeddings)  /  text.yaml (compressed embedding)  /  336 px dual."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from distillclip_amd import synth

pytestmark = pytest.mark.gpu
os.environ['DCLIP_SYNTHETIC_TEACHER'] = '1'
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))       # tests/real_cases.py

S_IMG = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24, mlp_ratio=4.0,
             qkv_bias=True, repeated_times=2, use_transform=True)


def T(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


def rel(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def test_image_yaml_one_tower_freeze_embed():
    """config/final_config/image.yaml: DistillModel(model_type='image', freeze_embed=True, teacher_need_layers=[0,1,10,11])"""
    from distillclip_amd.model import DistillModel
    from distillclip_amd.model.component import RepeatVisionTransformer
    seed, B = 5, 4
    tsd = T(synth.teacher_image_state(seed))
    tsd.update(T(synth.teacher_text_state(seed)))
    student = RepeatVisionTransformer(**S_IMG)
    sd = T(synth.student_image_state(seed, **S_IMG))
    student.load_state_dict(sd)
    m = DistillModel(student, dict(loss_name=['out_l1', 'out_cos']), './.cache', freeze_embed=True,
                     teacher_need_layers=[0, 1, 10, 11], model_type='image', weight_decay=1e-2, lr=5e-3,
                     teacher_state_dict=tsd).cuda()
    image = torch.from_numpy(synth.images(seed, B))
    loss = m.training_step(image.cuda())
    loss.backward()
    # oracle: same students with the teacher's patch / class / positional embeddings copied in (distil_model.py:197-213)
    sd = {k: v.clone() for k, v in sd.items()}
    sd['patch_embed.proj.weight'] = tsd['visual.conv1.weight'].clone()
    sd['cls_token'] = tsd['visual.class_embedding'].view(1, 1, -1).clone()
    sd['pos_embed'] = tsd['visual.positional_embedding'].unsqueeze(0).clone()
    for v in sd.values():
        v.requires_grad_(True)
    so = oracle.student_image_forward(sd, image, 24)
    with torch.no_grad():
        to = oracle.teacher_image_forward({k: v for k, v in tsd.items() if k.startswith('visual.')}, image)
    ol, _ = oracle.LossOracle(['out_l1', 'out_cos'])(so, to, 'image')
    ol.backward()
    assert abs(loss.item() - ol.item()) <= 2e-2 * abs(ol.item()), (loss.item(), ol.item())
    named = dict(m.student.named_parameters())
    for frozen in ('patch_embed.proj.weight', 'cls_token', 'pos_embed'):
        assert named[frozen].grad is None and not named[frozen].requires_grad
    # (gradients are held parameter by parameter to the reference's own run and to the rounding-matched oracle below:
    #  test_image_yaml_real_shapes_vs_reference_golden; here only that the shipped loss set's backward is finite everywhere)
    assert all(torch.isfinite(p.grad).all() for n, p in named.items() if p.requires_grad)
    # optimizer step runs on the trainable ranges only
    (opt,), _ = m.configure_optimizers()
    before = named['cls_token'].detach().clone()
    opt.step()
    assert torch.equal(named['cls_token'], before)


def test_text_yaml_compressed_embedding():
    """config/final_config/text.yaml: RepeatTextTransformer(depth=4, R=2, use_transform, compression_embedding=True)"""
    from distillclip_amd.model import DistillModel
    from distillclip_amd.model.component import RepeatTextTransformer
    seed, B = 6, 6
    cfg = dict(depth=4, repeated_times=2, use_transform=True, compression_embedding=True)
    tsd = T(synth.teacher_image_state(seed))
    tsd.update(T(synth.teacher_text_state(seed)))
    student = RepeatTextTransformer(**cfg)
    sd = T(synth.student_text_state(seed, **cfg))
    student.load_state_dict(sd)
    m = DistillModel(student, dict(loss_name=['out_l1', 'out_cos']), './.cache', model_type='text', teacher_state_dict=tsd).cuda()
    text = torch.from_numpy(synth.captions(seed, B))
    loss = m.training_step(text.cuda())
    loss.backward()
    for v in sd.values():
        v.requires_grad_(True)
    so = oracle.student_text_forward(sd, text, 12)
    with torch.no_grad():
        to = oracle.teacher_text_forward({k: v for k, v in tsd.items() if not k.startswith('visual.')}, text)
    ol, _ = oracle.LossOracle(['out_l1', 'out_cos'])(so, to, 'text')
    ol.backward()
    assert abs(loss.item() - ol.item()) <= 2e-2 * abs(ol.item()), (loss.item(), ol.item())
    named = dict(m.student.named_parameters())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in named.values())


def test_l_clip_336px_dual():
    """BASELINE configs[4]: 336 px images -> N = (336 // 32)^2 + 1 = 101 tokens (conv floor), synthetic [101, 768] teacher pos-embed"""
    from distillclip_amd.model import DualDistillModel
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    seed, B = 7, 3
    s_img_cfg = dict(S_IMG, img_size=336)
    s_txt_cfg = dict(depth=4, repeated_times=2, use_transform=True)
    tsd = T(synth.teacher_image_state(seed, resolution=336))
    tsd.update(T(synth.teacher_text_state(seed)))
    sdi, sdt = T(synth.student_image_state(seed, **s_img_cfg)), T(synth.student_text_state(seed, **s_txt_cfg))
    si, st = RepeatVisionTransformer(**s_img_cfg), RepeatTextTransformer(**s_txt_cfg)
    si.load_state_dict(sdi)
    st.load_state_dict(sdt)
    m = DualDistillModel(si, st, dict(loss_name=['out_l1', 'out_cos', 'cos_diff'], loss_scale={'cos_diff': 0.1}), 15, 300, 1e-3,
                         1e-4, './.cache', teacher_state_dict=tsd).cuda()
    assert m.teacher.image_encoder._tower.cfg.tokens == 101 and si._tower.cfg.tokens == 101
    image = torch.from_numpy(synth.images(seed, B, 336))
    text = torch.from_numpy(synth.captions(seed, B))
    loss = m.training_step([image.cuda(), text.cuda()])
    m.backward_and_sync(loss)
    for v in list(sdi.values()) + list(sdt.values()):
        v.requires_grad_(True)
    oi, ot = oracle.student_image_forward(sdi, image, 24), oracle.student_text_forward(sdt, text, 12)
    with torch.no_grad():
        ti = oracle.teacher_image_forward({k: v for k, v in tsd.items() if k.startswith('visual.')}, image)
        tt = oracle.teacher_text_forward({k: v for k, v in tsd.items() if not k.startswith('visual.')}, text)
    ol, _ = oracle.LossOracle(['out_l1', 'out_cos', 'cos_diff'], {'cos_diff': 0.1})(oracle.clip_forward(oi, ot), oracle.clip_forward(ti, tt), 'all')
    ol.backward()
    assert abs(loss.item() - ol.item()) <= 2e-2 * abs(ol.item()), (loss.item(), ol.item())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.student.parameters())
    # all four embeddings (the teacher is built from a checkpoint whose positional table only says 10 x 10 patches = 320 px:
    # the tower must read the 336 px input with its own row stride, reference conv floor semantics _common.py:196)
    so_, to_ = m.forward([image.cuda(), text.cuda()])
    for got, want, tag in ((so_.visual_output, oi, 's_img'), (so_.text_output, ot, 's_txt'), (to_.visual_output, ti, 't_img'),
                           (to_.text_output, tt, 't_txt')):
        assert rel(got.last_representation, want['last_representation']) < 2e-2, (tag, rel(got.last_representation, want['last_representation']))
    # (the gradient of every parameter of both 101- / 77-token students is held to the reference's own run below:
    #  test_l_clip_336px_real_shapes_vs_reference_golden)


# ---- round 5: the same three configurations against the REFERENCE's own runs at real shapes (tests/golden/real_b4_{image1,textc,336}.npz,
# tools/golden/gen_golden.py) and against the rounding-matched oracle, the gradient of EVERY trainable parameter ---------------------------
# Bounds: vs the reference (fp32) on the smooth objective <= 5e-2 per parameter (head-mix / bias / norm parameters, whose gradients are sums
# of many small terms of both signs, <= 8e-2: the bounds of the l_clip towers in tests/test_towers_gpu.py), norms <= 2e-2; vs
# oracle.bf16_matched() <= 2.5e-2.  Embeddings rel-L2 <= 2e-2, losses 2e-2.
def _bound(name):
    return 8e-2 if any(t in name for t in ('.attn.conv_', '.bias', 'norm')) else 5e-2


def _hold_to_reference(g, prefix, grads, expect):
    import real_cases as rc
    samples, norms, n = rc.gradient_errors(g, prefix, grads)
    assert n == expect, (prefix, n)
    bad = {k: round(v, 4) for k, v in samples.items() if v > _bound(k)}
    assert not bad, (prefix, bad)
    bad = {k: round(v, 4) for k, v in norms.items() if v > 2e-2}
    assert not bad, (prefix, 'norms', bad)
    return max(samples.values()), max(norms.values())


def _hold_to_matched(named, sd, skip=()):
    import real_cases as rc
    errs = {}
    for n, p in named.items():
        if n in skip or not p.requires_grad:
            continue
        ref = sd[n].grad
        if n.endswith('attn.qkv.bias'):        # the k third has a zero true gradient (softmax shift invariance): q and v thirds only
            D = ref.numel() // 3
            got, want = p.grad.detach().cpu().reshape(-1), ref.reshape(-1)
            errs[n] = max(rc.rel_l2(got[:D], want[:D]), rc.rel_l2(got[2 * D:], want[2 * D:]))
        else:
            errs[n] = rc.rel_l2(p.grad.detach().cpu().numpy(), ref.numpy())
    bad = {k: round(v, 4) for k, v in errs.items() if v > 2.5e-2}
    assert not bad, bad
    return max(errs.values())


def test_image_yaml_real_shapes_vs_reference_golden(golden_dir):
    """image.yaml through DistillModel (freeze_embed applied by the model, teacher_need_layers [0, 1, 10, 11]) at B = 4"""
    import real_cases as rc
    from distillclip_amd.model import DistillModel, LossCalculator
    from distillclip_amd.model.component import RepeatVisionTransformer
    g = rc.load(golden_dir, 'real_b4_image1.npz')
    seed = int(g['seed'])
    image, tsd_img, sd_frozen = rc.image1_inputs(g)
    tsd = dict(tsd_img)
    tsd.update(T(synth.teacher_text_state(seed)))
    student = RepeatVisionTransformer(**rc.S_IMG)
    student.load_state_dict(T(synth.student_image_state(seed, **rc.S_IMG)))      # (the model copies the teacher's embeddings in itself)
    m = DistillModel(student, dict(loss_name=['out_l1', 'out_cos']), './.cache', freeze_embed=True, teacher_need_layers=[0, 1, 10, 11],
                     model_type='image', weight_decay=1e-2, lr=5e-3, teacher_state_dict=tsd).cuda()
    so, to = m.forward(image.cuda())
    assert rel(so.last_representation, torch.from_numpy(g['img1.s.last_representation'])) < 2e-2
    assert rel(to.last_representation, torch.from_numpy(g['img1.t.last_representation'])) < 1e-2
    loss = m.training_step(image.cuda())
    assert abs(loss.item() - float(g['img1.loss'])) <= 2e-2 * float(g['img1.loss'])
    named = dict(m.student.named_parameters())
    assert {n for n, p in named.items() if not p.requires_grad} == set(rc.FROZEN_IMAGE)
    so, to = m.forward(image.cuda())
    l2, _ = LossCalculator(['out_cos'])(so, to, 'image')
    assert abs(l2.item() - float(g['img1.cos.loss'])) <= 2e-2 * float(g['img1.cos.loss'])
    l2.backward()
    grads = {n: p.grad for n, p in named.items() if p.requires_grad}
    worst = _hold_to_reference(g, 'img1.cos', grads, 68 - 3)
    # rounding-matched oracle on the same inputs
    for n, v in sd_frozen.items():
        v.requires_grad_(n not in rc.FROZEN_IMAGE)
    with oracle.bf16_matched():
        with torch.no_grad():
            ot = oracle.teacher_image_forward(tsd_img, image)
        os_ = oracle.student_image_forward(sd_frozen, image, 24)
        ol, _ = oracle.LossOracle(['out_cos'])(os_, ot, 'image')
        ol.backward()
    assert rel(so.last_representation, os_['last_representation']) < 5e-3 and rel(to.last_representation, ot['last_representation']) < 5e-3
    wm = _hold_to_matched(named, sd_frozen)
    print('image.yaml real shapes: worst gradient vs reference (samples, norms)', worst, 'vs matched oracle', wm)


def test_text_yaml_real_shapes_vs_reference_golden(golden_dir):
    """text.yaml (compressed embedding student) through DistillModel at B = 4"""
    import real_cases as rc
    from distillclip_amd.model import DistillModel, LossCalculator
    from distillclip_amd.model.component import RepeatTextTransformer
    g = rc.load(golden_dir, 'real_b4_textc.npz')
    seed = int(g['seed'])
    text, tsd_txt, sd = rc.textc_inputs(g)
    tsd = dict(tsd_txt)
    tsd.update(T(synth.teacher_image_state(seed)))
    student = RepeatTextTransformer(**rc.S_TXTC)
    student.load_state_dict(sd)
    m = DistillModel(student, dict(loss_name=['out_l1', 'out_cos']), './.cache', model_type='text', teacher_state_dict=tsd).cuda()
    loss = m.training_step(text.cuda())
    assert abs(loss.item() - float(g['txtc.loss'])) <= 2e-2 * float(g['txtc.loss'])
    so, to = m.forward(text.cuda())
    assert rel(so.last_representation, torch.from_numpy(g['txtc.s.last_representation'])) < 2e-2
    assert rel(to.last_representation, torch.from_numpy(g['txtc.t.last_representation'])) < 1e-2
    l2, _ = LossCalculator(['out_cos'])(so, to, 'text')
    assert abs(l2.item() - float(g['txtc.cos.loss'])) <= 2e-2 * float(g['txtc.cos.loss'])
    l2.backward()
    named = dict(m.student.named_parameters())
    worst = _hold_to_reference(g, 'txtc.cos', {n: p.grad for n, p in named.items()}, len(named))
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    with oracle.bf16_matched():
        with torch.no_grad():
            ot = oracle.teacher_text_forward(tsd_txt, text)
        os_ = oracle.student_text_forward(sdo, text, 12)
        ol, _ = oracle.LossOracle(['out_cos'])(os_, ot, 'text')
        ol.backward()
    assert rel(so.last_representation, os_['last_representation']) < 5e-3 and rel(to.last_representation, ot['last_representation']) < 5e-3
    wm = _hold_to_matched(named, sdo)
    print('text.yaml real shapes: worst gradient vs reference (samples, norms)', worst, 'vs matched oracle', wm)


def test_l_clip_336px_real_shapes_vs_reference_golden(golden_dir):
    """l_clip dual at 336 px (101 tokens) through DualDistillModel at B = 4"""
    import real_cases as rc
    from distillclip_amd.model import DualDistillModel, LossCalculator
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    g = rc.load(golden_dir, 'real_b4_336.npz')
    image, text, tsd, sdi, sdt = rc.l336_inputs(g)
    si, st = RepeatVisionTransformer(**dict(rc.S_IMG, img_size=336)), RepeatTextTransformer(**rc.S_TXT)
    si.load_state_dict(sdi)
    st.load_state_dict(sdt)
    m = DualDistillModel(si, st, dict(loss_name=['out_l1', 'out_cos', 'cos_diff'], loss_scale={'cos_diff': 0.1}), 15, 300, 1e-3,
                         1e-4, './.cache', teacher_state_dict=tsd).cuda()
    batch = [image.cuda(), text.cuda()]
    loss = m.training_step(batch)
    assert abs(loss.item() - float(g['loss'])) <= 2e-2 * float(g['loss'])
    so, to = m.forward(batch)
    for got, tag, tol in ((so.visual_output, 's_img', 2e-2), (so.text_output, 's_txt', 2e-2), (to.visual_output, 't_img', 1e-2),
                          (to.text_output, 't_txt', 1e-2)):
        assert rel(got.last_representation, torch.from_numpy(g[f'{tag}.last_representation'])) < tol, tag
    l2, _ = LossCalculator(['out_cos'])(so, to, 'all')
    assert abs(l2.item() - float(g['cos.loss'])) <= 2e-2 * float(g['cos.loss'])
    l2.backward()
    ni, nt = dict(si.named_parameters()), dict(st.named_parameters())
    wi = _hold_to_reference(g, 'cos.s_img', {n: p.grad for n, p in ni.items()}, 68)
    wt = _hold_to_reference(g, 'cos.s_txt', {n: p.grad for n, p in nt.items()}, 44)
    print('l_clip 336 px real shapes: worst gradient vs reference (samples, norms): image', wi, 'text', wt)
