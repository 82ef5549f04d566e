"""Ragged / minimal batches through the whole dual distill step (tiny towers): B = 1, 2, 5, 17, 33 against the oracle on the same
seeded inputs — the tilings of every kernel on the path (GEMM row tiles, attention per-(b, h) waves, loss stripes of 16 rows,
softmax rows per wave, wgrad splits) see partial tiles here.  Also captions of extreme lengths (EOT at position 1 / at the end)."""
import numpy as np
import pytest
import torch

import oracle
from distillclip_amd import synth

pytestmark = pytest.mark.gpu

S_IMG = dict(img_size=32, patch_size=8, in_chans=3, out_dim=64, embed_dim=128, depth=4, num_heads=4,
             mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
S_TXT = dict(vocab_size=97, context_length=13, out_dim=64, embed_dim=128, depth=2, num_heads=2,
             mlp_ratio=4.0, qkv_bias=False, repeated_times=2, use_transform=True)
T = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
NAMES = ['out_cos', 'out_kl', 'cos_diff', 'hard_label', 'soft_label', 'logits_mse']


def _build(seed, names=NAMES):
    from distillclip_amd.model import DualDistillModel
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    sd_i, sd_t = T(synth.student_image_state(seed, **S_IMG)), T(synth.student_text_state(seed, **S_TXT))
    tsd = synth.teacher_image_state(seed, 128, 2, 8, 32, 64)
    tsd.update(synth.teacher_text_state(seed, 128, 2, 13, 97, 64))
    tsd = T(tsd)
    s_img, s_txt = RepeatVisionTransformer(**S_IMG), RepeatTextTransformer(**S_TXT)
    s_img.load_state_dict(sd_i)
    s_txt.load_state_dict(sd_t)
    m = DualDistillModel(s_img, s_txt, dict(loss_name=names, temperature=2.0, loss_scale={'cos_diff': 0.1} if 'cos_diff' in names else None), warm_steps=1,
                         total_steps=5, weight_decay=0.0, lr=1e-3, download_root='.', teacher_state_dict=tsd).cuda()
    return m, sd_i, sd_t, tsd


def _oracle_loss(sd_i, sd_t, tsd, image, text, names=NAMES):
    for v in list(sd_i.values()) + list(sd_t.values()):
        v.requires_grad_(True)
        v.grad = None
    oi, ot = oracle.student_image_forward(sd_i, image, 4), oracle.student_text_forward(sd_t, text, 2)
    with torch.no_grad():
        ti = oracle.teacher_image_forward({k: v for k, v in tsd.items() if k.startswith('visual.')}, image)
        tt = oracle.teacher_text_forward({k: v for k, v in tsd.items() if not k.startswith('visual.')}, text)
    ref, res = oracle.LossOracle(names, {'cos_diff': 0.1} if 'cos_diff' in names else None, temperature=2.0)(oracle.clip_forward(oi, ot), oracle.clip_forward(ti, tt), 'all')
    ref.backward()
    return ref, res


@pytest.mark.parametrize('B', [1, 2, 5, 17, 33])
def test_ragged_batch_dual_step(B):
    # B = 1: the reference's cos_diff averages over an empty set of negatives (NaN upstream as well), so it is left out there
    names = NAMES if B > 1 else [n for n in NAMES if n != 'cos_diff']
    m, sd_i, sd_t, tsd = _build(31, names)
    image = torch.from_numpy(synth.images(40 + B, B, 32))
    text = torch.from_numpy(synth.captions(40 + B, B, 13, 97, 3, 9))
    loss = m.training_step([image.cuda(), text.cuda()])
    loss.backward()
    ref, res = _oracle_loss(sd_i, sd_t, tsd, image, text, names)
    assert abs(loss.item() - ref.item()) <= 2e-2 * abs(ref.item()) + 1e-5, (B, loss.item(), ref.item())
    for k, v in m.last_cal_res.items():
        assert torch.isfinite(v).all(), k
    named = dict(m.student.named_parameters())
    for key, want in (('image_encoder.head.weight', sd_i['head.weight'].grad), ('text_encoder.head.weight', sd_t['head.weight'].grad),
                      ('image_encoder.blocks.0.block.mlp.fc1.weight', sd_i['blocks.0.block.mlp.fc1.weight'].grad)):
        g = named[key].grad.cpu()
        assert torch.isfinite(g).all()
        assert (g - want).norm().item() <= 1.2e-1 * want.norm().item() + 1e-7, (B, key, (g - want).norm().item() / want.norm().item())


def test_caption_length_extremes():
    """EOT directly after SOT, and a caption that fills the context: the argmax pooling row is 1 resp. L - 1"""
    m, sd_i, sd_t, tsd = _build(32)
    B, L = 4, 13
    text = torch.zeros(B, L, dtype=torch.int64)
    sot, eot = 95, 96                                             # synth.captions layout: highest id = EOT (argmax pooling)
    text[0, 0], text[0, 1] = sot, eot
    text[1, 0], text[1, 1:L - 1], text[1, L - 1] = sot, torch.arange(1, L - 1), eot
    text[2, 0], text[2, 1:4], text[2, 4] = sot, torch.tensor([7, 8, 9]), eot
    text[3] = text[1]
    image = torch.from_numpy(synth.images(77, B, 32))
    loss = m.training_step([image.cuda(), text.cuda()])
    loss.backward()
    ref, _ = _oracle_loss(sd_i, sd_t, tsd, image, text)
    assert abs(loss.item() - ref.item()) <= 2e-2 * abs(ref.item()) + 1e-5, (loss.item(), ref.item())
    assert all(torch.isfinite(p.grad).all() for p in m.student.parameters() if p.grad is not None)


def test_single_sample_cos_diff_is_nan_like_the_reference():
    """reference clip_cos_diff.py:16-23 at B = 1: the mean over the (empty) set of negatives is NaN; the fused loss agrees"""
    m, sd_i, sd_t, tsd = _build(33, ['out_cos', 'cos_diff'])
    image = torch.from_numpy(synth.images(5, 1, 32))
    text = torch.from_numpy(synth.captions(5, 1, 13, 97, 3, 9))
    loss = m.training_step([image.cuda(), text.cuda()])
    ref, _ = _oracle_loss(sd_i, sd_t, tsd, image, text, ['out_cos', 'cos_diff'])
    assert torch.isnan(ref) and torch.isnan(loss)


def test_real_shapes_tiny_batch_forward():
    """the shipped l_clip towers (ViT-B/32 teacher, 6x768/24h and 4x768/12h students) at B = 1 and 3: GEMM row counts of 50 / 77
    (below one tile) through every tower, embeddings against the oracle"""
    from distillclip_amd.model.component import (RepeatVisionTransformer, RepeatTextTransformer, ImageEncoder, TextEncoder)
    seed = 17
    s_img_cfg = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24,
                     mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
    s_txt_cfg = dict(depth=4, repeated_times=2, use_transform=True)
    sd_si, sd_st = T(synth.student_image_state(seed, **s_img_cfg)), T(synth.student_text_state(seed, **s_txt_cfg))
    sd_ti, sd_tt = T(synth.teacher_image_state(seed)), T(synth.teacher_text_state(seed))
    s_img, s_txt = RepeatVisionTransformer(**s_img_cfg), RepeatTextTransformer(**s_txt_cfg)
    s_img.load_state_dict(sd_si)
    s_txt.load_state_dict(sd_st)
    t_img = ImageEncoder(False, dict(input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512))
    t_img.load_state_dict(sd_ti)
    t_txt = TextEncoder(512, 12, 8, 77, None, 49408, 512, is_student=False)
    t_txt.load_state_dict(sd_tt)
    s_img, s_txt, t_img, t_txt = s_img.cuda(), s_txt.cuda(), t_img.cuda(), t_txt.cuda()
    for B in (1, 3):
        image = torch.from_numpy(synth.images(seed, B, 224))
        text = torch.from_numpy(synth.captions(seed, B))
        with torch.no_grad():
            got = {'s_img': s_img(image.cuda()).last_representation, 's_txt': s_txt(text.cuda()).last_representation,
                   't_img': t_img(image.cuda()).last_representation, 't_txt': t_txt(text.cuda()).last_representation}
            want = {'s_img': oracle.student_image_forward(sd_si, image, 24)['last_representation'],
                    's_txt': oracle.student_text_forward(sd_st, text, 12)['last_representation'],
                    't_img': oracle.teacher_image_forward(sd_ti, image)['last_representation'],
                    't_txt': oracle.teacher_text_forward(sd_tt, text)['last_representation']}
        for k in got:
            g, w = got[k].float().cpu(), want[k]
            rel = ((g - w).norm() / w.norm()).item()
            assert rel < 2e-2, (B, k, rel)
