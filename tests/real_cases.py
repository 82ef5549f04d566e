"""Inputs of the real-shape golden files real_b4_image1 / real_b4_textc / real_b4_336 (tools/golden/gen_golden.py, round 5: BASELINE.json
configs[1], [2], [4] run through the reference's own modules at B = 4), rebuilt from the seeds the files carry, and the comparison of a
gradient set with what the reference produced.  Shared by tests/test_oracle_golden.py (CPU: oracle vs reference) and
tests/test_configs_gpu.py (HIP path vs reference, HIP path vs rounding-matched oracle)."""
import os

import numpy as np
import torch

from distillclip_amd import synth

S_IMG = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24, mlp_ratio=4.0,
             qkv_bias=True, repeated_times=2, use_transform=True)
S_TXT = dict(depth=4, repeated_times=2, use_transform=True)
S_TXTC = dict(depth=4, repeated_times=2, use_transform=True, compression_embedding=True)
FROZEN_IMAGE = ('patch_embed.proj.weight', 'cls_token', 'pos_embed')       # reference distil_model.py:200


def T(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def image1_inputs(g):
    """image.yaml: (image, teacher state, student state with the teacher's conv1 / class / positional embeddings copied in as
    DistillModel.freeze_image_embedding does, reference distil_model.py:197-213)"""
    seed, B = int(g['seed']), int(g['B'])
    assert tuple(g['frozen']) == FROZEN_IMAGE
    tsd = T(synth.teacher_image_state(seed))
    sd = T(synth.student_image_state(seed, **S_IMG))
    sd['patch_embed.proj.weight'] = tsd['visual.conv1.weight'].clone()
    sd['cls_token'] = tsd['visual.class_embedding'].view(1, 1, -1).clone()
    sd['pos_embed'] = tsd['visual.positional_embedding'].unsqueeze(0).clone()
    return torch.from_numpy(synth.images(seed, B, 224)), tsd, sd


def textc_inputs(g):
    seed, B = int(g['seed']), int(g['B'])
    return torch.from_numpy(synth.captions(seed, B)), T(synth.teacher_text_state(seed)), T(synth.student_text_state(seed, **S_TXTC))


def l336_inputs(g):
    seed, B = int(g['seed']), int(g['B'])
    tsd = T(synth.teacher_image_state(seed, resolution=336))
    tsd.update(T(synth.teacher_text_state(seed)))
    return (torch.from_numpy(synth.images(seed, B, 336)), torch.from_numpy(synth.captions(seed, B)), tsd,
            T(synth.student_image_state(seed, **dict(S_IMG, img_size=336))), T(synth.student_text_state(seed, **S_TXT)))


# plain CLIP encoders in the student role (tools/golden/gen_golden.py clip_student_tiny / clip_student_real)
CLIPSTU_TINY = dict(res=32, patch=8, ctx=13, vocab=97, out_dim=64, width=128, layers=2, heads=2, tea_width=192, tea_layers=2, tea_heads=3,
                    need_layers=None)
CLIPSTU_REAL = dict(res=224, patch=32, ctx=77, vocab=49408, out_dim=512, width=512, layers=4, heads=8, tea_width=None, tea_layers=12,
                    tea_heads=None, need_layers=[2, 5, 8, 11])
CLIPSTU_LOSSES = ['out_l1', 'out_cos', 'cos_diff', 'hidden_rep_mse', 'embedding_mse']
CLIPSTU_SMOOTH = ['out_cos', 'hidden_rep_mse', 'embedding_mse']


def clipstu_inputs(g, c):
    """-> image, text, teacher image state, teacher text state, student image state, student text state (the generator's recipe:
    teacher from `seed`, students from `seed + 1`)"""
    seed, B = int(g['seed']), int(g['B'])
    if c is CLIPSTU_TINY:
        image = torch.from_numpy(synth.images(seed, B, c['res']))
        text = torch.from_numpy(synth.captions(seed, B, c['ctx'], c['vocab'], 3, 9))
        tsd_i = T(synth.teacher_image_state(seed, c['tea_width'], c['tea_layers'], c['patch'], c['res'], c['out_dim']))
        tsd_t = T(synth.teacher_text_state(seed, c['tea_width'], c['tea_layers'], c['ctx'], c['vocab'], c['out_dim']))
        tw_i = tw_t = c['tea_width']
    else:
        image, text = torch.from_numpy(synth.images(seed, B, 224)), torch.from_numpy(synth.captions(seed, B))
        tsd_i, tsd_t = T(synth.teacher_image_state(seed)), T(synth.teacher_text_state(seed))
        tw_i, tw_t = 768, 512
    sd_i, sd_t = synth.clip_student_states(seed + 1, c['width'], c['layers'], c['patch'], c['res'], c['ctx'], c['vocab'], c['out_dim'], tw_i, tw_t)
    return image, text, tsd_i, tsd_t, T(sd_i), T(sd_t)


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def gradient_errors(g, prefix, grads, skip=()):
    """grads: {parameter name: gradient tensor}.  -> ({name: error of the 512-element sample (first 256 elements + 256 spread over the tensor)},
    {name: relative error of the norm}, number of parameters the file holds under `prefix`).
    Sample error = |got - ref| / max(|ref|, sqrt(512) * RMS of the whole reference gradient): the plain rel-L2 of the sample when the sample is
    as large as the tensor's typical entries, and an error relative to the TENSOR's scale when the sampled entries happen to be small (the
    first 256 inputs of head.weight's row 0 are 16 x below that tensor's RMS: rel-L2 against such a sample measures bf16 noise of the other
    entries' magnitude, 7.7e-2 where the same error is 5e-3 of the tensor's scale).  Norms are compared against the largest norm of the set
    when the reference norm is (near) zero."""
    names = [k[len(prefix) + 7:] for k in g if k.startswith(prefix + '.gnorm.')]
    scale = max(float(g[f'{prefix}.gnorm.{n}']) for n in names)
    samples, norms = {}, {}
    for n in names:
        if n in skip:
            continue
        gr = grads[n].detach().float().cpu().reshape(-1).numpy()
        ref_norm = float(g[f'{prefix}.gnorm.{n}'])
        step = max(1, gr.size // 256)
        got = np.concatenate([gr[:256], gr[::step][:256]]).astype(np.float64)
        ref = np.concatenate([g[f'{prefix}.ghead.{n}'], g[f'{prefix}.gspread.{n}']]).astype(np.float64)
        typical = np.sqrt(len(ref)) * max(ref_norm, 1e-7 * scale) / np.sqrt(gr.size)
        samples[n] = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), typical))
        norms[n] = abs(float(np.linalg.norm(gr)) - ref_norm) / (ref_norm + 1e-6 * scale)
    return samples, norms, len(names)
