#!/usr/bin/env python3
"""Step time of a dual distill step whose students are PLAIN CLIP encoders (ImageEncoder / TextEncoder with is_student=True: tower kind 2,
DESIGN.md section 7.7) under the ViT-B/32 teacher pair — the path no shipped YAML uses, measured so that it has a number next to its parity tests.

    python tools/diag/clip_student_step.py [--batch 512] [--layers 6] [--width 512] [--steps 20] [--hidden]

Prints one JSON line: ms per step (HIP events around the timed steps), pairs/s, the per-kernel-family time of one traced step."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--layers', type=int, default=6)
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--hidden', action='store_true', help='add hidden_rep_mse (projected hidden states of every layer against teacher layers)')
    a = ap.parse_args()
    import numpy as np
    import torch
    from distillclip_amd import synth
    from distillclip_amd.model import DualDistillModel
    from distillclip_amd.model.component import ImageEncoder, TextEncoder
    T = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
    seed, B, W, L = 2022, a.batch, a.width, a.layers
    tsd = synth.teacher_image_state(seed)
    tsd.update(synth.teacher_text_state(seed))
    s_img = ImageEncoder(True, dict(input_resolution=224, patch_size=32, width=W, layers=L, heads=W // 64, output_dim=512), 768)
    s_txt = TextEncoder(W, L, W // 64, 77, None, 49408, 512, tea_transformer_width=512, is_student=True)
    sd_i, sd_t = synth.clip_student_states(seed + 1, W, L, 32, 224, 77, 49408, 512, 768, 512)
    s_img.load_state_dict(T(sd_i))
    s_txt.load_state_dict(T(sd_t))
    names = ['out_l1', 'out_cos', 'cos_diff'] + (['hidden_rep_mse'] if a.hidden else [])
    need = [int(round((i + 1) * 12 / L)) - 1 for i in range(L)] if a.hidden else None
    model = DualDistillModel(s_img, s_txt, dict(loss_name=names, loss_scale={'cos_diff': 0.1}), 10, 200, 1e-3, 1e-3, None,
                             teacher_need_layers=need, teacher_state_dict=T(tsd)).cuda()
    (opt,), _ = model.configure_optimizers()
    opt.lr = 1e-4
    image = torch.from_numpy(synth.images(seed, B, 224)).cuda()
    text = torch.from_numpy(synth.captions(seed, B)).cuda()

    def step():
        loss = model.training_step([image, text])
        opt.zero_grad()
        model.backward_and_sync(loss, defer_wait=True)
        opt.step(zero_grad=True, overlap=True, join=False)
        return loss

    for _ in range(a.warmup):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    n_par = sum(p.numel() for p in model.student.parameters())
    print(json.dumps({'workload': f'dual distill, plain CLIP students {L} x {W} ({W // 64} heads) under ViT-B/32, B = {B}, losses {names}',
                      'ms_per_step': round(ms, 3), 'pairs_per_s': round(B / ms * 1e3, 1), 'student_parameters': n_par,
                      'loss': float(loss)}))


if __name__ == '__main__':
    main()
