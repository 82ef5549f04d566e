#!/usr/bin/env python3
"""Random shapes through the trainable CLIP towers (tower kind 2) against the fp32 oracle: width, head count / head dim, depth, patch grid
(1 .. 128 image tokens), context length (causal, 4 .. 128), batch 1 .. 6.  argv: [cases] [seed].  Prints the worst gradient error per case;
exit status 1 if any case exceeds the tolerance of tests/test_clip_student_gpu.py::test_edge_shapes_against_the_oracle (3e-2)."""
import os
import random
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import oracle                                         # noqa: E402  (diagnostic tool: the oracle is the checker)
from distillclip_amd import synth                     # noqa: E402
from distillclip_amd.model.component import ImageEncoder, TextEncoder, ControlOutput    # noqa: E402

T = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
rel = lambda a, b: ((a.detach().float().cpu() - b.detach().float().cpu()).norm() / (b.detach().float().cpu().norm() + 1e-12)).item()
cases, seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = random.Random(seed0)
bad = 0
for case in range(cases):
    hd = rng.choice([32, 64])
    heads = rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 16])
    while heads * hd > 1024 or (heads * hd) % 64:
        heads = rng.choice([1, 2, 3, 4, 6, 8])
    width = heads * hd
    layers = rng.randint(1, 3)
    patch = rng.choice([8, 16, 32])
    grid = rng.randint(1, 11 if patch == 8 else (9 if patch == 16 else 6))
    res = patch * grid + rng.choice([0, 0, 4])                      # (the conv floors: a few extra pixels are ignored)
    if (res % 4):
        res = patch * grid
    ctx = rng.randint(4, 128)
    B = rng.randint(1, 6)
    E = rng.choice([64, 128, 512])
    vocab = rng.choice([97, 1000])
    seed = 1000 + case
    tw = rng.choice([width, width + 64])                            # no_trans on / off for the image tower
    sd_i, sd_t = synth.clip_student_states(seed, width, layers, patch, patch * grid, ctx, vocab, E, tw, tw)
    sd_i, sd_t = T(sd_i), T(sd_t)
    s_img = ImageEncoder(True, dict(input_resolution=patch * grid, patch_size=patch, width=width, layers=layers, heads=heads, output_dim=E), tw)
    s_txt = TextEncoder(width, layers, heads, ctx, None, vocab, E, tea_transformer_width=tw, is_student=True)
    s_img.load_state_dict(sd_i); s_txt.load_state_dict(sd_t)
    s_img, s_txt = s_img.cuda(), s_txt.cuda()
    image = torch.from_numpy(synth.images(seed, B, res))
    text = torch.from_numpy(synth.captions(seed, B, ctx, vocab, 1, max(1, ctx - 2)))
    ti, tt = torch.from_numpy(synth.normal(seed, 'ti', (B, E))), torch.from_numpy(synth.normal(seed, 'tt', (B, E)))
    co = ControlOutput(need_rep=True, need_emb=True)
    oi, ot = s_img(image.cuda(), co), s_txt(text.cuda(), co)
    cs = torch.nn.functional.cosine_similarity
    obj = lambda a, b, x, y: ((1 - cs(a['last_representation'], x)).mean() + (1 - cs(b['last_representation'], y)).mean()
                              + sum(r.pow(2).mean() for r in a['representations']) + b['embedding'].pow(2).mean()
                              + sum(r.pow(2).mean() for r in b['representations']) + a['embedding'].pow(2).mean())
    d = lambda o: dict(last_representation=o.last_representation, representations=o.representations, embedding=o.embedding)
    loss = obj(d(oi), d(ot), ti.cuda(), tt.cuda())
    loss.backward()
    o_i = {k: v.clone().requires_grad_(True) for k, v in sd_i.items()}
    o_t = {k: v.clone().requires_grad_(True) for k, v in sd_t.items()}
    ri = oracle.clip_student_image_forward(o_i, image[..., :patch * grid, :patch * grid] if res != patch * grid else image, heads, True, True, no_trans=s_img.no_trans)
    rt = oracle.clip_student_text_forward(o_t, text, heads, True, True, no_trans=s_txt.no_trans)
    ref = obj(ri, rt, ti, tt)
    ref.backward()
    errs = {}
    for tag, m, sd in (('img', s_img, o_i), ('txt', s_txt, o_t)):
        for n, p in m.named_parameters():
            if sd[n].grad is None:
                assert p.grad is None, n
                continue
            errs[f'{tag}.{n}'] = rel(p.grad, sd[n].grad)
    worst = max(errs.items(), key=lambda kv: kv[1])
    e_emb = max(rel(oi.last_representation, ri['last_representation']), rel(ot.last_representation, rt['last_representation']))
    ok = worst[1] <= 3e-2 and e_emb <= 2e-2 and abs(loss.item() - ref.item()) <= 1e-2 * abs(ref.item())
    bad += not ok
    print(f'case {case:2d} width {width:4d} heads {heads:2d} hd {hd} layers {layers} image {res:3d}px/{patch:2d} ({grid * grid + 1:3d} tok) ctx {ctx:3d} B {B} E {E:3d}: '
          f'embedding {e_emb:.1e} loss {abs(loss.item() - ref.item()) / abs(ref.item()):.1e} worst gradient {worst[1]:.1e} ({worst[0]}) {"ok" if ok else "FAIL"}', flush=True)
    del s_img, s_txt
sys.exit(1 if bad else 0)
