"""localise the 336 px (N = 101) discrepancy: teacher / student image towers vs the oracle, B = 4 and inside B = 512"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from distillclip_amd import synth
from distillclip_amd.model.utils import teacher_load
from distillclip_amd.model.component import ControlOutput, RepeatVisionTransformer

T = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
rel = lambda a, b: ((a.detach().float().cpu() - b.detach().float().cpu()).norm() / b.detach().float().cpu().norm()).item()
res = int(sys.argv[1]) if len(sys.argv) > 1 else 336
seed = 2022
tsd = T(synth.teacher_image_state(seed, resolution=res)); tsd.update(T(synth.teacher_text_state(seed)))
enc = teacher_load('ViT-B/32', './.cache', 'image', need_layers=list(range(12)), state_dict=tsd).cuda()
small = torch.from_numpy(synth.images(seed, 4, res))
co = ControlOutput(need_rep=True, need_emb=True)
cap = {}
with torch.no_grad():
    ref = oracle.teacher_image_forward({k: v for k, v in tsd.items() if k.startswith('visual.')}, small, need_layers=list(range(12)), need_rep=True, need_emb=True)
    o4 = enc(small.cuda(), co)
print('teacher B=4 vs oracle: emb', rel(o4.embedding, ref['embedding']), 'last', rel(o4.last_representation, ref['last_representation']))
print('  reps', [round(rel(a, b), 4) for a, b in zip(o4.representations, ref['representations'])])
for B in (64, 512):
    full = torch.from_numpy(synth.images(seed + 1, B, res)); idx = [3, 130 % B, 255 % B, 77 % B]; full[idx] = small
    with torch.no_grad():
        oB = enc(full.cuda(), co)
    print(f'teacher B={B} vs B=4: emb', rel(oB.embedding[idx], o4.embedding), 'last', rel(oB.last_representation[idx], o4.last_representation))
    print('  reps', [round(rel(a[idx], b), 4) for a, b in zip(oB.representations, o4.representations)])
    print('  per-sample last', [round(rel(oB.last_representation[i], o4.last_representation[j]), 4) for j, i in enumerate(idx)])
# student
cfg = dict(img_size=res, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24, mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
sd = T(synth.student_image_state(seed, **cfg))
s = RepeatVisionTransformer(**cfg); s.load_state_dict(sd); s = s.cuda()
with torch.no_grad():
    so = s(small.cuda(), co)
    sr = oracle.student_image_forward(sd, small, 24, need_rep=True)
print('student B=4 vs oracle: last', rel(so.last_representation, sr['last_representation']), 'reps', [round(rel(a, b), 4) for a, b in zip(so.representations, sr['representations'])])
