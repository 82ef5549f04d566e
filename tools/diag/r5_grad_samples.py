"""per-sample detail of the every-parameter gradient comparison against the reference goldens (real_b4_image1 / real_b4_336): for the
worst parameters, the head / spread sample errors, the sample's RMS against the tensor's RMS, and the same against the rounding-matched oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
os.environ['DCLIP_SYNTHETIC_TEACHER'] = '1'
import numpy as np, torch
import oracle, real_cases as rc
from distillclip_amd import synth
from distillclip_amd.model import DistillModel, LossCalculator
from distillclip_amd.model.component import RepeatVisionTransformer

g = rc.load(os.path.join(ROOT, 'tests', 'golden'), 'real_b4_image1.npz')
seed = int(g['seed'])
image, tsd_img, sd_frozen = rc.image1_inputs(g)
tsd = dict(tsd_img); tsd.update(rc.T(synth.teacher_text_state(seed)))
student = RepeatVisionTransformer(**rc.S_IMG)
student.load_state_dict(rc.T(synth.student_image_state(seed, **rc.S_IMG)))
m = DistillModel(student, dict(loss_name=['out_l1', 'out_cos']), './.cache', freeze_embed=True, teacher_need_layers=[0, 1, 10, 11],
                 model_type='image', weight_decay=1e-2, lr=5e-3, teacher_state_dict=tsd).cuda()
if len(sys.argv) > 1:
    _ = m.training_step(image.cuda())       # a training forward whose backward never runs, as in the test
so, to = m.forward(image.cuda())
l2, _ = LossCalculator(['out_cos'])(so, to, 'image')
l2.backward()
named = dict(m.student.named_parameters())
for n, v in sd_frozen.items():
    v.requires_grad_(n not in rc.FROZEN_IMAGE)
with oracle.bf16_matched():
    with torch.no_grad():
        ot = oracle.teacher_image_forward(tsd_img, image)
    os_ = oracle.student_image_forward(sd_frozen, image, 24)
    ol, _ = oracle.LossOracle(['out_cos'])(os_, ot, 'image')
    ol.backward()
rows = []
for n, p in named.items():
    if not p.requires_grad:
        continue
    gr = p.grad.detach().float().cpu().reshape(-1).numpy()
    mo = sd_frozen[n].grad.reshape(-1).numpy()
    step = max(1, gr.size // 256)
    rms = float(g[f'img1.cos.gnorm.{n}']) / np.sqrt(gr.size)
    for kind, sl in (('head', slice(0, 256)), ('spread', slice(0, None, step))):
        ref = g[f'img1.cos.g{kind}.{n}']
        got, om = gr[sl][:256], mo[sl][:256]
        rows.append((rc.rel_l2(got, ref), n, kind, float(np.sqrt((ref ** 2).mean())) / (rms + 1e-30), rc.rel_l2(got, om), rc.rel_l2(om, ref),
                     float(np.linalg.norm(got - ref) / (np.sqrt(len(ref)) * rms + 1e-30))))
rows.sort(reverse=True)
print('rel-L2 vs reference | parameter | sample | sample RMS / tensor RMS | HIP vs matched oracle | matched oracle vs reference | error / tensor RMS')
for r in rows[:14]:
    print('%.4f  %-50s %-6s %.3f  %.4f  %.4f  %.4f' % r)
print('whole-tensor rel-L2 vs matched oracle, worst:', sorted(((rc.rel_l2(p.grad.detach().cpu().numpy(), sd_frozen[n].grad.numpy()), n) for n, p in named.items() if p.requires_grad), reverse=True)[:5])
