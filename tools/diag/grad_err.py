"""diagnostic: per-parameter gradient error of the HIP towers vs the oracle for several loss sets (tiny config)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from distillclip_amd import synth
from tests.test_towers_gpu import TINY, T, _tiny_modules, rel_l2
from distillclip_amd.model import LossCalculator
from distillclip_amd.model.component import CLIPModel

c = TINY
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
image = torch.from_numpy(synth.images(c['seed'], B, c['res']))
text = torch.from_numpy(synth.captions(c['seed'], B, c['ctx'], c['vocab'], 3, 9))
for names, tau in ((['out_cos'], None), (['out_l1'], None), (['out_kl'], 1.5), (['out_cos', 'soft_label'], 1.5)):
    s_img, s_txt, t_img, t_txt = _tiny_modules()
    student, teacher = CLIPModel(True, s_img, s_txt), CLIPModel(False, t_img, t_txt)
    lc = LossCalculator(names, temperature=tau)
    so = student(text.cuda(), image.cuda())
    loss, _ = lc(so, teacher(text.cuda(), image.cuda()), 'all')
    loss.backward()
    sd_i = {k: v.clone().requires_grad_(True) for k, v in T(synth.student_image_state(c['seed'], **c['s_img'])).items()}
    sd_t = {k: v.clone().requires_grad_(True) for k, v in T(synth.student_text_state(c['seed'], **c['s_txt'])).items()}
    with torch.no_grad():
        ti = oracle.teacher_image_forward(T(synth.teacher_image_state(c['seed'], 128, 2, c['patch'], c['res'], c['out_dim'])), image)
        tt = oracle.teacher_text_forward(T(synth.teacher_text_state(c['seed'], 128, 2, c['ctx'], c['vocab'], c['out_dim'])), text)
    oi, ot = oracle.student_image_forward(sd_i, image, 4), oracle.student_text_forward(sd_t, text, 2)
    ol, _ = oracle.LossOracle(names, temperature=tau)(oracle.clip_forward(oi, ot), oracle.clip_forward(ti, tt), 'all')
    ol.backward()
    print('====', names, 'B', B, 'loss', loss.item(), ol.item(), 'emb err', rel_l2(so.visual_output.last_representation, oi['last_representation']),
          rel_l2(so.text_output.last_representation, ot['last_representation']))
    for tag, mod, sd in (('img', s_img, sd_i), ('txt', s_txt, sd_t)):
        for n, p in mod.named_parameters():
            if sd[n].grad is not None and sd[n].grad.abs().max() > 0:
                print(f'  {tag} {n:50s} {rel_l2(p.grad, sd[n].grad):.4f}  |g|={sd[n].grad.norm().item():.3e}')
