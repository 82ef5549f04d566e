import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from distillclip_amd import synth
from tests.test_towers_gpu import TINY, T, _tiny_modules, rel_l2
from distillclip_amd.model import LossCalculator
from distillclip_amd.model.component import RepeatTextTransformer

c = TINY
B = 3
text = torch.from_numpy(synth.captions(c['seed'], B, c['ctx'], c['vocab'], 3, 9))
for comp in (True, False):
  for seed in (c['seed'] + 1, c['seed']):
    for names in (['out_cos'], ['out_l1', 'out_cos']):
        _, _, _, t_txt = _tiny_modules()
        cfg = dict(c['s_txt'], compression_embedding=comp, embedding_compression_dim=64)
        s = RepeatTextTransformer(**cfg)
        s.load_state_dict(T(synth.student_text_state(seed, **cfg)))
        s = s.cuda()
        lc = LossCalculator(names)
        so = s(text.cuda())
        loss, _ = lc(so, t_txt(text.cuda()), 'text')
        loss.backward()
        sd = {k: v.clone().requires_grad_(True) for k, v in T(synth.student_text_state(seed, **cfg)).items()}
        with torch.no_grad():
            tt = oracle.teacher_text_forward(T(synth.teacher_text_state(c['seed'], 128, 2, c['ctx'], c['vocab'], c['out_dim'])), text)
        cap = {}
        ot = oracle.student_text_forward(sd, text, 2, cap=cap)
        ol, _ = oracle.LossOracle(names)(ot, tt, 'text')
        ol.backward()
        print('==== comp', comp, 'seed', seed, names, 'loss', loss.item(), ol.item(), 'emb err', rel_l2(so.last_representation, ot['last_representation']))
        print('   oracle |x_out| per block', [cap[k].norm().item() for k in cap if k.endswith('.out')], 'max prob', max(cap[k].max().item() for k in cap if k.endswith('.probs')))
        for n, p in s.named_parameters():
            if sd[n].grad is not None and sd[n].grad.abs().max() > 0 and ('weight' in n and 'norm' not in n or 'pos' in n):
                print(f'   {n:50s} {rel_l2(p.grad, sd[n].grad):.4f}  |g|={sd[n].grad.norm().item():.3e}')
