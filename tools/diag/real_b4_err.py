import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from distillclip_amd import synth
from tests.test_towers_gpu import T, rel_l2
from distillclip_amd.model import LossCalculator
from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer, ImageEncoder, TextEncoder, CLIPModel
g = dict(np.load(os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden', 'real_b4.npz')))
seed, B = int(g['seed']), int(g['B'])
s_img_cfg = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24, mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
s_txt_cfg = dict(depth=4, repeated_times=2, use_transform=True)
s_img = RepeatVisionTransformer(**s_img_cfg); s_img.load_state_dict(T(synth.student_image_state(seed, **s_img_cfg)))
s_txt = RepeatTextTransformer(**s_txt_cfg); s_txt.load_state_dict(T(synth.student_text_state(seed, **s_txt_cfg)))
t_img = ImageEncoder(False, dict(input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512)); t_img.load_state_dict(T(synth.teacher_image_state(seed)))
t_txt = TextEncoder(512, 12, 8, 77, None, 49408, 512, is_student=False); t_txt.load_state_dict(T(synth.teacher_text_state(seed)))
student, teacher = CLIPModel(True, s_img.cuda(), s_txt.cuda()), CLIPModel(False, t_img.cuda(), t_txt.cuda())
image = torch.from_numpy(synth.images(seed, B, 224)).cuda(); text = torch.from_numpy(synth.captions(seed, B)).cuda()
lc = LossCalculator(['out_l1', 'out_cos', 'cos_diff'], {'cos_diff': 0.1})
so, to = student(text, image), teacher(text, image)
for tag, o in (('s_img', so.visual_output), ('s_txt', so.text_output), ('t_img', to.visual_output), ('t_txt', to.text_output)):
    print(tag, 'emb rel_l2', rel_l2(o.last_representation, g[f'{tag}.last_representation']))
loss, res = lc(so, to, 'all'); print('loss', loss.item(), float(g['loss']))
loss.backward()
for tag, m in (('s_img', s_img), ('s_txt', s_txt)):
    errs = {}
    for n, p in m.named_parameters():
        ref = float(g[f'{tag}.gnorm.{n}'])
        if ref > 0: errs[n] = abs(p.grad.norm().item() - ref) / ref
    w = sorted(errs.items(), key=lambda x: -x[1])[:4]
    print(tag, 'worst gnorm', [(k, round(v, 4)) for k, v in w])
    for key in [k for k in g if k.startswith(f'{tag}.gslice.')]:
        n = key[len(tag) + 8:]
        print('   slice', n, round(rel_l2(dict(m.named_parameters())[n].grad.reshape(-1)[:256], g[key]), 4))
