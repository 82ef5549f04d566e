#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

    python tools/golden/gen_golden.py            # needs /root/reference; writes tests/golden/

The reference cannot travel to the GPU box, so only numeric fixtures (seeds, inputs, expected outputs) are
committed.  Weights/inputs come from distillclip_amd.synth (deterministic, torch-RNG independent) and are
loaded into the reference modules with load_state_dict.  timm/easydict are absent from the image: the reference's
weight_share_model.py executes against tools/golden/ref_shims (our restatement of the five timm symbols).
"""
import os
import sys
import io
import contextlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
REF = os.environ.get('DCLIP_REFERENCE', '/root/reference')
sys.dont_write_bytecode = True
sys.path[:0] = [REF, os.path.join(HERE, 'ref_shims'), REPO]

import numpy as np   # noqa: E402
import torch         # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from model.component.image_encoder import ImageEncoder                  # noqa: E402
    from model.component.text_encoder import TextEncoder                    # noqa: E402
    from model.component.clip_model import CLIPModel                        # noqa: E402
    from model.component.output import ControlOutput                        # noqa: E402
    from model.component.weight_share_model import RepeatVisionTransformer, RepeatTextTransformer  # noqa: E402
    from model._loss import LossCalculator                                  # noqa: E402
from distillclip_amd import synth                                           # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')
torch.manual_seed(0)
torch.set_num_threads(8)


def T(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build_teacher_image(seed, width, layers, patch, res, out_dim, need_layers=None):
    paras = dict(input_resolution=res, patch_size=patch, width=width, layers=layers, heads=width // 64,
                 output_dim=out_dim, need_layers=need_layers, drop_out=0.)
    m = ImageEncoder(is_student=False, vit_paras=paras)
    m.load_state_dict(T(synth.teacher_image_state(seed, width, layers, patch, res, out_dim)))
    return m.eval()


def build_teacher_text(seed, width, layers, ctx, vocab, out_dim, need_layers=None):
    m = TextEncoder(transformer_width=width, transformer_layers=layers, transformer_heads=width // 64,
                    context_length=ctx, need_layers=need_layers, vocab_size=vocab, embed_dim=out_dim,
                    is_student=False)
    m.load_state_dict(T(synth.teacher_text_state(seed, width, layers, ctx, vocab, out_dim)))
    return m.eval()


def build_student_image(seed, **cfg):
    m = quiet(RepeatVisionTransformer, **cfg)
    m.load_state_dict(T(synth.student_image_state(seed, **cfg)))
    return m.train()       # reference trains the student in train() mode; all dropouts are p=0


def build_student_text(seed, **cfg):
    m = quiet(RepeatTextTransformer, **cfg)
    m.load_state_dict(T(synth.student_text_state(seed, **cfg)))
    return m.train()


def np_(t):
    return t.detach().cpu().numpy()


def grads_of(module, prefix=''):
    return {prefix + 'grad.' + n: np_(p.grad) for n, p in module.named_parameters() if p.grad is not None}


TINY = dict(
    seed=11, B=3, res=32, patch=8, ctx=13, vocab=97, out_dim=64,
    t_img=dict(width=128, layers=2), t_txt=dict(width=128, layers=2),
    s_img=dict(img_size=32, patch_size=8, in_chans=3, out_dim=64, embed_dim=128, depth=4, num_heads=4,
               mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True),
    s_txt=dict(vocab_size=97, context_length=13, out_dim=64, embed_dim=128, depth=2, num_heads=2,
               mlp_ratio=4.0, qkv_bias=False, repeated_times=2, use_transform=True),
)
ALL_LOSSES = ['out_l1', 'out_cos', 'out_kl', 'cos_diff', 'hard_label', 'soft_label', 'logits_mse']


def hook_intermediates(student, store, prefix):
    """Capture per-(block, repeat) hidden states and attention internals through forward hooks/ControlOutput."""
    # hidden states and raw scores / probs come out through the reference's own ControlOutput flags.
    return ControlOutput(need_emb=True, need_attn_score=True, need_attn_prob=True, need_rep=True)


def tiny_dual():
    c = TINY
    seed, B = c['seed'], c['B']
    image = torch.from_numpy(synth.images(seed, B, c['res']))
    text = torch.from_numpy(synth.captions(seed, B, c['ctx'], c['vocab'], 3, 9))
    t_img = build_teacher_image(seed, c['t_img']['width'], c['t_img']['layers'], c['patch'], c['res'], c['out_dim'])
    t_txt = build_teacher_text(seed, c['t_txt']['width'], c['t_txt']['layers'], c['ctx'], c['vocab'], c['out_dim'])
    s_img = build_student_image(seed, **c['s_img'])
    s_txt = build_student_text(seed, **c['s_txt'])
    student = CLIPModel(True, s_img, s_txt, False)
    teacher = CLIPModel(False, t_img, t_txt, False)
    for p in teacher.parameters():
        p.requires_grad = False
    out = {'image': np_(image), 'text': np_(text)}

    # pass 1: intermediates through the reference's ControlOutput
    co = ControlOutput(need_emb=True, need_attn_score=True, need_attn_prob=True, need_rep=True)
    so = student(text, image, co)
    to = teacher(text, image, co)
    for tag, o in (('s_img', so.visual_output), ('s_txt', so.text_output),
                   ('t_img', to.visual_output), ('t_txt', to.text_output)):
        out[f'{tag}.last_representation'] = np_(o.last_representation)
        out[f'{tag}.last_layer_output'] = np_(o.last_layer_output)
        out[f'{tag}.embedding'] = np_(o.embedding)
        for i, r in enumerate(o.representations):
            out[f'{tag}.rep{i}'] = np_(r)
        for i, r in enumerate(o.attention_scores):
            out[f'{tag}.scores{i}'] = np_(r)
        for i, r in enumerate(o.attention_probs):
            out[f'{tag}.probs{i}'] = np_(r)
    out['s.i2t_logits'] = np_(so.i2t_logits)
    out['t.i2t_logits'] = np_(to.i2t_logits)

    # pass 2: training-step semantics (dual_distill_model.py:106-127), every tier-1/2 loss enabled
    lc = quiet(LossCalculator, loss_name=list(ALL_LOSSES), loss_scale={'cos_diff': 0.1, 'soft_label': 0.5},
               temperature=2.0)
    co = lc.get_control_output()
    so = student(text, image, co)
    to = teacher(text, image, co)
    loss, res = lc(so, to, 'all')
    loss.backward()
    out['all.loss'] = np_(loss)
    for k, v in res.items():
        out['all.term.' + k] = np_(v)
    out.update(grads_of(s_img, 'all.s_img.'))
    out.update(grads_of(s_txt, 'all.s_txt.'))
    student.zero_grad()

    # pass 3: the l_clip.yaml loss set (l_clip.yaml:29-32)
    lc = quiet(LossCalculator, loss_name=['out_l1', 'out_cos', 'cos_diff'], loss_scale={'cos_diff': 0.1})
    so = student(text, image, lc.get_control_output())
    to = teacher(text, image, lc.get_control_output())
    loss, res = lc(so, to, 'all')
    loss.backward()
    out['lclip.loss'] = np_(loss)
    for k, v in res.items():
        out['lclip.term.' + k] = np_(v)
    out.update(grads_of(s_img, 'lclip.s_img.'))
    out.update(grads_of(s_txt, 'lclip.s_txt.'))
    student.zero_grad()

    # pass 4: one-tower runs (distil_model.py:81-102) with the image.yaml / text.yaml loss set + feature terms
    lc = quiet(LossCalculator, loss_name=['out_l1', 'out_cos', 'hidden_rep_mse', 'embedding_mse'])
    co = lc.get_control_output()
    t4 = build_teacher_image(seed, c['t_img']['width'], c['t_img']['layers'], c['patch'], c['res'], c['out_dim'],
                             need_layers=[0, 1])
    so = s_img(image, co)
    with torch.no_grad():
        to = t4(image, co)
    # the student emits 4 reps (depth 4) and the teacher 2: the reference zips them (hidden_mse.py:11) -> first 2
    loss, res = lc(so, to, 'image')
    loss.backward()
    out['img1.loss'] = np_(loss)
    for k, v in res.items():
        out['img1.term.' + k] = np_(v)
    out.update(grads_of(s_img, 'img1.s_img.'))
    s_img.zero_grad()

    lc = quiet(LossCalculator, loss_name=['out_l1', 'out_cos'])
    so = s_txt(text, lc.get_control_output())
    with torch.no_grad():
        to = t_txt(text, lc.get_control_output())
    loss, res = lc(so, to, 'text')
    loss.backward()
    out['txt1.loss'] = np_(loss)
    for k, v in res.items():
        out['txt1.term.' + k] = np_(v)
    out.update(grads_of(s_txt, 'txt1.s_txt.'))
    s_txt.zero_grad()

    # compressed-embedding text student (text.yaml:10)
    cfg = dict(c['s_txt'], compression_embedding=True, embedding_compression_dim=64)
    s_txt_c = build_student_text(seed + 1, **cfg)
    so = s_txt_c(text, lc.get_control_output())
    loss, res = lc(so, to, 'text')
    loss.backward()
    out['txtc.loss'] = np_(loss)
    out['txtc.last_representation'] = np_(so.last_representation)
    out.update(grads_of(s_txt_c, 'txtc.s_txt.'))
    np.savez_compressed(os.path.join(OUT, 'tiny.npz'), **out)
    print('tiny.npz', len(out), 'arrays')


def loss_only():
    """LossCalculator + loss_component goldens on random embeddings (incl. eps / edge behaviour)."""
    from types import SimpleNamespace as NS
    out = {}
    for case, (B, D, scale) in {'b8': (8, 64, 1.0), 'b37': (37, 512, 1.0), 'small': (5, 32, 1e-7)}.items():
        e = {k: torch.from_numpy(synth.normal(5, f'{case}.{k}', (B, D), scale)) for k in ('si', 'st', 'ti', 'tt')}
        e['si'].requires_grad_(True)
        e['st'].requires_grad_(True)

        def clip_out(i, t):
            fi = i / i.norm(dim=1, keepdim=True)
            ft = t / t.norm(dim=1, keepdim=True)
            lg = fi @ ft.t()
            return NS(visual_output=NS(last_representation=i), text_output=NS(last_representation=t),
                      i2t_logits=lg, t2i_logits=lg.T)
        names = list(ALL_LOSSES) + ['out_ce']
        lc = quiet(LossCalculator, loss_name=names, loss_scale={'cos_diff': 0.1, 'hard_label': 2.0},
                   temperature=0.5)
        loss, res = lc(clip_out(e['si'], e['st']), clip_out(e['ti'], e['tt']), 'all')
        loss.backward()
        for k, v in e.items():
            out[f'{case}.{k}'] = np_(v)
        out[f'{case}.loss'] = np_(loss)
        for k, v in res.items():
            out[f'{case}.term.{k}'] = np_(v)
        out[f'{case}.grad.si'] = np_(e['si'].grad)
        out[f'{case}.grad.st'] = np_(e['st'].grad)
    np.savez_compressed(os.path.join(OUT, 'loss.npz'), **out)
    print('loss.npz', len(out), 'arrays')


def real_shapes():
    """ViT-B/32 teacher + the shipped l_clip students, B=4: outputs only (SURVEY.md §8c family 2)."""
    seed, B = 2022, 4
    image = torch.from_numpy(synth.images(seed, B, 224))
    text = torch.from_numpy(synth.captions(seed, B))
    t_img = build_teacher_image(seed, 768, 12, 32, 224, 512)
    t_txt = build_teacher_text(seed, 512, 12, 77, 49408, 512)
    s_img_cfg = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24,
                     mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
    s_txt_cfg = dict(depth=4, repeated_times=2, use_transform=True)
    s_img = build_student_image(seed, **s_img_cfg)
    s_txt = build_student_text(seed, **s_txt_cfg)
    student = CLIPModel(True, s_img, s_txt, False)
    teacher = CLIPModel(False, t_img, t_txt, False)
    for p in teacher.parameters():
        p.requires_grad = False
    lc = quiet(LossCalculator, loss_name=['out_l1', 'out_cos', 'cos_diff'], loss_scale={'cos_diff': 0.1})
    so = student(text, image, lc.get_control_output())
    with torch.no_grad():
        to = teacher(text, image, lc.get_control_output())
    loss, res = lc(so, to, 'all')
    loss.backward()
    out = {'seed': np.int64(seed), 'B': np.int64(B), 'loss': np_(loss)}
    for k, v in res.items():
        out['term.' + k] = np_(v)
    out['s_img.last_representation'] = np_(so.visual_output.last_representation)
    out['s_txt.last_representation'] = np_(so.text_output.last_representation)
    out['t_img.last_representation'] = np_(to.visual_output.last_representation)
    out['t_txt.last_representation'] = np_(to.text_output.last_representation)
    out['s.i2t_logits'] = np_(so.i2t_logits)
    out['t.i2t_logits'] = np_(to.i2t_logits)
    # gradient fingerprints: L2 norm per parameter + a small slice of a few
    for tag, m in (('s_img', s_img), ('s_txt', s_txt)):
        for n, p in m.named_parameters():
            g = p.grad
            out[f'{tag}.gnorm.{n}'] = np_(g.norm())
        for n in ('head.weight', 'blocks.0.block.attn.qkv.weight', 'blocks.0.block.attn.conv_l.instances.1.weight',
                  'blocks.1.block.mlp.fc2.weight', 'pos_embed'):
            g = dict(m.named_parameters())[n].grad
            out[f'{tag}.gslice.{n}'] = np_(g.reshape(-1)[:256])
    # second objective: smooth (out_cos only) -> gradients are not subject to the sign / relu discontinuities of
    # out_l1 / cos_diff, so slices can be compared tightly
    student.zero_grad()
    lc2 = quiet(LossCalculator, loss_name=['out_cos'])
    so = student(text, image, lc2.get_control_output())
    loss2, _ = lc2(so, to, 'all')
    loss2.backward()
    out['cos.loss'] = np_(loss2)
    for tag, m in (('s_img', s_img), ('s_txt', s_txt)):
        for n, p in m.named_parameters():
            out[f'cos.{tag}.gnorm.{n}'] = np_(p.grad.norm())
        for n in ('head.weight', 'blocks.0.block.attn.qkv.weight', 'blocks.0.block.attn.conv_l.instances.1.weight',
                  'blocks.1.block.mlp.fc2.weight', 'pos_embed'):
            out[f'cos.{tag}.gslice.{n}'] = np_(dict(m.named_parameters())[n].grad.reshape(-1)[:256])
    np.savez_compressed(os.path.join(OUT, 'real_b4.npz'), **out)
    print('real_b4.npz', len(out), 'arrays', 'loss', float(loss))


def real_shapes_cos_slices():
    """Round 4: gradient slices of EVERY student parameter at real shapes (ViT-B/32 teacher + the shipped l_clip students, B = 4, the
    smooth out_cos objective, as in real_shapes()): the first 256 elements and 256 elements spread evenly over the tensor.  A separate
    file, so that real_b4.npz stays byte-identical."""
    seed, B = 2022, 4
    image = torch.from_numpy(synth.images(seed, B, 224))
    text = torch.from_numpy(synth.captions(seed, B))
    t_img = build_teacher_image(seed, 768, 12, 32, 224, 512)
    t_txt = build_teacher_text(seed, 512, 12, 77, 49408, 512)
    s_img_cfg = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24,
                     mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
    s_txt_cfg = dict(depth=4, repeated_times=2, use_transform=True)
    s_img = build_student_image(seed, **s_img_cfg)
    s_txt = build_student_text(seed, **s_txt_cfg)
    student = CLIPModel(True, s_img, s_txt, False)
    teacher = CLIPModel(False, t_img, t_txt, False)
    for p in teacher.parameters():
        p.requires_grad = False
    lc2 = quiet(LossCalculator, loss_name=['out_cos'])
    so = student(text, image, lc2.get_control_output())
    with torch.no_grad():
        to = teacher(text, image, lc2.get_control_output())
    loss2, _ = lc2(so, to, 'all')
    loss2.backward()
    out = {'seed': np.int64(seed), 'B': np.int64(B), 'cos.loss': np_(loss2)}
    for tag, m in (('s_img', s_img), ('s_txt', s_txt)):
        for n, p in m.named_parameters():
            g = p.grad.reshape(-1)
            out[f'cos.{tag}.gnorm.{n}'] = np_(g.norm())
            out[f'cos.{tag}.ghead.{n}'] = np_(g[:256])
            step = max(1, g.numel() // 256)
            out[f'cos.{tag}.gspread.{n}'] = np_(g[::step][:256])
    np.savez_compressed(os.path.join(OUT, 'real_b4_cos.npz'), **out)
    print('real_b4_cos.npz', len(out), 'arrays', 'loss', float(loss2))


def _train(student, teacher, s_img, s_txt, batches, lr, wd, warm, total, steps_per_epoch, tag, out):
    """The reference's training loop arithmetic without the Lightning shell: dual_distill_model.py:120-127 (training_step:
    student fwd, teacher fwd, LossCalculator) + what Lightning's automatic optimisation does around it (zero_grad, backward,
    optimizer.step) + :194-202 (AdamW over filter(requires_grad, parameters()), HF cosine-with-warmup stepped per EPOCH)."""
    from transformers.optimization import get_cosine_schedule_with_warmup
    lc = quiet(LossCalculator, loss_name=['out_l1', 'out_cos', 'cos_diff'], loss_scale={'cos_diff': 0.1})     # l_clip.yaml:29-32
    opt = torch.optim.AdamW(filter(lambda p: p.requires_grad, student.parameters()), lr=lr, weight_decay=wd)
    sched = get_cosine_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total)
    losses, lrs, terms = [], [], {}
    for i, (image, text) in enumerate(batches):
        so = student(text, image, lc.get_control_output())
        to = teacher(text, image, lc.get_control_output())
        loss, res = lc(so, to, 'all')
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
        lrs.append(opt.param_groups[0]['lr'])
        for k, v in res.items():
            terms.setdefault(k, []).append(float(v))
        if (i + 1) % steps_per_epoch == 0:
            sched.step()
    out[f'{tag}.loss'] = np.asarray(losses, dtype=np.float64)
    out[f'{tag}.lr'] = np.asarray(lrs, dtype=np.float64)
    for k, v in terms.items():
        out[f'{tag}.term.{k}'] = np.asarray(v, dtype=np.float64)
    out[f'{tag}.s_img.last_representation'] = np_(so.visual_output.last_representation)
    out[f'{tag}.s_txt.last_representation'] = np_(so.text_output.last_representation)
    return opt


def trajectory():
    """the l_clip plumbing run — B = 4, 16 steps over 64 synthetic pairs — as a loss TRAJECTORY of the
    reference with its optimizer and per-epoch schedule (tiny config: all 16 steps = 4 epochs of 4; real shapes: 4 steps).
    Pins forward + backward + AdamW + schedule end to end.  Inputs are synth.images / synth.captions of the stored seed."""
    out = {}
    # ---- tiny: 16 steps, 4 epochs x 4 batches, warm-up 2 epochs (epoch 0 runs at lr 0: HF's multiplier is 0 / warm) -------
    c = TINY
    seed, B, n = 31, 4, 64
    images = torch.from_numpy(synth.images(seed, n, c['res']))
    texts = torch.from_numpy(synth.captions(seed, n, c['ctx'], c['vocab'], 3, 9))
    t_img = build_teacher_image(seed, c['t_img']['width'], c['t_img']['layers'], c['patch'], c['res'], c['out_dim'])
    t_txt = build_teacher_text(seed, c['t_txt']['width'], c['t_txt']['layers'], c['ctx'], c['vocab'], c['out_dim'])
    s_img, s_txt = build_student_image(seed, **c['s_img']), build_student_text(seed, **c['s_txt'])
    student, teacher = CLIPModel(True, s_img, s_txt, False), CLIPModel(False, t_img, t_txt, False)
    for p in teacher.parameters():
        p.requires_grad = False
    batches = [(images[i:i + B], texts[i:i + B]) for i in range(0, n, B)]
    out.update({'tiny.seed': np.int64(seed), 'tiny.B': np.int64(B), 'tiny.steps': np.int64(len(batches)), 'tiny.lr': np.float64(2e-3),
                'tiny.wd': np.float64(1e-2), 'tiny.warm': np.int64(2), 'tiny.total': np.int64(8), 'tiny.steps_per_epoch': np.int64(4)})
    base_lr = 2e-3
    opt = _train(student, teacher, s_img, s_txt, batches, base_lr, 1e-2, 2, 8, 4, 'tiny', out)
    out['tiny.base_lr'] = np.float64(base_lr)
    for tag, m in (('s_img', s_img), ('s_txt', s_txt)):
        for name, p in m.named_parameters():
            out[f'tiny.{tag}.final.{name}'] = np_(p)
    # ---- real shapes: ViT-B/32 teacher + the shipped l_clip students, 4 steps at B = 4, l_clip.yaml's lr / wd, constant lr --------
    seed, B, n = 2023, 4, 16
    images = torch.from_numpy(synth.images(seed, n, 224))
    texts = torch.from_numpy(synth.captions(seed, n))
    t_img, t_txt = build_teacher_image(seed, 768, 12, 32, 224, 512), build_teacher_text(seed, 512, 12, 77, 49408, 512)
    s_img_cfg = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24,
                     mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
    s_txt_cfg = dict(depth=4, repeated_times=2, use_transform=True)
    s_img, s_txt = build_student_image(seed, **s_img_cfg), build_student_text(seed, **s_txt_cfg)
    before = {('s_img', k): v.detach().clone() for k, v in s_img.named_parameters()}
    before.update({('s_txt', k): v.detach().clone() for k, v in s_txt.named_parameters()})
    student, teacher = CLIPModel(True, s_img, s_txt, False), CLIPModel(False, t_img, t_txt, False)
    for p in teacher.parameters():
        p.requires_grad = False
    batches = [(images[i:i + B], texts[i:i + B]) for i in range(0, n, B)]
    out.update({'real.seed': np.int64(seed), 'real.B': np.int64(B), 'real.steps': np.int64(len(batches)),
                'real.base_lr': np.float64(1e-3), 'real.wd': np.float64(1e-3)})
    _train(student, teacher, s_img, s_txt, batches, 1e-3, 1e-3, 0, 300, 10 ** 9, 'real', out)
    for tag, m in (('s_img', s_img), ('s_txt', s_txt)):
        for name, p in m.named_parameters():
            d = p.detach() - before[(tag, name)]
            out[f'real.{tag}.dnorm.{name}'] = np_(d.norm())            # how far 4 AdamW steps moved every parameter
        for name in ('head.weight', 'blocks.0.block.attn.qkv.weight', 'blocks.1.block.mlp.fc2.weight', 'norm.weight'):
            out[f'real.{tag}.final_slice.{name}'] = np_(dict(m.named_parameters())[name].reshape(-1)[:256])
    np.savez_compressed(os.path.join(OUT, 'trajectory.npz'), **out)
    print('trajectory.npz', len(out), 'arrays; tiny losses', np.round(out['tiny.loss'], 5), 'real losses', np.round(out['real.loss'], 5))


# ---- round 5: the remaining BASELINE.json configurations at their real shapes, pinned to the reference itself -------------------------
S_IMG_REAL = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24,
                  mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)


def _every_gradient(out, prefix, module):
    """norm, first 256 elements and 256 evenly spread elements of the gradient of every parameter that has one"""
    for n, p in module.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.reshape(-1)
        out[f'{prefix}.gnorm.{n}'] = np_(g.norm())
        out[f'{prefix}.ghead.{n}'] = np_(g[:256])
        out[f'{prefix}.gspread.{n}'] = np_(g[::max(1, g.numel() // 256)][:256])


def _one_tower(student, teacher, x, model_type, out, tag):
    """the arithmetic of DistillModel.forward / training_step (reference distil_model.py:82-103) for the shipped one-tower loss set
    ['out_l1', 'out_cos'] (image.yaml:26, text.yaml:13), then the smooth objective (out_cos alone) for the gradient comparison"""
    lc = quiet(LossCalculator, loss_name=['out_l1', 'out_cos'])
    ctrl = lc.get_control_output()
    so = student(x, ctrl)
    with torch.no_grad():
        to = teacher(x, ctrl)
    loss, res = lc(so, to, model_type)
    loss.backward()
    out[f'{tag}.loss'] = np_(loss)
    for k, v in res.items():
        out[f'{tag}.term.{k}'] = np_(v)
    out[f'{tag}.s.last_representation'] = np_(so.last_representation)
    out[f'{tag}.t.last_representation'] = np_(to.last_representation)
    _every_gradient(out, f'{tag}.l1cos', student)
    student.zero_grad()
    lc2 = quiet(LossCalculator, loss_name=['out_cos'])
    so = student(x, lc2.get_control_output())
    loss2, _ = lc2(so, to, model_type)
    loss2.backward()
    out[f'{tag}.cos.loss'] = np_(loss2)
    _every_gradient(out, f'{tag}.cos', student)


def real_image1():
    """BASELINE.json configs[1] = config/final_config/image.yaml at its real shapes, B = 4: ViT-B/32 image teacher with
    need_layers [0, 1, 10, 11] (image.yaml:30) -> RepeatVisionTransformer(6 = 3 x 2, 24 heads), one-tower losses out_l1 + out_cos,
    freeze_embed applied as DistillModel.freeze_image_embedding does (distil_model.py:197-213: the teacher's conv1 / class / positional
    embeddings copied into patch_embed.proj.weight / cls_token / pos_embed and those three frozen).  The Lightning shell itself cannot
    be imported here (pytorch_lightning, wandb absent): the key mapping is restated in these six lines, everything numeric is the
    reference's modules."""
    seed, B = 2023, 4
    image = torch.from_numpy(synth.images(seed, B, 224))
    teacher = build_teacher_image(seed, 768, 12, 32, 224, 512, need_layers=[0, 1, 10, 11])
    student = build_student_image(seed, **S_IMG_REAL)
    for p in teacher.parameters():
        p.requires_grad = False
    sw, tw = student.state_dict(), teacher.state_dict()
    stu_keys = ['patch_embed.proj.weight', 'cls_token', 'pos_embed']
    sw['patch_embed.proj.weight'] = tw['visual.conv1.weight']
    sw['cls_token'] = tw['visual.class_embedding'].unsqueeze(0).unsqueeze(0)
    sw['pos_embed'] = tw['visual.positional_embedding'].unsqueeze(0)
    student.load_state_dict(sw)
    for n, p in student.named_parameters():
        if n in stu_keys:
            p.requires_grad = False
    out = {'seed': np.int64(seed), 'B': np.int64(B), 'frozen': np.array(stu_keys)}
    _one_tower(student, teacher, image, 'image', out, 'img1')
    assert not any(k.endswith(tuple('.' + f for f in stu_keys)) and '.ghead.' in k for k in out)
    np.savez_compressed(os.path.join(OUT, 'real_b4_image1.npz'), **out)
    print('real_b4_image1.npz', len(out), 'arrays', 'loss', float(out['img1.loss']))


def real_textc():
    """BASELINE.json configs[2] = config/final_config/text.yaml at its real shapes, B = 4: CLIP text teacher (12 x 512, causal) ->
    RepeatTextTransformer(depth 4 = 2 x 2, compression_embedding=True: text.yaml:10), one-tower losses out_l1 + out_cos."""
    seed, B = 2024, 4
    text = torch.from_numpy(synth.captions(seed, B))
    teacher = build_teacher_text(seed, 512, 12, 77, 49408, 512)
    cfg = dict(depth=4, repeated_times=2, use_transform=True, compression_embedding=True)
    student = build_student_text(seed, **cfg)
    for p in teacher.parameters():
        p.requires_grad = False
    out = {'seed': np.int64(seed), 'B': np.int64(B)}
    _one_tower(student, teacher, text, 'text', out, 'txtc')
    np.savez_compressed(os.path.join(OUT, 'real_b4_textc.npz'), **out)
    print('real_b4_textc.npz', len(out), 'arrays', 'loss', float(out['txtc.loss']))


def real_336():
    """BASELINE.json configs[4]: l_clip dual at 336 px, B = 4.  (336 // 32)^2 + 1 = 101 tokens; no ViT-B/32@336 archive exists, so the
    teacher's [101, 768] positional table is synthetic — built at input_resolution 320 (10 x 10 patches), the 336 px
    images being floored by the stride-32 conv exactly as the reference's conv1 floors them (_common.py:176,196)."""
    seed, B = 2025, 4
    image = torch.from_numpy(synth.images(seed, B, 336))
    text = torch.from_numpy(synth.captions(seed, B))
    paras = dict(input_resolution=336, patch_size=32, width=768, layers=12, heads=12, output_dim=512, need_layers=None, drop_out=0.)
    t_img = ImageEncoder(is_student=False, vit_paras=paras)
    t_img.load_state_dict(T(synth.teacher_image_state(seed, 768, 12, 32, 336, 512)))
    t_img.eval()
    t_txt = build_teacher_text(seed, 512, 12, 77, 49408, 512)
    s_img = build_student_image(seed, **dict(S_IMG_REAL, img_size=336))
    s_txt = build_student_text(seed, depth=4, repeated_times=2, use_transform=True)
    student = CLIPModel(True, s_img, s_txt, False)
    teacher = CLIPModel(False, t_img, t_txt, False)
    for p in teacher.parameters():
        p.requires_grad = False
    lc = quiet(LossCalculator, loss_name=['out_l1', 'out_cos', 'cos_diff'], loss_scale={'cos_diff': 0.1})
    so = student(text, image, lc.get_control_output())
    with torch.no_grad():
        to = teacher(text, image, lc.get_control_output())
    loss, res = lc(so, to, 'all')
    out = {'seed': np.int64(seed), 'B': np.int64(B), 'loss': np_(loss)}
    for k, v in res.items():
        out['term.' + k] = np_(v)
    for tag, o in (('s_img', so.visual_output), ('s_txt', so.text_output), ('t_img', to.visual_output), ('t_txt', to.text_output)):
        out[f'{tag}.last_representation'] = np_(o.last_representation)
    out['s.i2t_logits'], out['t.i2t_logits'] = np_(so.i2t_logits), np_(to.i2t_logits)
    lc2 = quiet(LossCalculator, loss_name=['out_cos'])
    so = student(text, image, lc2.get_control_output())
    loss2, _ = lc2(so, to, 'all')
    loss2.backward()
    out['cos.loss'] = np_(loss2)
    _every_gradient(out, 'cos.s_img', s_img)
    _every_gradient(out, 'cos.s_txt', s_txt)
    np.savez_compressed(os.path.join(OUT, 'real_b4_336.npz'), **out)
    print('real_b4_336.npz', len(out), 'arrays', 'loss', float(loss))



# ---- plain CLIP encoders in the student role (reference image_encoder.py:16-25,54-59 ; text_encoder.py:41-47,75-80) ------------------------
def _clip_students(seed, width, layers, patch, res, ctx, vocab, out_dim, tea_width):
    paras = dict(input_resolution=res, patch_size=patch, width=width, layers=layers, heads=width // 64, output_dim=out_dim,
                 need_layers=None, drop_out=0.)
    s_img = ImageEncoder(is_student=True, vit_paras=paras, tea_transformer_width=tea_width['img'])
    s_txt = TextEncoder(transformer_width=width, transformer_layers=layers, transformer_heads=width // 64, context_length=ctx,
                        need_layers=None, vocab_size=vocab, embed_dim=out_dim, tea_transformer_width=tea_width['txt'], is_student=True)
    sd_i, sd_t = synth.clip_student_states(seed, width, layers, patch, res, ctx, vocab, out_dim, tea_width['img'], tea_width['txt'])
    s_img.load_state_dict(T(sd_i))
    s_txt.load_state_dict(T(sd_t))
    return s_img.train(), s_txt.train()


def _clip_student_case(out, s_img, s_txt, t_img, t_txt, image, text, full):
    student, teacher = CLIPModel(True, s_img, s_txt, False), CLIPModel(False, t_img, t_txt, False)
    for p in teacher.parameters():
        p.requires_grad = False
    names = ['out_l1', 'out_cos', 'cos_diff', 'hidden_rep_mse', 'embedding_mse']
    lc = quiet(LossCalculator, loss_name=names, loss_scale={'cos_diff': 0.1})
    so = student(text, image, lc.get_control_output())
    with torch.no_grad():
        to = teacher(text, image, lc.get_control_output())
    loss, res = lc(so, to, 'all')
    out['loss'] = np_(loss)
    for k, v in res.items():
        out['term.' + k] = np_(v)
    for tag, o in (('s_img', so.visual_output), ('s_txt', so.text_output), ('t_img', to.visual_output), ('t_txt', to.text_output)):
        out[f'{tag}.last_representation'] = np_(o.last_representation)
        if full:
            out[f'{tag}.embedding'] = np_(o.embedding)
            for i, r in enumerate(o.representations):
                out[f'{tag}.rep{i}'] = np_(r)
    # the smooth objective for the gradient comparison: cosine + both feature-MSE terms (every projection linear gets a gradient)
    lc2 = quiet(LossCalculator, loss_name=['out_cos', 'hidden_rep_mse', 'embedding_mse'])
    so = student(text, image, lc2.get_control_output())
    loss2, res2 = lc2(so, to, 'all')
    loss2.backward()
    out['smooth.loss'] = np_(loss2)
    for k, v in res2.items():
        out['smooth.term.' + k] = np_(v)
    if full:
        out.update(grads_of(s_img, 'smooth.s_img.'))
        out.update(grads_of(s_txt, 'smooth.s_txt.'))
    else:
        _every_gradient(out, 'smooth.s_img', s_img)
        _every_gradient(out, 'smooth.s_txt', s_txt)


def clip_student_tiny():
    """ImageEncoder / TextEncoder with is_student=True (2 x 128, 2 heads) under a 2 x 192 teacher pair, B = 3: outputs, projected hidden
    states and embeddings, every loss term, the gradient of every parameter (whole tensors)."""
    seed, B, res, patch, ctx, vocab, E = 31, 3, 32, 8, 13, 97, 64
    image = torch.from_numpy(synth.images(seed, B, res))
    text = torch.from_numpy(synth.captions(seed, B, ctx, vocab, 3, 9))
    t_img = build_teacher_image(seed, 192, 2, patch, res, E)
    t_txt = build_teacher_text(seed, 192, 2, ctx, vocab, E)
    s_img, s_txt = _clip_students(seed + 1, 128, 2, patch, res, ctx, vocab, E, {'img': 192, 'txt': 192})
    out = {'seed': np.int64(seed), 'B': np.int64(B), 'image': np_(image), 'text': np_(text)}
    _clip_student_case(out, s_img, s_txt, t_img, t_txt, image, text, full=True)
    np.savez_compressed(os.path.join(OUT, 'clip_student_tiny.npz'), **out)
    print('clip_student_tiny.npz', len(out), 'arrays', 'loss', float(out['loss']))


def clip_student_real():
    """the same pair at real shapes, B = 4: 4 x 512 (8 heads) CLIP students, patch 32 at 224 px / 77 tokens, under the ViT-B/32 teacher
    pair with need_layers [2, 5, 8, 11]; gradient samples of every parameter."""
    seed, B = 2026, 4
    image = torch.from_numpy(synth.images(seed, B, 224))
    text = torch.from_numpy(synth.captions(seed, B))
    t_img = build_teacher_image(seed, 768, 12, 32, 224, 512, need_layers=[2, 5, 8, 11])
    t_txt = build_teacher_text(seed, 512, 12, 77, 49408, 512, need_layers=[2, 5, 8, 11])
    s_img, s_txt = _clip_students(seed + 1, 512, 4, 32, 224, 77, 49408, 512, {'img': 768, 'txt': 512})
    out = {'seed': np.int64(seed), 'B': np.int64(B)}
    _clip_student_case(out, s_img, s_txt, t_img, t_txt, image, text, full=False)
    np.savez_compressed(os.path.join(OUT, 'real_b4_clipstu.npz'), **out)
    print('real_b4_clipstu.npz', len(out), 'arrays', 'loss', float(out['loss']))

if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['tiny', 'loss', 'real', 'trajectory']
    if 'trajectory' in which:
        trajectory()
    if 'tiny' in which:
        tiny_dual()
    if 'loss' in which:
        loss_only()
    if 'real' in which:
        real_shapes()
    if 'real_cos' in which:
        real_shapes_cos_slices()
    if 'real_image1' in which:
        real_image1()
    if 'real_textc' in which:
        real_textc()
    if 'real_336' in which:
        real_336()
    if 'clip_student' in which:
        clip_student_tiny()
        clip_student_real()
