"""Plain CLIP encoders in the STUDENT role on the HIP path (tower kind 2, include/dclip.h): reference model/component/image_encoder.py:16-25,
54-59 and text_encoder.py:41-47,75-80 — ImageEncoder / TextEncoder with is_student=True, trained under a CLIP teacher pair, with their
embedding_projection / hidden_projection linears on the exported hidden states.

Held to (a) the reference's own runs (tests/golden/clip_student_tiny.npz, real_b4_clipstu.npz: tools/golden/gen_golden.py clip_student),
(b) the rounding-matched oracle (oracle.bf16_matched: bf16 where the HIP path stores bf16, the saved QuickGELU derivative as 8-bit fixed point).
Tolerances: embeddings / exported states rel-L2 <= 2e-2 against the fp32 reference, loss terms 2e-2, parameter gradients <= 3e-2 (5e-2 for
bias / norm vectors) against the reference (measured: <= 1.2e-2 tiny, <= 1.0e-2 at real shapes) and <= 1.5e-2 against the matched oracle
(measured: <= 6.2e-3 tiny, <= 8.9e-3 at real shapes)."""
import numpy as np
import pytest
import torch

import oracle
import real_cases as rc
from distillclip_amd import synth

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.detach().float().cpu().reshape(-1), torch.as_tensor(b).detach().float().cpu().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _modules(c, tsd_i, tsd_t, sd_i, sd_t):
    from distillclip_amd.model.component import ImageEncoder, TextEncoder, CLIPModel
    tw_i, tw_t = tsd_i['visual.conv1.weight'].shape[0], tsd_t['positional_embedding'].shape[1]
    tea_heads_i, tea_heads_t = c['tea_heads'] or tw_i // 64, c['tea_heads'] or tw_t // 64
    t_img = ImageEncoder(False, dict(input_resolution=c['res'], patch_size=c['patch'], width=tw_i, layers=c['tea_layers'], heads=tea_heads_i,
                                     output_dim=c['out_dim'], need_layers=c['need_layers']))
    t_txt = TextEncoder(tw_t, c['tea_layers'], tea_heads_t, c['ctx'], c['need_layers'], c['vocab'], c['out_dim'], is_student=False)
    s_img = ImageEncoder(True, dict(input_resolution=c['res'], patch_size=c['patch'], width=c['width'], layers=c['layers'], heads=c['heads'],
                                    output_dim=c['out_dim'], need_layers=None), tea_transformer_width=tw_i)
    s_txt = TextEncoder(c['width'], c['layers'], c['heads'], c['ctx'], None, c['vocab'], c['out_dim'], tea_transformer_width=tw_t, is_student=True)
    for m, sd in ((t_img, tsd_i), (t_txt, tsd_t), (s_img, sd_i), (s_txt, sd_t)):
        m.load_state_dict(sd)
    student, teacher = CLIPModel(True, s_img.cuda(), s_txt.cuda()), CLIPModel(False, t_img.cuda(), t_txt.cuda())
    for p in teacher.parameters():
        p.requires_grad = False
    return s_img, s_txt, student, teacher


def _case(golden_dir, case):
    c = rc.CLIPSTU_TINY if case == 'tiny' else rc.CLIPSTU_REAL
    g = rc.load(golden_dir, 'clip_student_tiny.npz' if case == 'tiny' else 'real_b4_clipstu.npz')
    return (c, g) + rc.clipstu_inputs(g, c)


def test_state_dict_keys_and_parameter_order_are_the_references(golden_dir):
    c, g, image, text, tsd_i, tsd_t, sd_i, sd_t = _case(golden_dir, 'tiny')
    s_img, s_txt, _, _ = _modules(c, tsd_i, tsd_t, sd_i, sd_t)
    assert set(s_img.state_dict()) == set(sd_i) and set(s_txt.state_dict()) == set(sd_t)
    assert [n for n, _ in s_img.named_parameters()][-4:] == ['embedding_projection.weight', 'embedding_projection.bias',
                                                              'hidden_projection.weight', 'hidden_projection.bias']
    assert s_img._tower.cfg.kind == 2 and s_txt._tower.cfg.kind == 2 and s_txt._tower.cfg.causal == 1
    assert not s_img.no_trans and not s_txt.no_trans


@pytest.mark.parametrize('case', ['tiny', 'real'])
def test_training_step_vs_reference_golden(golden_dir, case):
    from distillclip_amd.model import LossCalculator
    c, g, image, text, tsd_i, tsd_t, sd_i, sd_t = _case(golden_dir, case)
    s_img, s_txt, student, teacher = _modules(c, tsd_i, tsd_t, sd_i, sd_t)
    lc = LossCalculator(rc.CLIPSTU_LOSSES, {'cos_diff': 0.1})
    so = student(text.cuda(), image.cuda(), lc.get_control_output())
    to = teacher(text.cuda(), image.cuda(), lc.get_control_output())
    for tag, o in (('s_img', so.visual_output), ('s_txt', so.text_output), ('t_img', to.visual_output), ('t_txt', to.text_output)):
        assert rel_l2(o.last_representation, g[f'{tag}.last_representation']) < 2e-2, tag
        assert len(o.representations) == c['layers']
        if case == 'tiny':
            assert rel_l2(o.embedding, g[f'{tag}.embedding']) < 2e-2, tag
            for i, r in enumerate(o.representations):
                assert rel_l2(r, g[f'{tag}.rep{i}']) < 2e-2, (tag, i)
    loss, res = lc(so, to, 'all')
    assert abs(loss.item() - float(g['loss'])) <= 2e-2 * abs(float(g['loss']))
    assert set(res) == {k[5:] for k in g if k.startswith('term.')}
    for k, v in res.items():
        r = float(g['term.' + k])
        assert abs(v.item() - r) <= 6e-2 * abs(r) + 1e-3, (k, v.item(), r)
    # gradients: the smooth objective (cosine + both feature-MSE terms: every projection linear takes part)
    lc2 = LossCalculator(rc.CLIPSTU_SMOOTH)
    loss2, _ = lc2(student(text.cuda(), image.cuda(), lc2.get_control_output()), to, 'all')
    assert abs(loss2.item() - float(g['smooth.loss'])) <= 1e-2 * abs(float(g['smooth.loss']))
    loss2.backward()
    for tag, m, n_par in (('s_img', s_img, 5 + 12 * c['layers'] + 3 + 4), ('s_txt', s_txt, 2 + 12 * c['layers'] + 3 + 4)):
        grads = {n: p.grad for n, p in m.named_parameters()}
        assert all(v is not None for v in grads.values()) and len(grads) == n_par
        if case == 'tiny':
            errs = {n: rel_l2(v, g[f'smooth.{tag}.grad.{n}']) for n, v in grads.items()}
            norms = {}
        else:
            errs, norms, n = rc.gradient_errors(g, f'smooth.{tag}', grads)
            assert n == len(grads)
        print(case, tag, 'worst gradient errors vs the reference', sorted(errs.items(), key=lambda kv: -kv[1])[:5])
        bad = {k: v for k, v in errs.items() if v > (5e-2 if ('bias' in k or 'ln_' in k or k.endswith('embedding')) else 3e-2)}
        assert not bad, (tag, bad)
        assert not norms or max(norms.values()) < 2e-2, sorted(norms.items(), key=lambda kv: -kv[1])[:4]


@pytest.mark.parametrize('case', ['tiny', 'real'])
def test_backward_vs_rounding_matched_oracle(golden_dir, case):
    from distillclip_amd.model import LossCalculator
    c, g, image, text, tsd_i, tsd_t, sd_i, sd_t = _case(golden_dir, case)
    s_img, s_txt, student, teacher = _modules(c, tsd_i, tsd_t, sd_i, sd_t)
    names = ['out_cos', 'out_kl', 'hidden_rep_mse', 'embedding_mse']
    lc = LossCalculator(names, temperature=1.5)
    so = student(text.cuda(), image.cuda(), lc.get_control_output())
    to = teacher(text.cuda(), image.cuda(), lc.get_control_output())
    loss, _ = lc(so, to, 'all')
    loss.backward()
    oi_sd = {k: v.clone().requires_grad_(True) for k, v in sd_i.items()}
    ot_sd = {k: v.clone().requires_grad_(True) for k, v in sd_t.items()}
    with oracle.bf16_matched():
        with torch.no_grad():
            ti = oracle.teacher_image_forward(tsd_i, image, c['tea_heads'], c['need_layers'], True, True)
            tt = oracle.teacher_text_forward(tsd_t, text, c['tea_heads'], c['need_layers'], True, True)
        oi = oracle.clip_student_image_forward(oi_sd, image, c['heads'], True, True)
        ot = oracle.clip_student_text_forward(ot_sd, text, c['heads'], True, True)
        ref, _ = oracle.LossOracle(names, temperature=1.5)(oracle.clip_forward(oi, ot), oracle.clip_forward(ti, tt), 'all')
        ref.backward()
    assert abs(loss.item() - ref.item()) <= 5e-3 * abs(ref.item()), (loss.item(), ref.item())
    assert rel_l2(so.visual_output.last_representation, oi['last_representation']) < 5e-3
    assert rel_l2(so.text_output.last_representation, ot['last_representation']) < 5e-3
    errs = {}
    for tag, m, sd in (('s_img', s_img, oi_sd), ('s_txt', s_txt, ot_sd)):
        for n, p in m.named_parameters():
            w = sd[n].grad
            if n == 'token_embedding.weight':        # rows of tokens that do not occur are exactly zero on both sides
                assert torch.equal(p.grad.cpu() != 0, w != 0) or rel_l2(p.grad, w) < 1.5e-2
            errs[f'{tag}.{n}'] = rel_l2(p.grad, w)
    print(case, 'worst gradient errors vs the matched oracle', sorted(errs.items(), key=lambda kv: -kv[1])[:6])
    bad = {k: v for k, v in errs.items() if v > 1.5e-2}
    assert not bad, bad


def test_fused_adamw_updates_the_projection_linears_like_torch_adamw(golden_dir):
    """the four projection tensors live outside the towers' flat buffers (FusedAdamW extra_params): same update as torch.optim.AdamW from the
    same gradients, state_dict in torch's layout with them included, and a reload continues identically"""
    from distillclip_amd.model import LossCalculator
    from distillclip_amd.optim import FusedAdamW
    c, g, image, text, tsd_i, tsd_t, sd_i, sd_t = _case(golden_dir, 'tiny')
    s_img, s_txt, student, teacher = _modules(c, tsd_i, tsd_t, sd_i, sd_t)
    lc = LossCalculator(rc.CLIPSTU_SMOOTH)
    towers = [s_img._tower, s_txt._tower]
    for tw in towers:
        tw.materialize(torch.device('cuda', torch.cuda.current_device()))
    extras = s_img.extra_parameters() + s_txt.extra_parameters()
    assert len(extras) == 8
    opt = FusedAdamW(towers, lr=2e-3, weight_decay=1e-2, extra_params=extras)
    shadow = [p.detach().clone().requires_grad_(True) for p in student.parameters()]
    ref_opt = torch.optim.AdamW(shadow, lr=2e-3, weight_decay=1e-2)
    to = teacher(text.cuda(), image.cuda(), lc.get_control_output())
    for step in range(3):
        opt.zero_grad()
        loss, _ = lc(student(text.cuda(), image.cuda(), lc.get_control_output()), to, 'all')
        loss.backward()
        for q, p in zip(shadow, student.parameters()):
            q.grad = p.grad.detach().clone()
        opt.step()
        ref_opt.step()
        for (n, p), q in zip(student.named_parameters(), shadow):
            assert rel_l2(p, q) < 1e-5, (step, n)
            q.data.copy_(p.data)                      # (keep the trajectories together: this test is about one step's arithmetic)
    sd = opt.state_dict(params=list(student.parameters()))
    assert len(sd['param_groups'][0]['params']) == len(shadow) == len(sd['state'])
    ref_sd = ref_opt.state_dict()
    for i in range(len(shadow)):
        assert rel_l2(sd['state'][i]['exp_avg'], ref_sd['state'][i]['exp_avg']) < 1e-4, i
        assert tuple(sd['state'][i]['exp_avg_sq'].shape) == tuple(shadow[i].shape)
    opt2 = FusedAdamW(towers, lr=1.0, extra_params=extras)
    opt2.load_state_dict(sd, params=list(student.parameters()))
    assert opt2.step_count == 3 and opt2.lr == 2e-3
    for p in extras:
        assert torch.equal(opt2._extra_state[id(p)][0], opt._extra_state[id(p)][0])


def test_one_tower_image_student_with_frozen_embedding(golden_dir):
    """DistillModel with an ImageEncoder student of the teacher's width and freeze_embed (reference distil_model.py:211-219): the teacher's
    conv1 / class / positional embeddings are copied in and frozen, no_trans switches the projections off, one training step matches the
    oracle and leaves the frozen tensors untouched."""
    from distillclip_amd.model import DistillModel
    from distillclip_amd.model.component import ImageEncoder
    c = rc.CLIPSTU_TINY
    seed, B = 41, 5
    tsd = rc.T(synth.teacher_image_state(seed, 128, 2, c['patch'], c['res'], c['out_dim']))
    sd_i, _ = synth.clip_student_states(seed + 1, 128, 2, c['patch'], c['res'], c['ctx'], c['vocab'], c['out_dim'], 128, 128)
    sd_i = rc.T(sd_i)
    stu = ImageEncoder(True, dict(input_resolution=c['res'], patch_size=c['patch'], width=128, layers=2, heads=2, output_dim=c['out_dim']), 128)
    stu.load_state_dict(sd_i)
    m = DistillModel(stu, dict(loss_name=['out_l1', 'out_cos', 'hidden_rep_mse']), None, teacher_name='synthetic', freeze_embed=True,
                     model_type='image', lr=1e-3, teacher_state_dict=tsd).cuda()
    frozen = ['visual.conv1.weight', 'visual.class_embedding', 'visual.positional_embedding']
    assert stu.no_trans
    for k in frozen:
        assert torch.equal(stu.state_dict()[k].cpu(), tsd[k]) and not dict(stu.named_parameters())[k].requires_grad
    (opt,), _ = m.configure_optimizers()
    opt.lr = 1e-3                                   # (the schedule's warm-up starts the run at lr = 0)
    image = torch.from_numpy(synth.images(seed, B, c['res']))
    before = {k: v.clone() for k, v in stu.state_dict().items()}
    opt.zero_grad()
    loss = m.training_step(image.cuda())
    loss.backward()
    o_sd = {k: v.clone().requires_grad_(k not in frozen) for k, v in sd_i.items()}
    for k in frozen:
        o_sd[k] = tsd[k].clone()
    with torch.no_grad():
        to = oracle.teacher_image_forward(tsd, image, 2, None, True, False)
    so = oracle.clip_student_image_forward(o_sd, image, 2, True, False, no_trans=True)
    ref, _ = oracle.LossOracle(['out_l1', 'out_cos', 'hidden_rep_mse'])(so, to, 'image')
    assert abs(loss.item() - ref.item()) <= 2e-2 * abs(ref.item()), (loss.item(), ref.item())
    params = dict(stu.named_parameters())
    assert all(params[k].grad is None for k in frozen)
    assert params['hidden_projection.weight'].grad is None          # no_trans: the projections are not part of the graph
    opt.step()
    after = stu.state_dict()
    for k in frozen + ['hidden_projection.weight', 'embedding_projection.bias']:
        assert torch.equal(after[k], before[k]), k
    assert not torch.equal(after['visual.transformer.resblocks.0.attn.in_proj_weight'], before['visual.transformer.resblocks.0.attn.in_proj_weight'])
    assert not torch.equal(after['visual.ln_pre.weight'], before['visual.ln_pre.weight'])


@pytest.mark.parametrize('width,heads,res,patch,ctx,B', [(64, 1, 32, 8, 13, 1), (64, 2, 320, 32, 77, 2), (128, 4, 64, 8, 128, 2), (192, 3, 48, 16, 21, 5), (384, 6, 32, 16, 16, 1)])
def test_edge_shapes_against_the_oracle(width, heads, res, patch, ctx, B):
    """one / three / six heads (the softmax kernels without head mixing take any count), head dim 32 and 64, 5 / 10 / 65 / 101 image tokens, 13 .. 128 causal text tokens, a single sample: forward and every gradient of
    both trainable CLIP towers against the fp32 oracle (bf16 operand noise: <= 3e-2; 101 tokens = the 336 px grid of BASELINE config 5)"""
    from distillclip_amd.model import LossCalculator
    from distillclip_amd.model.component import ImageEncoder, TextEncoder
    seed, vocab, E, layers = 100 + width + ctx, 211, 64, 2
    sd_i, sd_t = synth.clip_student_states(seed, width, layers, patch, res, ctx, vocab, E, width, width)
    sd_i, sd_t = rc.T(sd_i), rc.T(sd_t)
    s_img = ImageEncoder(True, dict(input_resolution=res, patch_size=patch, width=width, layers=layers, heads=heads, output_dim=E), width)
    s_txt = TextEncoder(width, layers, heads, ctx, None, vocab, E, tea_transformer_width=width, is_student=True)
    s_img.load_state_dict(sd_i)
    s_txt.load_state_dict(sd_t)
    s_img, s_txt = s_img.cuda(), s_txt.cuda()
    image = torch.from_numpy(synth.images(seed, B, res))
    text = torch.from_numpy(synth.captions(seed, B, ctx, vocab, 3, ctx - 2))
    target_i, target_t = torch.from_numpy(synth.normal(seed, 'ti', (B, E))), torch.from_numpy(synth.normal(seed, 'tt', (B, E)))
    from distillclip_amd.model.component import ControlOutput
    co = ControlOutput(need_rep=True, need_emb=True)
    oi, ot = s_img(image.cuda(), co), s_txt(text.cuda(), co)
    obj = lambda a, b, ti, tt: ((1 - torch.nn.functional.cosine_similarity(a['last_representation'], ti)).mean()
                                + (1 - torch.nn.functional.cosine_similarity(b['last_representation'], tt)).mean()
                                + sum(r.pow(2).mean() for r in a['representations']) + b['embedding'].pow(2).mean()
                                + sum(r.pow(2).mean() for r in b['representations']) + a['embedding'].pow(2).mean())
    as_dict = lambda o: dict(last_representation=o.last_representation, representations=o.representations, embedding=o.embedding)
    loss = obj(as_dict(oi), as_dict(ot), target_i.cuda(), target_t.cuda())
    loss.backward()
    o_i = {k: v.clone().requires_grad_(True) for k, v in sd_i.items()}
    o_t = {k: v.clone().requires_grad_(True) for k, v in sd_t.items()}
    ri = oracle.clip_student_image_forward(o_i, image, heads, True, True, no_trans=s_img.no_trans)
    rt = oracle.clip_student_text_forward(o_t, text, heads, True, True, no_trans=s_txt.no_trans)
    ref = obj(ri, rt, target_i, target_t)
    ref.backward()
    assert rel_l2(oi.last_representation, ri['last_representation']) < 2e-2 and rel_l2(ot.last_representation, rt['last_representation']) < 2e-2
    assert abs(loss.item() - ref.item()) <= 1e-2 * abs(ref.item()), (loss.item(), ref.item())
    errs = {}
    for tag, m, sd in (('img', s_img, o_i), ('txt', s_txt, o_t)):
        for n, p in m.named_parameters():
            if sd[n].grad is None:
                assert p.grad is None, n               # (no_trans towers: text compares the layer count, so only the image projections rest)
                continue
            errs[f'{tag}.{n}'] = rel_l2(p.grad, sd[n].grad)
    bad = {k: v for k, v in errs.items() if v > 3e-2}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:6]
