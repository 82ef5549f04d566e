"""End-to-end parity of the HIP towers + fused loss (through the C ABI) against
  (a) the oracle run on the same seeded inputs / weights, and (b) the goldens emitted by the reference itself.

Tolerances (bf16 GEMM operands, f32 accumulate / LN / softmax / loss; judged against the f32 oracle):
  embeddings  rel-L2 <= 2e-2, cosine >= 0.999 ; loss |rel| <= 2e-2 ; parameter gradients rel-L2 <= 8e-2 on the tiny config
  (measured: 1.4e-2 at the head growing ~1e-2 per block execution towards the embedding — accumulated bf16 rounding of
  the saved activations and of the bf16 gradient operands; tools/diag/grad_err.py prints the profile).
"""
import os

import numpy as np
import pytest
import torch

import oracle
from distillclip_amd import synth

pytestmark = pytest.mark.gpu

TINY = dict(
    seed=11, B=3, res=32, patch=8, ctx=13, vocab=97, out_dim=64,
    s_img=dict(img_size=32, patch_size=8, in_chans=3, out_dim=64, embed_dim=128, depth=4, num_heads=4,
               mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True),
    s_txt=dict(vocab_size=97, context_length=13, out_dim=64, embed_dim=128, depth=2, num_heads=2,
               mlp_ratio=4.0, qkv_bias=False, repeated_times=2, use_transform=True),
)


def T(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


L1_TOL = 2e-1      # gradient rel-L2 bound for objectives containing out_l1 at B = 3 (see the comment in the dual test)


def rel_l2(a, b):
    a, b = a.detach().float().cpu().reshape(-1), torch.as_tensor(b).float().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def cosine(a, b):
    a, b = a.detach().float().cpu().reshape(-1), torch.as_tensor(b).float().reshape(-1)
    return (a @ b / (a.norm() * b.norm() + 1e-20)).item()


@pytest.fixture(scope='module')
def tiny(golden_dir):
    return dict(np.load(os.path.join(golden_dir, 'tiny.npz')))


def _tiny_modules():
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer, ImageEncoder, TextEncoder
    c = TINY
    s_img = RepeatVisionTransformer(**c['s_img'])
    s_img.load_state_dict(T(synth.student_image_state(c['seed'], **c['s_img'])))
    s_txt = RepeatTextTransformer(**c['s_txt'])
    s_txt.load_state_dict(T(synth.student_text_state(c['seed'], **c['s_txt'])))
    t_img = ImageEncoder(False, dict(input_resolution=c['res'], patch_size=c['patch'], width=128, layers=2, heads=2,
                                     output_dim=c['out_dim'], need_layers=None))
    t_img.load_state_dict(T(synth.teacher_image_state(c['seed'], 128, 2, c['patch'], c['res'], c['out_dim'])))
    t_txt = TextEncoder(128, 2, 2, c['ctx'], None, c['vocab'], c['out_dim'], is_student=False)
    t_txt.load_state_dict(T(synth.teacher_text_state(c['seed'], 128, 2, c['ctx'], c['vocab'], c['out_dim'])))
    return s_img.cuda(), s_txt.cuda(), t_img.cuda(), t_txt.cuda()


def test_tiny_forward_vs_reference_golden(tiny):
    s_img, s_txt, t_img, t_txt = _tiny_modules()
    image, text = torch.from_numpy(tiny['image']).cuda(), torch.from_numpy(tiny['text']).cuda()
    with torch.no_grad():
        outs = {'s_img': s_img(image), 's_txt': s_txt(text), 't_img': t_img(image), 't_txt': t_txt(text)}
    for tag, o in outs.items():
        ref = tiny[f'{tag}.last_representation']
        assert rel_l2(o.last_representation, ref) < 2e-2, (tag, rel_l2(o.last_representation, ref))
        assert cosine(o.last_representation, ref) > 0.999, tag


def test_tiny_last_layer_output_on_request_vs_reference_golden(tiny):
    """`last_layer_output` [B, N, E] (reference output.py:16-35; _common.py:212-215, weight_share_model.py:364-366, :504-506) is
    produced on request — by the ControlOutput extension flag or later from encoder.last_layer_output() — and its class / EOT
    row is last_representation."""
    from distillclip_amd.model.component import ControlOutput
    s_img, s_txt, t_img, t_txt = _tiny_modules()
    image, text = torch.from_numpy(tiny['image']).cuda(), torch.from_numpy(tiny['text']).cuda()
    co = ControlOutput(need_last_layer_output=True)
    outs = {'s_img': s_img(image, co), 's_txt': s_txt(text, co), 't_img': t_img(image, co), 't_txt': t_txt(text, co)}   # students in training mode
    eot = text.argmax(-1)
    for tag, o in outs.items():
        ref = tiny[f'{tag}.last_layer_output']
        assert tuple(o.last_layer_output.shape) == ref.shape
        assert rel_l2(o.last_layer_output, ref) < 2e-2, (tag, rel_l2(o.last_layer_output, ref))
        row = o.last_layer_output[:, 0] if 'img' in tag else o.last_layer_output[torch.arange(text.shape[0]), eot]
        assert rel_l2(row, o.last_representation.detach().cpu().numpy()) < 1e-3, tag     # same LN + GEMM, all rows vs one row
    with torch.no_grad():                       # inference-mode workspace layout + the explicit accessor
        o = s_img(image)
        assert o.last_layer_output is None
        assert rel_l2(s_img.last_layer_output(), tiny['s_img.last_layer_output']) < 2e-2
        t_txt(text)
        assert rel_l2(t_txt.last_layer_output(), tiny['t_txt.last_layer_output']) < 2e-2


def _grad_check(module, tiny, prefix, tol=5e-2, loose=2e-1):
    worst = {}
    for n, p in module.named_parameters():
        key = prefix + n
        if key not in tiny:
            continue
        assert p.grad is not None, n
        ref = tiny[key]
        if np.abs(ref).max() == 0:
            assert p.grad.abs().max().item() == 0, n
            continue
        worst[n] = rel_l2(p.grad, ref)
    assert len(worst) > 10
    bad = {n: e for n, e in worst.items() if e > (loose if ('conv_' in n or 'bias' in n or 'norm' in n) else tol)}
    assert not bad, bad
    return worst


@pytest.mark.parametrize('case', ['lclip', 'all'])
def test_tiny_dual_training_step_vs_reference_golden(tiny, case):
    from distillclip_amd.model import LossCalculator
    from distillclip_amd.model.component import CLIPModel
    s_img, s_txt, t_img, t_txt = _tiny_modules()
    student, teacher = CLIPModel(True, s_img, s_txt), CLIPModel(False, t_img, t_txt)
    for p in teacher.parameters():
        p.requires_grad = False
    if case == 'lclip':
        lc = LossCalculator(['out_l1', 'out_cos', 'cos_diff'], {'cos_diff': 0.1})
    else:
        lc = LossCalculator(['out_l1', 'out_cos', 'out_kl', 'cos_diff', 'hard_label', 'soft_label', 'logits_mse'],
                            {'cos_diff': 0.1, 'soft_label': 0.5}, temperature=2.0)
    image, text = torch.from_numpy(tiny['image']).cuda(), torch.from_numpy(tiny['text']).cuda()
    so = student(text, image, lc.get_control_output())
    to = teacher(text, image, lc.get_control_output())
    loss, res = lc(so, to, 'all')
    ref = float(tiny[f'{case}.loss'])
    assert abs(loss.item() - ref) <= 2e-2 * abs(ref), (loss.item(), ref)
    for k, v in res.items():
        r = float(tiny[f'{case}.term.{k}'])
        assert abs(v.item() - r) <= 6e-2 * abs(r) + 1e-3, (k, v.item(), r)   # 9-sample statistics at B=3 amplify bf16 noise
    assert set(res) == {k[len(case) + 6:] for k in tiny if k.startswith(f'{case}.term.')}
    loss.backward()
    # out_l1 (sign) and cos_diff (relu) gradients are discontinuous in the embeddings, and 'all' adds five more
    # 3x3-logit terms: bf16 forward noise moves the point the (exact) backward is evaluated at
    # (tools/diag/grad_err_txtc.py: 1-5 % with out_cos alone, up to 14 % once sign(s - t) of out_l1 flips on a few of the
    # B*E = 192 elements).  The smooth-loss test below pins the backward kernels tightly.
    _grad_check(s_img, tiny, f'{case}.s_img.grad.', L1_TOL, 2 * L1_TOL)
    _grad_check(s_txt, tiny, f'{case}.s_txt.grad.', L1_TOL, 2 * L1_TOL)


def test_shared_image_patches_between_teacher_and_student(tiny):
    """teacher and student image towers take their patch rows from ONE im2row when they cut the same images the same way
    (dclip_encoder_forward_patches / _backward_patches; reference dual_distill_model.py:107-109: both see `image`): embeddings and every
    student gradient — the patch-embedding wgrad reads those rows — equal the separately converted run (embeddings bit for bit)"""
    from distillclip_amd.model.component import _tower
    image = torch.from_numpy(tiny['image']).cuda()

    def run(share):
        s_img, _, t_img, _ = _tiny_modules()
        towers = [s_img._tower, t_img._tower] if share else []
        with _tower.shared_image_patches(image, towers) as sh:
            assert (sh.entry is not None) == share
            from distillclip_amd.model.component.output import ControlOutput
            co = ControlOutput()
            so = s_img(image, co)
            with torch.no_grad():
                to = t_img(image, co)
            used = (s_img._tower._patch_rows is not None)
        assert used == share
        (so.last_representation * to.last_representation).sum().backward()
        assert s_img._tower._patch_rows is None               # released by the backward
        return so.last_representation.detach().clone(), to.last_representation.clone(), {n: p.grad.clone() for n, p in s_img.named_parameters()}

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for n in a[2]:      # (same operands; LayerNorm / bias / small-wgrad sums are f32 atomics: equal up to summation order)
        assert rel_l2(a[2][n], b[2][n].cpu().numpy()) < 1e-5, n
    # a different tensor (same values) is not the tensor the rows were cut from: the tower converts it itself
    s_img, _, _, _ = _tiny_modules()
    with _tower.shared_image_patches(image, [s_img._tower, s_img._tower]):
        from distillclip_amd.model.component.output import ControlOutput
        s_img(image.clone(), ControlOutput())
        assert s_img._tower._patch_rows is None


def test_tiny_backward_smooth_loss_vs_oracle(tiny):
    """Backward accuracy with a smooth objective (out_cos + out_kl): isolates the kernels' own error from the
    sign / relu discontinuities of out_l1 and cos_diff."""
    from distillclip_amd.model import LossCalculator
    from distillclip_amd.model.component import CLIPModel
    c = TINY
    s_img, s_txt, t_img, t_txt = _tiny_modules()
    student, teacher = CLIPModel(True, s_img, s_txt), CLIPModel(False, t_img, t_txt)
    image, text = torch.from_numpy(tiny['image']), torch.from_numpy(tiny['text'])
    lc = LossCalculator(['out_cos', 'out_kl', 'soft_label'], temperature=1.5)
    loss, _ = lc(student(text.cuda(), image.cuda()), teacher(text.cuda(), image.cuda()), 'all')
    loss.backward()
    sd_i = {k: v.clone().requires_grad_(True) for k, v in T(synth.student_image_state(c['seed'], **c['s_img'])).items()}
    sd_t = {k: v.clone().requires_grad_(True) for k, v in T(synth.student_text_state(c['seed'], **c['s_txt'])).items()}
    with torch.no_grad():
        ti = oracle.teacher_image_forward(T(synth.teacher_image_state(c['seed'], 128, 2, c['patch'], c['res'], c['out_dim'])), image)
        tt = oracle.teacher_text_forward(T(synth.teacher_text_state(c['seed'], 128, 2, c['ctx'], c['vocab'], c['out_dim'])), text)
    so = oracle.clip_forward(oracle.student_image_forward(sd_i, image, 4), oracle.student_text_forward(sd_t, text, 2))
    ol, _ = oracle.LossOracle(['out_cos', 'out_kl', 'soft_label'], temperature=1.5)(so, oracle.clip_forward(ti, tt), 'all')
    ol.backward()
    assert abs(loss.item() - ol.item()) <= 1e-2 * abs(ol.item())
    for mod, sd in ((s_img, sd_i), (s_txt, sd_t)):
        errs = {n: rel_l2(p.grad, sd[n].grad) for n, p in mod.named_parameters() if sd[n].grad.abs().max() > 0}
        bad = {n: e for n, e in errs.items() if e > (1.5e-1 if ('conv_' in n or 'bias' in n or 'norm' in n) else 8e-2)}
        assert not bad, bad


def test_tiny_backward_vs_rounding_matched_oracle(tiny):
    """End-to-end gradient parity held TIGHT: the oracle in bf16_matched() mode rounds to bf16 at the same storage points as the
    HIP path (oracle/encoders.py header), so what is left is accumulation order — every parameter gradient, incl. the conv_l /
    conv_w head mixes, biases and LayerNorm affine parameters, agrees to ~1e-2 where the fp32 oracle allows only ~1e-1.  A backward
    error of a few per cent in any kernel fails here."""
    from distillclip_amd.model import LossCalculator
    from distillclip_amd.model.component import CLIPModel
    c = TINY
    s_img, s_txt, t_img, t_txt = _tiny_modules()
    student, teacher = CLIPModel(True, s_img, s_txt), CLIPModel(False, t_img, t_txt)
    image, text = torch.from_numpy(tiny['image']), torch.from_numpy(tiny['text'])
    lc = LossCalculator(['out_cos', 'out_kl', 'soft_label'], temperature=1.5)
    so_h = student(text.cuda(), image.cuda())
    loss, _ = lc(so_h, teacher(text.cuda(), image.cuda()), 'all')
    loss.backward()
    sd_i = {k: v.clone().requires_grad_(True) for k, v in T(synth.student_image_state(c['seed'], **c['s_img'])).items()}
    sd_t = {k: v.clone().requires_grad_(True) for k, v in T(synth.student_text_state(c['seed'], **c['s_txt'])).items()}
    with oracle.bf16_matched():
        with torch.no_grad():
            ti = oracle.teacher_image_forward(T(synth.teacher_image_state(c['seed'], 128, 2, c['patch'], c['res'], c['out_dim'])), image)
            tt = oracle.teacher_text_forward(T(synth.teacher_text_state(c['seed'], 128, 2, c['ctx'], c['vocab'], c['out_dim'])), text)
        so = oracle.clip_forward(oracle.student_image_forward(sd_i, image, 4), oracle.student_text_forward(sd_t, text, 2))
        ol, _ = oracle.LossOracle(['out_cos', 'out_kl', 'soft_label'], temperature=1.5)(so, oracle.clip_forward(ti, tt), 'all')
        ol.backward()
    e_emb = {'s_img': rel_l2(so_h.visual_output.last_representation, so['visual_output']['last_representation'].detach().numpy()),
             's_txt': rel_l2(so_h.text_output.last_representation, so['text_output']['last_representation'].detach().numpy())}
    errs = {}
    for tag, mod, sd in (('s_img', s_img, sd_i), ('s_txt', s_txt, sd_t)):
        for n, p in mod.named_parameters():
            if sd[n].grad is not None and sd[n].grad.abs().max() > 0:
                errs[f'{tag}.{n}'] = rel_l2(p.grad, sd[n].grad.numpy())
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print('matched-oracle parity: loss', abs(loss.item() - ol.item()) / abs(ol.item()), 'embeddings', e_emb, 'worst gradients', top)
    assert abs(loss.item() - ol.item()) <= 2e-3 * abs(ol.item())
    assert max(e_emb.values()) < 5e-3, e_emb
    # the k-third of attn.qkv.bias has a zero true gradient (softmax shift invariance): pure rounding noise, compared on q / v only
    bad = {}
    for n, e in errs.items():
        if n.endswith('attn.qkv.bias'):
            tag, name = n.split('.', 1)
            mod, sd = (s_img, sd_i) if tag == 's_img' else (s_txt, sd_t)
            g, r = dict(mod.named_parameters())[name].grad.detach().cpu(), sd[name].grad
            D = g.numel() // 3
            e = max(rel_l2(g[:D], r[:D].numpy()), rel_l2(g[2 * D:], r[2 * D:].numpy()))
        # measured <= 1.5e-2 (fp32 oracle: 8e-2 / 1.5e-1 in the test above)
        if e > 2.5e-2:
            bad[n] = e
    assert not bad, bad


def test_tiny_one_tower_text_compressed(tiny):
    from distillclip_amd.model import LossCalculator
    from distillclip_amd.model.component import RepeatTextTransformer
    c = TINY
    _, _, _, t_txt = _tiny_modules()
    cfg = dict(c['s_txt'], compression_embedding=True, embedding_compression_dim=64)
    s = RepeatTextTransformer(**cfg)
    s.load_state_dict(T(synth.student_text_state(c['seed'] + 1, **cfg)))
    s = s.cuda()
    text = torch.from_numpy(tiny['text']).cuda()
    lc = LossCalculator(['out_l1', 'out_cos'])
    so = s(text, lc.get_control_output())
    to = t_txt(text)
    assert rel_l2(so.last_representation, tiny['txtc.last_representation']) < 2e-2
    loss, _ = lc(so, to, 'text')
    assert abs(loss.item() - float(tiny['txtc.loss'])) <= 2e-2 * abs(float(tiny['txtc.loss']))
    loss.backward()
    _grad_check(s, tiny, 'txtc.s_txt.grad.', L1_TOL, 2 * L1_TOL)


def test_tiny_one_tower_image_feature_mse(tiny):
    """hidden_rep_mse + embedding_mse (SURVEY.md §2.1 tier 2): hidden states / embeddings exported by both towers, the
    gradients re-enter the student's backward at every block execution.  Golden = the reference run `img1`."""
    from distillclip_amd.model import LossCalculator
    from distillclip_amd.model.component import ImageEncoder
    c = TINY
    s_img, _, _, _ = _tiny_modules()
    t_img = ImageEncoder(False, dict(input_resolution=c['res'], patch_size=c['patch'], width=128, layers=2, heads=2,
                                     output_dim=c['out_dim'], need_layers=[0, 1]))
    t_img.load_state_dict(T(synth.teacher_image_state(c['seed'], 128, 2, c['patch'], c['res'], c['out_dim'])))
    t_img = t_img.cuda()
    lc = LossCalculator(['out_l1', 'out_cos', 'hidden_rep_mse', 'embedding_mse'])
    co = lc.get_control_output()
    assert co.need_rep and co.need_emb
    image = torch.from_numpy(tiny['image']).cuda()
    so, to = s_img(image, co), t_img(image, co)
    assert len(so.representations) == 4 and len(to.representations) == 2          # student: every execution; teacher: need_layers
    for i in range(4):
        assert rel_l2(so.representations[i], tiny[f's_img.rep{i}']) < 2e-2
    for i in range(2):
        assert rel_l2(to.representations[i], tiny[f't_img.rep{i}']) < 2e-2
    assert rel_l2(so.embedding, tiny['s_img.embedding']) < 1e-2 and rel_l2(to.embedding, tiny['t_img.embedding']) < 1e-2
    loss, res = lc(so, to, 'image')
    assert abs(loss.item() - float(tiny['img1.loss'])) <= 2e-2 * abs(float(tiny['img1.loss'])), (loss.item(), float(tiny['img1.loss']))
    assert set(res) == {'out_l1', 'out_cos', 'hidden_rep_mse', 'embedding_mse'}
    for k, v in res.items():
        r = float(tiny['img1.term.' + k])
        assert abs(v.item() - r) <= 4e-2 * abs(r) + 1e-4, (k, v.item(), r)
    loss.backward()
    _grad_check(s_img, tiny, 'img1.s_img.grad.', L1_TOL, 2 * L1_TOL)


def test_feature_mse_needs_matching_widths():
    """S-txt (768) vs T-txt (512): the reference shape-errors here as well"""
    from distillclip_amd.model._loss import _FeatureMSEFn
    with pytest.raises(RuntimeError, match='must match'):
        _FeatureMSEFn.apply(torch.zeros(2, 4, 768, device='cuda'), torch.zeros(2, 4, 512, device='cuda'))


def test_hip_vs_oracle_same_inputs_new_seed():
    """independent of the committed goldens: oracle and HIP path on a fresh seed / batch size"""
    c = TINY
    seed, B = 77, 5
    s_img, s_txt, t_img, t_txt = _tiny_modules()
    sd_i = T(synth.student_image_state(seed, **c['s_img']))
    sd_t = T(synth.student_text_state(seed, **c['s_txt']))
    s_img.load_state_dict(sd_i)
    s_txt.load_state_dict(sd_t)
    image = torch.from_numpy(synth.images(seed, B, c['res']))
    text = torch.from_numpy(synth.captions(seed, B, c['ctx'], c['vocab'], 3, 9))
    with torch.no_grad():
        hi = s_img(image.cuda()).last_representation
        ht = s_txt(text.cuda()).last_representation
        oi = oracle.student_image_forward(sd_i, image, 4)['last_representation']
        ot = oracle.student_text_forward(sd_t, text, 2)['last_representation']
    assert rel_l2(hi, oi) < 2e-2 and rel_l2(ht, ot) < 2e-2


def test_real_shapes_b4_vs_reference_golden(golden_dir):
    """ViT-B/32 teacher + shipped l_clip students, B=4 (tools/golden/gen_golden.py:real_shapes)."""
    from distillclip_amd.model import LossCalculator
    from distillclip_amd.model.component import (RepeatVisionTransformer, RepeatTextTransformer, ImageEncoder, TextEncoder,
                                                 CLIPModel)
    g = dict(np.load(os.path.join(golden_dir, 'real_b4.npz')))
    seed, B = int(g['seed']), int(g['B'])
    s_img_cfg = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24,
                     mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
    s_txt_cfg = dict(depth=4, repeated_times=2, use_transform=True)
    s_img = RepeatVisionTransformer(**s_img_cfg)
    s_img.load_state_dict(T(synth.student_image_state(seed, **s_img_cfg)))
    s_txt = RepeatTextTransformer(**s_txt_cfg)
    s_txt.load_state_dict(T(synth.student_text_state(seed, **s_txt_cfg)))
    t_img = ImageEncoder(False, dict(input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512))
    t_img.load_state_dict(T(synth.teacher_image_state(seed)))
    t_txt = TextEncoder(512, 12, 8, 77, None, 49408, 512, is_student=False)
    t_txt.load_state_dict(T(synth.teacher_text_state(seed)))
    student, teacher = CLIPModel(True, s_img.cuda(), s_txt.cuda()), CLIPModel(False, t_img.cuda(), t_txt.cuda())
    for p in teacher.parameters():
        p.requires_grad = False
    image = torch.from_numpy(synth.images(seed, B, 224)).cuda()
    text = torch.from_numpy(synth.captions(seed, B)).cuda()
    lc = LossCalculator(['out_l1', 'out_cos', 'cos_diff'], {'cos_diff': 0.1})
    so, to = student(text, image), teacher(text, image)
    for tag, o in (('s_img', so.visual_output), ('s_txt', so.text_output), ('t_img', to.visual_output), ('t_txt', to.text_output)):
        ref = g[f'{tag}.last_representation']
        assert rel_l2(o.last_representation, ref) < 2e-2, (tag, rel_l2(o.last_representation, ref))
        assert cosine(o.last_representation, ref) > 0.999
    loss, res = lc(so, to, 'all')
    assert abs(loss.item() - float(g['loss'])) <= 2e-2 * abs(float(g['loss'])), (loss.item(), float(g['loss']))
    for k, v in res.items():
        r = float(g['term.' + k])
        assert abs(v.item() - r) <= 6e-2 * abs(r) + 1e-3, (k, v.item(), r)   # 9-sample statistics at B=3 amplify bf16 noise
    loss.backward()

    def check(prefix, slice_tol):
        worst = {}
        for tag, m in (('s_img', s_img), ('s_txt', s_txt)):
            for n, p in m.named_parameters():
                ref = float(g[f'{prefix}{tag}.gnorm.{n}'])
                if ref > 0:
                    worst[f'{tag}.{n}'] = abs(p.grad.norm().item() - ref) / ref
            for key in [k for k in g if k.startswith(f'{prefix}{tag}.gslice.')]:
                n = key[len(prefix) + len(tag) + 8:]
                got = dict(m.named_parameters())[n].grad.reshape(-1)[:256]
                assert rel_l2(got, g[key]) < slice_tol, (key, rel_l2(got, g[key]))
        bad = {k: v for k, v in worst.items() if v > 6e-2}
        assert not bad, bad
    # l_clip objective: norms are stable; 256-element slices of deep-layer gradients move by several % when ONE of the
    # B*E = 2048 sign(s - t) terms of out_l1 flips under a 1e-7 perturbation of the forward -> loose bound on slices
    check('', L1_TOL)
    # smooth objective (out_cos only, generated by the reference as well): tight bound on the same slices
    for p in student.parameters():
        p.grad = None
    lc2 = LossCalculator(['out_cos'])
    loss2, _ = lc2(student(text, image), to, 'all')
    assert abs(loss2.item() - float(g['cos.loss'])) <= 1e-2 * abs(float(g['cos.loss']))
    loss2.backward()
    check('cos.', 6e-2)


def _real_b4_modules(seed):
    from distillclip_amd.model.component import (RepeatVisionTransformer, RepeatTextTransformer, ImageEncoder, TextEncoder, CLIPModel)
    s_img_cfg = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24,
                     mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
    s_txt_cfg = dict(depth=4, repeated_times=2, use_transform=True)
    s_img = RepeatVisionTransformer(**s_img_cfg)
    s_img.load_state_dict(T(synth.student_image_state(seed, **s_img_cfg)))
    s_txt = RepeatTextTransformer(**s_txt_cfg)
    s_txt.load_state_dict(T(synth.student_text_state(seed, **s_txt_cfg)))
    t_img = ImageEncoder(False, dict(input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512))
    t_img.load_state_dict(T(synth.teacher_image_state(seed)))
    t_txt = TextEncoder(512, 12, 8, 77, None, 49408, 512, is_student=False)
    t_txt.load_state_dict(T(synth.teacher_text_state(seed)))
    student, teacher = CLIPModel(True, s_img.cuda(), s_txt.cuda()), CLIPModel(False, t_img.cuda(), t_txt.cuda())
    for p in teacher.parameters():
        p.requires_grad = False
    return s_img, s_txt, student, teacher, s_img_cfg, s_txt_cfg


def test_real_shapes_b4_every_parameter_gradient_vs_reference_golden(golden_dir):
    """round 4: the gradient of EVERY student parameter at the shipped shapes against the reference's own (real_b4_cos.npz: head and a
    spread sample of each tensor, smooth out_cos objective) — before, only five slices were pinned at real shapes."""
    from distillclip_amd.model import LossCalculator
    g = dict(np.load(os.path.join(golden_dir, 'real_b4_cos.npz')))
    seed, B = int(g['seed']), int(g['B'])
    s_img, s_txt, student, teacher, _, _ = _real_b4_modules(seed)
    image = torch.from_numpy(synth.images(seed, B, 224)).cuda()
    text = torch.from_numpy(synth.captions(seed, B)).cuda()
    loss, _ = LossCalculator(['out_cos'])(student(text, image), teacher(text, image), 'all')
    assert abs(loss.item() - float(g['cos.loss'])) <= 1e-2 * abs(float(g['cos.loss']))
    loss.backward()
    errs, norms = {}, {}
    for tag, m in (('s_img', s_img), ('s_txt', s_txt)):
        for n, p in m.named_parameters():
            gr = p.grad.reshape(-1)
            ref_norm = float(g[f'cos.{tag}.gnorm.{n}'])
            if ref_norm == 0:
                continue
            norms[f'{tag}.{n}'] = abs(gr.norm().item() - ref_norm) / ref_norm
            step = max(1, gr.numel() // 256)
            got = torch.cat([gr[:256], gr[::step][:256]])
            ref = np.concatenate([g[f'cos.{tag}.ghead.{n}'], g[f'cos.{tag}.gspread.{n}']])
            if n.endswith('attn.qkv.bias'):      # the k-third has a zero true gradient: compare where the reference is not noise
                keep = np.abs(ref) > 1e-3 * np.abs(ref).max()
                got, ref = got[torch.from_numpy(keep).to(got.device)], ref[keep]
            errs[f'{tag}.{n}'] = rel_l2(got, ref)
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    print('real-shape gradient parity vs the reference (fp32): worst slices', top, 'worst norms', sorted(norms.items(), key=lambda kv: -kv[1])[:4])
    assert len(errs) >= 100
    # measured (default path): slices <= 2.2e-2, norms <= 5e-3
    bad = {k: v for k, v in errs.items() if v > (8e-2 if ('conv_' in k or 'bias' in k or 'norm' in k) else 5e-2)}
    assert not bad, bad
    assert max(norms.values()) < 2e-2, sorted(norms.items(), key=lambda kv: -kv[1])[:4]


def test_real_shapes_b4_backward_vs_rounding_matched_oracle():
    """round 4: the TIGHT end-to-end bound (every parameter gradient <= 2.5e-2 against the oracle that rounds to bf16 where the HIP path
    stores bf16) at the shipped student shapes — H = 24 / hd = 32 / N = 50 and H = 12 / hd = 64 / N = 77, B = 4 — not only on the tiny
    configuration."""
    from distillclip_amd.model import LossCalculator
    seed, B = 2022, 4
    s_img, s_txt, student, teacher, s_img_cfg, s_txt_cfg = _real_b4_modules(seed)
    image, text = torch.from_numpy(synth.images(seed, B, 224)), torch.from_numpy(synth.captions(seed, B))
    names = ['out_cos', 'out_kl', 'soft_label']
    so_h = student(text.cuda(), image.cuda())
    loss, _ = LossCalculator(names, temperature=1.5)(so_h, teacher(text.cuda(), image.cuda()), 'all')
    loss.backward()
    sd_i = {k: v.clone().requires_grad_(True) for k, v in T(synth.student_image_state(seed, **s_img_cfg)).items()}
    sd_t = {k: v.clone().requires_grad_(True) for k, v in T(synth.student_text_state(seed, **s_txt_cfg)).items()}
    with oracle.bf16_matched():
        with torch.no_grad():
            to = oracle.clip_forward(oracle.teacher_image_forward(T(synth.teacher_image_state(seed)), image),
                                     oracle.teacher_text_forward(T(synth.teacher_text_state(seed)), text))
        so = oracle.clip_forward(oracle.student_image_forward(sd_i, image, 24), oracle.student_text_forward(sd_t, text, 12))
        ol, _ = oracle.LossOracle(names, temperature=1.5)(so, to, 'all')
        ol.backward()
    e_emb = {'s_img': rel_l2(so_h.visual_output.last_representation, so['visual_output']['last_representation'].detach().numpy()),
             's_txt': rel_l2(so_h.text_output.last_representation, so['text_output']['last_representation'].detach().numpy())}
    errs = {}
    for tag, mod, sd in (('s_img', s_img, sd_i), ('s_txt', s_txt, sd_t)):
        for n, p in mod.named_parameters():
            r = sd[n].grad
            if r is None or r.abs().max() == 0:
                continue
            gq = p.grad.detach().cpu()
            if n.endswith('attn.qkv.bias'):
                D = gq.numel() // 3
                errs[f'{tag}.{n}'] = max(rel_l2(gq[:D], r[:D].numpy()), rel_l2(gq[2 * D:], r[2 * D:].numpy()))
            elif n == 'token_embedding.weight':
                rows = torch.unique(text.reshape(-1))                      # the other 49 k rows have no gradient
                errs[f'{tag}.{n}'] = rel_l2(gq[rows], r[rows].numpy())
            else:
                errs[f'{tag}.{n}'] = rel_l2(gq, r.numpy())
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    print('real-shape matched-oracle parity: loss', abs(loss.item() - ol.item()) / abs(ol.item()), 'embeddings', e_emb, 'worst gradients', top)
    assert abs(loss.item() - ol.item()) <= 2e-3 * abs(ol.item())
    assert max(e_emb.values()) < 5e-3, e_emb
    # measured <= 1.3e-2 on the default path.  The matched oracle rounds where THAT path stores bf16; the unfused fallback
    # (DCLIP_ATTN_MIX=0: scores and probabilities through HBM) rounds at other points of the score stage: <= 3.1e-2 on the conv_l weights
    bound = 2.5e-2 if os.environ.get('DCLIP_ATTN_MIX', '1') != '0' else 4e-2
    bad = {n: e for n, e in errs.items() if e > bound}
    assert not bad, bad


def test_c_abi_backward_is_self_contained_on_a_repeated_or_foreign_call(tiny):
    """dclip_encoder_backward starts from a residual-stream gradient accumulator that the TRAINING FORWARD leaves cleared (the fills run beside
    the other towers instead of on the critical path).  A caller outside the Python wrapper may call the backward twice on one forward
    (several d_out: gradient checks), retry it, or use another workspace in between; round 4 then accumulated onto stale residual gradients
    and returned wrong parameter gradients with rc = 0.  The handle now remembers which workspace its last training forward prepared and the
    backward clears the seeds itself when it does not find its workspace there (include/dclip.h, dclip_encoder_backward): every variant below
    returns the gradients of a fresh forward + backward."""
    s_img, s_txt, _, _ = _tiny_modules()
    image = torch.from_numpy(tiny['image']).cuda()
    text = torch.from_numpy(tiny['text']).cuda()
    for enc, x in ((s_img, image), (s_txt, text)):
        tw = enc._tower
        gen = torch.Generator().manual_seed(5)

        def grads_of(d):
            tw.flat_grad.zero_()
            tw.backward(xin, d)
            torch.cuda.synchronize()
            return tw.flat_grad.clone()

        out, xin, _, _ = tw.forward(x, training=True)
        d1 = torch.randn(out.shape, generator=gen).cuda()
        d2 = torch.randn(out.shape, generator=gen).cuda()
        g1 = grads_of(d1)                                   # the ordinary call: consumes the forward's cleared seeds
        tw._saved_batch = x.shape[0]                        # (the Python wrapper refuses a second backward; the C ABI must cope with it)
        g2 = grads_of(d2)                                   # second backward on the same forward, another d_out
        tw._saved_batch = x.shape[0]
        g1_again = grads_of(d1)                             # ... and the first one repeated
        # reference: each d_out on a forward of its own
        tw.forward(x, training=True)
        r1 = grads_of(d1)
        tw.forward(x, training=True)
        r2 = grads_of(d2)
        scale = r1.abs().max().item()
        for got, want, tag in ((g1, r1, 'first'), (g2, r2, 'second backward, other d_out'), (g1_again, r1, 'third backward, first d_out again')):
            err = (got - want).abs().max().item()
            assert err <= 2e-5 * scale, (type(enc).__name__, tag, err, scale)      # (f32 atomics of the small wgrads / column sums: order noise only)
        assert (g2 - r1).abs().max().item() > 1e-2 * scale  # the two d_out really give different gradients


def test_teacher_text_prefix_is_exact():
    """causal teacher text tower on the prefix that holds every EOT == on all 77 positions (the EOT row cannot see later tokens)"""
    from distillclip_amd.model.component import TextEncoder
    seed, B = 9, 16
    enc = TextEncoder(512, 12, 8, 77, None, 49408, 512, is_student=False)
    sd = T(synth.teacher_text_state(seed))
    enc.load_state_dict(sd)
    enc = enc.cuda()
    caps = synth.captions(seed, B)
    text = torch.from_numpy(caps).cuda()
    full = enc(text).last_representation.clone()
    n_eff = int((caps != 0).sum(1).max())
    assert n_eff < 50
    enc.max_tokens = n_eff
    short = enc(text).last_representation
    assert rel_l2(short, full.cpu()) < 2e-3, rel_l2(short, full.cpu())         # same kernels, different tile boundaries
    with torch.no_grad():
        ref = oracle.teacher_text_forward(sd, torch.from_numpy(caps))['last_representation']
    assert rel_l2(short, ref) < 2e-2 and cosine(short, ref) > 0.999
    from distillclip_amd._lib import lib
    with pytest.raises(ValueError, match='tokens_eff'):            # only the causal text teacher accepts a prefix
        from distillclip_amd.model.component import RepeatTextTransformer
        s = RepeatTextTransformer(depth=4, repeated_times=2, use_transform=True).cuda()
        s._tower.forward(text, training=False, tokens_eff=40)


def test_unfused_score_stage_path_matches_the_same_goldens():
    """The default student path is the register-resident score stage (attention_mix.hip).  The unfused kernels of rounds 1-2
    (attn_nt + softmax, scores through HBM) remain as the fallback for head counts dclip_attn_mix_supported rejects and behind
    DCLIP_ATTN_MIX=0; the knob is read once per process, so the forward / training-step / backward golden tests of this file
    are re-run in ONE child process with it set (the parent keeps no GPU work in flight meanwhile)."""
    import subprocess
    import sys
    if os.environ.get('DCLIP_TEST_CHILD'):
        pytest.skip('already inside a child test process')
    torch.cuda.synchronize()
    env = dict(os.environ, DCLIP_ATTN_MIX='0', DCLIP_TEST_CHILD='1')
    # (tiny configuration: forward, training step, backward against the matched oracle; real shapes: the every-parameter gradient golden.
    #  The unfused kernels are also what a trainable CLIP tower runs by default: tests/test_clip_student_gpu.py)
    sel = 'tiny_forward_vs_reference_golden or tiny_dual_training_step or tiny_backward_vs_rounding_matched or real_shapes_b4_every'
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-x', '-q', '-k', sel, '-p', 'no:cacheprovider'],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout, r.stdout[-2000:]
