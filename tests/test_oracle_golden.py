"""Pin the oracle (oracle/) against goldens produced by the reference's own modules (tools/golden/gen_golden.py).

CPU only.  Tolerances are fp32 re-association noise (the restatement uses einsum for the 1x1 head-mixing conv etc.).
"""
import os

import numpy as np
import pytest
import torch

import oracle
from distillclip_amd import synth

TINY = dict(
    seed=11, B=3, res=32, patch=8, ctx=13, vocab=97, out_dim=64,
    s_img=dict(img_size=32, patch_size=8, in_chans=3, out_dim=64, embed_dim=128, depth=4, num_heads=4,
               mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True),
    s_txt=dict(vocab_size=97, context_length=13, out_dim=64, embed_dim=128, depth=2, num_heads=2,
               mlp_ratio=4.0, qkv_bias=False, repeated_times=2, use_transform=True),
)


def T(d, grad=False):
    out = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
    if grad:
        for v in out.values():
            v.requires_grad_(True)
    return out


@pytest.fixture(scope='module')
def tiny(golden_dir):
    return dict(np.load(os.path.join(golden_dir, 'tiny.npz')))


@pytest.fixture(scope='module')
def tiny_models():
    c = TINY
    return dict(
        t_img=T(synth.teacher_image_state(c['seed'], 128, 2, c['patch'], c['res'], c['out_dim'])),
        t_txt=T(synth.teacher_text_state(c['seed'], 128, 2, c['ctx'], c['vocab'], c['out_dim'])),
        s_img=T(synth.student_image_state(c['seed'], **c['s_img']), grad=True),
        s_txt=T(synth.student_text_state(c['seed'], **c['s_txt']), grad=True),
    )


def close(a, b, rtol=2e-5, atol=2e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_inputs_regenerate(tiny):
    c = TINY
    np.testing.assert_array_equal(synth.images(c['seed'], c['B'], c['res']), tiny['image'])
    np.testing.assert_array_equal(synth.captions(c['seed'], c['B'], c['ctx'], c['vocab'], 3, 9), tiny['text'])


def _forward_all(tiny, m, cap=None, need_rep=False):
    image, text = torch.from_numpy(tiny['image']), torch.from_numpy(tiny['text'])
    cap = cap if cap is not None else {}
    s_img = oracle.student_image_forward(m['s_img'], image, 4, cap=cap.setdefault('s_img', {}), need_rep=need_rep)
    s_txt = oracle.student_text_forward(m['s_txt'], text, 2, cap=cap.setdefault('s_txt', {}), need_rep=need_rep)
    with torch.no_grad():
        t_img = oracle.teacher_image_forward(m['t_img'], image, need_rep=need_rep, need_emb=True)
        t_txt = oracle.teacher_text_forward(m['t_txt'], text, need_rep=need_rep, need_emb=True)
    return s_img, s_txt, t_img, t_txt


def test_forward_and_intermediates(tiny, tiny_models):
    cap = {}
    outs = dict(zip(('s_img', 's_txt', 't_img', 't_txt'), _forward_all(tiny, tiny_models, cap, need_rep=True)))
    for tag, o in outs.items():
        close(o['last_representation'], tiny[f'{tag}.last_representation'])
        close(o['last_layer_output'], tiny[f'{tag}.last_layer_output'])
        close(o['embedding'], tiny[f'{tag}.embedding'])
        for i, r in enumerate(o['representations']):
            close(r, tiny[f'{tag}.rep{i}'])
    # attention internals of the students: raw scaled scores (pre conv_l) and probs (post softmax, pre conv_w)
    for tag, nb in (('s_img', 2), ('s_txt', 1)):
        i = 0
        for b in range(nb):
            for r in range(2):
                close(cap[tag][f'sblock{b}.{r}.scores'], tiny[f'{tag}.scores{i}'])
                close(cap[tag][f'sblock{b}.{r}.probs'], tiny[f'{tag}.probs{i}'])
                i += 1
    so = oracle.clip_forward(outs['s_img'], outs['s_txt'])
    to = oracle.clip_forward(outs['t_img'], outs['t_txt'])
    close(so['i2t_logits'], tiny['s.i2t_logits'])
    close(to['i2t_logits'], tiny['t.i2t_logits'])


def _zero_grads(m):
    for sd in (m['s_img'], m['s_txt']):
        for v in sd.values():
            v.grad = None


def _check_grads(sd, tiny, prefix, rtol=2e-4):
    n = 0
    for k, v in tiny.items():
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        g = sd[name].grad
        assert g is not None, name
        scale = max(float(np.abs(v).max()), 1e-12)
        np.testing.assert_allclose(g.numpy(), v, rtol=rtol, atol=rtol * scale, err_msg=name)
        n += 1
    assert n > 0
    return n


@pytest.mark.parametrize('case', ['all', 'lclip'])
def test_dual_loss_and_grads(tiny, tiny_models, case):
    _zero_grads(tiny_models)
    s_img, s_txt, t_img, t_txt = _forward_all(tiny, tiny_models)
    if case == 'all':
        lc = oracle.LossOracle(['out_l1', 'out_cos', 'out_kl', 'cos_diff', 'hard_label', 'soft_label', 'logits_mse'],
                               {'cos_diff': 0.1, 'soft_label': 0.5}, temperature=2.0)
    else:
        lc = oracle.LossOracle(['out_l1', 'out_cos', 'cos_diff'], {'cos_diff': 0.1})
    loss, res = lc(oracle.clip_forward(s_img, s_txt), oracle.clip_forward(t_img, t_txt), 'all')
    close(loss, tiny[f'{case}.loss'])
    terms = {k[len(case) + 6:]: v for k, v in tiny.items() if k.startswith(f'{case}.term.')}
    assert set(terms) == set(res)
    for k, v in terms.items():
        close(res[k], v)
    loss.backward()
    assert _check_grads(tiny_models['s_img'], tiny, f'{case}.s_img.grad.') >= 30
    assert _check_grads(tiny_models['s_txt'], tiny, f'{case}.s_txt.grad.') >= 15


def test_one_tower_image_with_feature_terms(tiny, tiny_models):
    _zero_grads(tiny_models)
    c = TINY
    image = torch.from_numpy(tiny['image'])
    so = oracle.student_image_forward(tiny_models['s_img'], image, 4, need_rep=True)
    with torch.no_grad():
        to = oracle.teacher_image_forward(tiny_models['t_img'], image, need_layers=[0, 1], need_rep=True, need_emb=True)
    lc = oracle.LossOracle(['out_l1', 'out_cos', 'hidden_rep_mse', 'embedding_mse'])
    loss, res = lc(so, to, 'image')
    close(loss, tiny['img1.loss'])
    for k in res:
        close(res[k], tiny['img1.term.' + k])
    loss.backward()
    _check_grads(tiny_models['s_img'], tiny, 'img1.s_img.grad.')


def test_one_tower_text_and_compressed(tiny, tiny_models):
    _zero_grads(tiny_models)
    c = TINY
    text = torch.from_numpy(tiny['text'])
    with torch.no_grad():
        to = oracle.teacher_text_forward(tiny_models['t_txt'], text)
    lc = oracle.LossOracle(['out_l1', 'out_cos'])
    so = oracle.student_text_forward(tiny_models['s_txt'], text, 2)
    loss, res = lc(so, to, 'text')
    close(loss, tiny['txt1.loss'])
    loss.backward()
    _check_grads(tiny_models['s_txt'], tiny, 'txt1.s_txt.grad.')
    cfg = dict(c['s_txt'], compression_embedding=True, embedding_compression_dim=64)
    sd = T(synth.student_text_state(c['seed'] + 1, **cfg), grad=True)
    so = oracle.student_text_forward(sd, text, 2)
    close(so['last_representation'], tiny['txtc.last_representation'])
    loss, _ = lc(so, to, 'text')
    close(loss, tiny['txtc.loss'])
    loss.backward()
    _check_grads(sd, tiny, 'txtc.s_txt.grad.')


@pytest.mark.parametrize('case', ['b8', 'b37', 'small'])
def test_loss_components(golden_dir, case):
    g = dict(np.load(os.path.join(golden_dir, 'loss.npz')))
    e = {k: torch.from_numpy(g[f'{case}.{k}']) for k in ('si', 'st', 'ti', 'tt')}
    e['si'].requires_grad_(True)
    e['st'].requires_grad_(True)
    mk = lambda i, t: oracle.clip_forward({'last_representation': i}, {'last_representation': t})
    names = ['out_l1', 'out_cos', 'out_kl', 'cos_diff', 'hard_label', 'soft_label', 'logits_mse', 'out_ce']
    lc = oracle.LossOracle(names, {'cos_diff': 0.1, 'hard_label': 2.0}, temperature=0.5)
    loss, res = lc(mk(e['si'], e['st']), mk(e['ti'], e['tt']), 'all')
    close(loss, g[f'{case}.loss'], rtol=5e-5)
    for k, v in res.items():
        close(v, g[f'{case}.term.{k}'], rtol=5e-5, atol=1e-6)
    loss.backward()
    for k in ('si', 'st'):
        ref = g[f'{case}.grad.{k}']
        np.testing.assert_allclose(e[k].grad.numpy(), ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max())


def test_real_shapes_gradients_of_every_parameter_match_the_reference(golden_dir):
    """real_b4_cos.npz (round 4; tools/golden/gen_golden.py:real_shapes_cos_slices): the reference's gradients of EVERY student parameter
    at the shipped shapes (H = 24 / hd = 32 / N = 50 and H = 12 / hd = 64 / N = 77), B = 4, smooth objective — the head of each tensor
    and 256 elements spread over it.  Pins the oracle's backward at real shapes; the HIP path is held to the same file on the GPU."""
    g = dict(np.load(os.path.join(golden_dir, 'real_b4_cos.npz')))
    seed, B = int(g['seed']), int(g['B'])
    s_img_cfg = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24,
                     mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
    s_txt_cfg = dict(depth=4, repeated_times=2, use_transform=True)
    image = torch.from_numpy(synth.images(seed, B, 224))
    text = torch.from_numpy(synth.captions(seed, B))
    sd_i = T(synth.student_image_state(seed, **s_img_cfg), grad=True)
    sd_t = T(synth.student_text_state(seed, **s_txt_cfg), grad=True)
    with torch.no_grad():
        to = oracle.clip_forward(oracle.teacher_image_forward(T(synth.teacher_image_state(seed)), image),
                                 oracle.teacher_text_forward(T(synth.teacher_text_state(seed)), text))
    so = oracle.clip_forward(oracle.student_image_forward(sd_i, image, 24), oracle.student_text_forward(sd_t, text, 12))
    loss, _ = oracle.LossOracle(['out_cos'])(so, to, 'all')
    close(loss, g['cos.loss'], rtol=2e-5)
    loss.backward()
    n_checked = 0
    for tag, sd in (('s_img', sd_i), ('s_txt', sd_t)):
        for n, p in sd.items():
            key = f'cos.{tag}.gnorm.{n}'
            if key not in g:
                continue
            gr = p.grad.reshape(-1)
            ref_norm = float(g[key])
            assert abs(float(gr.norm()) - ref_norm) <= 2e-4 * ref_norm + 1e-12, (tag, n)
            step = max(1, gr.numel() // 256)
            for kind, got in (('ghead', gr[:256]), ('gspread', gr[::step][:256])):
                ref = g[f'cos.{tag}.{kind}.{n}']
                np.testing.assert_allclose(got.numpy(), ref, rtol=5e-4, atol=5e-4 * max(np.abs(ref).max(), 1e-12), err_msg=f'{tag}.{n}.{kind}')
            n_checked += 1
    assert n_checked == 68 + 44


# ---- round 5: BASELINE.json configs[1], [2], [4] at their real shapes, pinned to reference runs (tools/golden/gen_golden.py real_image1 /
# real_textc / real_336).  The HIP path is held to the same three files on the GPU (tests/test_configs_gpu.py). ------------------------------
def _assert_grads(g, prefix, grads, expect, tol=5e-4):
    import real_cases as rc
    samples, norms, n = rc.gradient_errors(g, prefix, grads)
    assert n == expect, (prefix, n)
    bad = {k: v for k, v in samples.items() if v > tol}
    assert not bad, (prefix, bad)
    bad = {k: v for k, v in norms.items() if v > tol}
    assert not bad, (prefix, 'norms', bad)


def _grads(sd):
    return {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in sd.items()}


def test_image_yaml_real_shapes_match_the_reference(golden_dir):
    """config/final_config/image.yaml (one image tower, freeze_embed, teacher_need_layers [0, 1, 10, 11], losses out_l1 + out_cos):
    embeddings, loss dict and the gradient of every trainable parameter, for the shipped loss set and for the smooth objective."""
    import real_cases as rc
    g = rc.load(golden_dir, 'real_b4_image1.npz')
    image, tsd, sd = rc.image1_inputs(g)
    for n, v in sd.items():
        v.requires_grad_(n not in rc.FROZEN_IMAGE)
    with torch.no_grad():
        to = oracle.teacher_image_forward(tsd, image, need_layers=[0, 1, 10, 11])
    so = oracle.student_image_forward(sd, image, 24)
    close(so['last_representation'], g['img1.s.last_representation'], rtol=2e-4, atol=2e-5)
    close(to['last_representation'], g['img1.t.last_representation'], rtol=2e-4, atol=2e-5)
    loss, res = oracle.LossOracle(['out_l1', 'out_cos'])(so, to, 'image')
    close(loss, g['img1.loss'], rtol=2e-5)
    for k in res:
        close(res[k], g['img1.term.' + k], rtol=2e-5)
    loss.backward()
    trainable = {n: p for n, p in sd.items() if p.requires_grad}
    _assert_grads(g, 'img1.l1cos', _grads(trainable), 68 - 3, tol=2e-3)      # (out_l1: sign(s - t) / n, piecewise constant)
    for p in sd.values():
        p.grad = None
    loss2, _ = oracle.LossOracle(['out_cos'])(oracle.student_image_forward(sd, image, 24), to, 'image')
    close(loss2, g['img1.cos.loss'], rtol=2e-5)
    loss2.backward()
    _assert_grads(g, 'img1.cos', _grads(trainable), 68 - 3)
    assert all(sd[n].grad is None for n in rc.FROZEN_IMAGE)


def test_text_yaml_real_shapes_match_the_reference(golden_dir):
    """config/final_config/text.yaml (one text tower, compression_embedding=True, losses out_l1 + out_cos)"""
    import real_cases as rc
    g = rc.load(golden_dir, 'real_b4_textc.npz')
    text, tsd, sd = rc.textc_inputs(g)
    for v in sd.values():
        v.requires_grad_(True)
    with torch.no_grad():
        to = oracle.teacher_text_forward(tsd, text)
    so = oracle.student_text_forward(sd, text, 12)
    close(so['last_representation'], g['txtc.s.last_representation'], rtol=2e-4, atol=2e-5)
    close(to['last_representation'], g['txtc.t.last_representation'], rtol=2e-4, atol=2e-5)
    loss, res = oracle.LossOracle(['out_l1', 'out_cos'])(so, to, 'text')
    close(loss, g['txtc.loss'], rtol=2e-5)
    for k in res:
        close(res[k], g['txtc.term.' + k], rtol=2e-5)
    loss.backward()
    _assert_grads(g, 'txtc.l1cos', _grads(sd), len(sd), tol=2e-3)
    for p in sd.values():
        p.grad = None
    loss2, _ = oracle.LossOracle(['out_cos'])(oracle.student_text_forward(sd, text, 12), to, 'text')
    close(loss2, g['txtc.cos.loss'], rtol=2e-5)
    loss2.backward()
    _assert_grads(g, 'txtc.cos', _grads(sd), len(sd))


def test_l_clip_336px_real_shapes_match_the_reference(golden_dir):
    """l_clip dual at 336 px (101 tokens: the stride-32 conv floors, reference _common.py:176,196), losses out_l1 + out_cos + 0.1 cos_diff"""
    import real_cases as rc
    g = rc.load(golden_dir, 'real_b4_336.npz')
    image, text, tsd, sdi, sdt = rc.l336_inputs(g)
    for v in list(sdi.values()) + list(sdt.values()):
        v.requires_grad_(True)
    with torch.no_grad():
        ti = oracle.teacher_image_forward({k: v for k, v in tsd.items() if k.startswith('visual.')}, image)
        tt = oracle.teacher_text_forward({k: v for k, v in tsd.items() if not k.startswith('visual.')}, text)
        to = oracle.clip_forward(ti, tt)
    oi, ot = oracle.student_image_forward(sdi, image, 24), oracle.student_text_forward(sdt, text, 12)
    so = oracle.clip_forward(oi, ot)
    for tag, o in (('s_img', oi), ('s_txt', ot), ('t_img', ti), ('t_txt', tt)):
        close(o['last_representation'], g[f'{tag}.last_representation'], rtol=2e-4, atol=2e-5)
    close(so['i2t_logits'], g['s.i2t_logits'], rtol=2e-4, atol=2e-5)
    loss, res = oracle.LossOracle(['out_l1', 'out_cos', 'cos_diff'], {'cos_diff': 0.1})(so, to, 'all')
    close(loss, g['loss'], rtol=2e-5)
    for k in res:
        close(res[k], g['term.' + k], rtol=2e-4, atol=1e-6)
    loss2, _ = oracle.LossOracle(['out_cos'])(so, to, 'all')
    close(loss2, g['cos.loss'], rtol=2e-5)
    loss2.backward()
    _assert_grads(g, 'cos.s_img', _grads(sdi), 68)
    _assert_grads(g, 'cos.s_txt', _grads(sdt), 44)


def _clip_student_oracle(c, image, text, tsd_i, tsd_t, sd_i, sd_t, names, scale=None):
    with torch.no_grad():
        to = oracle.clip_forward(oracle.teacher_image_forward(tsd_i, image, c['tea_heads'], c['need_layers'], True, True),
                                 oracle.teacher_text_forward(tsd_t, text, c['tea_heads'], c['need_layers'], True, True))
    oi = oracle.clip_student_image_forward(sd_i, image, c['heads'], True, True)
    ot = oracle.clip_student_text_forward(sd_t, text, c['heads'], True, True)
    so = oracle.clip_forward(oi, ot)
    loss, res = oracle.LossOracle(names, scale)(so, to, 'all')
    return so, to, loss, res


@pytest.mark.parametrize('case', ['tiny', 'real'])
def test_plain_clip_encoders_as_students_match_the_reference(golden_dir, case):
    """ImageEncoder / TextEncoder with is_student=True (reference image_encoder.py:16-25,54-59 ; text_encoder.py:41-47,75-80): the CLIP
    architecture with gradients + embedding_projection / hidden_projection on the exported states, under a CLIP teacher pair; five loss
    terms (the two feature-MSE terms read the projected states), then the gradient of EVERY parameter for the smooth objective."""
    import real_cases as rc
    c = rc.CLIPSTU_TINY if case == 'tiny' else rc.CLIPSTU_REAL
    g = rc.load(golden_dir, 'clip_student_tiny.npz' if case == 'tiny' else 'real_b4_clipstu.npz')
    image, text, tsd_i, tsd_t, sd_i, sd_t = rc.clipstu_inputs(g, c)
    if case == 'tiny':
        np.testing.assert_array_equal(image.numpy(), g['image'])
        np.testing.assert_array_equal(text.numpy(), g['text'])
    for v in list(sd_i.values()) + list(sd_t.values()):
        v.requires_grad_(True)
    so, to, loss, res = _clip_student_oracle(c, image, text, tsd_i, tsd_t, sd_i, sd_t, rc.CLIPSTU_LOSSES, {'cos_diff': 0.1})
    for tag, o in (('s_img', so['visual_output']), ('s_txt', so['text_output']), ('t_img', to['visual_output']), ('t_txt', to['text_output'])):
        close(o['last_representation'], g[f'{tag}.last_representation'], rtol=2e-4, atol=2e-5)
        if case == 'tiny':
            close(o['embedding'], g[f'{tag}.embedding'], rtol=2e-4, atol=2e-5)
            assert len(o['representations']) == c['layers']
            for i, r in enumerate(o['representations']):
                close(r, g[f'{tag}.rep{i}'], rtol=2e-4, atol=5e-5)
    close(loss, g['loss'], rtol=2e-5)
    assert set(res) == {k[5:] for k in g if k.startswith('term.')}
    for k in res:
        close(res[k], g['term.' + k], rtol=2e-4, atol=1e-6)
    so, to, loss2, res2 = _clip_student_oracle(c, image, text, tsd_i, tsd_t, sd_i, sd_t, rc.CLIPSTU_SMOOTH)
    close(loss2, g['smooth.loss'], rtol=2e-5)
    loss2.backward()
    if case == 'tiny':
        n = 0
        for tag, sd in (('s_img', sd_i), ('s_txt', sd_t)):
            for name, p in sd.items():
                ref = g[f'smooth.{tag}.grad.{name}']
                np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-3, atol=2e-4 * max(np.abs(ref).max(), 1e-12), err_msg=f'{tag}.{name}')
                n += 1
        assert n == len(sd_i) + len(sd_t) == (5 + 24 + 3 + 4) + (2 + 24 + 3 + 4)
    else:
        _assert_grads(g, 'smooth.s_img', _grads(sd_i), len(sd_i))
        _assert_grads(g, 'smooth.s_txt', _grads(sd_t), len(sd_t))


def test_metrics_known_answers():
    """validation metrics restatement (oracle/metrics.py) on hand-computable cases: a permutation structure fixes every
    rank, and the diagonal scores follow from the logits in closed form."""
    from oracle import metrics as om
    n = 8
    # one-hot rows scaled arbitrarily (norm_and_logits must remove the scale); row i matches caption i exactly
    img = torch.eye(n) * torch.arange(1, n + 1)[:, None].float()
    txt = torch.eye(n) * 3.0
    m = om.retrieval_metrics(img, txt, k_list=(1, 3))
    assert m['acc_top1'].item() == 1.0 and m['acc_top3'].item() == 1.0
    assert m['ranks'].tolist() == [0] * n
    assert abs(m['mean_score'].item() - 1.0) < 1e-12
    # logits row = (1, 0, ..., 0) up to position: softmax diagonal = e / (e + n - 1)
    e = float(np.e)
    assert abs(m['softmax_mean_score'].item() - e / (e + n - 1)) < 1e-12
    # rank structure: caption j scores c - |i - j| * step against image i, shifted so that image i's best caption is i + s
    ang = torch.linspace(0.0, 1.0, n)
    base = torch.stack([torch.cos(ang), torch.sin(ang)], 1)               # unit vectors on an arc, angle gap 1/7
    shift = 2
    m2 = om.retrieval_metrics(base, base.roll(-shift, 0), k_list=(1, 3, 5))
    # image i best matches the caption equal to itself, which sits at index i - shift: for i >= shift the label caption i
    # (vector i + shift) is `shift` steps away: captions at distance < shift on both sides beat it
    lg, _ = om.norm_and_logits(base.double(), base.roll(-shift, 0).double())
    want = (lg > torch.diagonal(lg)[:, None]).sum(1)
    assert m2['ranks'].tolist() == want.tolist()
    for k in (1, 3, 5):
        assert abs(m2[f'acc_top{k}'].item() - (want < k).double().mean().item()) < 1e-12
    assert m2['acc_top1'].item() < m2['acc_top3'].item() <= m2['acc_top5'].item()


# ---------------------------------------------------------------------------------------------------------------------
# B = 4 plumbing run as a loss TRAJECTORY of the reference (tools/golden/gen_golden.py:trajectory):
# forward + backward + AdamW + per-epoch cosine/warm-up schedule, 16 steps (tiny) / 4 steps (real shapes)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def traj(golden_dir):
    return dict(np.load(os.path.join(golden_dir, 'trajectory.npz')))


def _oracle_train(sd_i, sd_t, t_i, t_t, heads_i, heads_t, batches, lr, wd, warm, total, steps_per_epoch):
    from distillclip_amd.optim import cosine_with_warmup
    params = list(sd_i.values()) + list(sd_t.values())
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=wd)
    lc = oracle.LossOracle(['out_l1', 'out_cos', 'cos_diff'], {'cos_diff': 0.1})
    losses, epoch = [], 0
    for i, (image, text) in enumerate(batches):
        for g in opt.param_groups:
            g['lr'] = lr * cosine_with_warmup(epoch, warm, total)
        so = oracle.clip_forward(oracle.student_image_forward(sd_i, image, heads_i), oracle.student_text_forward(sd_t, text, heads_t))
        with torch.no_grad():
            to = oracle.clip_forward(oracle.teacher_image_forward(t_i, image), oracle.teacher_text_forward(t_t, text))
        loss, _ = lc(so, to, 'all')
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if (i + 1) % steps_per_epoch == 0:
            epoch += 1
    return losses


def test_tiny_16_step_trajectory_matches_reference(traj):
    c = TINY
    seed, B, n = int(traj['tiny.seed']), int(traj['tiny.B']), 64
    images = torch.from_numpy(synth.images(seed, n, c['res']))
    texts = torch.from_numpy(synth.captions(seed, n, c['ctx'], c['vocab'], 3, 9))
    sd_i = T(synth.student_image_state(seed, **c['s_img']), grad=True)
    sd_t = T(synth.student_text_state(seed, **c['s_txt']), grad=True)
    t_i = T(synth.teacher_image_state(seed, 128, 2, c['patch'], c['res'], c['out_dim']))
    t_t = T(synth.teacher_text_state(seed, 128, 2, c['ctx'], c['vocab'], c['out_dim']))
    batches = [(images[i:i + B], texts[i:i + B]) for i in range(0, n, B)]
    losses = _oracle_train(sd_i, sd_t, t_i, t_t, 4, 2, batches, float(traj['tiny.base_lr']), float(traj['tiny.wd']),
                           int(traj['tiny.warm']), int(traj['tiny.total']), int(traj['tiny.steps_per_epoch']))
    assert len(losses) == 16
    np.testing.assert_allclose(losses, traj['tiny.loss'], rtol=2e-4)
    # epoch 0 of a warm-up schedule runs at lr 0 (HF multiplier 0 / warm): the first 4 batches see the initial weights
    assert traj['tiny.lr'][0] == 0.0 and traj['tiny.lr'][4] > 0
    # final weights after 12 effective AdamW steps.  Per-tensor relative L2: the k-part of attn.qkv.bias has an exactly-zero
    # true gradient (softmax is shift-invariant), so Adam normalises pure rounding noise there and its elements move by
    # +-lr per step in a run-dependent direction; measured over whole tensors that is < 1e-2 of the bias norm.
    worst = {}
    for tag, sd in (('s_img', sd_i), ('s_txt', sd_t)):
        for name, p in sd.items():
            ref = traj[f'tiny.{tag}.final.{name}']
            worst[f'{tag}.{name}'] = float(np.linalg.norm(p.detach().numpy() - ref) / (np.linalg.norm(ref) + 1e-12))
    bad = {k: v for k, v in worst.items() if v > (2e-2 if 'qkv.bias' in k else 1e-3)}
    assert not bad, bad


def test_real_shapes_4_step_trajectory_matches_reference(traj):
    seed, B, n = int(traj['real.seed']), int(traj['real.B']), 16
    cfg_i = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24, mlp_ratio=4.0,
                 qkv_bias=True, repeated_times=2, use_transform=True)
    cfg_t = dict(depth=4, repeated_times=2, use_transform=True)
    images = torch.from_numpy(synth.images(seed, n, 224))
    texts = torch.from_numpy(synth.captions(seed, n))
    sd_i, sd_t = T(synth.student_image_state(seed, **cfg_i), grad=True), T(synth.student_text_state(seed, **cfg_t), grad=True)
    t_i, t_t = T(synth.teacher_image_state(seed)), T(synth.teacher_text_state(seed))
    batches = [(images[i:i + B], texts[i:i + B]) for i in range(0, n, B)]
    losses = _oracle_train(sd_i, sd_t, t_i, t_t, 24, 12, batches, float(traj['real.base_lr']), float(traj['real.wd']), 0, 300, 10 ** 9)
    np.testing.assert_allclose(losses, traj['real.loss'], rtol=5e-4)


def test_bf16_matched_mode_stays_within_bf16_noise_of_the_pinned_oracle(tiny, tiny_models):
    """oracle.bf16_matched() only inserts bf16 round-trips at the HIP path's storage points (oracle/encoders.py header): the
    arithmetic between them is the pinned restatement's, so values and gradients stay within bf16 noise of the fp32 goldens."""
    image, text = torch.from_numpy(tiny['image']), torch.from_numpy(tiny['text'])
    m = {k: {n: v.detach().clone().requires_grad_(v.requires_grad) for n, v in sd.items()} for k, sd in tiny_models.items()}
    lc = oracle.LossOracle(['out_l1', 'out_cos', 'cos_diff'], {'cos_diff': 0.1})
    with oracle.bf16_matched():
        so = oracle.clip_forward(oracle.student_image_forward(m['s_img'], image, 4), oracle.student_text_forward(m['s_txt'], text, 2))
        with torch.no_grad():
            to = oracle.clip_forward(oracle.teacher_image_forward(m['t_img'], image), oracle.teacher_text_forward(m['t_txt'], text))
        loss, _ = lc(so, to, 'all')
        loss.backward()
    for tag, o in (('s_img', so['visual_output']), ('s_txt', so['text_output']), ('t_img', to['visual_output']), ('t_txt', to['text_output'])):
        ref = tiny[f'{tag}.last_representation']
        e = np.linalg.norm(o['last_representation'].detach().numpy() - ref) / np.linalg.norm(ref)
        assert 1e-5 < e < 2e-2, (tag, e)                     # it DOES round (e > 0) and stays at bf16 level
    assert abs(loss.item() - float(tiny['lclip.loss'])) < 2e-2 * float(tiny['lclip.loss'])
    g = m['s_img']['head.weight'].grad.numpy()
    ref = tiny['lclip.s_img.grad.head.weight']
    assert np.linalg.norm(g - ref) / np.linalg.norm(ref) < 2e-1
    # and the mode is scoped: outside the context the restatement is exact fp32 again
    out = oracle.student_image_forward(tiny_models['s_img'], image, 4)
    close(out['last_representation'], tiny['s_img.last_representation'])
