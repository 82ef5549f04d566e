"""Retrieval-metrics kernel (dclip_retrieval_metrics through the C ABI) vs oracle/metrics.py (reference
dual_distill_model.py:204-224, :271-275)."""
import pytest
import torch

from oracle import metrics as om

pytestmark = pytest.mark.gpu


def _emb(n, E, seed, corr=0.6):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(n, E, generator=g)
    txt = corr * img + (1 - corr) * torch.randn(n, E, generator=g)         # matching pairs correlated, like trained towers
    return img * torch.rand(n, 1, generator=g).add(0.5), txt * 3.0         # arbitrary row scales: the kernel normalises


@pytest.mark.parametrize('n,E', [(1, 16), (7, 32), (16, 64), (100, 512), (257, 512), (1000, 512), (513, 1024)])
def test_metrics_match_oracle(n, E):
    from distillclip_amd.metrics import retrieval_metrics
    img, txt = _emb(n, E, 1000 + n, corr=0.15 if n > 50 else 0.5)
    want = om.retrieval_metrics(img, txt)
    got = retrieval_metrics(img.cuda(), txt.cuda(), return_ranks=True)
    # integer work is bit-exact unless two f32 logits tie to the last bit against the f64 oracle: allow none at these sizes
    assert got['ranks'].cpu().tolist() == want['ranks'].tolist()
    for k in (1, 3, 5, 10, 20, 50):
        assert abs(got[f'acc_top{k}'].item() - want[f'acc_top{k}'].item()) < 1e-6
    assert abs(got['softmax_mean_score'].item() - want['softmax_mean_score'].item()) < 2e-6
    assert abs(got['mean_score'].item() - want['mean_score'].item()) < 2e-6


def test_metrics_val_set_size_properties():
    """n = 5000 (the COCO val split of the reference's validation loop): size-independent properties"""
    from distillclip_amd.metrics import retrieval_metrics
    n, E = 5000, 512
    img, txt = _emb(n, E, 77, corr=0.12)
    got = retrieval_metrics(img.cuda(), txt.cuda(), return_ranks=True)
    accs = [got[f'acc_top{k}'].item() for k in (1, 3, 5, 10, 20, 50)]
    assert all(a <= b + 1e-7 for a, b in zip(accs, accs[1:]))            # monotone in k
    r = got['ranks'].cpu()
    assert r.min().item() >= 0 and r.max().item() < n
    for k, a in zip((1, 3, 5, 10, 20, 50), accs):
        assert abs(a - (r < k).float().mean().item()) < 1e-6             # accuracies are the rank histogram
    # permuting the pairs together permutes the ranks
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(5))
    got2 = retrieval_metrics(img[perm].cuda(), txt[perm].cuda(), return_ranks=True)
    assert torch.equal(got2['ranks'].cpu(), r[perm])
    # a sample of rows against the f64 oracle
    rows = torch.arange(0, n, 97)
    lg = (img.double() / img.double().norm(dim=1, keepdim=True))[rows] @ (txt.double() / txt.double().norm(dim=1, keepdim=True)).t()
    d = lg[torch.arange(len(rows)), rows]
    assert ((lg > d[:, None]).sum(1) == r[rows]).all()
    # identical towers: every row is its own best match
    same = retrieval_metrics(img.cuda(), img.cuda(), return_ranks=True)
    assert same['acc_top1'].item() == 1.0 and same['ranks'].abs().sum().item() == 0
    assert abs(same['mean_score'].item() - 1.0) < 1e-5


def test_metrics_argument_errors():
    from distillclip_amd.metrics import retrieval_metrics
    x = torch.randn(8, 24, device='cuda')
    with pytest.raises(ValueError):
        retrieval_metrics(x, x)                                           # E % 16 != 0
    with pytest.raises(ValueError):
        retrieval_metrics(torch.randn(8, 32, device='cuda'), torch.randn(9, 32, device='cuda'))
    with pytest.raises(ValueError):
        retrieval_metrics(torch.randn(8, 32, device='cuda'), torch.randn(8, 32, device='cuda'), k_list=range(1, 11))
    with pytest.raises(RuntimeError):
        retrieval_metrics(torch.randn(8, 32), torch.randn(8, 32))


def test_dual_model_validation_hooks_vs_oracle():
    """DualDistillModel.validation_step / validation_epoch_end (reference dual_distill_model.py:129-187): the logged keys
    and values, from oracle towers + oracle metrics on the same seeded inputs (tiny towers, E = 64)."""
    import numpy as np
    import oracle
    from distillclip_amd import synth
    from distillclip_amd.model import DualDistillModel
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    seed, B = 23, 12
    s_img_cfg = dict(img_size=32, patch_size=8, in_chans=3, out_dim=64, embed_dim=128, depth=4, num_heads=4,
                     mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
    s_txt_cfg = dict(vocab_size=97, context_length=13, out_dim=64, embed_dim=128, depth=2, num_heads=2,
                     mlp_ratio=4.0, qkv_bias=False, repeated_times=2, use_transform=True)
    T = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
    sd_i, sd_t = T(synth.student_image_state(seed, **s_img_cfg)), T(synth.student_text_state(seed, **s_txt_cfg))
    tsd = synth.teacher_image_state(seed, 128, 2, 8, 32, 64)
    tsd.update(synth.teacher_text_state(seed, 128, 2, 13, 97, 64))
    tsd = T(tsd)
    s_img, s_txt = RepeatVisionTransformer(**s_img_cfg), RepeatTextTransformer(**s_txt_cfg)
    s_img.load_state_dict(sd_i)
    s_txt.load_state_dict(sd_t)
    model = DualDistillModel(s_img, s_txt, dict(loss_name=['out_l1', 'out_cos', 'cos_diff'], loss_scale={'cos_diff': 0.1}),
                             warm_steps=1, total_steps=10, weight_decay=1e-3, lr=1e-4, download_root='.',
                             teacher_state_dict=tsd).cuda().eval()
    outs, logs = [], []
    reps = {k: [] for k in ('si', 'st', 'ti', 'tt')}
    for it in range(2):
        image = torch.from_numpy(synth.images(seed + it, B, 32))
        text = torch.from_numpy(synth.captions(seed + it, B, 13, 97, 3, 9))
        o, lg = model.validation_step([image.cuda(), text.cuda()], it)
        outs.append(o)
        logs.append(lg)
        with torch.no_grad():
            oi, ot = oracle.student_image_forward(sd_i, image, 4), oracle.student_text_forward(sd_t, text, 2)
            ti = oracle.teacher_image_forward({k: v for k, v in tsd.items() if k.startswith('visual.')}, image)
            tt = oracle.teacher_text_forward({k: v for k, v in tsd.items() if not k.startswith('visual.')}, text)
            ref, _ = oracle.LossOracle(['out_l1', 'out_cos', 'cos_diff'], {'cos_diff': 0.1})(
                oracle.clip_forward(oi, ot), oracle.clip_forward(ti, tt), 'all')
        for k, v in zip(('si', 'st', 'ti', 'tt'), (oi, ot, ti, tt)):
            reps[k].append(v['last_representation'])
        assert abs(lg['val_loss/loss'].item() - ref.item()) <= 2e-2 * abs(ref.item())
        # per-batch metrics computed on the model's own (bf16-operand) representations vs the oracle metrics of those
        want = om.retrieval_metrics(o['stu_image_outs'].cpu(), o['stu_text_outs'].cpu())
        for k in (1, 3, 5, 10, 20, 50):
            assert abs(lg[f'val_step/stu_acc_top{k}'].item() - want[f'acc_top{k}'].item()) < 1e-6
        assert abs(lg['val_step/stu_softmax_mean_score'].item() - want['softmax_mean_score'].item()) < 2e-6
        assert abs(lg['val_step/stu_mean_score'].item() - want['mean_score'].item()) < 2e-6
        assert 'val_step/tea_acc_top1' in lg and 'val_step/tea_mean_score' not in lg
    ep = model.validation_epoch_end(outs)
    expect_keys = {f'val_stu_acc/stu_acc_top{k}' for k in model.k_list} | \
        {f'val_stu_image_tea_text/stu_image_tea_text_acc_top{k}' for k in model.k_list} | \
        {f'val_stu_text_tea_image/stu_text_tea_image_acc_top{k}' for k in model.k_list} | \
        {'val_stu_score/stu_softmax_mean_score', 'val_stu_score/stu_mean_score', 'val_tea_score/tea_softmax_mean_score',
         'val_tea_score/tea_mean_score'} | {f'val_tea_acc/tea_acc_top{k}' for k in model.k_list}
    assert set(ep) == expect_keys
    cat = {k: torch.cat(v) for k, v in reps.items()}
    # scores from the oracle towers (fp32) vs the HIP towers: representation-level tolerance
    want = om.retrieval_metrics(cat['si'], cat['st'])
    assert abs(ep['val_stu_score/stu_mean_score'].item() - want['mean_score'].item()) < 2e-2
    assert abs(ep['val_stu_score/stu_softmax_mean_score'].item() - want['softmax_mean_score'].item()) < 2e-3
    want_t = om.retrieval_metrics(cat['ti'], cat['tt'])
    assert abs(ep['val_tea_score/tea_mean_score'].item() - want_t['mean_score'].item()) < 2e-2
    model.current_epoch = 1
    assert not any(k.startswith('val_tea') for k in model.validation_epoch_end(outs))
