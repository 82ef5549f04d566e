"""SURVEY.md §8d config 1 on the HIP path: the B = 4 plumbing run as a loss TRAJECTORY against the reference's own
(tests/golden/trajectory.npz, written by tools/golden/gen_golden.py:trajectory from the reference's modules + torch AdamW + the HF
cosine/warm-up schedule stepped per epoch).  The only end-to-end pin of forward + backward + fused AdamW + schedule over time."""
import os

import numpy as np
import pytest
import torch

from distillclip_amd import synth

pytestmark = pytest.mark.gpu

TINY = dict(
    res=32, patch=8, ctx=13, vocab=97, out_dim=64,
    s_img=dict(img_size=32, patch_size=8, in_chans=3, out_dim=64, embed_dim=128, depth=4, num_heads=4,
               mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True),
    s_txt=dict(vocab_size=97, context_length=13, out_dim=64, embed_dim=128, depth=2, num_heads=2,
               mlp_ratio=4.0, qkv_bias=False, repeated_times=2, use_transform=True),
)
LOSS = dict(loss_name=['out_l1', 'out_cos', 'cos_diff'], loss_scale={'cos_diff': 0.1})


def T(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


@pytest.fixture(scope='module')
def traj(golden_dir):
    return dict(np.load(os.path.join(golden_dir, 'trajectory.npz')))


def _run(model, batches, steps_per_epoch):
    (opt,), (sched,) = model.configure_optimizers()
    losses, lrs = [], []
    for i, (image, text) in enumerate(batches):
        loss = model.training_step([image.cuda(), text.cuda()])
        opt.zero_grad()
        model.backward_and_sync(loss)
        opt.step()
        losses.append(loss.item())
        lrs.append(opt.lr)
        if (i + 1) % steps_per_epoch == 0:
            sched.step()
    torch.cuda.synchronize()
    return losses, lrs


def test_tiny_16_step_loss_trajectory_vs_reference(traj):
    from distillclip_amd.model import DualDistillModel
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    c = TINY
    seed, B, n = int(traj['tiny.seed']), int(traj['tiny.B']), 64
    images = torch.from_numpy(synth.images(seed, n, c['res']))
    texts = torch.from_numpy(synth.captions(seed, n, c['ctx'], c['vocab'], 3, 9))
    s_img, s_txt = RepeatVisionTransformer(**c['s_img']), RepeatTextTransformer(**c['s_txt'])
    s_img.load_state_dict(T(synth.student_image_state(seed, **c['s_img'])))
    s_txt.load_state_dict(T(synth.student_text_state(seed, **c['s_txt'])))
    tsd = synth.teacher_image_state(seed, 128, 2, c['patch'], c['res'], c['out_dim'])
    tsd.update(synth.teacher_text_state(seed, 128, 2, c['ctx'], c['vocab'], c['out_dim']))
    model = DualDistillModel(s_img, s_txt, LOSS, warm_steps=int(traj['tiny.warm']), total_steps=int(traj['tiny.total']),
                             weight_decay=float(traj['tiny.wd']), lr=float(traj['tiny.base_lr']), download_root='.',
                             teacher_state_dict=T(tsd)).cuda()
    batches = [(images[i:i + B], texts[i:i + B]) for i in range(0, n, B)]
    losses, lrs = _run(model, batches, int(traj['tiny.steps_per_epoch']))
    ref = traj['tiny.loss']
    np.testing.assert_allclose(lrs, traj['tiny.lr'], rtol=1e-6, atol=1e-12)         # per-epoch HF cosine/warm-up, epoch 0 at lr 0
    err = np.abs(np.asarray(losses) - ref) / np.abs(ref)
    print('tiny trajectory rel err per step', np.round(err, 4))
    # the first epoch runs at lr 0 (4 different batches, initial weights): pure forward parity
    assert err[:4].max() < 5e-3, err[:4]              # measured <= 7e-4
    # 12 AdamW steps later the bf16 path still follows the fp32 reference (errors compound through the weights)
    assert err.max() < 2e-2, err                      # measured <= 4e-3
    assert losses[-1] < 0.5 * losses[0]                                              # and it trains: 0.74 -> 0.26 in the reference
    # final weights: relative L2 of what training moved (delta from the initial weights) for the large tensors
    init = {'s_img': T(synth.student_image_state(seed, **c['s_img'])), 's_txt': T(synth.student_text_state(seed, **c['s_txt']))}
    for tag, mod in (('s_img', model.student.image_encoder), ('s_txt', model.student.text_encoder)):
        for name in ('head.weight', 'blocks.0.block.mlp.fc1.weight', 'blocks.0.block.attn.proj.weight'):
            got = dict(mod.named_parameters())[name].detach().cpu().numpy() - init[tag][name].numpy()
            want = traj[f'tiny.{tag}.final.{name}'] - init[tag][name].numpy()
            e = np.linalg.norm(got - want) / np.linalg.norm(want)
            assert e < 0.35, (tag, name, e)     # Adam's sign-like update amplifies small gradient differences; the loss curve is the pin


def test_real_shapes_4_step_loss_trajectory_vs_reference(traj):
    from distillclip_amd.model import DualDistillModel
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    seed, B, n = int(traj['real.seed']), int(traj['real.B']), 16
    cfg_i = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24, mlp_ratio=4.0,
                 qkv_bias=True, repeated_times=2, use_transform=True)
    cfg_t = dict(depth=4, repeated_times=2, use_transform=True)
    images = torch.from_numpy(synth.images(seed, n, 224))
    texts = torch.from_numpy(synth.captions(seed, n))
    s_img, s_txt = RepeatVisionTransformer(**cfg_i), RepeatTextTransformer(**cfg_t)
    s_img.load_state_dict(T(synth.student_image_state(seed, **cfg_i)))
    s_txt.load_state_dict(T(synth.student_text_state(seed, **cfg_t)))
    tsd = synth.teacher_image_state(seed)
    tsd.update(synth.teacher_text_state(seed))
    model = DualDistillModel(s_img, s_txt, LOSS, warm_steps=0, total_steps=300, weight_decay=float(traj['real.wd']),
                             lr=float(traj['real.base_lr']), download_root='.', teacher_state_dict=T(tsd)).cuda()
    batches = [(images[i:i + B], texts[i:i + B]) for i in range(0, n, B)]
    losses, _ = _run(model, batches, 10 ** 9)
    ref = traj['real.loss']
    err = np.abs(np.asarray(losses) - ref) / np.abs(ref)
    print('real trajectory', np.round(losses, 5), 'ref', np.round(ref, 5), 'rel err', np.round(err, 4))
    assert err[0] < 5e-3 and err.max() < 2e-2, err     # measured 3e-4 / 5e-3
    # how far 4 AdamW steps moved every parameter (norm of the delta): the optimizer's arithmetic at real shapes
    init = {'s_img': T(synth.student_image_state(seed, **cfg_i)), 's_txt': T(synth.student_text_state(seed, **cfg_t))}
    worst = {}
    for tag, mod in (('s_img', model.student.image_encoder), ('s_txt', model.student.text_encoder)):
        for name, p in mod.named_parameters():
            d = (p.detach().cpu() - init[tag][name]).norm().item()
            want = float(traj[f'real.{tag}.dnorm.{name}'])
            worst[f'{tag}.{name}'] = abs(d - want) / (want + 1e-12)
    # (the k-part of attn.qkv.bias has a zero true gradient: Adam normalises rounding noise there, run-dependent direction)
    bad = {k: v for k, v in worst.items() if v > (0.35 if 'qkv.bias' in k else 0.1)}
    assert not bad, bad
