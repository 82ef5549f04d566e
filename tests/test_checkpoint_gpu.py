"""Checkpoint layout + resume (reference dual_distill_model.py:22-38 load_weight, Lightning's
ModelCheckpoint layout, distil_model.py:160-169 optimizer construction)."""
import io
import os

import numpy as np
import pytest
import torch

from distillclip_amd import synth, checkpoint

pytestmark = pytest.mark.gpu

S_IMG = dict(img_size=32, patch_size=8, in_chans=3, out_dim=64, embed_dim=128, depth=4, num_heads=4,
             mlp_ratio=4.0, qkv_bias=True, repeated_times=2, use_transform=True)
S_TXT = dict(vocab_size=97, context_length=13, out_dim=64, embed_dim=128, depth=2, num_heads=2,
             mlp_ratio=4.0, qkv_bias=False, repeated_times=2, use_transform=True)
T = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


def _teacher_sd(seed):
    tsd = synth.teacher_image_state(seed, 128, 2, 8, 32, 64)
    tsd.update(synth.teacher_text_state(seed, 128, 2, 13, 97, 64))
    return T(tsd)


def _dual(seed, lr=1e-3):
    from distillclip_amd.model import DualDistillModel
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    s_img, s_txt = RepeatVisionTransformer(**S_IMG), RepeatTextTransformer(**S_TXT)
    s_img.load_state_dict(T(synth.student_image_state(seed, **S_IMG)))
    s_txt.load_state_dict(T(synth.student_text_state(seed, **S_TXT)))
    m = DualDistillModel(s_img, s_txt, dict(loss_name=['out_cos', 'cos_diff'], loss_scale={'cos_diff': 0.1}),
                         warm_steps=2, total_steps=10, weight_decay=1e-2, lr=lr, download_root='.',
                         teacher_state_dict=_teacher_sd(7)).cuda()
    (opt,), (sched,) = m.configure_optimizers()
    return m, opt, sched


def _batch(i, B=6):
    return [torch.from_numpy(synth.images(100 + i, B, 32)).cuda(), torch.from_numpy(synth.captions(100 + i, B, 13, 97, 3, 9)).cuda()]


def _step(m, opt, i):
    opt.zero_grad()
    loss = m.training_step(_batch(i))
    m.backward_and_sync(loss)
    opt.step()
    return loss.item()


def test_save_resume_continues_the_run(tmp_path):
    from distillclip_amd.checkpoint import save_checkpoint, load_checkpoint
    m, opt, sched = _dual(3)
    for i in range(2):
        _step(m, opt, i)
    sched.step()
    path = os.path.join(tmp_path, 'epoch0.ckpt')
    save_checkpoint(path, m, opt, sched, epoch=1, global_step=2)
    l3 = _step(m, opt, 2)
    want = {k: v.detach().clone() for k, v in m.student.state_dict().items()}

    m2, opt2, sched2 = _dual(99)                       # different initial weights: everything must come from the file
    epoch, gstep = load_checkpoint(path, m2, opt2, sched2)
    assert (epoch, gstep) == (1, 2) and m2.current_epoch == 1
    assert opt2.step_count == 2 and abs(opt2.lr - opt.lr) < 1e-12 and sched2.epoch == 1
    l3b = _step(m2, opt2, 2)
    assert abs(l3 - l3b) <= 1e-5 * abs(l3), (l3, l3b)
    got = m2.student.state_dict()
    for k, v in want.items():
        err = (got[k] - v).norm().item() / (v.norm().item() + 1e-12)
        assert err < 2e-4, (k, err)                      # wgrad accumulates with f32 atomics: not bit-reproducible


def test_layout_is_the_references(tmp_path):
    """keys carry Lightning's 'student.' / 'teacher.' prefixes; the optimizer entry loads into torch.optim.AdamW built the
    way the reference builds it, and that AdamW then takes the same step as the fused kernel"""
    from distillclip_amd.checkpoint import checkpoint_dict, student_state_dict, trainable_parameters
    m, opt, sched = _dual(5)
    _step(m, opt, 0)
    ck = checkpoint_dict(m, opt, sched, epoch=0, global_step=1)
    buf = io.BytesIO()
    torch.save(ck, buf)
    buf.seek(0)
    ck = torch.load(buf, map_location='cpu')             # weights_only load: plain containers and tensors only
    keys = list(ck['state_dict'])
    assert all(k.startswith(('student.', 'teacher.')) for k in keys)
    assert 'student.image_encoder.blocks.0.block.attn.conv_l.instances.1.weight' in keys
    assert 'student.text_encoder.patch_embed.weight' in keys
    assert 'teacher.image_encoder.visual.transformer.resblocks.1.attn.in_proj_weight' in keys
    assert 'teacher.text_encoder.token_embedding.weight' in keys
    stu = student_state_dict(ck)
    assert set(stu) == {k for k in m.student.state_dict()}
    # torch AdamW over CPU copies of the trainable parameters, in the reference's order
    params = trainable_parameters(m)
    cpu = [torch.nn.Parameter(p.detach().cpu().clone()) for p in params]
    ref = torch.optim.AdamW(cpu, lr=m.hparams.lr, weight_decay=m.hparams.weight_decay)
    ref.load_state_dict(ck['optimizer_states'][0])
    assert len(ref.state) == len(cpu)
    # next step with identical gradients on both sides
    opt.zero_grad()
    loss = m.training_step(_batch(1))
    m.backward_and_sync(loss)
    for c, p in zip(cpu, params):
        c.grad = p.grad.detach().cpu().clone()
    opt.step()
    ref.step()
    for c, p in zip(cpu, params):
        err = (p.detach().cpu() - c.detach()).abs().max().item()
        assert err <= 1e-6 + 1e-5 * c.detach().abs().max().item(), err


def test_stage1_files_feed_load_weight(tmp_path):
    """reference l_clip.yaml load_path: two one-tower stage-1 checkpoints -> dual students (dual_distill_model.py:22-38)"""
    from distillclip_amd.checkpoint import save_checkpoint
    from distillclip_amd.model import DistillModel
    from distillclip_amd.model.dual_distill_model import load_weight
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    tsd = _teacher_sd(7)
    paths = {}
    for kind, cls, cfg, state in (('image', RepeatVisionTransformer, S_IMG, synth.student_image_state),
                                  ('text', RepeatTextTransformer, S_TXT, synth.student_text_state)):
        stu = cls(**cfg)
        stu.load_state_dict(T(state(11, **cfg)))
        one = DistillModel(stu, dict(loss_name=['out_cos']), '.', model_type=kind, teacher_state_dict=tsd)
        paths[kind] = os.path.join(tmp_path, f'{kind}.ckpt')
        save_checkpoint(paths[kind], one)
    a, b = RepeatVisionTransformer(**S_IMG), RepeatTextTransformer(**S_TXT)
    a, b = load_weight(a, b, paths)
    for k, v in T(synth.student_image_state(11, **S_IMG)).items():
        assert torch.equal(a.state_dict()[k], v)
    for k, v in T(synth.student_text_state(11, **S_TXT)).items():
        assert torch.equal(b.state_dict()[k], v)
    with pytest.raises(ValueError):
        load_weight(a, b, {'image': None, 'text': paths['text']})


def test_frozen_teacher_reload_refreshes_weight_cache():
    """load_state_dict after a forward must reach the bf16 weight cache of a frozen tower"""
    m, opt, _ = _dual(3)
    batch = _batch(0)
    with torch.no_grad():
        _, t0 = m.forward(batch)
        r0 = t0.visual_output.last_representation.clone()
        sd = {k: v.clone() for k, v in m.teacher.state_dict().items()}
        sd['image_encoder.visual.proj'] = sd['image_encoder.visual.proj'] * 2.0
        m.teacher.load_state_dict(sd)
        _, t1 = m.forward(batch)
    assert torch.allclose(t1.visual_output.last_representation, 2.0 * r0, rtol=2e-2, atol=1e-4)


def test_norm_true_pre_normalised_representations():
    """DualDistillModel(norm=True) (reference dual_distill_model.py:110-111, :278-284): L2-normalised representations feed the
    loss; loss and a gradient against the oracle on the same inputs"""
    import oracle
    from distillclip_amd.model import DualDistillModel
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    seed = 13
    sd_i, sd_t = T(synth.student_image_state(seed, **S_IMG)), T(synth.student_text_state(seed, **S_TXT))
    s_img, s_txt = RepeatVisionTransformer(**S_IMG), RepeatTextTransformer(**S_TXT)
    s_img.load_state_dict(sd_i)
    s_txt.load_state_dict(sd_t)
    tsd = _teacher_sd(7)
    m = DualDistillModel(s_img, s_txt, dict(loss_name=['out_cos', 'out_kl'], temperature=2.0), warm_steps=1, total_steps=5,
                         weight_decay=0.0, lr=1e-3, download_root='.', norm=True, teacher_state_dict=tsd).cuda()
    image, text = _batch(0)
    loss = m.training_step([image, text])
    loss.backward()
    for v in list(sd_i.values()) + list(sd_t.values()):
        v.requires_grad_(True)
    oi, ot = oracle.student_image_forward(sd_i, image.cpu(), 4), oracle.student_text_forward(sd_t, text.cpu(), 2)
    with torch.no_grad():
        ti = oracle.teacher_image_forward({k: v for k, v in tsd.items() if k.startswith('visual.')}, image.cpu())
        tt = oracle.teacher_text_forward({k: v for k, v in tsd.items() if not k.startswith('visual.')}, text.cpu())
    for o in (oi, ot, ti, tt):
        o['last_representation'] = o['last_representation'] / o['last_representation'].norm(dim=-1, keepdim=True)
    ref, _ = oracle.LossOracle(['out_cos', 'out_kl'], temperature=2.0)(oracle.clip_forward(oi, ot), oracle.clip_forward(ti, tt), 'all')
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 2e-2 * abs(ref.item()) + 1e-6, (loss.item(), ref.item())
    g, w = dict(m.student.image_encoder.named_parameters())['head.weight'].grad.cpu(), sd_i['head.weight'].grad
    assert (g - w).norm().item() <= 8e-2 * w.norm().item()


def test_adamw_clears_consumed_gradients():
    """FusedAdamW.step(zero_grad=True): same update as step() + zero_grad(), gradients left at zero, later zero_grad() free"""
    m1, o1, _ = _dual(21)
    m2, o2, _ = _dual(21)
    for i in range(2):
        o1.zero_grad(); l1 = m1.training_step(_batch(i)); m1.backward_and_sync(l1); o1.step()
        o2.zero_grad(); l2 = m2.training_step(_batch(i)); m2.backward_and_sync(l2); o2.step(zero_grad=True)
        for tw in m2.towers():
            assert tw._grad_clean and float(tw.flat_grad.abs().max()) == 0.0
    a, b = m1.student.state_dict(), m2.student.state_dict()
    for k in a:
        assert (a[k] - b[k]).norm().item() <= 2e-4 * (a[k].norm().item() + 1e-12), k


def test_overlapped_per_tower_optimizer_step_matches_plain():
    """FusedAdamW.step(overlap=True): update + weight-cache refresh on each tower's own backward stream — same training
    trajectory as the plain step on the current stream (4 steps, losses and weights)"""
    m1, o1, _ = _dual(27)
    m2, o2, _ = _dual(27)
    o2.refresh_cache_in_step = True
    for i in range(4):
        o1.zero_grad(); l1 = m1.training_step(_batch(i)); m1.backward_and_sync(l1); o1.step()
        o2.zero_grad(); l2 = m2.training_step(_batch(i)); m2.backward_and_sync(l2); o2.step(zero_grad=True, overlap=True)
        assert abs(l1.item() - l2.item()) <= 1e-4 * abs(l1.item()), (i, l1.item(), l2.item())
    torch.cuda.synchronize()
    a, b = m1.student.state_dict(), m2.student.state_dict()
    for k in a:
        assert (a[k] - b[k]).norm().item() <= 5e-4 * (a[k].norm().item() + 1e-12), k
    assert all(not tw._prepare_always and not tw.wcache_dirty for tw in m2.towers())
    # a foreign in-place update of the masters after that needs an explicit refresh flag (documented contract)
    sd = m2.student.state_dict()
    m2.student.load_state_dict(sd)
    assert all(tw.wcache_dirty for tw in m2.towers())


def test_teacher_issued_ahead_gives_the_same_step():
    """DualDistillModel.teacher_forward_async + training_step(teacher=handle): the frozen teacher may be issued earlier (e.g. under
    the previous backward); loss and gradients equal the in-step teacher forward"""
    m, opt, _ = _dual(31)
    batch = _batch(3)
    opt.zero_grad()
    l1 = m.training_step(batch)
    l1.backward()
    g1 = {n: p.grad.detach().clone() for n, p in m.student.named_parameters()}
    opt.zero_grad()
    handle = m.teacher_forward_async(batch)
    l2 = m.training_step(batch, teacher=handle)
    l2.backward()
    torch.cuda.synchronize()
    assert abs(l1.item() - l2.item()) <= 1e-6 * abs(l1.item())
    for n, p in m.student.named_parameters():
        assert (p.grad - g1[n]).norm().item() <= 2e-4 * (g1[n].norm().item() + 1e-12), n
    m.multi_stream = False
    opt.zero_grad()
    l3 = m.training_step(batch, teacher=m.teacher_forward_async(batch))
    assert abs(l1.item() - l3.item()) <= 1e-6 * abs(l1.item())


def test_unjoined_overlapped_step_with_deferred_sync_matches_plain():
    """bench.py's loop: backward_and_sync(defer_wait=True) + step(overlap=True, join=False): dependencies ride on the tower streams
    only; same trajectory as the plain loop, also across a switch to single-stream towers and through a checkpoint"""
    from distillclip_amd.checkpoint import checkpoint_dict
    m1, o1, _ = _dual(29)
    m2, o2, _ = _dual(29)
    for i in range(5):
        if i == 3:
            m1.multi_stream = m2.multi_stream = False      # the next forward runs on the main stream: it must see the update
        o1.zero_grad(); l1 = m1.training_step(_batch(i)); m1.backward_and_sync(l1); o1.step()
        o2.zero_grad(); l2 = m2.training_step(_batch(i)); m2.backward_and_sync(l2, defer_wait=True)
        o2.step(zero_grad=True, overlap=True, join=False)
        assert abs(l1.item() - l2.item()) <= 1e-4 * abs(l1.item()), (i, l1.item(), l2.item())
    ck = checkpoint_dict(m2, o2)                           # joins before copying
    a = m1.student.state_dict()
    for k, v in a.items():
        w = ck['state_dict']['student.' + k]
        assert (v.cpu() - w).norm().item() <= 5e-4 * (w.norm().item() + 1e-12), k


def test_optimizer_keeps_the_trainable_set_it_was_built_with():
    """reference dual_distill_model.py:195 / distil_model.py:161: AdamW(filter(requires_grad, parameters())) is built once;
    parameters unfrozen later (unfreeze_embed) never enter it.  Here: frozen embeddings stay put after unfreeze_embed() and the
    optimizer's state_dict keeps its slot count (checkpoints stay loadable)."""
    from distillclip_amd.model import DistillModel
    from distillclip_amd.model.component import RepeatVisionTransformer
    cfg = dict(img_size=32, patch_size=8, in_chans=3, out_dim=64, embed_dim=128, depth=4, num_heads=4, mlp_ratio=4.0,
               qkv_bias=True, repeated_times=2, use_transform=True)
    seed = 5
    s = RepeatVisionTransformer(**cfg)
    s.load_state_dict(T(synth.student_image_state(seed, **cfg)))
    tsd = T(synth.teacher_image_state(seed, 128, 2, 8, 32, 64))
    tsd.update(T(synth.teacher_text_state(seed, 128, 2, 13, 97, 64)))
    m = DistillModel(s, dict(loss_name=['out_l1', 'out_cos']), '.', freeze_embed=True, model_type='image', lr=1e-2,
                     warm_steps=0, teacher_state_dict=tsd).cuda()
    (opt,), _ = m.configure_optimizers()
    n_slots = len(opt.state_dict(checkpoint.trainable_parameters(m))['param_groups'][0]['params'])
    image = torch.from_numpy(synth.images(seed, 4, 32)).cuda()
    m.unfreeze_embed()                                   # after the optimizer exists
    cls0 = dict(m.student.named_parameters())['cls_token'].detach().clone()
    head0 = dict(m.student.named_parameters())['head.weight'].detach().clone()
    for _ in range(2):
        loss = m.training_step(image)
        opt.zero_grad()
        m.backward_and_sync(loss)
        opt.step()
    torch.cuda.synchronize()
    named = dict(m.student.named_parameters())
    assert named['cls_token'].requires_grad and named['cls_token'].grad is not None     # autograd sees it ...
    assert torch.equal(named['cls_token'].detach(), cls0)                                # ... the optimizer does not
    assert not torch.equal(named['head.weight'].detach(), head0)
    sd = opt.state_dict(checkpoint.trainable_parameters(m))
    assert len(sd['param_groups'][0]['params']) == n_slots == len(sd['state'])


def test_reduce_scatter_sharded_path_over_rccl_world1_equals_plain_path():
    """The data-parallel step (bucket callbacks from inside the backward -> reduce-scatter -> sharded AdamW -> all-gather) on a
    1-rank RCCL group takes the same optimizer step as the single-process path: same code the N-GPU bench runs, one rank."""
    import torch.distributed as dist
    from distillclip_amd.model import DualDistillModel
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    from distillclip_amd.parallel import GradSync
    s_img_cfg = dict(img_size=32, patch_size=8, in_chans=3, out_dim=64, embed_dim=128, depth=4, num_heads=4, mlp_ratio=4.0,
                     qkv_bias=True, repeated_times=2, use_transform=True)
    s_txt_cfg = dict(vocab_size=97, context_length=13, out_dim=64, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4.0,
                     qkv_bias=False, repeated_times=2, use_transform=True)
    seed, B = 9, 6
    image = torch.from_numpy(synth.images(seed, B, 32)).cuda()
    text = torch.from_numpy(synth.captions(seed, B, 13, 97, 3, 9)).cuda()

    def build():
        si, st = RepeatVisionTransformer(**s_img_cfg), RepeatTextTransformer(**s_txt_cfg)
        si.load_state_dict(T(synth.student_image_state(seed, **s_img_cfg)))
        st.load_state_dict(T(synth.student_text_state(seed, **s_txt_cfg)))
        tsd = T(synth.teacher_image_state(seed, 128, 2, 8, 32, 64))
        tsd.update(T(synth.teacher_text_state(seed, 128, 2, 13, 97, 64)))
        return DualDistillModel(si, st, dict(loss_name=['out_cos', 'out_kl'], temperature=2.0), 0, 10, 1e-2, 1e-3, '.',
                                freeze_prefix=['image_encoder.pos_embed'], teacher_state_dict=tsd).cuda()

    def train(model, steps=3):
        (opt,), _ = model.configure_optimizers()
        for _ in range(steps):
            loss = model.training_step([image, text])
            opt.zero_grad()
            model.backward_and_sync(loss, defer_wait=True)
            opt.step(zero_grad=True, overlap=True, join=False)
        opt.join()
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in model.student.named_parameters()}, opt

    plain, _ = train(build())
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        model = build()
        model._sync = GradSync()
        model._sync.enabled = True                      # world 1: the collectives still run (RCCL), the shard is the whole bucket
        sharded, opt = train(model)
        tw = model.towers()[0]
        assert tw.dp is not None and len(tw.dp.buckets) == tw.cfg.layers + 2 and tw.gshard.numel() >= 1
        assert float(tw.flat_grad.abs().max()) == 0.0   # exchanged buckets are left clean
        sd = opt.state_dict(checkpoint.trainable_parameters(model))      # gathers the (1-rank) shards
        assert len(sd['state']) == len(sd['param_groups'][0]['params'])
    finally:
        dist.destroy_process_group()
    for k in plain:
        # wgrad atomics are not bit-reproducible between runs and Adam's m / sqrt(v) turns a rounding-level gradient
        # difference into up to +-lr per step on near-zero-gradient elements: 3 steps at lr 1e-3
        # (the k-third of attn.qkv.bias has an exactly-zero true gradient — softmax is shift-invariant — so there every element is
        #  such a near-zero-gradient element and only the +-lr-per-step bound holds)
        d = (plain[k] - sharded[k]).abs()
        assert d.max() < 3e-3 and d.mean() < (1e-3 if 'qkv.bias' in k else 2e-5), (k, d.max().item(), d.mean().item())


def test_adamw_multi_range_launch_equals_one_launch_per_range():
    """dclip_adamw_multi (round 5: the sharded data-parallel step updates the owned slice of every gradient bucket of a tower in ONE launch) gives
    bit for bit what one dclip_adamw launch per range gives — ranges of very different lengths (a 64-element norm slice next to a 2.4 M-element
    block slice), more ranges than one launch holds (24), gradients cleared when asked — and refuses misaligned / odd-length ranges on the host."""
    import ctypes
    from distillclip_amd._lib import lib
    from distillclip_amd.optim import FusedAdamW
    torch.manual_seed(3)
    lens = [64, 2_359_296, 768, 4096, 1_769_472, 256] + [1024 * (1 + i % 5) for i in range(22)]          # 28 ranges -> two launches
    mk = lambda: [torch.randn(n, device='cuda') for n in lens]
    p0, g0, m0 = mk(), mk(), mk()
    v0 = [t.abs() for t in mk()]

    class _Tw:                       # the smallest thing FusedAdamW._adamw_many needs: it only looks at the tensors it is handed
        flat = None
    for zero in (False, True):
        opt = FusedAdamW([], lr=3e-3, weight_decay=1e-2)
        opt.step_count = 7
        a = [[t.clone() for t in x] for x in (p0, g0, m0, v0)]
        b = [[t.clone() for t in x] for x in (p0, g0, m0, v0)]
        st = torch.cuda.current_stream().cuda_stream
        opt._adamw_many(list(zip(*a)), zero, st)
        for p, g, m, v in zip(*b):
            opt._adamw(p, g, m, v, zero, st)
        torch.cuda.synchronize()
        for x, y in zip(a, b):
            for t, u in zip(x, y):
                assert torch.equal(t, u)
        assert all((float(g.abs().max()) == 0.0) == zero for g in a[1])
    one = torch.zeros(66, device='cuda')
    arr = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
    with pytest.raises(ValueError, match='multiple of 4'):
        lib().dclip_adamw_multi(arr(one), arr(one), arr(one), arr(one), (ctypes.c_int64 * 1)(66), 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 0, None)
    with pytest.raises(ValueError, match='1..24'):
        lib().dclip_adamw_multi(arr(one), arr(one), arr(one), arr(one), (ctypes.c_int64 * 1)(64), 25, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 0, None)
