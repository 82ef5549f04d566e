"""world_size-2 rehearsal (gloo, CPU) of the data-parallel gradient exchange: same call pattern as the RCCL path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from distillclip_amd.parallel import GradSync
    sync = GradSync(bucket_elems=1000)              # several buckets + a ragged tail
    g = torch.arange(3333, dtype=torch.float32) * (rank + 1)
    h = torch.full((17,), float(rank))
    sync.launch(g)
    sync.launch(h)
    sync.wait()
    q.put((rank, g.clone(), h.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sync_averages_flat_buffers_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_g = torch.arange(3333, dtype=torch.float32) * 1.5
    for _, g, h in res:
        assert torch.allclose(g, want_g) and torch.allclose(h, torch.full((17,), 0.5))


def _gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from distillclip_amd.parallel import gather_embeddings
    a = torch.full((3, 4), float(rank)) + torch.arange(4)
    b = torch.full((3, 2), 10.0 * rank)
    (ga, gb), r, w = gather_embeddings([a, b])
    q.put((rank, r, w, ga.clone(), gb.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_embeddings_world2():
    """global-negative mode plumbing: one fused all-gather, rank-major row order, shard slice = own rows"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, r, w, ga, gb in res:
        assert (r, w) == (rank, 2) and ga.shape == (6, 4) and gb.shape == (6, 2)
        assert torch.equal(ga[:3], torch.zeros(3, 4) + torch.arange(4)) and torch.equal(ga[3:], torch.ones(3, 4) + torch.arange(4))
        assert torch.equal(gb[3:], torch.full((3, 2), 10.0))


def test_global_negative_gradient_identity():
    """W * (rows of the global-batch gradient) averaged over ranks == single-process gradient on the concatenated batch
    (the parity statement of SURVEY.md §8e), checked with the oracle on CPU."""
    import oracle
    g = torch.Generator().manual_seed(1)
    e = {k: torch.randn(8, 32, generator=g) for k in ('si', 'st', 'ti', 'tt')}
    lc = oracle.LossOracle(['out_cos', 'cos_diff', 'hard_label'], {'cos_diff': 0.5})
    si, st = e['si'].clone().requires_grad_(True), e['st'].clone().requires_grad_(True)
    full, _ = lc(oracle.clip_forward({'last_representation': si}, {'last_representation': st}),
                 oracle.clip_forward({'last_representation': e['ti']}, {'last_representation': e['tt']}), 'all')
    full.backward()
    W, B = 2, 4
    eff = torch.cat([W * si.grad[r * B:(r + 1) * B] for r in range(W)]) / W
    assert torch.allclose(eff, si.grad)


def test_grad_sync_is_a_noop_without_process_group():
    from distillclip_amd.parallel import GradSync
    s = GradSync()
    assert not s.enabled
    g = torch.ones(5)
    s.launch(g)
    s.wait()
    assert torch.equal(g, torch.ones(5))


def test_sharded_loss_equals_reference_ddp_semantics():
    """Local-negative mode (the reference's training_step): k ranks each compute the loss on their shard and DDP averages the
    gradients.  Emulated in one process with the oracle: mean of shard gradients == gradient of the mean of shard losses."""
    import oracle
    g = torch.Generator().manual_seed(0)
    e = {k: torch.randn(8, 32, generator=g) for k in ('si', 'st', 'ti', 'tt')}
    lc = oracle.LossOracle(['out_l1', 'out_cos', 'cos_diff'], {'cos_diff': 0.1})

    def loss_on(sl):
        si, st = e['si'][sl].clone().requires_grad_(True), e['st'][sl].clone().requires_grad_(True)
        stu = oracle.clip_forward({'last_representation': si}, {'last_representation': st})
        tea = oracle.clip_forward({'last_representation': e['ti'][sl]}, {'last_representation': e['tt'][sl]})
        l, _ = lc(stu, tea, 'all')
        l.backward()
        return l.item(), si.grad, st.grad
    l0, gi0, gt0 = loss_on(slice(0, 4))
    l1, gi1, gt1 = loss_on(slice(4, 8))
    lf, gif, gtf = loss_on(slice(0, 8))
    # per-sample tower terms are shard-additive; cos_diff uses per-shard negatives, so the full-batch loss differs
    assert abs(0.5 * (l0 + l1) - lf) > 1e-6
    assert gi0.shape == (4, 32) and gi1.shape == (4, 32)


def _gather_rows_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from distillclip_amd.metrics import gather_rows
    x = torch.arange(6, dtype=torch.float32).reshape(3, 2) + 100 * rank
    q.put((rank, gather_rows(x).clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_validation_gather_rows_world2():
    """validation_step's all_gather of the representations (reference dual_distill_model.py:141-146): rank-major rows,
    identical on every rank, so each rank's validation_epoch_end sees the whole validation set"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gather_rows_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = torch.arange(6, dtype=torch.float32).reshape(3, 2)
    for _, g in res:
        assert torch.equal(g, torch.cat([base, base + 100]))


def test_gather_rows_is_identity_without_process_group():
    from distillclip_amd.metrics import gather_rows
    x = torch.randn(4, 8)
    assert gather_rows(x) is x
