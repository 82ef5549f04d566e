"""world_size-2 and world_size-8 rehearsals (gloo, CPU) of the data-parallel gradient exchange: same call pattern as the RCCL path.
World size 8 is the target (one 8 x MI355X node, reference l_clip.yaml:56); no multi-GPU box is available to this build, so the shard plan,
the padded row-sparse segments, the global-negative row blocks and the launcher are exercised at that size here."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _to_numpy(o):
    if isinstance(o, torch.Tensor):
        return ('__tensor__', o.detach().cpu().numpy().copy())
    if isinstance(o, dict):
        return {k: _to_numpy(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_to_numpy(v) for v in o)
    return o


def _to_torch(o):
    if isinstance(o, tuple) and len(o) == 2 and isinstance(o[0], str) and o[0] == '__tensor__':
        return torch.from_numpy(o[1])
    if isinstance(o, dict):
        return {k: _to_torch(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_to_torch(v) for v in o)
    return o


class _ResultQueue:
    """what the ranks report through: tensors travel as numpy arrays by value.  (A torch tensor on a multiprocessing queue is a
    shared-memory handle served by the SENDING process; a rank that had already exited when the parent unpickled its report made
    the test fail with FileNotFoundError under load.)"""

    def __init__(self, q):
        self.q = q

    def put(self, item):
        self.q.put(_to_numpy(item))


def _run_ranks(target, world=2, timeout=120):
    """spawn `world` rank processes of `target(rank, world, rdzv, q)` and collect one queue item per rank.  Rendezvous is a
    FileStore in a private temporary directory: no port is picked, so there is no bind-then-close race and NO retry -- a queue
    timeout or a non-zero rank exit fails the test on the first attempt."""
    import queue as _queue
    import shutil
    import tempfile
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    d = tempfile.mkdtemp(prefix='dclip_rdzv_')
    ps = [ctx.Process(target=target, args=(r, world, os.path.join(d, 'store'), _ResultQueue(q))) for r in range(world)]
    for p in ps:
        p.start()
    try:
        try:
            res = [_to_torch(q.get(timeout=timeout)) for _ in ps]
        except _queue.Empty:
            raise AssertionError(f'ranks did not report within {timeout} s (exit codes so far {[p.exitcode for p in ps]})')
        codes = []
        for p in ps:
            p.join(timeout=60)
            codes.append(p.exitcode)
        assert all(c == 0 for c in codes), f'rank exit codes {codes}'
        return res
    finally:
        for p in ps:
            if p.is_alive():
                p.terminate()
            p.join(timeout=30)
        shutil.rmtree(d, ignore_errors=True)


def _worker(rank, world, rdzv, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', init_method='file://' + rdzv, rank=rank, world_size=world)
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
    res = _run_ranks(_worker, 2, timeout=120)
    want_g = torch.arange(3333, dtype=torch.float32) * 1.5
    for _, g, h in res:
        assert torch.allclose(g, want_g) and torch.allclose(h, torch.full((17,), 0.5))


def _gather_worker(rank, world, rdzv, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', init_method='file://' + rdzv, rank=rank, world_size=world)
    from distillclip_amd.parallel import gather_embeddings
    a = torch.full((3, 4), float(rank)) + torch.arange(4)
    b = torch.full((3, 2), 10.0 * rank)
    (ga, gb), r, w = gather_embeddings([a, b])
    q.put((rank, r, w, ga.clone(), gb.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_embeddings_world2():
    """global-negative mode plumbing: one fused all-gather, rank-major row order, shard slice = own rows"""
    res = _run_ranks(_gather_worker, 2, timeout=120)
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


def _global_rows_worker(rank, world, rdzv, q):
    """global-negative mode as the model runs it (dual_distill_model.py here, SURVEY.md 8e Collective 2): ONE fused all-gather of the four
    [B, E] embeddings, every rank evaluates the loss of ITS ROW BLOCK against all world * B columns — here with the oracle standing in for
    dclip_distill_loss_rows, which needs the GPU — and hands back W x (gradient rows of its own samples)"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', init_method='file://' + rdzv, rank=rank, world_size=world)
    import oracle
    from distillclip_amd.parallel import gather_embeddings, check_equal_batch
    B, E = 3, 32
    g = torch.Generator().manual_seed(7)
    full = {k: torch.randn(world * B, E, generator=g) for k in ('si', 'st', 'ti', 'tt')}      # the concatenated batch, same on every rank
    mine = {k: v[rank * B:(rank + 1) * B].clone() for k, v in full.items()}
    check_equal_batch(B, torch.device('cpu'))
    (gsi, gst, gti, gtt), r, w = gather_embeddings([mine['si'], mine['st'], mine['ti'], mine['tt']])
    assert (r, w) == (rank, world) and all(torch.equal(a, full[k]) for a, k in ((gsi, 'si'), (gst, 'st'), (gti, 'ti'), (gtt, 'tt')))
    si, st = gsi.clone().requires_grad_(True), gst.clone().requires_grad_(True)
    lc = oracle.LossOracle(['out_l1', 'out_cos', 'cos_diff', 'hard_label'], {'cos_diff': 0.1})
    loss, _ = lc(oracle.clip_forward({'last_representation': si}, {'last_representation': st}),
                 oracle.clip_forward({'last_representation': gti}, {'last_representation': gtt}), 'all')
    loss.backward()
    q.put((rank, float(loss), world * si.grad[rank * B:(rank + 1) * B].clone(), world * st.grad[rank * B:(rank + 1) * B].clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_global_negative_row_blocks_at_world_8_equal_one_process_on_the_concatenated_batch():
    """the parity statement of SURVEY.md 8e at the target world size: the DP AVERAGE over ranks of (W x the gradient rows a rank owns)
    is the single-process gradient on the concatenated batch, the gathered rows arrive in rank-major order on all 8 ranks"""
    import oracle
    world, B, E = 8, 3, 32
    res = sorted(_run_ranks(_global_rows_worker, world, timeout=240), key=lambda r: r[0])
    g = torch.Generator().manual_seed(7)
    full = {k: torch.randn(world * B, E, generator=g) for k in ('si', 'st', 'ti', 'tt')}
    si, st = full['si'].clone().requires_grad_(True), full['st'].clone().requires_grad_(True)
    lc = oracle.LossOracle(['out_l1', 'out_cos', 'cos_diff', 'hard_label'], {'cos_diff': 0.1})
    loss, _ = lc(oracle.clip_forward({'last_representation': si}, {'last_representation': st}),
                 oracle.clip_forward({'last_representation': full['ti']}, {'last_representation': full['tt']}), 'all')
    loss.backward()
    for rank, l, gi, gt in res:
        assert abs(l - loss.item()) <= 1e-6 * abs(loss.item())
        # what the tower's backward receives on rank r, divided by W by the DP average, is that rank's rows of the global gradient
        assert torch.allclose(gi / world, si.grad[rank * B:(rank + 1) * B], rtol=1e-6, atol=1e-8)
        assert torch.allclose(gt / world, st.grad[rank * B:(rank + 1) * B], rtol=1e-6, atol=1e-8)


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


def _gather_rows_worker(rank, world, rdzv, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', init_method='file://' + rdzv, rank=rank, world_size=world)
    from distillclip_amd.metrics import gather_rows
    x = torch.arange(6, dtype=torch.float32).reshape(3, 2) + 100 * rank
    q.put((rank, gather_rows(x).clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_validation_gather_rows_world2():
    """validation_step's all_gather of the representations (reference dual_distill_model.py:141-146): rank-major rows,
    identical on every rank, so each rank's validation_epoch_end sees the whole validation set"""
    res = _run_ranks(_gather_rows_worker, 2, timeout=120)
    base = torch.arange(6, dtype=torch.float32).reshape(3, 2)
    for _, g in res:
        assert torch.equal(g, torch.cat([base, base + 100]))


def test_gather_rows_is_identity_without_process_group():
    from distillclip_amd.metrics import gather_rows
    x = torch.randn(4, 8)
    assert gather_rows(x) is x


# ---------------------------------------------------------------------------------------------------------------------
# reduce-scatter -> sharded AdamW -> all-gather (SURVEY.md §8e Collective 1), rehearsed over gloo with the same call pattern
# the RCCL path uses: GradSync.plan / bucket_ready (what the backward's bucket callback calls) / finish, FusedAdamW.step
# ---------------------------------------------------------------------------------------------------------------------
class _FakeTower:
    """flat buffers + bucket layout of a tower, without the HIP runtime (the bookkeeping under test is host logic)"""

    def __init__(self, total, buckets, trainable, seed):
        g = torch.Generator().manual_seed(seed)
        self.flat = torch.randn(total, generator=g)
        self.flat_grad = torch.zeros(total)
        self._buckets, self._trainable = buckets, trainable
        self.sync = self.dp = self.gshard = None
        self.dp_released = 0
        self.wcache_dirty = False
        self.grads_ready = self.opt_done = self.bwd_stream = None
        self._grad_clean = False

        edges = sorted({0, total} | {e for r in trainable for e in r})
        self._offsets = edges[:-1]
        self._plist = [self.flat[a:b] for a, b in zip(edges[:-1], edges[1:])]      # one "parameter" per segment

    def _params(self):
        return self._plist

    def grad_buckets(self):
        return list(self._buckets)

    def trainable_ranges(self):
        return [list(r) for r in self._trainable]


def _torch_adamw(self, p, g, m, v, zero_grad, st):
    """torch restatement of dclip_adamw (torch.optim.AdamW semantics) for the CPU rehearsal"""
    b1, b2 = self.betas
    p.mul_(1.0 - self.lr * self.weight_decay)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** self.step_count, 1 - b2 ** self.step_count
    p.addcdiv_(m, (v.sqrt() / bc2 ** 0.5).add_(self.eps), value=-self.lr / bc1)
    if zero_grad:
        g.zero_()


# completion order: head bucket at the END of the flat layout, two blocks, a frozen-only bucket, embedding at the start
_TOTAL = 64 * 40
_BUCKETS = [(64 * 34, 64 * 40), (64 * 20, 64 * 34), (64 * 12, 64 * 20), (64 * 8, 64 * 12), (0, 64 * 8)]
_TRAINABLE = [[64 * 2, 64 * 8], [64 * 12, 64 * 25], [64 * 26, 64 * 40]]      # frozen: part of the embedding, bucket 3, one segment


def _sharded_worker(rank, world, rdzv, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', init_method='file://' + rdzv, rank=rank, world_size=world)
    from distillclip_amd.parallel import GradSync
    from distillclip_amd.optim import FusedAdamW
    FusedAdamW._adamw = _torch_adamw
    tw = _FakeTower(_TOTAL, _BUCKETS, _TRAINABLE, seed=3)              # same parameters on every rank
    sync = GradSync()
    assert sync.enabled and sync.sharded
    sync.attach([tw])
    opt = FusedAdamW([tw], lr=1e-2, weight_decay=1e-2)
    mask = torch.zeros(_TOTAL, dtype=torch.bool)
    for a, b in _TRAINABLE:
        mask[a:b] = True
    for step in range(3):
        g = torch.Generator().manual_seed(100 * step + rank)          # every rank its own gradient; frozen entries stay 0
        tw.flat_grad.add_(torch.randn(_TOTAL, generator=g) * mask)
        for i in range(len(_BUCKETS)):                                 # what the backward's bucket callback does, in order
            sync.bucket_ready(tw, i)
        sync.finish(tw)
        assert float(tw.flat_grad.abs().max()) == 0.0                  # exchanged buckets are left clean for the next backward
        opt.step()
    sd = opt.state_dict()                                              # collective: gathers the ranks' m / v shards
    assert tw.dp.shard_elems * world == sum(b1 - b0 for (b0, b1), b in zip(_BUCKETS, tw.dp.buckets) if b is not None)
    result = (rank, tw.flat.clone(), {k: {n: t.clone() for n, t in v.items()} for k, v in sd['state'].items()},
              opt._state[id(tw)][0].numel())
    # a second backward before step() (gradient accumulation) would replace the un-consumed shard average: refused
    tw.flat_grad.add_(1.0)
    sync.bucket_ready(tw, 0)
    try:
        sync.bucket_ready(tw, 0)
        twice = 'no error'
    except RuntimeError as e:
        twice = str(e)
    assert 'released twice' in twice, twice
    opt.step()                                                         # consumes it: the bucket may be released again
    sync.bucket_ready(tw, 0)
    # moments whose layout no longer matches the shard plan are never silently re-zeroed
    m0 = opt._state[id(tw)][0]
    opt._state[id(tw)] = (torch.zeros(m0.numel() + 64), torch.zeros(m0.numel() + 64))
    try:
        opt._moments(tw)
        resized = 'no error'
    except RuntimeError as e:
        resized = str(e)
    assert 'optimizer state of' in resized, resized
    q.put(result)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_reduce_scatter_sharded_adamw_all_gather_equals_unsharded(world):
    res = _run_ranks(_sharded_worker, world, timeout=240)
    # expected: torch.optim.AdamW on the AVERAGED gradients, trainable entries only (what DDP + the reference's optimizer do)
    ref = _FakeTower(_TOTAL, _BUCKETS, _TRAINABLE, seed=3)
    mask = torch.zeros(_TOTAL, dtype=torch.bool)
    for a, b in _TRAINABLE:
        mask[a:b] = True
    chunks = [torch.nn.Parameter(ref.flat[a:b].clone()) for a, b in _TRAINABLE]
    topt = torch.optim.AdamW(chunks, lr=1e-2, weight_decay=1e-2)
    for step in range(3):
        avg = sum(torch.randn(_TOTAL, generator=torch.Generator().manual_seed(100 * step + r)) * mask for r in range(world)) / world
        for c, (a, b) in zip(chunks, _TRAINABLE):
            c.grad = avg[a:b].clone()
        topt.step()
    want = ref.flat.clone()
    for c, (a, b) in zip(chunks, _TRAINABLE):
        want[a:b] = c.detach()
    for rank, flat, state, nshard in res:
        assert torch.allclose(flat, want, rtol=1e-5, atol=1e-6), (rank, (flat - want).abs().max())
        assert torch.equal(flat[~mask], ref.flat[~mask])               # frozen parameters untouched
        assert nshard * world < _TOTAL                                 # m / v exist for the owned shards only
    # the gathered optimizer state equals torch's (first trainable range = the first slots)
    states = [st for _, _, st, _ in sorted(res, key=lambda r: r[0])]
    assert len(states) == world and all(st.keys() == states[0].keys() and len(st) == len(_TRAINABLE) for st in states)
    tstate = topt.state_dict()['state']
    for i in range(len(_TRAINABLE)):
        for st in states:                                              # every rank sees the full (gathered) moments
            assert torch.allclose(st[i]['exp_avg'], tstate[i]['exp_avg'], rtol=1e-5, atol=1e-7)
            assert torch.allclose(st[i]['exp_avg_sq'], tstate[i]['exp_avg_sq'], rtol=1e-5, atol=1e-9)
            assert float(st[i]['step']) == 3.0


# ---------------------------------------------------------------------------------------------------------------------
# row-sparse exchange of the token-embedding gradient (reference weight_share_model.py:407: nn.Embedding(49408, 768))
# ---------------------------------------------------------------------------------------------------------------------
_SV, _SD = 40, 24                                   # table rows / width: the shard boundary (512) cuts row 21
_S_TOTAL = 1024 + 128
_S_BUCKETS = [(1024, 1152), (0, 1024)]              # a block bucket, then the embedding bucket: table [0, 960) + positions [960, 1024)


def _extras_worker(rank, world, rdzv, q):
    """parameters outside the tower buffers (the projection linears of a plain CLIP encoder in the student role): FusedAdamW averages their
    autograd-owned gradients over the ranks before the update, and saves / restores their moments in torch's layout"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', init_method='file://' + rdzv, rank=rank, world_size=world)
    from distillclip_amd.parallel import GradSync
    from distillclip_amd.optim import FusedAdamW
    FusedAdamW._adamw = _torch_adamw
    tw = _FakeTower(_TOTAL, _BUCKETS, _TRAINABLE, seed=3)
    g0 = torch.Generator().manual_seed(7)
    extras = [torch.nn.Parameter(torch.randn(16, 8, generator=g0)), torch.nn.Parameter(torch.randn(16, generator=g0)),
              torch.nn.Parameter(torch.randn(4, generator=g0), requires_grad=False)]
    sync = GradSync()
    sync.attach([tw])
    opt = FusedAdamW([tw], lr=1e-2, weight_decay=1e-2, extra_params=extras)
    assert len(opt.extras) == 2                                        # the frozen one never enters the optimizer
    for step in range(3):
        g = torch.Generator().manual_seed(100 * step + rank)
        for i in range(len(_BUCKETS)):
            sync.bucket_ready(tw, i)
        sync.finish(tw)
        for p in extras[:2]:
            p.grad = torch.randn(p.shape, generator=g)
        opt.step()
        opt.zero_grad()
        assert all(p.grad is None for p in extras)
    sd = opt.state_dict()
    opt2 = FusedAdamW([tw], lr=1.0, extra_params=extras)
    opt2.load_state_dict(sd)
    same = all(torch.equal(opt2._extra_state[id(p)][k], opt._extra_state[id(p)][k]) for p in extras[:2] for k in (0, 1))
    q.put((rank, [p.detach().clone() for p in extras], len(sd['param_groups'][0]['params']), same))
    dist.barrier()
    dist.destroy_process_group()


def test_extra_parameters_are_averaged_and_updated_like_torch_adamw_world2():
    res = _run_ranks(_extras_worker, 2, timeout=240)
    g0 = torch.Generator().manual_seed(7)
    ref = [torch.nn.Parameter(torch.randn(16, 8, generator=g0)), torch.nn.Parameter(torch.randn(16, generator=g0))]
    frozen = torch.randn(4, generator=g0)
    topt = torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2)
    for step in range(3):
        gens = [torch.Generator().manual_seed(100 * step + r) for r in range(2)]
        for p in ref:
            p.grad = sum(torch.randn(p.shape, generator=g) for g in gens) / 2
        topt.step()
    n_slots = None
    for rank, got, n, same in res:
        assert same
        n_slots = n
        for a, b in zip(got[:2], ref):
            assert torch.allclose(a, b.detach(), rtol=1e-5, atol=1e-6), rank
        assert torch.equal(got[2], frozen)
    assert n_slots == 3 + 2                                            # the tower's three trainable segments + the two extras


class _FakeTextTower(_FakeTower):
    def sparse_spec(self):
        return 1, 0, _SV, _SD


def _sparse_worker(rank, world, rdzv, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', init_method='file://' + rdzv, rank=rank, world_size=world)
    from distillclip_amd.parallel import GradSync
    out = {}
    for mode in ('sparse', 'dense'):
        tw = _FakeTextTower(_S_TOTAL, _S_BUCKETS, [[0, _S_TOTAL]], seed=5)
        sync = GradSync()
        sync.sparse_embedding = mode == 'sparse'
        sync.attach([tw])
        gen = torch.Generator().manual_seed(40 + rank)
        ids = torch.randint(0, _SV, (3, 4), generator=gen)              # this rank's token ids: at most 12 of the 40 rows
        g = torch.zeros(_S_TOTAL)
        rows = torch.unique(ids.reshape(-1))
        g[:_SV * _SD].view(_SV, _SD)[rows] = torch.randn(len(rows), _SD, generator=gen)
        g[_SV * _SD:] = torch.randn(_S_TOTAL - _SV * _SD, generator=gen)
        tw.flat_grad.copy_(g)
        sync.note_token_ids(tw, ids)                                    # what HipTower.forward does
        assert (getattr(tw, '_sparse', None) is not None) == (mode == 'sparse')
        for i in range(len(_S_BUCKETS)):
            sync.bucket_ready(tw, i)
        sync.finish(tw)
        assert float(tw.flat_grad.abs().max()) == 0.0                   # exchanged buckets are left clean either way
        out[mode] = tw.gshard.clone()
        if mode == 'sparse':
            out['rows'] = tw.sparse_rows_last
            out['ids'] = ids
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_row_sparse_embedding_exchange_equals_dense_reduce_scatter(world):
    """(world 8: eight destination segments padded to the longest, most table rows cut by a shard boundary: 128 elements per shard, 24 per row)"""
    res = sorted(_run_ranks(_sparse_worker, world, timeout=240), key=lambda r: r[0])
    union = torch.unique(torch.cat([o['ids'].reshape(-1) for _, o in res]))
    for rank, o in res:
        if world == 2:
            assert torch.equal(o['sparse'], o['dense']), (rank, (o['sparse'] - o['dense']).abs().max())     # bit for bit: a + b
        else:
            # eight addends: a ring reduction adds them in an order that depends on where an element sits in the exchanged buffer, and the
            # sparse and the dense exchange place the same element differently — equal up to the order of seven f32 additions
            assert torch.allclose(o['sparse'], o['dense'], rtol=0, atol=4e-7), (rank, (o['sparse'] - o['dense']).abs().max())
            assert torch.equal(o['sparse'] == 0, o['dense'] == 0)                                          # and exactly zero in the same places
        assert o['rows'] == len(union) <= _SV                           # only the touched rows travelled
    assert world > 2 or len(union) < _SV
    # ... and the untouched rows of the owned table slice are exactly zero in the shard
    per = 1024 // world
    for rank, o in res:
        emb = o['sparse'][128 // world:]                                # shard layout: block bucket first, then the embedding bucket
        lo = rank * per
        for r in range(_SV):
            a, b = max(r * _SD, lo), min((r + 1) * _SD, lo + per)
            if a < b and r not in union.tolist():
                assert float(emb[a - lo:b - lo].abs().max()) == 0.0


def _sparse_edge_worker(rank, world, rdzv, q):
    """the sparse exchange's assumptions, broken on purpose: two grad-enabled forwards before one backward; a ragged (smaller) batch on
    one rank; a batch larger than the first one"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', init_method='file://' + rdzv, rank=rank, world_size=world)
    from distillclip_amd.parallel import GradSync
    out = {}

    def grads(tw, id_sets, gen):
        g = torch.zeros(_S_TOTAL)
        for ids in id_sets:
            rows = torch.unique(ids.reshape(-1))
            g[:_SV * _SD].view(_SV, _SD)[rows] += torch.randn(len(rows), _SD, generator=gen)
        g[_SV * _SD:] = torch.randn(_S_TOTAL - _SV * _SD, generator=gen)
        tw.flat_grad.copy_(g)

    def release(sync, tw):
        for i in range(len(_S_BUCKETS)):
            sync.bucket_ready(tw, i)
        sync.finish(tw)
        assert float(tw.flat_grad.abs().max()) == 0.0
        tw.dp_unstepped.clear()                                          # (what optimizer.step() does)
        return tw.gshard.clone()

    for mode in ('sparse', 'dense'):
        tw = _FakeTextTower(_S_TOTAL, _S_BUCKETS, [[0, _S_TOTAL]], seed=5)
        sync = GradSync()
        sync.sparse_embedding = mode == 'sparse'
        sync.attach([tw])
        gen = torch.Generator().manual_seed(70 + rank)
        # step 1: TWO forwards (different ids) before one backward: the gradient holds the rows of both
        ids_a = torch.randint(0, _SV, (3, 4), generator=gen)
        ids_b = torch.randint(0, _SV, (3, 4), generator=gen)
        grads(tw, [ids_a, ids_b], gen)
        sync.note_token_ids(tw, ids_a)
        sync.note_token_ids(tw, ids_b)
        if mode == 'sparse':
            assert tw._sparse.get('dense')                              # marked dense: the union of the last call alone would be wrong
        out[mode + '_two_forwards'] = release(sync, tw)
        assert getattr(tw, '_sparse', None) is None
        # step 2: a forward whose backward never runs, then a normal step: the stale entry makes that step dense, the one after sparse
        sync.note_token_ids(tw, ids_a)
        ids_c = torch.randint(0, _SV, (3, 4), generator=gen)
        grads(tw, [ids_c], gen)
        sync.note_token_ids(tw, ids_c)
        out[mode + '_after_skipped'] = release(sync, tw)
        # step 3: ragged batch: rank 1 holds 2 captions instead of 3 (padded with the "no row" id on the wire)
        ids_d = torch.randint(0, _SV, (2 if rank == 1 else 3, 4), generator=gen)
        grads(tw, [ids_d], gen)
        sync.note_token_ids(tw, ids_d)
        if mode == 'sparse':
            assert tw._sparse is not None and not tw._sparse.get('dense')
        out[mode + '_ragged'] = release(sync, tw)
        if mode == 'sparse':
            out['ragged_rows'] = tw.sparse_rows_last
            out['ragged_ids'] = ids_d
            # a batch larger than the first forward's cannot be announced to the other ranks: refused on this rank, before any collective
            try:
                sync.note_token_ids(tw, torch.zeros(4, 4, dtype=torch.int64))
                big = 'no error'
            except RuntimeError as e:
                big = str(e)
            assert 'sized for 12' in big, big
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sparse_exchange_survives_two_forwards_a_skipped_backward_and_a_ragged_batch_world2():
    res = sorted(_run_ranks(_sparse_edge_worker, 2, timeout=120), key=lambda r: r[0])
    for rank, o in res:
        for case in ('two_forwards', 'after_skipped', 'ragged'):
            assert torch.equal(o['sparse_' + case], o['dense_' + case]), (rank, case, (o['sparse_' + case] - o['dense_' + case]).abs().max())
    union = torch.unique(torch.cat([o['ragged_ids'].reshape(-1) for _, o in res]))
    assert all(o['ragged_rows'] == len(union) for _, o in res)           # the padding ids are not rows


def test_shard_plan_at_world_8_partitions_every_live_bucket():
    """the plan of the target world size on the bucket layout of the shipped l_clip students (sizes from the host-side encoder plan,
    no device): every live bucket is cut into 8 equal owned slices that tile it in rank order, the owned trainable ranges of the ranks tile
    the bucket's trainable ranges exactly, and every rank holds 1/8 of the exchanged elements"""
    from distillclip_amd.parallel import _Shards
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    towers = [RepeatVisionTransformer(img_size=224, patch_size=32, out_dim=512, embed_dim=768, depth=6, num_heads=24, qkv_bias=True,
                                      repeated_times=2, use_transform=True)._tower,
              RepeatTextTransformer(depth=4, repeated_times=2, use_transform=True)._tower]
    for tw in towers:
        offs = tw.param_offsets()
        total = offs[-1]
        buckets = tw.grad_buckets()
        assert sorted(buckets) == sorted(set(buckets)) and sum(b1 - b0 for b0, b1 in buckets) == total      # a partition of the flat buffer
        trainable = [[0, total]]
        plans = [_Shards(buckets, trainable, rank=r, world=8) for r in range(8)]
        assert len({p.shard_elems for p in plans}) == 1 and plans[0].shard_elems * 8 == total
        for i, (b0, b1) in enumerate(buckets):
            own = [p.buckets[i][2:4] for p in plans]
            assert own[0][0] == b0 and own[-1][1] == b1 and all(own[r][1] == own[r + 1][0] for r in range(7))
            assert all(o1 - o0 == (b1 - b0) // 8 for o0, o1 in own)
            assert all(p.buckets[i][5] == [(p.buckets[i][2], p.buckets[i][3])] for p in plans)
        # frozen embeddings (image.yaml freeze_embed): the frozen ranges drop out of every rank's owned trainable ranges
        frozen_end = offs[1]
        plans = [_Shards(buckets, [[frozen_end, total]], rank=r, world=8) for r in range(8)]
        covered = sorted((a, b) for p in plans for bk in p.live() for a, b in bk[5])
        assert covered[0][0] == frozen_end and covered[-1][1] == total or covered[0][0] >= frozen_end
        assert sum(b - a for a, b in covered) == total - frozen_end


def test_shard_plan_rejects_indivisible_world():
    from distillclip_amd.parallel import _Shards
    with pytest.raises(ValueError, match='not divisible'):
        _Shards([(0, 64)], [[0, 64]], rank=0, world=3)
    sh = _Shards(_BUCKETS, _TRAINABLE, rank=1, world=2)
    assert sh.buckets[3] is None                                        # frozen-only bucket: never exchanged
    b0, b1, o0, o1, off, own = sh.buckets[0]
    assert (o0, o1) == (b0 + (b1 - b0) // 2, b1) and off == 0


def _launcher_env():
    return {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'DCLIP_RDZV_FILE')}


def test_bench_self_launch_dry_run():
    """`python bench.py --gpus 2 --dry-launch`: the parent starts two rank processes before touching any GPU and relays rank 0's
    JSON line; a non-zero exit of a rank propagates."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = _launcher_env()
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-launch'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out == {'dry_launch': True, 'n_gpus': 2, 'ranks_seen': 2, 'backend': 'gloo'}
    # a failing rank (WORLD_SIZE mismatch is impossible here, so ask for an unknown config) -> non-zero exit of the launcher
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-launch', '--config', 'nope'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_bench_self_launch_dry_run_and_hung_rank_at_8_ranks():
    """`python bench.py --gpus 8 --dry-launch` (what the driver's SCALE run starts at N = 8): eight ranks rendezvous and rank 0 reports all
    eight; with rank 5 hung after the collective the launcher ends non-zero inside its deadline"""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = _launcher_env()
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--dry-launch'], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out == {'dry_launch': True, 'n_gpus': 8, 'ranks_seen': 8, 'backend': 'gloo'}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--dry-launch', '--launch-grace', '5',
                        '--launch-timeout', '240'], env=dict(env, DCLIP_DRY_HANG_RANK='5'), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'still running' in r.stderr, (r.returncode, r.stderr[-1000:])
    assert time.monotonic() - t0 < 200


def test_launcher_rendezvous_repeatedly_without_retry():
    """The launcher's rendezvous (FileStore in a private directory: no port to race for) 8 times in a row from one process (20 in rounds 3-4: the world-8 launches of round 5 took over part of that time), every
    run on its first and only attempt."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('dclip_bench_launcher', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    saved = dict(os.environ)
    try:
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'DCLIP_RDZV_FILE'):
            os.environ.pop(k, None)
        devnull = os.open(os.devnull, os.O_WRONLY)
        keep = os.dup(1)
        os.dup2(devnull, 1)                       # rank 0's JSON line (inherited stdout) is not this test's output
        try:
            codes = [bench.launch_ranks(2, ['--gpus', '2', '--dry-launch'], timeout_s=240.0, grace_s=60.0) for _ in range(8)]
        finally:
            os.dup2(keep, 1)
            os.close(keep)
            os.close(devnull)
    finally:
        os.environ.clear()
        os.environ.update(saved)
    assert codes == [0] * 8, codes


def test_launcher_deadline_ends_a_hung_rank():
    """A rank that never exits (stuck after the collective) must not block the launcher: once the first rank has exited the others
    get --launch-grace seconds, then are terminated and the launcher exits non-zero."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(_launcher_env(), DCLIP_DRY_HANG_RANK='1')
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-launch', '--launch-grace', '5',
                        '--launch-timeout', '120'], env=env, capture_output=True, text=True, timeout=300)
    dt = time.monotonic() - t0
    assert r.returncode != 0 and 'still running' in r.stderr, (r.returncode, r.stderr[-1000:])
    assert dt < 90, dt
    # ... and the overall deadline alone (every rank hangs, nobody exits) does the same
    env = dict(_launcher_env(), DCLIP_DRY_HANG_RANK='all')
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-launch', '--launch-timeout', '20'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'deadline' in r.stderr, (r.returncode, r.stderr[-1000:])
    assert time.monotonic() - t0 < 120


def _ragged_worker(rank, world, rdzv, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', init_method='file://' + rdzv, rank=rank, world_size=world)
    from distillclip_amd.parallel import check_equal_batch
    check_equal_batch(8, torch.device('cpu'))                 # equal: passes on every rank
    try:
        check_equal_batch(8 - rank, torch.device('cpu'))      # a loader without drop_last: last batch ragged
        q.put((rank, 'no error'))
    except ValueError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_global_negatives_reject_unequal_batches_on_every_rank():
    res = _run_ranks(_ragged_worker, 2, timeout=90)
    assert all('same per-rank batch' in msg and '7..8' in msg for _, msg in res), res
