"""Data-parallel exchange over RCCL (torch.distributed backend "nccl" on ROCm), one process per GPU.

Reference behaviour (SURVEY.md §2.2 C1, §8e Collective 1): Lightning DDP (`ddp_find_unused_parameters_false`,
config/final_config/l_clip.yaml:56) all-reduces the student's f32 gradients in buckets released by autograd hooks and every
rank runs the same AdamW (dual_distill_model.py:194-196).  Here every tower keeps its gradients in ONE flat f32 buffer whose
layout follows the backward's completion order, so the exchange is

    per gradient bucket (final norm + head, blocks L-1 .. 0, embedding), as soon as the backward has enqueued its last writer:
        reduce-scatter (AVG) of the bucket  ->  this rank's 1/W shard                          [side stream, under the backward]
    optimizer step:
        fused AdamW on the owned shard of every bucket (m / v exist only for the shard)  ->  all-gather of the parameters

i.e. reduce-scatter -> sharded AdamW -> all-gather instead of all-reduce + W identical full-size updates: the same bytes on
the xGMI links, 1/W of the optimizer's HBM traffic and state.  `GradSync.launch` (flat all-reduce) remains as the fallback for
world sizes that do not divide the 64-element parameter alignment.
"""
import os

import torch
import torch.distributed as dist


def _dist_on():
    return dist.is_available() and dist.is_initialized()


def _native():
    """RCCL ("nccl") has reduce-scatter / all-gather / AVG on device tensors; gloo (CPU rehearsal, or the one-GPU multi-process test
    of tests/test_parallel_gpu.py) gets the same results from all_reduce, which it supports for every tensor type"""
    return dist.get_backend() == 'nccl'


def reduce_scatter_avg(out, inp, group=None):
    """out <- this rank's 1/W slice of the element-wise average of `inp` over the ranks"""
    world, rank = dist.get_world_size(), dist.get_rank()
    if _native():
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.AVG, group=group)
        return
    n = out.numel()
    if not inp.is_cuda:
        try:
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group)
            out.mul_(1.0 / world)
            return
        except RuntimeError:
            pass
    tmp = inp.clone()
    dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
    out.copy_(tmp[rank * n:(rank + 1) * n]).mul_(1.0 / world)


def all_reduce_avg(t, group=None):
    """t <- element-wise average over the ranks, in place"""
    if _native():
        dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
        return
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t.mul_(1.0 / dist.get_world_size())


def all_gather_flat(out, inp, async_op=False, group=None):
    """out[r * n:(r + 1) * n] <- rank r's `inp` (out may contain inp in place)"""
    if _native() or not inp.is_cuda:
        return dist.all_gather_into_tensor(out, inp, async_op=async_op, group=group)
    rank, n = dist.get_rank(), inp.numel()
    tmp = torch.zeros_like(out)
    tmp[rank * n:(rank + 1) * n].copy_(inp)
    dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
    out.copy_(tmp)
    return None


class _Shards:
    """Per-tower plan: which flat ranges travel together and which slice of each this rank owns."""

    def __init__(self, buckets, trainable, rank, world):
        # buckets: [(begin, end)] flat element ranges in backward-completion order; trainable: [[begin, end)] ranges
        self.rank, self.world = rank, world
        self.buckets = []                       # (begin, end, own_begin, own_end, shard_offset, [trainable ranges inside own])
        off = 0
        for (b0, b1) in buckets:
            tr = [(max(a, b0), min(b, b1)) for a, b in trainable if max(a, b0) < min(b, b1)]
            if b1 <= b0 or not tr:
                self.buckets.append(None)       # empty or frozen: never written by the backward, nothing to exchange
                continue
            n = b1 - b0
            if n % world:
                raise ValueError(f'bucket of {n} elements is not divisible by the world size {world}')
            per = n // world
            o0, o1 = b0 + rank * per, b0 + (rank + 1) * per
            own_tr = [(max(a, o0), min(b, o1)) for a, b in tr if max(a, o0) < min(b, o1)]
            self.buckets.append((b0, b1, o0, o1, off, own_tr))
            off += per
        self.shard_elems = off

    def live(self):
        return [b for b in self.buckets if b is not None]


class GradSync:
    """Gradient exchange of the student towers.  `sharded` (default, DCLIP_DP_MODE != 'allreduce'): bucketed reduce-scatter
    released from inside the backward + sharded optimizer + parameter all-gather; otherwise the flat all-reduce of round 1."""

    def __init__(self, bucket_elems=32 * 1024 * 1024, sharded=None):
        mode = os.environ.get('DCLIP_DP_MODE', 'reduce_scatter')
        # DCLIP_DP_MODE=off: somebody else owns the gradient exchange (Lightning's DDP wrapper over p.grad, a torch optimizer):
        # nothing here touches the gradients
        self.enabled = mode != 'off' and _dist_on() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if _dist_on() else 1
        self.rank = dist.get_rank() if _dist_on() else 0
        self.bucket = bucket_elems
        if sharded is None:
            sharded = mode != 'allreduce'
        self.sharded = bool(sharded) and 64 % max(self.world, 1) == 0      # parameter segments are 64-element aligned
        # The per-bucket release from INSIDE a tower's backward (reduce-scatter + clearing of the exchanged bucket) happens only
        # while armed, i.e. inside backward_and_sync(): a bare loss.backward() leaves p.grad whole for whoever consumes it
        # (gradient clipping, grad-norm logging, a DDP wrapper, a torch optimizer).
        self.armed = False
        # one exchange stream AND one communicator per tower: the image tower's buckets do not queue behind the text tower's
        # embedding bucket (collectives of one communicator execute in issue order)
        self._streams = {}
        self._groups = {}
        self._nkeys = 0                          # towers are keyed by the order they were attached in (the same on every rank)
        self._stream = None                      # (flat all-reduce fallback)
        self._pending = []
        # token-embedding table of the (uncompressed) text student: exchange only the rows the global batch touched
        self.sparse_embedding = os.environ.get('DCLIP_DP_SPARSE_EMBED', '1') != '0'

    @staticmethod
    def current(existing=None):
        """the GradSync a model should use now: torch.distributed may have been initialised after the last call"""
        if existing is None or (not existing.enabled and _dist_on() and dist.get_world_size() > 1):
            return GradSync()
        return existing

    def attach(self, towers):
        """bind the student towers: in sharded mode, plan the buckets / shards of every materialised tower once"""
        for tw in towers:
            tw.sync = self
            if getattr(tw, '_dp_key', None) is None or getattr(tw, '_dp_owner', None) is not self:
                # (not id(tw): an id can be reused by another object once a tower is collected)
                tw._dp_key, tw._dp_owner = self._nkeys, self
                self._nkeys += 1
            if self.enabled and self.sharded and tw.dp is None and tw.flat is not None:
                self.plan(tw, tw.trainable_ranges())
            if self.enabled and self.sharded and tw._dp_key not in self._groups:
                # collective: every rank attaches its towers in the same order
                self._groups[tw._dp_key] = dist.new_group() if os.environ.get('DCLIP_DP_TOWER_GROUPS', '0') != '0' else None
        return self

    def _key(self, tw):
        k = getattr(tw, '_dp_key', None)
        if k is None or getattr(tw, '_dp_owner', None) is not self:
            self.attach([tw])
            k = tw._dp_key
        return k

    def group_for(self, tw):
        return self._groups.get(self._key(tw))

    def close(self):
        """destroy the per-tower communicators (collective: every rank, same order) and drop the streams"""
        for k in sorted(self._groups):
            g = self._groups[k]
            if g is not None and _dist_on():
                dist.destroy_process_group(g)
        self._groups, self._streams = {}, {}

    # ---- stream plumbing (no-ops for the gloo / CPU rehearsal of the same call pattern) ---------------------------
    def stream_for(self, t, tw=None):
        if not t.is_cuda:
            return None
        if tw is not None:
            k = self._key(tw)
            s = self._streams.get(k)
            if s is None:
                s = self._streams[k] = torch.cuda.Stream(device=t.device)
            return s
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=t.device)
        return self._stream

    class _On:
        def __init__(self, stream):
            self.ctx = torch.cuda.stream(stream) if stream is not None else None

        def __enter__(self):
            if self.ctx is not None:
                self.ctx.__enter__()

        def __exit__(self, *a):
            if self.ctx is not None:
                self.ctx.__exit__(*a)

    # ---- sharded mode ------------------------------------------------------------------------------------------------
    def plan(self, tw, trainable):
        """attach the shard plan of tower `tw` (HipTower or any object with .flat / .flat_grad / .grad_buckets())."""
        sh = _Shards(tw.grad_buckets(), trainable, self.rank, self.world)
        tw.dp = sh
        tw.gshard = torch.zeros(max(sh.shard_elems, 1), dtype=torch.float32, device=tw.flat.device)
        tw.dp_released = 0
        tw.dp_unstepped = set()
        return sh

    def bucket_ready(self, tw, i, after=None):
        """Called (on the thread that issues the backward, current stream = the backward's stream) as soon as every launch that
        writes bucket i of `tw.flat_grad` has been enqueued: average it across ranks into this rank's shard.  `after`: event that
        marks the bucket complete (default: everything enqueued on the current stream so far)."""
        sh = tw.dp
        b = sh.buckets[i]
        tw.dp_released = max(tw.dp_released, i + 1)
        if b is None:
            return
        if i in tw.dp_unstepped:
            # the shard holds an average the optimizer has not consumed yet: a second backward before step() (gradient
            # accumulation) would silently replace it
            raise RuntimeError('GradSync: gradient bucket released twice before optimizer.step(): the sharded exchange supports '
                               'accumulate_grad_batches = 1 only (use DCLIP_DP_MODE=allreduce to accumulate)')
        tw.dp_unstepped.add(i)
        b0, b1, o0, o1, off, _ = b
        g = tw.flat_grad
        s = self.stream_for(g, tw)
        if s is not None:
            if after is None:
                after = torch.cuda.Event()
                after.record(torch.cuda.current_stream())
            s.wait_event(after)
        with GradSync._On(s):
            sp = getattr(tw, '_sparse', None)
            if sp is not None and sp['bucket'] == i and not sp.get('dense'):
                self._release_sparse(tw, b, sp)
                tw._sparse = None
            else:
                if sp is not None and sp['bucket'] == i:
                    tw._sparse = None                   # (a step marked dense by note_token_ids: the whole bucket travels and is cleared)
                reduce_scatter_avg(tw.gshard[off:off + (o1 - o0)], g[b0:b1], group=self.group_for(tw))
                # (the backward accumulates (+=): the exchanged buffer is cleared ONCE, behind the last bucket's collective, in finish() —
                #  ten fills per tower and step before)

    # ---- row-sparse exchange of the token-embedding gradient (reference weight_share_model.py:407) ----
    def note_token_ids(self, tw, ids):
        """Forward time, text student with a plain (uncompressed) embedding table: the gradient of the [V, D] table is non-zero only
        in the rows of the token ids of the GLOBAL batch.  The ranks exchange their ids ([B, 77] int32 each), every rank builds
        the same sorted union with static shapes (sort, first-occurrence flags, prefix sum: no host synchronisation), counts the
        union rows that fall into each rank's shard of the bucket, and those W counts travel to pinned host memory; by the time
        the embedding bucket is released at the end of the backward they have long arrived, and the collective over the compacted
        rows can be sized on the host."""
        spec = tw.sparse_spec() if hasattr(tw, 'sparse_spec') else None
        if spec is None or not (self.enabled and self.sharded and self.sparse_embedding) or tw.dp is None:
            return
        bucket, t0, V, D = spec
        bk = tw.dp.buckets[bucket]
        if bk is None:
            return
        if getattr(tw, '_sparse', None) is not None:
            # a second grad-enabled forward before the bucket was released (two text forwards per step, or a forward whose
            # backward never ran): the union of THIS call would miss the rows of the earlier one, which would then neither travel
            # nor be cleared.  Every rank sees the same call sequence, so every rank takes the dense bucket for this step.
            tw._sparse = dict(bucket=bucket, dense=True)
            return
        b0, b1 = bk[0], bk[1]
        n = ids.numel()
        cap = getattr(tw, '_sparse_cap', None)
        if cap is None:
            # ids per rank and forward: fixed by the FIRST forward, which also checks (one host synchronisation, once) that every rank
            # holds the same count.  Later, smaller batches (a ragged last batch) are padded with the "no row" id V; a larger one
            # cannot be announced to the other ranks without a collective they do not expect.
            check_equal_batch(n, ids.device)
            cap = tw._sparse_cap = n
        if n > cap:
            raise RuntimeError(f'GradSync: {n} token ids in this forward, but the row-sparse embedding exchange was sized for {cap} by '
                               'the first forward: keep the per-rank batch <= the first one, or set DCLIP_DP_SPARSE_EMBED=0')
        s = self.stream_for(ids, tw)
        if s is not None:
            s.wait_stream(torch.cuda.current_stream())
        with GradSync._On(s):
            # (everything the exchange stream reads is allocated ON it: a tensor made on the compute stream and freed at the end of
            #  this function could be handed out again by the caching allocator before this stream has read it)
            loc = torch.full((cap,), V, dtype=torch.int32, device=ids.device)
            loc[:n].copy_(ids.reshape(-1))
            bounds = self._segment_bounds(tw, b0, b1, t0, V, D, ids.device)
            allids = torch.empty(self.world * cap, dtype=torch.int32, device=loc.device)
            all_gather_flat(allids, loc, group=self.group_for(tw))
            srt = torch.sort(allids.long()).values
            first = torch.ones_like(srt, dtype=torch.bool)
            first[1:] = srt[1:] != srt[:-1]
            pos = torch.cumsum(first.to(torch.int64), 0) - 1
            uniq = torch.full((min(V, self.world * cap) + 1,), V, dtype=torch.int64, device=loc.device)   # V = "no row" (sorts last)
            uniq.scatter_(0, pos, srt)                   # duplicates write the same value to the same slot
            cuts = torch.searchsorted(uniq, bounds)      # [2 W]: first union index >= r_lo / >= r_hi of every segment
            if loc.is_cuda:
                host = torch.empty(2 * self.world, dtype=torch.int64, pin_memory=True)
                host.copy_(cuts, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(s)
            else:
                host, ev = cuts.clone(), None
        tw._sparse = dict(bucket=bucket, t0=t0, V=V, D=D, uniq=uniq, cuts=host, cuts_dev=cuts, event=ev)

    def _segment_bounds(self, tw, b0, b1, t0, V, D, device):
        """[2 W] table rows [r_lo, r_hi) that intersect rank k's shard [b0 + k per, b0 + (k + 1) per) of the embedding bucket — a shard
        boundary may cut a row, which then belongs to both neighbours' segments.  Built once per tower and plan and kept on the
        device: a host list -> device tensor per step would be a blocking pageable copy in stream order, i.e. the host would wait in
        every forward for everything enqueued so far."""
        key = (b0, b1, t0, V, D, self.world, str(device))
        cached = getattr(tw, '_sparse_bounds', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        per = (b1 - b0) // self.world
        t1 = t0 + V * D
        seg = []
        for k in range(self.world):
            lo, hi = max(b0 + k * per, t0), min(b0 + (k + 1) * per, t1)
            seg += [0, 0] if lo >= hi else [(lo - t0) // D, (hi - t0 + D - 1) // D]
        bounds = torch.tensor(seg, dtype=torch.int64, device=device)
        tw._sparse_bounds = (key, bounds)
        return bounds

    def _release_sparse(self, tw, b, sp):
        """(on the tower's exchange stream) a reduce-scatter over the TOUCHED table rows only: every rank contributes, per destination
        rank, the union rows of that rank's shard (segments padded to the longest one, so the collective has equal parts), receives
        the average of its own segment and writes it into its shard (zeros in the untouched rows), then clears the touched rows.
        Bytes on the wire: (W - 1) / W x W x longest segment x D x 4 ~ (W - 1) / W x touched rows x D x 4, never more than the dense
        reduce-scatter of V rows."""
        b0, b1, o0, o1, off, _ = b
        if sp['event'] is not None:
            sp['event'].synchronize()                   # recorded at forward time: done long ago
        cuts = [int(x) for x in sp['cuts'].tolist()]
        W, rank = self.world, self.rank
        starts, counts = cuts[0::2], [cuts[2 * k + 1] - cuts[2 * k] for k in range(W)]
        t0, V, D = sp['t0'], sp['V'], sp['D']
        t1 = t0 + V * D
        g = tw.flat_grad
        grp = self.group_for(tw)
        shard = tw.gshard[off:off + (o1 - o0)]
        shard.zero_()
        uniq = sp['uniq']
        table = g[t0:t1].view(V, D)
        mseg = max(max(counts), 1)
        ar = torch.arange(mseg, device=uniq.device)
        # (segment starts / counts come from the device copy of the cuts: a host list -> device tensor here would be a synchronous
        #  copy in stream order, i.e. the host would wait for the backward that this stream waits for)
        cd = sp['cuts_dev'].view(W, 2)
        st = cd[:, 0:1]
        ct = cd[:, 1:2] - cd[:, 0:1]
        valid = ar.unsqueeze(0) < ct                                             # [W, mseg]
        idx = (st + ar.unsqueeze(0)).clamp_(max=uniq.numel() - 1)
        rows_all = torch.where(valid, uniq[idx], torch.zeros_like(idx))          # padding reads row 0 and is masked to zero
        send = table.index_select(0, rows_all.reshape(-1)).view(W, mseg, D) * valid.unsqueeze(2).to(table.dtype)
        recv = torch.empty((mseg, D), dtype=table.dtype, device=table.device)
        reduce_scatter_avg(recv.view(-1), send.view(-1), group=grp)
        # owned part of the table: flat [lo, hi) -> rows [r_lo, r_hi) (the shard boundary may cut a row)
        lo, hi = max(o0, t0), min(o1, t1)
        n_own = counts[rank]
        if lo < hi and n_own > 0:
            r_lo, r_hi = (lo - t0) // D, (hi - t0 + D - 1) // D
            rows = rows_all[rank, :n_own]
            buf = torch.zeros((r_hi - r_lo, D), dtype=table.dtype, device=table.device)
            buf.index_copy_(0, rows - r_lo, recv[:n_own])
            start = lo - (t0 + r_lo * D)
            shard[lo - o0:hi - o0].copy_(buf.view(-1)[start:start + (hi - lo)])
        touched = uniq[:max(cuts)]                       # the union (the segments cover every table row of the bucket)
        # the rest of the bucket (positional embedding, ...) is small and dense
        for a, e in ((b0, t0), (t1, b1)):
            if a < e:
                rest = g[a:e]
                all_reduce_avg(rest, group=grp)
                x0, x1 = max(a, o0), min(e, o1)
                if x0 < x1:
                    shard[x0 - o0:x1 - o0].copy_(g[x0:x1])
        tw.sparse_rows_last = int(touched.numel())       # (diagnostics / tests: rows exchanged instead of V)
        tw.sparse_segment_rows_last = mseg               # rows per destination rank in the padded collective

    def finish(self, tw):
        """after the backward call: release whatever the callbacks did not (a backward without per-bucket callbacks)"""
        for i in range(tw.dp_released, len(tw.dp.buckets)):
            self.bucket_ready(tw, i, after=getattr(tw, 'bwd_done', None))
        tw.dp_released = 0
        s = self._streams.get(self._key(tw))
        # every bucket has been handed to its collective on the exchange stream: clear the whole gradient buffer behind them, once
        # (the next backward of this tower is ordered after this stream through the optimizer's opt_done / grads_ready events)
        with GradSync._On(s if tw.flat_grad.is_cuda else None):
            tw.flat_grad.zero_()
        if tw.flat_grad.is_cuda and s is not None:
            done = torch.cuda.Event()
            done.record(s)
            return done
        return None

    def gather_params(self, tw, i):
        """all-gather bucket i of the parameters from the ranks' updated shards (call on the exchange stream)"""
        b = tw.dp.buckets[i]
        if b is None:
            return
        b0, b1, o0, o1, _, _ = b
        all_gather_flat(tw.flat[b0:b1], tw.flat[o0:o1], group=self.group_for(tw))

    def gather_full(self, tw, shard):
        """[shard_elems] per-rank optimizer state -> full flat-layout tensor (checkpointing; collective)"""
        full = torch.zeros_like(tw.flat)
        for b in tw.dp.live():
            b0, b1, o0, o1, off, _ = b
            all_gather_flat(full[b0:b1], shard[off:off + (o1 - o0)].contiguous(), group=self.group_for(tw))
        return full

    # ---- flat all-reduce (fallback; the reference's DDP semantics literally) -----------------------------------------------
    def launch(self, flat_grad, after=None):
        """average `flat_grad` across ranks, asynchronously with respect to the compute stream.  `after`: CUDA event that
        marks the buffer complete (a tower's end-of-backward event); default = everything enqueued on the current stream."""
        if not self.enabled or flat_grad is None:
            return None
        if flat_grad.is_cuda:
            s = self.stream_for(flat_grad)
            if after is not None:
                s.wait_event(after)
            else:
                s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for b in range(0, flat_grad.numel(), self.bucket):
                    chunk = flat_grad[b:b + self.bucket]
                    dist.all_reduce(chunk, op=dist.ReduceOp.AVG)      # RCCL averages in the collective itself
            self._pending.append(flat_grad)
            done = torch.cuda.Event()
            done.record(s)              # this buffer's average is complete: a per-tower optimizer step may wait on it
            return done
        else:   # gloo / CPU rehearsal of the same call pattern
            for b in range(0, flat_grad.numel(), self.bucket):
                chunk = flat_grad[b:b + self.bucket]
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
                chunk.mul_(1.0 / self.world)

    def wait(self):
        for s in [self._stream] + list(self._streams.values()):
            if s is not None:
                torch.cuda.current_stream().wait_stream(s)
        self._pending = []

    def forget(self):
        """the caller carries the dependency itself (per-buffer events returned by launch / finish)"""
        self._pending = []


def world_size():
    return dist.get_world_size() if _dist_on() else 1


def check_equal_batch(B, device):
    """Global-negative mode addresses gathered rows as rank * B: every rank must hold the same B (a loader without drop_last
    would desynchronise the collectives).  One tiny all-reduce of (B, -B); raises the same error on every rank."""
    if not _dist_on() or dist.get_world_size() == 1:
        return
    t = torch.tensor([float(B), -float(B)], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    bmax, bmin = int(t[0].item()), int(-t[1].item())
    if bmax != bmin:
        raise ValueError(f'global negatives need the same per-rank batch on every rank (got {bmin}..{bmax}): use drop_last=True')


def gather_embeddings(tensors):
    """All-gather a list of [B, E] tensors across ranks with ONE collective: -> ([W*B, E] tensors, rank, world).

    Global-negative mode (not in the reference's training_step): every rank evaluates the fused loss on the
    gathered batch and keeps the gradient rows of its own shard, so no reduce-scatter of embedding gradients is needed; the
    gradient is scaled by `world` so that the DDP *average* of parameter gradients equals the single-process gradient of the
    loss on the concatenated batch."""
    if not _dist_on():
        return list(tensors), 0, 1
    world, rank = dist.get_world_size(), dist.get_rank()
    packed = torch.cat([t.detach().float() for t in tensors], dim=1).contiguous()          # [B, k*E]: one fused gather
    out = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
    all_gather_flat(out.view(-1), packed.view(-1))
    full = out.view(world * packed.shape[0], packed.shape[1])
    widths = [t.shape[1] for t in tensors]
    return [c.contiguous() for c in torch.split(full, widths, dim=1)], rank, world
