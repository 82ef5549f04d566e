"""Fused AdamW over the towers' flat parameter buffers (reference distil_model.py:160-169,
dual_distill_model.py:194-202: AdamW over every requires_grad parameter in one group + HF cosine-with-warmup
stepped per epoch)."""
import math

import torch

from ._lib import lib


def cosine_with_warmup(step, warm, total):
    """transformers.get_cosine_schedule_with_warmup's lr multiplier (num_cycles = 0.5)."""
    if step < warm:
        return float(step) / float(max(1, warm))
    prog = float(step - warm) / float(max(1, total - warm))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


class FusedAdamW:
    """One dclip_adamw launch per contiguous trainable range of each tower's flat buffer.

    The trainable set is fixed when the optimizer is built, like the reference's AdamW(filter(requires_grad, parameters()))
    (dual_distill_model.py:195, distil_model.py:161): parameters unfrozen later (unfreeze_embed) do not enter it.

    Data-parallel runs (tower.dp set by parallel.GradSync.plan): each rank updates only its 1/W shard of every gradient bucket
    from the reduce-scattered average, keeps m / v for that shard only, and the updated parameters are all-gathered."""

    def __init__(self, towers, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, extra_params=()):
        """extra_params: trainable parameters that live outside the towers' flat buffers — the embedding_projection / hidden_projection
        linears of a plain CLIP encoder in the student role (reference image_encoder.py:23-25, text_encoder.py:45-47): four small tensors
        whose gradients autograd produces; each gets its own dclip_adamw launch (after an all-reduce of its gradient in a data-parallel
        run whose exchange this package owns)."""
        self.towers = list(towers)
        self.extras = [p for p in extra_params if p.requires_grad]
        self._extra_state = {}
        self.base_lr = self.lr = lr
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.step_count = 0
        self._state = {}
        self._fixed_ranges = {}
        for tw in self.towers:
            if tw.flat is not None:
                self._fixed_ranges[id(tw)] = [list(r) for r in tw.trainable_ranges()]
        # overlap mode only: also re-cast the bf16 weight cache right after the update.  Off by default: the re-cast of the NEXT
        # forward runs under the teacher towers' forward (4 streams wide), which hides it better than the end of the step does
        self.refresh_cache_in_step = False

    def _ranges(self, tw):
        """contiguous [begin, end) element ranges of the parameters this optimizer was built over"""
        r = self._fixed_ranges.get(id(tw))
        if r is None:                                  # tower materialised after construction: captured at first use
            r = self._fixed_ranges[id(tw)] = [list(x) for x in tw.trainable_ranges()]
        return r

    def _adamw_hip(self, p, g, m, v, zero_grad, st):
        """p, g, m, v: equally long 1-D f32 views.  (tests/test_parallel_cpu.py substitutes a torch version as `_adamw` to rehearse the
        sharded bookkeeping over gloo; the product path is the HIP kernel.)"""
        lib().dclip_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), self.lr, self.betas[0],
                          self.betas[1], self.eps, self.weight_decay, self.step_count, 1 if zero_grad else 0, st)

    _adamw = _adamw_hip

    def _adamw_many(self, items, zero_grad, st):
        """items: [(p, g, m, v)] of equally long 1-D f32 views — ONE dclip_adamw_multi launch per 24 ranges (the sharded step has one
        owned slice per gradient bucket: nine launches per step for the two l_clip students became one per tower)"""
        import ctypes
        if type(self)._adamw is not FusedAdamW._adamw_hip or any(p.numel() % 4 or (p.data_ptr() | g.data_ptr() | m.data_ptr() | v.data_ptr()) % 16
                                                                  for p, g, m, v in items):
            for p, g, m, v in items:
                self._adamw(p, g, m, v, zero_grad, st)
            return
        for i in range(0, len(items), 24):
            chunk = items[i:i + 24]
            n = len(chunk)
            arr = lambda k: (ctypes.c_void_p * n)(*[t[k].data_ptr() for t in chunk])
            lens = (ctypes.c_int64 * n)(*[t[0].numel() for t in chunk])
            lib().dclip_adamw_multi(arr(0), arr(1), arr(2), arr(3), lens, n, self.lr, self.betas[0], self.betas[1], self.eps,
                                    self.weight_decay, self.step_count, 1 if zero_grad else 0, st)

    def zero_grad(self, set_to_none=False):
        from .model.component._tower import autograd_params_mode
        for tw in self.towers:
            if tw.flat is not None and autograd_params_mode(tw):
                # the gradients belong to autograd (AccumulateGrad / a DDP reducer): drop them, as torch's zero_grad(set_to_none) does
                for p in tw._params():
                    if p is not None and p.requires_grad:
                        p.grad = None
            if tw.flat_grad is not None and not getattr(tw, '_grad_clean', False):
                self.join()
                tw.flat_grad.zero_()
        for p in self.extras:
            p.grad = None

    @staticmethod
    def _pack_autograd_grads(tw):
        """DCLIP_DP_MODE=off / tower.autograd_params: the backward handed the parameter gradients to autograd, so p.grad are autograd's own
        tensors (or a DistributedDataParallel reducer's bucket views) and tower.flat_grad, which the kernel below reads, was never
        written.  Copy them into the flat buffer (one multi-tensor copy); a trainable parameter without a gradient counts as zero."""
        dst, src = [], []
        live = [p for p in tw._params() if p is not None]
        for p, off in zip(live, tw._offsets):
            if not p.requires_grad:
                continue
            view = tw.flat_grad[off:off + p.numel()]
            g = p.grad
            if g is None:
                view.zero_()
            elif g.data_ptr() != view.data_ptr():
                if g.dtype != torch.float32 or g.device != view.device:
                    raise RuntimeError('FusedAdamW: parameter gradients must be f32 tensors on the tower\'s device')
                dst.append(view)
                src.append(g.reshape(-1))
        if dst:
            torch._foreach_copy_(dst, src)

    def _ranges_cover_everything(self, tw):
        r = self._ranges(tw)
        return len(r) == 1 and r[0][0] == 0 and r[0][1] >= tw.flat.numel()

    @staticmethod
    def _sharded(tw):
        return getattr(tw, 'dp', None) is not None and getattr(tw, 'sync', None) is not None and tw.sync.enabled

    def _moments(self, tw):
        key = id(tw)
        n = tw.dp.shard_elems if self._sharded(tw) else tw.flat.numel()
        if key in self._state and self._state[key][0].numel() != max(n, 1):
            # the layout of the moments changed under us (torch.distributed initialised, or the shard plan swapped, after state
            # was created or loaded): re-zeroing would silently drop trained / restored Adam moments
            raise RuntimeError(f'FusedAdamW: optimizer state of {self._state[key][0].numel()} elements, but the tower now needs '
                               f'{max(n, 1)} (sharded={self._sharded(tw)}): build the optimizer / load its state after '
                               'torch.distributed and the data-parallel plan are set up')
        if key not in self._state:
            self._state[key] = (torch.zeros(max(n, 1), dtype=torch.float32, device=tw.flat.device),
                                torch.zeros(max(n, 1), dtype=torch.float32, device=tw.flat.device))
        return self._state[key]

    def _step_sharded(self, tw):
        """reduce-scattered gradient shards -> AdamW on the owned slices -> all-gather of the updated parameters, all on the
        exchange stream behind the tower's reduce-scatters (which were released from inside its backward)."""
        from .parallel import all_gather_flat
        sync = tw.sync
        m, v = self._moments(tw)
        s = sync.stream_for(tw.flat, tw)
        grp = sync.group_for(tw)
        if s is not None:
            # whatever the caller's stream holds before step() must be visible on the exchange stream — in particular the
            # zero-fill of freshly allocated m / v (first step): without this edge AdamW read uninitialised moments at world
            # size 2 (NaN weights after one step; found by tests/test_parallel_gpu.py)
            s.wait_stream(torch.cuda.current_stream())
        works = []
        with sync._On(s):
            st = s.cuda_stream if s is not None else None
            items = []
            for b in tw.dp.buckets:
                if b is None:
                    continue
                b0, b1, o0, o1, off, own_tr = b
                for a, e in own_tr:
                    lo, hi = off + a - o0, off + e - o0
                    items.append((tw.flat[a:e], tw.gshard[lo:hi], m[lo:hi], v[lo:hi]))
            if items:
                self._adamw_many(items, False, st)
            for b in tw.dp.buckets:
                if b is not None:
                    b0, b1, o0, o1, off, own_tr = b
                    works.append(all_gather_flat(tw.flat[b0:b1], tw.flat[o0:o1], async_op=True, group=grp))
            for w in works:
                if w is not None:
                    w.wait()
        tw.wcache_dirty = True
        tw.grads_ready = None
        tw.dp_unstepped.clear()                     # the exchanged averages are consumed: the next backward may release again
        if s is not None:
            tw.opt_done = torch.cuda.Event()
            tw.opt_done.record(s)
        return s

    @torch.no_grad()
    def step(self, zero_grad=False, overlap=False, join=True):
        """zero_grad=True: the kernel clears each gradient element once it has consumed it (saves the separate 306 MB fill that
        otherwise runs alone on the main stream); the next zero_grad() is then free.  Frozen ranges are never written by the
        backward, so a fully trainable tower stays clean until its next backward.

        overlap=True: each tower is updated on the stream its backward ran on, as soon as that backward (and, under data
        parallelism, that tower's gradient exchange) is done, followed by the re-cast of its bf16 weight cache — the shorter
        tower's update then runs under the longer tower's backward instead of alone at the end of the step.  The current stream
        is ordered after every tower's update before step() returns, unless join=False: then only the tower streams carry the
        dependency (backward -> exchange -> update -> next forward of that tower) and the next step's frozen teacher towers
        may start while the last gradient exchange and update are still running; join() orders the current stream after them
        (zero_grad() and state_dict() call it)."""
        from .model.component._tower import autograd_params_mode
        self.step_count += 1
        main = torch.cuda.current_stream() if torch.cuda.is_available() else None
        joined = []
        for tw in self.towers:
            if tw.flat is None:
                continue
            if self._sharded(tw):
                s = self._step_sharded(tw)
                if s is not None:
                    joined.append(s)
                continue
            m, v = self._moments(tw)
            through_autograd = autograd_params_mode(tw)
            # (gradients that went through autograd may have been written by anybody's stream — a DDP reducer's —: take them on `main`)
            stream = tw.bwd_stream if (overlap and not through_autograd and getattr(tw, 'bwd_stream', None) is not None) else main
            if stream != main:
                stream.wait_stream(main)                     # whatever the caller enqueued before step() (e.g. zero_grad of others)
            if getattr(tw, 'grads_ready', None) is not None:
                stream.wait_event(tw.grads_ready)            # this tower's gradient average (RCCL side stream), whichever stream updates
                tw.grads_ready = None
            with torch.cuda.stream(stream):
                st = stream.cuda_stream
                if through_autograd:
                    self._pack_autograd_grads(tw)
                self._adamw_many([(tw.flat[b:e], tw.flat_grad[b:e], m[b:e], v[b:e]) for b, e in self._ranges(tw)], zero_grad, st)
                tw.wcache_dirty = True
                tw._grad_clean = bool(zero_grad) and self._ranges_cover_everything(tw)
                if overlap and self.refresh_cache_in_step:
                    tw._prepare_always = False               # from now on this optimizer keeps the bf16 cache in step
                    tw.prepare()
            if stream != main:
                joined.append(stream)
                tw.opt_done = torch.cuda.Event()
                tw.opt_done.record(stream)
        self._step_extras(zero_grad, main)
        if join:
            for stream in joined:
                main.wait_stream(stream)
            for tw in self.towers:
                tw.opt_done = None

    def _extra_moments(self, p):
        st = self._extra_state.get(id(p))
        if st is None:
            st = self._extra_state[id(p)] = (torch.zeros(p.numel(), dtype=torch.float32, device=p.device),
                                             torch.zeros(p.numel(), dtype=torch.float32, device=p.device))
        return st

    def _step_extras(self, zero_grad, main):
        """the parameters outside the tower buffers, on the current stream (their gradients were written by autograd on it)"""
        if not self.extras:
            return
        from .parallel import all_reduce_avg
        sync = next((tw.sync for tw in self.towers if getattr(tw, 'sync', None) is not None), None)
        items = []
        for p in self.extras:
            if p.grad is None:
                continue                                  # torch.optim.AdamW skips parameters without a gradient
            if not (p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32 and p.grad.dtype == torch.float32):
                raise RuntimeError('FusedAdamW: extra parameters and their gradients must be contiguous f32 tensors')
            if sync is not None and sync.enabled:
                all_reduce_avg(p.grad)
            m, v = self._extra_moments(p)
            items.append((p.data.view(-1), p.grad.view(-1), m, v))
        if items:
            self._adamw_many(items, zero_grad, main.cuda_stream if main is not None else None)

    def join(self):
        """order the current stream after every tower's pending (un-joined) update"""
        if not torch.cuda.is_available():
            return
        cur = torch.cuda.current_stream()
        for tw in self.towers:
            ev = getattr(tw, 'opt_done', None)
            if ev is not None:
                cur.wait_event(ev)
                tw.opt_done = None

    def _full_moments(self, tw):
        """(m, v) in the flat layout of the tower; collective in a sharded data-parallel run (all-gather of the ranks' shards)"""
        m, v = self._moments(tw)
        if self._sharded(tw):
            return tw.sync.gather_full(tw, m), tw.sync.gather_full(tw, v)
        return m, v

    # ---- torch.optim.AdamW-compatible (de)serialisation: what a Lightning checkpoint stores under 'optimizer_states' ----
    def _slots(self, params=None):
        """[(tower, offset, numel, shape)] of the trainable parameters, in `params` order (default: tower order)."""
        where = {}
        for tw in self.towers:
            if tw.flat is None:
                continue
            live = [p for p in tw._params() if p is not None]
            rng = self._ranges(tw)
            for p, off in zip(live, tw._offsets):
                if any(a <= off < b for a, b in rng):        # the set captured at construction, not the live flags
                    where[p.data_ptr()] = (tw, off, p.numel(), tuple(p.shape))
        for p in self.extras:
            where[p.data_ptr()] = (p, None, p.numel(), tuple(p.shape))
        if params is None:
            return list(where.values())
        out = []
        for p in params:
            if p.data_ptr() not in where and not p.requires_grad:
                continue
            if p.data_ptr() not in where:
                if any(p.data_ptr() == q.data_ptr() for tw in self.towers if tw.flat is not None
                       for q in tw._params() if q is not None):
                    continue                                  # unfrozen after the optimizer was built: not one of its slots
                raise ValueError('FusedAdamW.state_dict: a trainable parameter is not a view of a tower buffer')
            out.append(where[p.data_ptr()])
        return out

    def state_dict(self, params=None):
        """`params`: the iteration order torch.optim.AdamW would have been built with (reference distil_model.py:161,
        dual_distill_model.py:195: filter(requires_grad, self.parameters())); default = canonical tower order.

        COLLECTIVE in a sharded data-parallel run: every rank holds 1/W of m / v, so every rank must call this (the moments are
        all-gathered per bucket); checkpoint.save_checkpoint does that and lets rank 0 alone write the file."""
        self.join()
        slots = self._slots(params)
        state = {}
        full = {id(tw): self._full_moments(tw) for tw in self.towers if id(tw) in self._state}
        for i, (tw, off, n, shape) in enumerate(slots):
            if off is None:                               # a parameter outside the tower buffers
                if id(tw) in self._extra_state:
                    m, v = self._extra_state[id(tw)]
                    state[i] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': m.view(shape).clone(),
                                'exp_avg_sq': v.view(shape).clone()}
                continue
            if id(tw) in full:
                m, v = full[id(tw)]
                state[i] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': m[off:off + n].view(shape).clone(),
                            'exp_avg_sq': v[off:off + n].view(shape).clone()}
        group = {'lr': self.lr, 'initial_lr': self.base_lr, 'betas': tuple(self.betas), 'eps': self.eps,
                 'weight_decay': self.weight_decay, 'amsgrad': False, 'maximize': False, 'foreach': None,
                 'capturable': False, 'differentiable': False, 'fused': None, 'params': list(range(len(slots)))}
        return {'state': state, 'param_groups': [group]}

    @torch.no_grad()
    def load_state_dict(self, sd, params=None):
        slots = self._slots(params)
        group = sd['param_groups'][0]
        if len(group['params']) != len(slots):
            raise ValueError(f"FusedAdamW.load_state_dict: {len(group['params'])} saved parameters, {len(slots)} trainable here")
        self.lr = group['lr']
        self.base_lr = group.get('initial_lr', self.base_lr)
        self.betas, self.eps, self.weight_decay = tuple(group['betas']), group['eps'], group['weight_decay']
        steps = set()
        for i, (tw, off, n, shape) in enumerate(slots):
            st = sd['state'].get(i, sd['state'].get(str(i)))
            if st is None:
                continue
            if tuple(st['exp_avg'].shape) != shape:
                raise ValueError(f"FusedAdamW.load_state_dict: parameter {i} has shape {shape}, saved {tuple(st['exp_avg'].shape)}")
            if off is None:                               # a parameter outside the tower buffers
                m, v = self._extra_moments(tw)
                m.copy_(st['exp_avg'].reshape(-1))
                v.copy_(st['exp_avg_sq'].reshape(-1))
                steps.add(int(float(st['step'])))
                continue
            m, v = self._moments(tw)
            if self._sharded(tw):
                # keep the slices of this parameter that fall into the shards this rank owns
                for b in tw.dp.live():
                    b0, b1, o0, o1, soff, _ = b
                    lo, hi = max(off, o0), min(off + n, o1)
                    if lo < hi:
                        m[soff + lo - o0:soff + hi - o0].copy_(st['exp_avg'].reshape(-1)[lo - off:hi - off])
                        v[soff + lo - o0:soff + hi - o0].copy_(st['exp_avg_sq'].reshape(-1)[lo - off:hi - off])
            else:
                m[off:off + n].copy_(st['exp_avg'].reshape(-1))
                v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('FusedAdamW.load_state_dict: parameters with different step counts (one fused step counter here)')
        self.step_count = steps.pop() if steps else 0


class EpochCosineSchedule:
    """lr = base_lr * cosine_with_warmup(epoch): stepped once per epoch like the reference's Lightning default."""

    def __init__(self, optimizer, warm_steps, total_steps):
        self.opt, self.warm, self.total, self.epoch = optimizer, warm_steps, total_steps, 0
        self.opt.lr = self.opt.base_lr * cosine_with_warmup(0, warm_steps, total_steps)

    def step(self):
        self.epoch += 1
        self.opt.lr = self.opt.base_lr * cosine_with_warmup(self.epoch, self.warm, self.total)

    def get_last_lr(self):
        return [self.opt.lr]

    def state_dict(self):
        """LambdaLR-shaped (what the reference's scheduler — transformers.get_cosine_schedule_with_warmup, a LambdaLR — saves
        and what its load_state_dict updates from: base_lrs, last_epoch, _step_count, _last_lr, lr_lambdas=[None] for a
        plain function), plus the two schedule constants under their own keys"""
        return {'base_lrs': [self.opt.base_lr], 'last_epoch': self.epoch, 'verbose': False, '_step_count': self.epoch + 1,
                '_get_lr_called_within_step': False, '_last_lr': [self.opt.lr], 'lr_lambdas': [None],
                'warm_steps': self.warm, 'total_steps': self.total}

    def load_state_dict(self, sd):
        self.epoch = sd['last_epoch']
        if sd.get('base_lrs'):
            self.opt.base_lr = sd['base_lrs'][0]
        self.opt.lr = self.opt.base_lr * cosine_with_warmup(self.epoch, self.warm, self.total)
