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
    """One dclip_adamw launch per contiguous trainable range of each tower's flat buffer."""

    def __init__(self, towers, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.towers = list(towers)
        self.base_lr = self.lr = lr
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.step_count = 0
        self._state = {}
        # overlap mode only: also re-cast the bf16 weight cache right after the update.  Off by default: the re-cast of the NEXT
        # forward runs under the teacher towers' forward (4 streams wide), which hides it better than the end of the step does
        self.refresh_cache_in_step = False

    def _ranges(self, tw):
        """contiguous [begin, end) element ranges of trainable parameters inside tw.flat"""
        live = [p for p in tw._params() if p is not None]
        out = []
        for p, off in zip(live, tw._offsets):
            if not p.requires_grad:
                continue
            end = off + (p.numel() + 63) // 64 * 64
            if out and out[-1][1] == off:
                out[-1][1] = end
            else:
                out.append([off, end])
        return out

    def zero_grad(self, set_to_none=False):
        for tw in self.towers:
            if tw.flat_grad is not None and not getattr(tw, '_grad_clean', False):
                self.join()
                tw.flat_grad.zero_()

    def _ranges_cover_everything(self, tw):
        r = self._ranges(tw)
        return len(r) == 1 and r[0][0] == 0 and r[0][1] >= tw.flat.numel()

    @torch.no_grad()
    def step(self, zero_grad=False, overlap=False, join=True):
        """zero_grad=True: the kernel clears each gradient element once it has consumed it (saves the separate 306 MB fill that
        otherwise runs alone on the main stream); the next zero_grad() is then free.  Frozen ranges are never written by the
        backward, so a fully trainable tower stays clean until its next backward.

        overlap=True: each tower is updated on the stream its backward ran on, as soon as that backward (and, under data
        parallelism, that tower's gradient all-reduce) is done, followed by the re-cast of its bf16 weight cache — the shorter
        tower's update then runs under the longer tower's backward instead of alone at the end of the step.  The current stream
        is ordered after every tower's update before step() returns, unless join=False: then only the tower streams carry the
        dependency (backward -> all-reduce -> update -> next forward of that tower) and the next step's frozen teacher towers
        may start while the last gradient exchange and update are still running; join() orders the current stream after them
        (zero_grad() and state_dict() call it)."""
        self.step_count += 1
        main = torch.cuda.current_stream()
        joined = []
        for tw in self.towers:
            if tw.flat is None:
                continue
            key = id(tw)
            if key not in self._state or self._state[key][0].numel() != tw.flat.numel():
                self._state[key] = (torch.zeros_like(tw.flat), torch.zeros_like(tw.flat))
            m, v = self._state[key]
            stream = tw.bwd_stream if (overlap and getattr(tw, 'bwd_stream', None) is not None) else main
            if stream != main:
                stream.wait_stream(main)                     # whatever the caller enqueued before step() (e.g. zero_grad of others)
            if getattr(tw, 'grads_ready', None) is not None:
                stream.wait_event(tw.grads_ready)            # this tower's gradient average (RCCL side stream), whichever stream updates
                tw.grads_ready = None
            with torch.cuda.stream(stream):
                st = stream.cuda_stream
                for b, e in self._ranges(tw):
                    lib().dclip_adamw(tw.flat.data_ptr() + b * 4, tw.flat_grad.data_ptr() + b * 4, m.data_ptr() + b * 4,
                                      v.data_ptr() + b * 4, e - b, self.lr, self.betas[0], self.betas[1], self.eps,
                                      self.weight_decay, self.step_count, 1 if zero_grad else 0, st)
                tw.wcache_dirty = True
                tw._grad_clean = bool(zero_grad) and self._ranges_cover_everything(tw)
                if overlap and self.refresh_cache_in_step:
                    tw._prepare_always = False               # from now on this optimizer keeps the bf16 cache in step
                    tw.prepare()
            if stream != main:
                joined.append(stream)
                tw.opt_done = torch.cuda.Event()
                tw.opt_done.record(stream)
        if join:
            for stream in joined:
                main.wait_stream(stream)
            for tw in self.towers:
                tw.opt_done = None

    def join(self):
        """order the current stream after every tower's pending (un-joined) update"""
        cur = torch.cuda.current_stream()
        for tw in self.towers:
            ev = getattr(tw, 'opt_done', None)
            if ev is not None:
                cur.wait_event(ev)
                tw.opt_done = None

    # ---- torch.optim.AdamW-compatible (de)serialisation: what a Lightning checkpoint stores under 'optimizer_states' ----
    def _slots(self, params=None):
        """[(tower, offset, numel, shape)] of the trainable parameters, in `params` order (default: tower order)."""
        where = {}
        for tw in self.towers:
            if tw.flat is None:
                continue
            live = [p for p in tw._params() if p is not None]
            for p, off in zip(live, tw._offsets):
                if p.requires_grad:
                    where[p.data_ptr()] = (tw, off, p.numel(), tuple(p.shape))
        if params is None:
            return list(where.values())
        out = []
        for p in params:
            if not p.requires_grad:
                continue
            if p.data_ptr() not in where:
                raise ValueError('FusedAdamW.state_dict: a trainable parameter is not a view of a tower buffer')
            out.append(where[p.data_ptr()])
        return out

    def state_dict(self, params=None):
        """`params`: the iteration order torch.optim.AdamW would have been built with (reference distil_model.py:161,
        dual_distill_model.py:195: filter(requires_grad, self.parameters())); default = canonical tower order."""
        self.join()
        slots = self._slots(params)
        state = {}
        for i, (tw, off, n, shape) in enumerate(slots):
            if id(tw) in self._state:
                m, v = self._state[id(tw)]
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
            if id(tw) not in self._state:
                self._state[id(tw)] = (torch.zeros_like(tw.flat), torch.zeros_like(tw.flat))
            m, v = self._state[id(tw)]
            if tuple(st['exp_avg'].shape) != shape:
                raise ValueError(f"FusedAdamW.load_state_dict: parameter {i} has shape {shape}, saved {tuple(st['exp_avg'].shape)}")
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
        return {'last_epoch': self.epoch, 'warm_steps': self.warm, 'total_steps': self.total, '_last_lr': [self.opt.lr]}

    def load_state_dict(self, sd):
        self.epoch = sd['last_epoch']
        self.opt.lr = self.opt.base_lr * cosine_with_warmup(self.epoch, self.warm, self.total)
