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
            if tw.flat_grad is not None:
                tw.flat_grad.zero_()

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        st = torch.cuda.current_stream().cuda_stream
        for tw in self.towers:
            if tw.flat is None:
                continue
            key = id(tw)
            if key not in self._state or self._state[key][0].numel() != tw.flat.numel():
                self._state[key] = (torch.zeros_like(tw.flat), torch.zeros_like(tw.flat))
            m, v = self._state[key]
            for b, e in self._ranges(tw):
                lib().dclip_adamw(tw.flat.data_ptr() + b * 4, tw.flat_grad.data_ptr() + b * 4, m.data_ptr() + b * 4,
                                  v.data_ptr() + b * 4, e - b, self.lr, self.betas[0], self.betas[1], self.eps,
                                  self.weight_decay, self.step_count, st)
            tw.wcache_dirty = True


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
