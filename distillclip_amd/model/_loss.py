"""LossCalculator (reference model/_loss.py:17-216) on the fused HIP loss kernel.

Constructor arguments, percent / scale bookkeeping, error behaviour, `get_control_output()` and the returned
`(loss, dict)` (dict holds the *scaled* terms, tower terms prefixed `image_` / `text_`) follow the reference.
One `dclip_distill_loss` call computes every enabled term and d loss / d student embeddings; backward just scales them.
"""
from typing import Dict, List

import torch
from torch import nn

from .component.output import ControlOutput
from .. import ops

IMAGE_TEXT_LOSS = ['hard_label', 'soft_label', 'logits_mse', 'fine_grain', 'cos_diff']     # reference _loss.py:14
_TOWER_FUSED = ('out_l1', 'out_cos', 'out_kl', 'out_ce')
_CROSS_FUSED = ('cos_diff', 'hard_label', 'soft_label', 'logits_mse')
_FEATURE = ('hidden_rep_mse', 'embedding_mse')
_KNOWN_UNSUPPORTED = ('attention_score_mse', 'attention_probs_mse', 'attention_probs_kl', 'last_value_map_kl', 'vit_kd',
                      'fine_grain', 'smd')
_SLOT_TOWER = {'out_l1': 1, 'out_cos': 2, 'out_kl': 3, 'out_ce': 4}
_SLOT_CROSS = {'cos_diff': 9, 'hard_label': 10, 'soft_label': 11, 'logits_mse': 12}


class _FusedLossFn(torch.autograd.Function):
    # (autocast-safe: inputs arrive as fp32 whatever Lightning's `precision: 16` autocast did to the caller, autocast is off inside)
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, s_img, s_txt, t_img, t_txt, weights, temperature, global_negatives=False):
        two = s_txt is not None
        if global_negatives and two:
            # opt-in north-star mode: in-batch negatives over the GLOBAL batch (all ranks), one fused all-gather over RCCL
            from ..parallel import gather_embeddings, check_equal_batch, world_size
            B = s_img.shape[0]
            # rows are addressed as rank * B and the fused kernel holds at most 4096 gathered rows: fail on EVERY rank, before
            # the first collective, rather than desynchronise it (ragged last batch) or fail on one rank (kernel limit)
            check_equal_batch(B, s_img.device)
            if world_size() * B > 4096:
                raise ValueError(f'global negatives: world * batch = {world_size() * B} exceeds the fused loss kernel\'s 4096 '
                                 f'gathered rows (include/dclip.h: dclip_distill_loss_rows)')
            (gsi, gti, gst, gtt), rank, world = gather_embeddings([s_img, t_img, s_txt, t_txt])
            import torch.distributed as dist
            gstats = None
            if weights.get('hard_label', 0) or weights.get('soft_label', 0):
                # the softmax statistics of every row (both directions) are needed: own rows first, then one small all-gather
                st = ops.distill_loss(gsi, gti, gst, gtt, weights=weights, temperature=temperature, row0=rank * B, rows=B,
                                      stats_only=True)
                if world > 1:
                    allst = torch.empty((world, 6, B), dtype=torch.float32, device=st.device)
                    from ..parallel import all_gather_flat
                    all_gather_flat(allst.view(-1), st.contiguous().view(-1))
                    gstats = allst.permute(1, 0, 2).reshape(6, world * B).contiguous()
                else:
                    gstats = st
            # own row block [B, world * B] only (1 / world of the logits work); the 16 scalars are shares that add up
            scal, d_i, d_t = ops.distill_loss(gsi, gti, gst, gtt, weights=weights, temperature=temperature,
                                              row0=rank * B, rows=B, gathered_stats=gstats)
            if world > 1:
                dist.all_reduce(scal, op=dist.ReduceOp.SUM)
            d_i = d_i * float(world)
            d_t = d_t * float(world)
        else:
            scal, d_i, d_t = ops.distill_loss(s_img.detach().float().contiguous(), t_img.detach().float().contiguous(),
                                              s_txt.detach().float().contiguous() if two else None,
                                              t_txt.detach().float().contiguous() if two else None,
                                              weights=weights, temperature=temperature)
        ctx.save_for_backward(d_i, d_t if two else d_i)
        ctx.two = two
        ctx.mark_non_differentiable(scal)
        return scal[0].clone(), scal

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g_loss, _g_scal):
        d_i, d_t = ctx.saved_tensors
        return d_i * g_loss, (d_t * g_loss) if ctx.two else None, None, None, None, None, None


class _FeatureMSEFn(torch.autograd.Function):
    """mean((s - t)^2) with the gradient 2 (s - t) / n, both by dclip_feature_mse (hidden_mse.py / embed_mse.py)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, s, t):
        s = s.contiguous().float()
        t = t.detach().contiguous().float()
        if s.shape != t.shape:
            raise RuntimeError(f'The size of tensor a {tuple(s.shape)} must match the size of tensor b {tuple(t.shape)}')
        val, ds = ops.feature_mse(s.detach(), t)
        ctx.save_for_backward(ds)
        return val

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g):
        (ds,) = ctx.saved_tensors
        return ds * g, None


class LossCalculator(nn.Module):
    def __init__(self, loss_name: List, loss_scale: dict = None, temperature=None, percent=None, smd_tau: float = 0.04,
                 vit_kd_para: Dict = None):
        super().__init__()
        self.loss_name = loss_name
        self.loss_scale = {}
        if loss_scale is None:
            loss_scale = {n: 1 for n in self.loss_name}
        for n in loss_name:
            self.loss_scale[n] = loss_scale.get(n, 1)                                   # reference :24-27
        if percent is None:
            percent = {n: 1 / len(loss_name) for n in self.loss_name}                   # :29-31
        self.percent = percent
        default_value = (1 - sum(self.percent.values())) / len(self.percent)
        if len(loss_name) != len(self.percent.keys()) and default_value <= 0:
            raise ValueError(f'there are some loss default percent is negative. Please check the sum of the percent {percent}'
                             f'the default_value is {default_value} = (1 - sum(percent.values())) / len(percent)')
        for n in loss_name:
            if n not in self.percent:
                self.percent[n] = default_value
        assert abs(sum(self.percent.values()) - 1) <= 1e-5                              # :42
        self.temperature = temperature
        self.smd_tau = smd_tau
        self.vit_kd_para = vit_kd_para
        # extra knob with a reference-preserving default: False = per-rank (local) negatives like the reference's training_step
        self.global_negatives = False
        for n in loss_name:                                                             # :57-98
            if n in _KNOWN_UNSUPPORTED:
                raise NotImplementedError(f"loss '{n}' is registered by the reference but used by no shipped config; it is "
                                          f'outside the HIP hot path (SURVEY.md §2.1)')
            if n not in _TOWER_FUSED and n not in _CROSS_FUSED and n not in _FEATURE:
                raise ValueError('Invalid Loss Type!')

    def get_control_output(self):
        # reference :100-116
        return ControlOutput(need_emb='embedding_mse' in self.loss_name, need_rep='hidden_rep_mse' in self.loss_name)

    def _feature_terms(self, stu, tea):
        """-> (weighted sum, {name: scaled value}) of the hidden-state / embedding MSE terms of one tower"""
        total, res = 0, {}
        for n in self.loss_name:
            if n == 'hidden_rep_mse':
                # reference hidden_mse.py:9-17: pairs by zip(), divided by the number of STUDENT hidden states
                val = sum(_FeatureMSEFn.apply(s, t) for s, t in zip(stu.representations, tea.representations))
                val = val / max(len(stu.representations), 1)
            elif n == 'embedding_mse':
                val = _FeatureMSEFn.apply(stu.embedding, tea.embedding)
            else:
                continue
            res[n] = val * self.loss_scale[n]
            total = total + res[n] * self.percent[n]
        return total, res

    def set_percent(self, new_percent):
        self.percent = new_percent

    def set_scale(self, new_scale):
        self.loss_scale = new_scale

    def _weights(self, two_tower):
        w = {}
        for n in self.loss_name:
            if n in _TOWER_FUSED or (two_tower and n in _CROSS_FUSED):
                w[n] = self.loss_scale[n] * self.percent[n]
        if 'out_kl' in w or 'soft_label' in w:
            assert self.temperature, 'You should give the temperature for the kl loss'     # reference :133,:166
        return w

    def _scale_vector(self, device):
        """loss_scale of each raw term in the kernel's 16-slot output order (cached on the device; rebuilt when the scales change)"""
        key = (str(device), tuple(sorted(self.loss_scale.items())))
        if getattr(self, '_scale_cache', (None,))[0] != key:
            v = [1.0] * 16
            for n, s in self.loss_scale.items():
                if n in _SLOT_TOWER:
                    v[_SLOT_TOWER[n]] = v[_SLOT_TOWER[n] + 4] = float(s)
                elif n in _SLOT_CROSS:
                    v[_SLOT_CROSS[n]] = float(s)
            self._scale_cache = (key, torch.tensor(v, dtype=torch.float32, device=device))
        return self._scale_cache[1]

    def cal_tow_tower_loss(self, stu_out, tea_out):
        loss, scal = _FusedLossFn.apply(stu_out.visual_output.last_representation, stu_out.text_output.last_representation,
                                        tea_out.visual_output.last_representation, tea_out.text_output.last_representation,
                                        self._weights(True), self.temperature, self.global_negatives)
        res = {}
        scaled = scal * self._scale_vector(scal.device)          # ONE launch for every logged term (they sit between the
        for prefix, off, so, to in (('image_', 0, stu_out.visual_output, tea_out.visual_output),   # forward and the backward)
                                    ('text_', 4, stu_out.text_output, tea_out.text_output)):
            for n in self.loss_name:
                if n in _TOWER_FUSED:
                    res[prefix + n] = scaled[_SLOT_TOWER[n] + off]
            ft, fres = self._feature_terms(so, to)
            if torch.is_tensor(ft):
                loss = loss + 0.5 * ft                                   # reference :148: 0.5 * (image_loss + text_loss)
            res.update({prefix + k: v for k, v in fres.items()})
        for n in self.loss_name:
            if n in _CROSS_FUSED:
                res[n] = scaled[_SLOT_CROSS[n]]
        return loss, res

    def cal_one_tower_loss(self, stu_out, tea_out):
        loss, scal = _FusedLossFn.apply(stu_out.last_representation, None, tea_out.last_representation, None,
                                        self._weights(False), self.temperature)
        scaled = scal * self._scale_vector(scal.device)
        res = {n: scaled[_SLOT_TOWER[n]] for n in self.loss_name if n in _TOWER_FUSED}
        ft, fres = self._feature_terms(stu_out, tea_out)
        res.update(fres)
        return (loss + ft) if torch.is_tensor(ft) else loss, res

    def forward(self, stu_out, tea_out, model_type: str):
        if model_type == 'all':
            return self.cal_tow_tower_loss(stu_out, tea_out)
        return self.cal_one_tower_loss(stu_out, tea_out)
