"""Functional fp32 restatement of the reference LossCalculator (TEST INFRASTRUCTURE — see __init__)."""
import torch
import torch.nn.functional as F

IMAGE_TEXT_LOSS = ['hard_label', 'soft_label', 'logits_mse', 'fine_grain', 'cos_diff']   # reference _loss.py:14


def out_l1(s, t):
    return (s - t).abs().mean()                                     # out_l1.py:9-10 (nn.L1Loss, mean)


def out_cos(s, t):
    # out_cos.py:10-11 = CosineEmbeddingLoss(target=+1): eps 1e-12 added to each squared norm
    dot = (s * t).sum(1)
    den = torch.sqrt(((s * s).sum(1) + 1e-12) * ((t * t).sum(1) + 1e-12))
    return (1 - dot / den).mean()


def _neg(x):
    n = x.shape[0]
    return x.flatten()[:-1].view(n - 1, n + 1)[:, 1:].flatten()     # clip_cos_diff.py:5-8


def cos_diff(s_logits, t_logits):
    # clip_cos_diff.py:16-23
    pos = F.relu(torch.diagonal(t_logits) - torch.diagonal(s_logits)).mean()
    neg = F.relu(_neg(s_logits) - _neg(t_logits)).mean()
    return neg + pos


def hard_label(s_logits):
    # hard_label.py:10-12
    return F.cross_entropy(s_logits, torch.arange(s_logits.shape[0]))


def kl_sum(s, t, tau):
    # soft_label.py:11-16 / out_kl.py:12-16: KLDiv(sum)(log softmax(s/tau), softmax(t/tau)) * tau^2
    lp_s = F.log_softmax(s / tau, dim=1)
    p_t = F.softmax(t / tau, dim=1)
    lp_t = F.log_softmax(t / tau, dim=1)
    return (p_t * (lp_t - lp_s)).sum() * tau ** 2


def mse(s, t):
    return ((s - t) ** 2).mean()                                    # logits_mse.py / embed_mse.py


def hidden_mse(s_list, t_list):
    return sum(mse(a, b) for a, b in zip(s_list, t_list)) / len(s_list)   # hidden_mse.py:9-17


def out_ce(s, t):
    return -(F.softmax(t, dim=1) * F.log_softmax(s, dim=1)).sum(1).mean()  # out_ce.py:9-13


LOSS_FUNCS = dict(out_l1=out_l1, out_cos=out_cos, cos_diff=cos_diff, hard_label=hard_label, kl_sum=kl_sum,
                  mse=mse, hidden_mse=hidden_mse, out_ce=out_ce)


class LossOracle:
    """reference _loss.py:17-216; the subset of terms in SURVEY.md §2.1 tiers ★1/★2 (+ out_ce)."""

    def __init__(self, loss_name, loss_scale=None, temperature=None, percent=None):
        self.loss_name = list(loss_name)
        loss_scale = loss_scale or {}
        self.loss_scale = {n: loss_scale.get(n, 1) for n in self.loss_name}          # :24-27
        if percent is None:
            percent = {n: 1 / len(self.loss_name) for n in self.loss_name}           # :29-31
        percent = dict(percent)
        default = (1 - sum(percent.values())) / len(percent)                         # :32
        if len(self.loss_name) != len(percent) and default <= 0:
            raise ValueError('negative default percent')                             # :33-38
        for n in self.loss_name:
            percent.setdefault(n, default)
        assert abs(sum(percent.values()) - 1) <= 1e-5                                # :42
        self.percent = percent
        self.temperature = temperature

    def one_tower(self, stu, tea):
        # :155-202
        res = {}
        for n in self.loss_name:
            s, t = stu['last_representation'], tea['last_representation']
            if n == 'out_l1':
                res[n] = out_l1(s, t)
            elif n == 'out_cos':
                res[n] = out_cos(s, t)
            elif n == 'out_kl':
                assert self.temperature
                res[n] = kl_sum(s, t, self.temperature)
            elif n == 'out_ce':
                res[n] = out_ce(s, t)
            elif n == 'embedding_mse':
                res[n] = mse(stu['embedding'], tea['embedding'])
            elif n == 'hidden_rep_mse':
                res[n] = hidden_mse(stu['representations'], tea['representations'])
        loss = 0
        for n, scale in self.loss_scale.items():
            if n in IMAGE_TEXT_LOSS:
                continue
            res[n] = res[n] * scale
            loss = loss + res[n] * self.percent[n]
        return loss, res

    def two_tower(self, stu, tea):
        # :118-153
        res = {}
        il, ires = self.one_tower(stu['visual_output'], tea['visual_output'])
        tl, tres = self.one_tower(stu['text_output'], tea['text_output'])
        res.update({'image_' + k: v for k, v in ires.items()})
        res.update({'text_' + k: v for k, v in tres.items()})
        si, st, ti, tt = stu['i2t_logits'], stu['t2i_logits'], tea['i2t_logits'], tea['t2i_logits']
        for n in self.loss_name:
            if n == 'hard_label':
                res[n] = 0.5 * (hard_label(si) + hard_label(st))
            elif n == 'soft_label':
                assert self.temperature
                res[n] = 0.5 * (kl_sum(si, ti, self.temperature) + kl_sum(st, tt, self.temperature))
            elif n == 'logits_mse':
                res[n] = 0.5 * (mse(si, ti) + mse(st, tt))
            elif n == 'cos_diff':
                res[n] = 0.5 * (cos_diff(si, ti) + cos_diff(st, tt))
        loss = 0.5 * (il + tl)
        for n, scale in self.loss_scale.items():
            if n in IMAGE_TEXT_LOSS:
                res[n] = res[n] * scale
                loss = loss + res[n] * self.percent[n]
        return loss, res

    def __call__(self, stu, tea, model_type):
        return self.two_tower(stu, tea) if model_type == 'all' else self.one_tower(stu, tea)
