"""Fused HIP loss (fwd + bwd) vs the oracle's LossCalculator restatement and the reference goldens (tests/golden/loss.npz)."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

NAMES = ['out_l1', 'out_cos', 'out_kl', 'out_ce', 'cos_diff', 'hard_label', 'soft_label', 'logits_mse']
SLOT_IMG = {'out_l1': 1, 'out_cos': 2, 'out_kl': 3, 'out_ce': 4}
SLOT_X = {'cos_diff': 9, 'hard_label': 10, 'soft_label': 11, 'logits_mse': 12}


def _run_oracle(e, names, scale, tau, two=True):
    si, st = e['si'].clone().requires_grad_(True), e.get('st', e['si']).clone().requires_grad_(True)
    lc = oracle.LossOracle(names, scale, temperature=tau)
    if two:
        stu = oracle.clip_forward({'last_representation': si}, {'last_representation': st})
        tea = oracle.clip_forward({'last_representation': e['ti']}, {'last_representation': e['tt']})
        loss, res = lc(stu, tea, 'all')
    else:
        loss, res = lc({'last_representation': si}, {'last_representation': e['ti']}, 'image')
    loss.backward()
    return loss.detach(), res, si.grad, st.grad, lc


def _run_hip(e, lc, tau, two=True):
    from distillclip_amd import ops
    w = {n: lc.loss_scale[n] * lc.percent[n] for n in lc.loss_name}
    g = {k: v.cuda() for k, v in e.items()}
    if two:
        return ops.distill_loss(g['si'], g['ti'], g['st'], g['tt'], weights=w, temperature=tau)
    return ops.distill_loss(g['si'], g['ti'], weights=w, temperature=tau)


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize('B,E,names,tau', [
    (8, 64, NAMES, 0.5), (37, 512, NAMES, 2.0), (64, 512, ['out_l1', 'out_cos', 'cos_diff'], None),
    (300, 512, NAMES, 1.0), (512, 512, ['out_l1', 'out_cos', 'cos_diff', 'hard_label', 'soft_label'], 0.7), (1, 64, ['out_l1', 'out_cos'], None)])
def test_two_tower_vs_oracle(B, E, names, tau):
    g = torch.Generator().manual_seed(B * 7 + E)
    e = {k: torch.randn(B, E, generator=g) * (1 + i) for i, k in enumerate(('si', 'st', 'ti', 'tt'))}
    e['ti'] = 0.7 * e['ti'] + 0.5 * e['si']          # correlated teacher so cos_diff has both signs
    scale = {'cos_diff': 0.1, 'hard_label': 2.0}
    loss, res, gi, gt, lc = _run_oracle(e, names, scale, tau)
    out, di, dt = _run_hip(e, lc, tau)
    out = out.cpu()
    assert abs(out[0].item() - loss.item()) <= 2e-5 * max(1.0, abs(loss.item())), (out[0].item(), loss.item())
    for n in names:
        if n in SLOT_X:
            got, ref = out[SLOT_X[n]].item() * lc.loss_scale[n], res[n].item()
            assert abs(got - ref) <= 3e-5 * max(1.0, abs(ref)), (n, got, ref)
        else:
            for tow, off in (('image_', 0), ('text_', 4)):
                got, ref = out[SLOT_IMG[n] + off].item() * lc.loss_scale[n], res[tow + n].item()
                assert abs(got - ref) <= 3e-5 * max(1.0, abs(ref)), (tow + n, got, ref)
    assert _rel(di.cpu(), gi) < 2e-4, _rel(di.cpu(), gi)
    assert _rel(dt.cpu(), gt) < 2e-4, _rel(dt.cpu(), gt)


def test_one_tower_vs_oracle():
    g = torch.Generator().manual_seed(5)
    e = {k: torch.randn(100, 512, generator=g) for k in ('si', 'ti')}
    loss, res, gi, _, lc = _run_oracle(e, ['out_l1', 'out_cos', 'out_kl'], None, 3.0, two=False)
    out, di, _ = _run_hip(e, lc, 3.0, two=False)
    assert abs(out[0].item() - loss.item()) <= 2e-5 * max(1.0, abs(loss.item()))
    assert _rel(di.cpu(), gi) < 2e-4


@pytest.mark.parametrize('case', ['b8', 'b37', 'small'])
def test_against_reference_golden(golden_dir, case):
    """loss.npz was produced by the reference's LossCalculator itself (tools/golden/gen_golden.py:loss_only)."""
    gd = dict(np.load(os.path.join(golden_dir, 'loss.npz')))
    e = {k: torch.from_numpy(gd[f'{case}.{k}']) for k in ('si', 'st', 'ti', 'tt')}
    lc = oracle.LossOracle(NAMES, {'cos_diff': 0.1, 'hard_label': 2.0}, temperature=0.5)
    out, di, dt = _run_hip(e, lc, 0.5)
    ref = float(gd[f'{case}.loss'])
    assert abs(out[0].item() - ref) <= 5e-5 * max(1.0, abs(ref)), (out[0].item(), ref)
    for k, r in (('si', di), ('st', dt)):
        want = torch.from_numpy(gd[f'{case}.grad.{k}'])
        assert _rel(r.cpu(), want) < 5e-4, (k, _rel(r.cpu(), want))


def test_bad_arguments():
    from distillclip_amd import ops
    x = torch.randn(4, 40, device='cuda')          # E % 16 != 0
    with pytest.raises(ValueError):
        ops.distill_loss(x, x, weights={'out_l1': 1.0})
    with pytest.raises(RuntimeError):
        ops.distill_loss(x.cpu(), x.cpu(), weights={'out_l1': 1.0})


@pytest.mark.parametrize('B,world,names', [(24, 3, ['out_l1', 'out_cos', 'cos_diff']), (64, 4, ['out_cos', 'out_kl', 'cos_diff', 'logits_mse']),
                                           (512, 8, ['out_l1', 'out_cos', 'cos_diff']),
                                           (4096, 8, ['out_l1', 'out_cos', 'cos_diff'])])      # 8 ranks x 512: the N = 8 bench shape
def test_row_blocks_add_up_to_the_whole_batch(B, world, names):
    """dclip_distill_loss_rows (data-parallel global negatives): every rank's row block [B / world, B] — the scalar shares
    sum to the whole-batch scalars and the gradient rows equal the whole-batch gradient rows (oracle = single process on the
    concatenated batch)"""
    from distillclip_amd import ops
    E = 512 if B >= 512 else 64
    g = torch.Generator().manual_seed(B + world)
    e = {k: (torch.randn(B, E, generator=g) * (1 + i)).cuda() for i, k in enumerate(('si', 'st', 'ti', 'tt'))}
    e['ti'] = 0.7 * e['ti'] + 0.5 * e['si']
    scale = {'cos_diff': 0.1}
    lc = oracle.LossOracle(names, scale, temperature=2.0)
    w = {n: lc.loss_scale[n] * lc.percent[n] for n in lc.loss_name}
    full, di, dt = ops.distill_loss(e['si'], e['ti'], e['st'], e['tt'], weights=w, temperature=2.0)
    per = B // world
    tot = torch.zeros(16, device='cuda')
    for r in range(world):
        sc, gi, gt = ops.distill_loss(e['si'], e['ti'], e['st'], e['tt'], weights=w, temperature=2.0, row0=r * per, rows=per)
        tot += sc
        assert _rel(gi, di[r * per:(r + 1) * per]) < 2e-5 and _rel(gt, dt[r * per:(r + 1) * per]) < 2e-5, r
    assert torch.allclose(tot, full, rtol=2e-5, atol=1e-6), (tot, full)
    # and against the oracle on the concatenated batch
    loss, res, ogi, ogt, _ = _run_oracle({k: v.cpu() for k, v in e.items()}, names, scale, 2.0)
    assert abs(tot[0].item() - loss.item()) <= 2e-5 * max(1.0, abs(loss.item()))
    assert _rel(di.cpu(), ogi) < 2e-4 and _rel(dt.cpu(), ogt) < 2e-4


@pytest.mark.parametrize('B,world', [(48, 3), (512, 4)])
def test_row_blocks_with_gathered_statistics(B, world):
    """hard_label / soft_label over global negatives: statistics pass per block -> gather -> gradient pass per block; shares and
    gradient rows against the whole-batch call and the oracle (every cross term enabled)"""
    from distillclip_amd import ops
    E = 64 if B < 512 else 512
    g = torch.Generator().manual_seed(B * 3 + world)
    e = {k: (torch.randn(B, E, generator=g) * (1 + i)).cuda() for i, k in enumerate(('si', 'st', 'ti', 'tt'))}
    e['ti'] = 0.7 * e['ti'] + 0.5 * e['si']
    scale = {'cos_diff': 0.1, 'hard_label': 2.0}
    lc = oracle.LossOracle(NAMES, scale, temperature=0.7)
    w = {n: lc.loss_scale[n] * lc.percent[n] for n in lc.loss_name}
    full, di, dt = ops.distill_loss(e['si'], e['ti'], e['st'], e['tt'], weights=w, temperature=0.7)
    per = B // world
    blocks = [ops.distill_loss(e['si'], e['ti'], e['st'], e['tt'], weights=w, temperature=0.7, row0=r * per, rows=per, stats_only=True)
              for r in range(world)]
    gstats = torch.stack(blocks).permute(1, 0, 2).reshape(6, B).contiguous()          # what the all-gather builds
    tot = torch.zeros(16, device='cuda')
    for r in range(world):
        sc, gi, gt = ops.distill_loss(e['si'], e['ti'], e['st'], e['tt'], weights=w, temperature=0.7, row0=r * per, rows=per,
                                      gathered_stats=gstats)
        tot += sc
        assert _rel(gi, di[r * per:(r + 1) * per]) < 5e-5 and _rel(gt, dt[r * per:(r + 1) * per]) < 5e-5, r
    assert torch.allclose(tot, full, rtol=5e-5, atol=1e-6), (tot, full)
    loss, res, ogi, ogt, _ = _run_oracle({k: v.cpu() for k, v in e.items()}, NAMES, scale, 0.7)
    assert abs(tot[0].item() - loss.item()) <= 5e-5 * max(1.0, abs(loss.item()))
    assert _rel(di.cpu(), ogi) < 2e-4 and _rel(dt.cpu(), ogt) < 2e-4


def test_row_block_rejects_terms_that_need_every_row():
    from distillclip_amd import ops
    x = torch.randn(32, 64, device='cuda')
    with pytest.raises(ValueError):
        ops.distill_loss(x, x, x, x, weights={'hard_label': 1.0}, row0=0, rows=16)        # neither statistics in nor out
    with pytest.raises(ValueError):
        ops.distill_loss(x, x, x, x, weights={'cos_diff': 1.0}, row0=24, rows=16)
