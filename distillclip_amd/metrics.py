"""Validation retrieval metrics: what the reference's `log_acc` / `log_diag_score` compute from a
logits matrix (dual_distill_model.py:204-224, distil_model.py:171-191), from the embeddings, on the HIP kernel
`dclip_retrieval_metrics` — the [n, n] logits never reach HBM and torchmetrics is not needed."""
import ctypes

import torch

from ._lib import lib

K_LIST = (1, 3, 5, 10, 20, 50)          # reference dual_distill_model.py:87, distil_model.py:65


def retrieval_metrics(rows, cols, k_list=K_LIST, return_ranks=False):
    """rows, cols: [n, E] embeddings (any float dtype, un-normalised); logits = norm(rows) @ norm(cols).T as the reference's
    norm_and_logits (dual_distill_model.py:271-275).  -> {'acc_top{k}': .., 'softmax_mean_score': .., 'mean_score': ..}
    as 0-dim CUDA tensors (views of one result vector; no host sync)."""
    if not rows.is_cuda or not cols.is_cuda:
        raise RuntimeError('distillclip_amd has no CPU path: retrieval_metrics needs CUDA (HIP) tensors')
    if rows.dim() != 2 or rows.shape != cols.shape:
        raise ValueError(f'retrieval_metrics: need two [n, E] matrices, got {tuple(rows.shape)} and {tuple(cols.shape)}')
    rows = rows.detach().float().contiguous()
    cols = cols.detach().float().contiguous()
    n, E = rows.shape
    ks = [int(k) for k in k_list]
    out = torch.empty(len(ks) + 2, dtype=torch.float32, device=rows.device)
    ranks = torch.empty(n, dtype=torch.int32, device=rows.device) if return_ranks else None
    nbytes = lib().dclip_retrieval_metrics_workspace(n, E)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=rows.device)
    karr = (ctypes.c_int32 * max(len(ks), 1))(*ks)
    lib().dclip_retrieval_metrics(rows.data_ptr(), cols.data_ptr(), n, E, karr, len(ks), out.data_ptr(),
                                  ranks.data_ptr() if ranks is not None else None, ws.data_ptr(), ws.numel(),
                                  torch.cuda.current_stream().cuda_stream)
    res = {f'acc_top{k}': out[i] for i, k in enumerate(ks)}
    res['softmax_mean_score'] = out[len(ks)]
    res['mean_score'] = out[len(ks) + 1]
    if return_ranks:
        res['ranks'] = ranks
    return res


def gather_rows(x, group=None):
    """all ranks' rows in rank order (the reference's `self.all_gather(...)` + reshape(-1, E), dual_distill_model.py:141-160);
    the identity without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x
    x = x.contiguous()
    out = torch.empty((dist.get_world_size(group) * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out
