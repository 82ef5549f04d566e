"""ORACLE (test infrastructure only — never imported by the product path): CPU restatement of the reference's validation
metrics.  Parity for this row is pinned by known-answer cases (tests/test_oracle_golden.py::test_metrics_known_answers):
the reference computes accuracy with torchmetrics, which is absent from this image, so no reference-run golden exists.

  norm_and_logits   reference model/dual_distill_model.py:271-275
  diag scores       reference model/dual_distill_model.py:204-212 (log_diag_score)
  top-k accuracy    reference model/dual_distill_model.py:220-224 (log_acc): torchmetrics multiclass accuracy(top_k=k),
                    micro average = fraction of rows whose label is among the k largest logits (torch.topk)
"""
import torch


def norm_and_logits(img, txt):
    img = img / img.norm(dim=1, keepdim=True)
    txt = txt / txt.norm(dim=1, keepdim=True)
    logits = img @ txt.t()
    return logits, logits.T


def diag_scores(logits):
    soft = torch.nn.functional.softmax(logits, dim=1)
    return torch.diagonal(soft).mean(), torch.diagonal(logits).mean()


def topk_accuracy(logits, k):
    n = logits.shape[0]
    label = torch.arange(n)
    top = torch.topk(logits, min(k, n), dim=1).indices
    return (top == label[:, None]).any(dim=1).float().mean()


def retrieval_metrics(img, txt, k_list=(1, 3, 5, 10, 20, 50)):
    logits, _ = norm_and_logits(img.double(), txt.double())
    out = {f'acc_top{k}': topk_accuracy(logits, k) for k in k_list}
    out['softmax_mean_score'], out['mean_score'] = diag_scores(logits)
    out['ranks'] = (logits > torch.diagonal(logits)[:, None]).sum(dim=1)
    return out
