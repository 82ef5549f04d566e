"""nn.Linear on token rows through the HIP GEMMs: the `embedding_projection` / `hidden_projection` of a plain CLIP encoder in the
student role (reference model/component/image_encoder.py:23-25,54-59, text_encoder.py:45-47,75-80), which lift the student's hidden
states / token embedding to the teacher's width before the feature-MSE terms compare them (_loss.py:100-116).

Forward  y = x W^T + b   : dclip_gemm_nt on bf16 copies of x and W, f32 accumulation and output.
Backward dx = dy W       : dclip_gemm_nt ; dW += dy^T x : dclip_gemm_tn_acc ; db += column sums of dy : dclip_colsum_acc.
"""
import torch
from torch import nn

from ... import ops
from ..._lib import lib


def _bf16(t):
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    lib().dclip_cast_bf16(t.data_ptr(), out.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
    return out


class _LinearFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias):
        if not x.is_cuda:
            raise RuntimeError('distillclip_amd projections run on MI355X only (no CPU fallback)')
        rows = x.reshape(-1, x.shape[-1]).contiguous()
        xb, wb = _bf16(rows), _bf16(weight.contiguous())
        y = ops.gemm_nt(xb, wb, bias=bias, out_dtype=torch.float32)
        ctx.save_for_backward(xb, weight)
        ctx.has_bias = bias is not None
        ctx.shape = x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, dy):
        xb, weight = ctx.saved_tensors
        dyb = _bf16(dy.reshape(-1, dy.shape[-1]).contiguous().float())
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = _bf16(weight.t().contiguous())                      # [in, out]: dx[M, in] = dy[M, out] @ wt[in, out]^T
            dx = ops.gemm_nt(dyb, wt, out_dtype=torch.float32).view(ctx.shape)
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_tn_acc(dyb, xb, torch.zeros_like(weight), splits=max(1, min(16, xb.shape[0] // 2048)))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum_acc(dyb, torch.zeros(weight.shape[0], dtype=torch.float32, device=weight.device))
        return dx, dw, db


class HipLinear(nn.Linear):
    """nn.Linear's parameters, initialisation and state_dict keys; the product runs on the HIP GEMMs (in / out features: multiples of 64 / 8)"""

    def forward(self, x):
        return _LinearFn.apply(x, self.weight, self.bias)
