"""Functional fp32 restatement of the reference encoders (TEST INFRASTRUCTURE — see __init__).

All functions take a plain dict `sd` of tensors keyed by the reference's state_dict names and return a dict
with the fields of the reference's output dataclasses (model/component/output.py:16-35).
`cap` (optional dict) receives intermediates for kernel-level parity tests.
"""
import contextlib
import math
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------------------------------
# Rounding-matched mode (test infrastructure for tight END-TO-END gradient parity).  The HIP path stores GEMM operands as
# bf16; against the plain fp32 oracle that costs ~1e-2 per block execution on gradients, which hides a
# backward error of a few per cent.  Inside `with bf16_matched():` this restatement rounds to bf16 at the SAME points —
# forward: weights, LayerNorm outputs, qkv, probabilities, mixed probabilities, context, gelu output, picked final row, im2row patches,
# and the frozen teacher's residual stream as fp16 (the type the reference's `precision: 16` autocast keeps it in; the HIP path stores it
# that way: DESIGN.md section 3); backward: the gradient of every one of those tensors plus the residual-stream gradient where it enters a
# GEMM (fc2 / proj / head / embedding outputs), the pre-mix scores, and gelu'(z) saved as 8-bit fixed point — while accumulation stays
# fp32 on both sides.  What remains is accumulation order and rare rounding flips: ~1e-3.  The arithmetic between the
# rounding points is unchanged, and tests/test_oracle_golden.py holds this mode to the pinned fp32 oracle within bf16 noise.
# ---------------------------------------------------------------------------------------------------------------------
_MATCHED = False


@contextlib.contextmanager
def bf16_matched(on=True):
    global _MATCHED
    old, _MATCHED = _MATCHED, bool(on)
    try:
        yield
    finally:
        _MATCHED = old


def _rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return _rb(x) if fwd else x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (_rb(g) if ctx.bwd else g), None, None


def Q(x):        # stored as bf16, and its gradient is stored as bf16 too
    return _Round.apply(x, True, True) if _MATCHED else x


def Qf(x):       # stored as bf16; gradient stays fp32 (weights: f32 wgrad accumulation)
    return _Round.apply(x, True, False) if _MATCHED else x


def Qb(x):       # fp32 value whose GRADIENT enters a GEMM as a bf16 operand
    return _Round.apply(x, False, True) if _MATCHED else x


def Qh(x):       # the frozen teacher's residual stream: stored as fp16 (inference only, no gradient)
    return x.to(torch.float16).to(torch.float32) if _MATCHED else x


# 8-bit fixed-point code of the saved gelu' (include/dclip.h, DCLIP_ACT_GELU_SAVE): q = rint((g' + 0.13) * 255 / 1.26)
_DG_LO, _DG_STEP = -0.13, 1.26 / 255.0


def dg_quantise(dg):
    q = torch.clamp(torch.round(dg * (255.0 / 1.26) + 0.13 * (255.0 / 1.26)), 0.0, 255.0)
    return q * _DG_STEP + _DG_LO


class _GeluSave(torch.autograd.Function):
    """fc1 epilogue of the training towers (include/dclip.h DCLIP_ACT_GELU_SAVE / DCLIP_ACT_MULAUX): u = bf16(gelu(z)),
    gelu'(z) saved as 8-bit fixed point, dz = bf16(du * gelu'(z))."""

    @staticmethod
    def forward(ctx, z):
        cdf = 0.5 * (1.0 + torch.erf(z * 0.7071067811865476))
        dg = cdf + z * torch.exp(-0.5 * z * z) * 0.3989422804014327
        ctx.save_for_backward(dg_quantise(dg))
        return _rb(z * cdf)

    @staticmethod
    def backward(ctx, g):
        (dg,) = ctx.saved_tensors
        return _rb(g * dg)


def _gelu(z):
    return _GeluSave.apply(z) if _MATCHED else F.gelu(z)


class _QuickGeluSave(torch.autograd.Function):
    """c_fc epilogue of a CLIP tower that trains (include/dclip.h DCLIP_ACT_QUICKGELU_SAVE / DCLIP_ACT_MULAUX): u = bf16(z s(1.702 z)),
    the derivative s + 1.702 z s (1 - s) saved as 8-bit fixed point, dz = bf16(du * derivative)."""

    @staticmethod
    def forward(ctx, z):
        sg = torch.sigmoid(1.702 * z)
        ctx.save_for_backward(dg_quantise(sg + 1.702 * z * sg * (1.0 - sg)))
        return _rb(z * sg)

    @staticmethod
    def backward(ctx, g):
        (dg,) = ctx.saved_tensors
        return _rb(g * dg)


def _lin(x, w, b=None, grad_operand=False):
    """F.linear with the bf16 weight cache; grad_operand: the output's gradient is a bf16 GEMM operand (bias added outside:
    bias gradients are column sums of the fp32 residual gradient)"""
    if not _MATCHED:
        return F.linear(x, w, b)
    y = F.linear(x, Qf(w))
    if grad_operand:
        y = Qb(y)
    return y if b is None else y + b


def _ln(x, sd, prefix, eps=1e-5):
    # reference _common.py:14-20 (fp32 LayerNorm, eps 1e-5) / nn.LayerNorm in weight_share_model.py:239
    return F.layer_norm(x.float(), (x.shape[-1],), sd[prefix + '.weight'], sd[prefix + '.bias'], eps)


def quick_gelu(x):
    # reference _common.py:23-25
    return x * torch.sigmoid(1.702 * x)


def _teacher_attention(h, sd, p, heads, mask, cap, tag, train=False):
    # reference _common.py:51-95
    B, N, D = h.shape
    hd = D // heads
    qkv = Q(_lin(h, sd[p + 'in_proj_weight'], sd[p + 'in_proj_bias']))
    q, k, v = qkv.chunk(3, dim=-1)
    sp = lambda t: t.view(B, N, heads, hd).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    scores = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if mask is not None:
        scores = scores + mask
    if _MATCHED and train:
        # a tower that trains runs the unfused kernels (scores f32 in memory, probabilities saved as bf16; the gradient of the scores
        # enters the dQ / dK products as a bf16 operand), and the gradient of the out-projection's output is a bf16 GEMM operand
        ctx = Q(Qb(scores).softmax(dim=-1)) @ v
    elif _MATCHED:
        # the HIP kernel (attn_fused_fwd, round 5) keeps the probabilities UNNORMALISED in bf16 — e = exp(s - max), the row maximum exactly 1 —
        # and applies 1 / sum(e) (f32, over the unrounded e) to the value product: same arithmetic, the rounding point moved
        e = (scores - scores.amax(dim=-1, keepdim=True)).exp()
        ctx = (Qf(e) @ v) / e.sum(dim=-1, keepdim=True)
    else:
        ctx = scores.softmax(dim=-1) @ v
    ctx = Q(ctx.permute(0, 2, 1, 3).reshape(B, N, D))
    if cap is not None:
        cap[tag + '.ctx'] = ctx
    return _lin(ctx, sd[p + 'out_proj.weight'], sd[p + 'out_proj.bias'], grad_operand=train)


def _teacher_blocks(x, sd, prefix, layers, heads, mask, cap, need_layers=None, need_rep=False, train=False):
    # reference _common.py:116-127 (block), :143-167 (stack).  train: the tower is a student (f32 residual stream, saved activations)
    reps = []
    Qs = (lambda t: t) if train else Qh
    for i in range(layers):
        p = f'{prefix}transformer.resblocks.{i}.'
        x = Qs(x + _teacher_attention(Q(_ln(x, sd, p + 'ln_1')), sd, p + 'attn.', heads, mask, cap, f'tblock{i}', train))
        h = Q(_ln(x, sd, p + 'ln_2'))
        z = _lin(h, sd[p + 'mlp.c_fc.weight'], sd[p + 'mlp.c_fc.bias'])
        u = _QuickGeluSave.apply(z) if (_MATCHED and train) else Q(quick_gelu(z))
        x = Qs(x + _lin(u, sd[p + 'mlp.c_proj.weight'], sd[p + 'mlp.c_proj.bias'], grad_operand=train))
        if cap is not None:
            cap[f'tblock{i}.out'] = x
        if need_rep and (need_layers is None or i in need_layers):
            reps.append(x)
    return x, reps


def teacher_image_forward(sd, image, heads=None, need_layers=None, need_rep=False, need_emb=False, cap=None, train=False):
    """reference _common.py:188-221 (VisionTransformer.forward) behind image_encoder.py:50-65.  train: the encoder is the STUDENT
    (image_encoder.py:16-25): same arithmetic with gradients; see clip_student_image_forward for its projection linears."""
    w = sd['visual.conv1.weight']
    width, patch = w.shape[0], w.shape[-1]
    layers = 1 + max(int(k.split('.')[3]) for k in sd if k.startswith('visual.transformer.resblocks.'))
    heads = heads or width // 64            # reference utils.py:126 (heads = width*32//64 // 32 ... = width//64)
    Qs = (lambda t: t) if train else Qh
    x = F.conv2d(Qf(image), Qf(w), stride=patch)                    # :196
    if train:
        x = Qb(x)                                                   # (its gradient is the bf16 operand of the conv1 weight gradient)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)      # :197-198
    cls = sd['visual.class_embedding'] + torch.zeros(x.shape[0], 1, width)
    x = Qs(torch.cat([cls, x], dim=1) + sd['visual.positional_embedding'])   # :199-202
    emb = x if need_emb else None
    x = Qs(_ln(x, sd, 'visual.ln_pre'))                              # :208
    x, reps = _teacher_blocks(x, sd, 'visual.', layers, heads, None, cap, need_layers, need_rep, train)
    x = Q(_ln(x, sd, 'visual.ln_post')) @ Qf(sd['visual.proj'])      # :210-213
    if train:
        x = Qb(x)
    return dict(last_representation=x[:, 0, :], last_layer_output=x, representations=reps, embedding=emb)


def causal_mask(n):
    # reference text_encoder.py:54-60
    return torch.full((n, n), float('-inf')).triu_(1)


def teacher_text_forward(sd, text, heads=None, need_layers=None, need_rep=False, need_emb=False, cap=None, train=False):
    """reference text_encoder.py:62-92 (TextEncoder.encode_text).  train: the encoder is the STUDENT (text_encoder.py:41-47)."""
    width = sd['positional_embedding'].shape[1]
    layers = 1 + max(int(k.split('.')[2]) for k in sd if k.startswith('transformer.resblocks.'))
    heads = heads or width // 64            # reference utils.py:94
    x = sd['token_embedding.weight'][text] + sd['positional_embedding']        # :65-66
    x = x if train else Qh(x)
    emb = x if need_emb else None
    x, reps = _teacher_blocks(x, sd, '', layers, heads, causal_mask(text.shape[1]), cap, need_layers, need_rep, train)
    x = Q(_ln(x, sd, 'ln_final')) @ Qf(sd['text_projection'])               # :69,72
    if train:
        x = Qb(x)
    pick = x[torch.arange(x.shape[0]), text.argmax(dim=-1)]                 # :86
    return dict(last_representation=pick, last_layer_output=x, representations=reps, embedding=emb)


def _project(t, sd, name):
    """embedding_projection / hidden_projection of a CLIP encoder in the student role (nn.Linear on bf16 operands on the HIP path)"""
    if not _MATCHED:
        return F.linear(t, sd[name + '.weight'], sd[name + '.bias'])
    return Qb(F.linear(Qf(t), Qf(sd[name + '.weight']))) + sd[name + '.bias']


def _clip_student(out, sd, no_trans, need_rep, need_emb):
    # reference image_encoder.py:54-59 / text_encoder.py:75-80
    if not no_trans:
        if need_rep:
            out['representations'] = [_project(r, sd, 'hidden_projection') for r in out['representations']]
        if need_emb:
            out['embedding'] = _project(out['embedding'], sd, 'embedding_projection')
    return out


def clip_student_image_forward(sd, image, heads=None, need_rep=False, need_emb=False, no_trans=False, cap=None):
    """reference image_encoder.py:50-65 with is_student=True: VisionTransformer + the two projection linears"""
    return _clip_student(teacher_image_forward(sd, image, heads, None, need_rep, need_emb, cap, train=True), sd, no_trans, need_rep, need_emb)


def clip_student_text_forward(sd, text, heads=None, need_rep=False, need_emb=False, no_trans=False, cap=None):
    """reference text_encoder.py:62-92 with is_student=True"""
    return _clip_student(teacher_text_forward(sd, text, heads, None, need_rep, need_emb, cap, train=True), sd, no_trans, need_rep, need_emb)


def _mini_attention(h, sd, p, r, heads, use_transform, cap, tag):
    # reference weight_share_model.py:88-140 (MiniAttention.forward), rpe disabled (rpe_config: null)
    B, N, C = h.shape
    hd = C // heads
    qkv = Q(_lin(h, sd[p + 'qkv.weight'], sd.get(p + 'qkv.bias')))
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]                   # :101 (q *= scale)
    attn = Qb(q @ k.transpose(-2, -1))                              # :103
    if cap is not None:
        cap[tag + '.scores'] = attn
    if use_transform:
        wl = sd[p + f'conv_l.instances.{r}.weight'].reshape(heads, heads)
        attn = torch.einsum('gh,bhij->bgij', wl, attn)             # :114-115 (1x1 conv over the head channel)
    attn = attn.softmax(dim=-1)                                     # :117 (the mix below reads the f32 probabilities on the HIP
    if cap is not None:                                             #       path too; only the saved copy for the backward is bf16)
        cap[tag + '.probs'] = attn
    if use_transform:
        ww = sd[p + f'conv_w.instances.{r}.weight'].reshape(heads, heads)
        attn = torch.einsum('gh,bhij->bgij', ww, attn)             # :120-121
    attn = Q(attn)
    if cap is not None:
        cap[tag + '.mixed'] = attn
    out = Q((attn @ v).transpose(1, 2).reshape(B, N, C))            # :125,131
    if cap is not None:
        cap[tag + '.ctx'] = out
    return _lin(out, sd[p + 'proj.weight'], sd[p + 'proj.bias'], grad_operand=True)  # :132


def _student_blocks(x, sd, heads, repeats, use_transform, cap, need_rep=False):
    # reference weight_share_model.py:199-218 (RepeatedMiniBlock) and :179-185 (MiniBlock)
    n_blocks = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
    reps = []
    for i in range(n_blocks):
        p = f'blocks.{i}.block.'
        for r in range(repeats):
            tag = f'sblock{i}.{r}'
            h = Q(_ln(x, sd, p + f'norm1.instances.{r}'))
            x = x + _mini_attention(h, sd, p + 'attn.', r, heads, use_transform, cap, tag)
            h = Q(_ln(x, sd, p + f'norm2.instances.{r}'))
            u = _gelu(_lin(h, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias']))         # timm Mlp, exact erf GELU
            x = x + _lin(u, sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'], grad_operand=True)
            if cap is not None:
                cap[tag + '.out'] = x
            if need_rep:
                reps.append(x)
    return x, reps


def student_image_forward(sd, image, num_heads, repeated_times=2, use_transform=True, need_rep=False, cap=None):
    """reference weight_share_model.py:336-372 (RepeatVisionTransformer.forward_features)."""
    w = sd['patch_embed.proj.weight']
    patch = w.shape[-1]
    if _MATCHED:
        x = (Qb(F.conv2d(Qf(image), Qf(w), stride=patch)) + sd['patch_embed.proj.bias'].view(1, -1, 1, 1)).flatten(2).transpose(1, 2)
    else:
        x = F.conv2d(image, w, sd['patch_embed.proj.bias'], stride=patch).flatten(2).transpose(1, 2)   # :344 (timm PatchEmbed)
    x = torch.cat((sd['cls_token'].expand(x.shape[0], -1, -1), x), dim=1) + sd['pos_embed']        # :346-349
    emb = x
    x, reps = _student_blocks(x, sd, num_heads, repeated_times, use_transform, cap, need_rep)
    x = _lin(Q(_ln(x, sd, 'norm')), sd['head.weight'], sd['head.bias'], grad_operand=True)          # :363-364
    return dict(last_representation=x[:, 0], last_layer_output=x, representations=reps, embedding=emb)


def student_text_forward(sd, text, num_heads, repeated_times=2, use_transform=True, need_rep=False, cap=None):
    """reference weight_share_model.py:482-512 (RepeatTextTransformer.forward_features); no attention mask."""
    if 'patch_embed.weight' in sd:
        x = sd['patch_embed.weight'][text]                                                         # :407
    else:                                                                                          # :402-405
        x = _lin(Qf(sd['patch_embed.0.weight'][text]), sd['patch_embed.1.weight'], sd['patch_embed.1.bias'], grad_operand=True)
    x = x + sd['pos_embed']                                                                        # :489
    emb = x
    x, reps = _student_blocks(x, sd, num_heads, repeated_times, use_transform, cap, need_rep)
    x = _lin(Q(_ln(x, sd, 'norm')), sd['head.weight'], sd['head.bias'], grad_operand=True)          # :503-504
    pick = x[torch.arange(x.shape[0]), text.argmax(dim=-1)]                                        # :506
    return dict(last_representation=pick, last_layer_output=x, representations=reps, embedding=emb)


def clip_forward(image_out, text_out):
    """reference clip_model.py:37-44: L2-normalise (dim=1), logits = img @ txt.T, t2i = logits.T. No logit scale."""
    i = image_out['last_representation']
    t = text_out['last_representation']
    i = i / i.norm(dim=1, keepdim=True)
    t = t / t.norm(dim=1, keepdim=True)
    logits = i @ t.t()
    return dict(visual_output=image_out, text_output=text_out, i2t_logits=logits, t2i_logits=logits.T)
