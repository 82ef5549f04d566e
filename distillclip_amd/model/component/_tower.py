"""HipTower: binds an nn.Module's parameters to one dclip_encoder handle (include/dclip.h, tower-level runtime).

Owns the flat f32 parameter / gradient buffers (parameters become views of one buffer so the optimizer and the
RCCL gradient exchange see a single contiguous tensor), the bf16 weight cache and the activation workspace.
"""
import ctypes
import os
import threading

import torch

from ..._lib import lib


class EncoderCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        'kind', 'modality', 'tokens', 'width', 'heads', 'layers', 'repeats', 'mlp_dim', 'out_dim', 'patch', 'resolution',
        'in_chans', 'vocab', 'embed_rank', 'head_mix', 'causal')]


_BUCKET_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int32)      # include/dclip.h: dclip_bucket_cb


_SHARE = threading.local()


class shared_image_patches:
    """Context: image towers that run inside it on the SAME image tensor with the same patch size take their patch rows from ONE
    dclip_im2row conversion (made here, on the current stream) instead of converting the batch once each.  The reference feeds
    teacher and student the same `image` (dual_distill_model.py:107-109) and both start with the same stride-p unfolding
    (_common.py:196-198, weight_share_model.py:344).  A tower keeps the rows alive until its backward has used them (patch-embedding
    wgrad).  No-op unless at least two of `towers` are image towers of equal patch size and channel count; DCLIP_SHARE_PATCHES=0
    switches it off."""

    def __init__(self, image, towers):
        self.image, self.entry = image, None
        tw = [t for t in towers if t is not None and t.cfg.modality == 0]
        ok = (len(tw) >= 2 and isinstance(image, torch.Tensor) and image.is_cuda and image.dtype == torch.float32 and image.dim() == 4
              and image.is_contiguous() and image.shape[2] == image.shape[3] and os.environ.get('DCLIP_SHARE_PATCHES', '1') != '0'
              and len({(int(t.cfg.patch), int(t.cfg.in_chans)) for t in tw}) == 1 and image.shape[1] == tw[0].cfg.in_chans
              and image.shape[-1] % 4 == 0 and tw[0].cfg.patch % 4 == 0)
        if ok:
            patch, chans, res, B = int(tw[0].cfg.patch), int(tw[0].cfg.in_chans), int(image.shape[-1]), int(image.shape[0])
            grid = res // patch
            rows = torch.empty((B * (grid * grid + 1), chans * patch * patch), dtype=torch.bfloat16, device=image.device)
            lib().dclip_im2row(image.data_ptr(), rows.data_ptr(), B, chans, res, patch, 1, torch.cuda.current_stream().cuda_stream)
            self.entry = dict(key=(image.data_ptr(), tuple(image.shape), image._version), patch=patch, chans=chans, rows=rows,
                              stream=torch.cuda.current_stream())

    def __enter__(self):
        self.prev = getattr(_SHARE, 'entry', None)
        _SHARE.entry = self.entry
        return self

    def __exit__(self, *exc):
        _SHARE.entry = self.prev
        return False


def _shared_rows_for(x, cfg):
    """the active share's patch rows if `x` is the tensor they were cut from and this tower cuts the same way, else None"""
    e = getattr(_SHARE, 'entry', None)
    if e is None or cfg.modality != 0 or e['key'] != (x.data_ptr(), tuple(x.shape), x._version):
        return None
    if int(cfg.patch) != e['patch'] or int(cfg.in_chans) != e['chans']:
        return None
    rows = e['rows']
    cur = torch.cuda.current_stream()
    if cur != e['stream']:
        rows.record_stream(cur)                # cut on the share's stream (the tower streams wait for it), read on this one
    return rows


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class HipTower:
    def __init__(self, module, cfg: EncoderCfg, param_names):
        """param_names: reference state_dict keys in the canonical order of include/dclip.h (None = absent)."""
        self.module = module
        self.cfg = cfg
        self.param_names = list(param_names)
        self._handle = lib().dclip_encoder_create(ctypes.byref(cfg))
        if not self._handle:
            raise ValueError(lib().dclip_last_error_string().decode())
        n = lib().dclip_encoder_num_params(self._handle)
        if n != len(self.param_names):
            raise RuntimeError(f'parameter table mismatch: runtime expects {n}, module lists {len(self.param_names)}')
        self.flat = None
        self.flat_grad = None
        self.wcache = None
        self.wcache_dirty = True
        self.workspace = None
        self._ws_key = None
        self._saved_batch = None
        self.bwd_done = None
        self._grad_clean = False                  # True after an optimizer step that cleared the gradients it consumed
        self.bwd_stream = None
        self.grads_ready = None
        self.opt_done = None                       # event of an un-joined FusedAdamW.step(overlap=True, join=False)
        self.sync = None                           # parallel.GradSync of a data-parallel run (set by the model)
        self.dp = None                             # parallel._Shards: bucket / shard plan of this tower's flat buffers
        self.gshard = None                         # this rank's averaged gradient shards (reduce-scatter output)
        self.dp_released = 0
        # trainable towers re-cast their bf16 weight cache at every forward unless an optimizer that maintains
        # `wcache_dirty` itself (FusedAdamW) has taken over; any other in-place update of the masters needs the re-cast
        self._prepare_always = True
        # load_state_dict copies into the flat buffer in place: frozen towers must re-cast their bf16 weight cache afterwards
        module.register_load_state_dict_post_hook(lambda m, incompatible: setattr(self, 'wcache_dirty', True))

    def __del__(self):
        try:
            if getattr(self, '_handle', None):
                lib().dclip_encoder_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    # ---- parameter plumbing -------------------------------------------------------------------------------------
    def _params(self):
        """reference-named parameters in the canonical order of include/dclip.h.  The module tree of a tower never changes
        after construction (load_state_dict / .cuda() keep the Parameter objects), so names are resolved once; a replaced
        Parameter object is caught by an identity check of the first one (looked up through its owning sub-module)."""
        cached = getattr(self, '_plist', None)
        if cached is not None and self._first_owner._parameters.get(self._first_key) is self._first_param:
            return cached
        table = dict(self.module.named_parameters())
        plist = [None if n is None else table[n] for n in self.param_names]
        first = next(n for n in self.param_names if n is not None)
        owner = self.module
        *path, key = first.split('.')
        for part in path:
            owner = getattr(owner, part)
        self._first_owner, self._first_key, self._first_param = owner, key, owner._parameters.get(key)
        self._plist = plist
        return plist

    def materialize(self, device):
        """(Re)build the flat buffers when the parameters are not (any more) views of them (first use, .to(), .cuda())."""
        ps = self._params()
        live = [p for p in ps if p is not None]
        ok = self.flat is not None and self.flat.device == device
        if ok:
            off = 0
            for p in live:
                if p.data_ptr() != self.flat.data_ptr() + off * 4 or p.dtype != torch.float32:
                    ok = False
                    break
                off += (p.numel() + 63) // 64 * 64
        if ok:
            return
        if device.type != 'cuda':
            raise RuntimeError('distillclip_amd towers run on MI355X only (no CPU fallback): move the module and its inputs to cuda')
        total = sum((p.numel() + 63) // 64 * 64 for p in live)
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        off = 0
        with torch.no_grad():
            for p in live:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
                p.data = flat[off:off + n].view(p.shape)
                p.grad = None
                off += (n + 63) // 64 * 64
        self.flat, self.flat_grad = flat, flat_grad
        self._offsets = []
        off = 0
        for p in live:
            self._offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        wbytes = lib().dclip_encoder_wcache_bytes(self._handle)
        self.wcache = torch.empty(wbytes, dtype=torch.uint8, device=device)
        self.wcache_dirty = True

    def param_offsets(self):
        """flat element offset of every canonical parameter index (absent parameters take the next live one's), + total"""
        ps = self._params()
        offs, off = [], 0
        for p in ps:
            offs.append(off)
            if p is not None:
                off += (p.numel() + 63) // 64 * 64
        offs.append(off)
        return offs

    def grad_buckets(self):
        """[(begin, end)] flat ranges in the order the backward completes them (include/dclip.h: dclip_encoder_grad_bucket)"""
        offs = self.param_offsets()
        first, end = ctypes.c_int32(), ctypes.c_int32()
        out = []
        for i in range(lib().dclip_encoder_num_grad_buckets(self._handle)):
            lib().dclip_encoder_grad_bucket(self._handle, i, ctypes.byref(first), ctypes.byref(end))
            out.append((offs[first.value], offs[end.value]))
        return out

    def trainable_ranges(self):
        """contiguous [begin, end) element ranges of the requires_grad parameters inside the flat buffers"""
        live = [p for p in self._params() if p is not None]
        out = []
        for p, off in zip(live, self._offsets):
            if not p.requires_grad:
                continue
            end = off + (p.numel() + 63) // 64 * 64
            if out and out[-1][1] == off:
                out[-1][1] = end
            else:
                out.append([off, end])
        return out

    def sparse_spec(self):
        """(bucket index, flat begin, rows, dim) of a token-embedding table whose gradient is row-sparse (the uncompressed text
        student: reference weight_share_model.py:407, nn.Embedding(49408, 768)), else None"""
        if self.cfg.modality != 1 or self.cfg.kind != 1 or self.cfg.embed_rank != 0:
            return None
        offs = self.param_offsets()
        return lib().dclip_encoder_num_grad_buckets(self._handle) - 1, offs[0], self.cfg.vocab, self.cfg.width

    def attach_grads(self):
        """Make p.grad views of the flat gradient buffer.  If any trainable p.grad was dropped (zero_grad(set_to_none)),
        the buffer is zeroed first — the kernels accumulate with +=, like autograd does into an existing .grad."""
        live = [p for p in self._params() if p is not None]
        # fast path (every step after the first): first and last trainable parameters still view the flat buffer
        tr = getattr(self, '_attached', None)
        if tr is not None and tr[0] == self.flat_grad.data_ptr():
            pa, pb = tr[1], tr[2]
            if pa.grad is not None and pb.grad is not None and pa.grad.data_ptr() == tr[3] and pb.grad.data_ptr() == tr[4] \
                    and tr[5] == sum(1 for p in live if p.requires_grad):
                return
        fresh = any(p.requires_grad and p.grad is None for p in live)
        if fresh:
            self.flat_grad.zero_()
        for p, off in zip(live, self._offsets):
            if p.requires_grad:
                want = self.flat_grad.data_ptr() + off * 4
                if p.grad is None or p.grad.data_ptr() != want:
                    p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)
        trainable = [p for p in live if p.requires_grad]
        self._attached = (self.flat_grad.data_ptr(), trainable[0], trainable[-1], trainable[0].grad.data_ptr(),
                          trainable[-1].grad.data_ptr(), len(trainable)) if trainable else None

    def _ensure_workspace(self, batch, training, device):
        key = (batch, bool(training))
        need = lib().dclip_encoder_workspace_bytes(self._handle, batch, 1 if training else 0)
        if self.workspace is None or self.workspace.numel() < need or self.workspace.device != device:
            self.workspace = torch.empty(need, dtype=torch.uint8, device=device)
        self._ws_key = key

    def prepare(self):
        ps = self._params()
        lib().dclip_encoder_prepare(self._handle, _ptr_array(ps), self.wcache.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
        self.wcache_dirty = False

    def _bind_image_size(self, x):
        """The patch conv floors (reference _common.py:176,196 / timm PatchEmbed: Conv2d(k=p, s=p) on a 336 px image with p = 32
        reads the top-left 320 x 320): the plan's `resolution` is the row stride of the INPUT, which a teacher built from a
        checkpoint only knows as patch * grid (utils.py:18-28 -> 320 for a [101, 768] positional table).  Re-plan (host-side,
        same token count, same buffers) when the images are larger than that; anything that changes the grid is an error."""
        if x.dim() != 4 or x.shape[1] != self.cfg.in_chans or x.shape[2] != x.shape[3]:
            raise ValueError(f'image tower expects [B, {self.cfg.in_chans}, R, R] inputs, got {tuple(x.shape)}')
        res = int(x.shape[-1])
        if res == self.cfg.resolution:
            return
        grid = res // self.cfg.patch
        if grid * grid + 1 != self.cfg.tokens:
            raise ValueError(f'{res} px images give {grid * grid + 1} tokens at patch {self.cfg.patch}; this tower has '
                             f'{self.cfg.tokens} (positional table)')
        cfg = EncoderCfg.from_buffer_copy(self.cfg)
        cfg.resolution = res
        handle = lib().dclip_encoder_create(ctypes.byref(cfg))
        if not handle:
            raise ValueError(lib().dclip_last_error_string().decode())
        lib().dclip_encoder_destroy(self._handle)
        self._handle, self.cfg = handle, cfg

    # ---- execution -----------------------------------------------------------------------------------------------
    def forward(self, x, training, need_rep=False, need_emb=False, rep_layers=None, tokens_eff=0):
        """-> (last_representation [B,E], input as passed to C, hidden states list, embedding or None)"""
        if not x.is_cuda:
            raise RuntimeError('distillclip_amd towers need CUDA(HIP) inputs; there is no CPU fallback')
        expect = torch.float32 if self.cfg.modality == 0 else torch.int64
        if x.dtype != expect:
            x = x.to(expect)
        x = x.contiguous()
        if self.cfg.modality == 0:
            self._bind_image_size(x)
        self.materialize(x.device)
        ev = getattr(self, 'opt_done', None)
        if ev is not None:                         # an optimizer update of these weights may still be running on another stream
            torch.cuda.current_stream().wait_event(ev)
        B = x.shape[0]
        if training and self.dp is not None and self.sync is not None and self.sync.enabled:
            self.sync.note_token_ids(self, x)      # text student: the global batch's token ids, for the row-sparse table exchange
        if self.wcache_dirty or (self._prepare_always and (training or any(p.requires_grad for p in self.module.parameters()))):
            self.prepare()      # trainable towers: the optimizer moved the f32 masters since the last cast
        self._ensure_workspace(B, training, x.device)
        out = torch.empty((B, self.cfg.out_dim), dtype=torch.float32, device=x.device)
        ps = self._params()
        nex = self.cfg.layers * self.cfg.repeats
        N, D = self.cfg.tokens, self.cfg.width
        reps, rep_arr = [], None
        if need_rep:
            want = range(nex) if rep_layers is None else [i for i in rep_layers if 0 <= i < nex]
            slots = [None] * nex
            for i in want:
                slots[i] = torch.empty((B, N, D), dtype=torch.float32, device=x.device)
            reps = [slots[i] for i in want]
            rep_arr = _ptr_array(slots)
        emb = torch.empty((B, N, D), dtype=torch.float32, device=x.device) if need_emb else None
        rows = _shared_rows_for(x, self.cfg) if not tokens_eff else None
        self._patch_rows = rows if training else None      # (kept until the backward: operand of the patch-embedding wgrad)
        if rows is not None:
            lib().dclip_encoder_forward_patches(self._handle, rows.data_ptr(), B, _ptr_array(ps), self.wcache.data_ptr(),
                                                self.workspace.data_ptr(), self.workspace.numel(), 1 if training else 0,
                                                out.data_ptr(), rep_arr, None if emb is None else emb.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream)
        else:
            lib().dclip_encoder_forward(self._handle, x.data_ptr(), B, _ptr_array(ps), self.wcache.data_ptr(),
                                        self.workspace.data_ptr(), self.workspace.numel(), 1 if training else 0,
                                        out.data_ptr(), rep_arr, None if emb is None else emb.data_ptr(), int(tokens_eff),
                                        torch.cuda.current_stream().cuda_stream)
        self._saved_batch = B if training else None
        self._last_fwd = (B, bool(training), int(tokens_eff))
        return out, x, reps, emb

    @torch.no_grad()
    def last_layer_output(self):
        """[B, N, E] f32: final norm + projection of EVERY token of the most recent forward (reference `last_layer_output`,
        output.py:16-35).  Computed on request from the residual stream still resident in the workspace; not part of the autograd
        graph (no loss term on the hot path consumes it)."""
        last = getattr(self, '_last_fwd', None)
        if last is None:
            raise RuntimeError('last_layer_output: run a forward first')
        B, training, tokens_eff = last
        if tokens_eff:
            raise RuntimeError('last_layer_output is not available when the text teacher ran on a caption prefix (max_tokens hint)')
        N, D, E = self.cfg.tokens, self.cfg.width, self.cfg.out_dim
        dev = self.workspace.device
        scratch = torch.empty((B * N, D), dtype=torch.bfloat16, device=dev)
        out = torch.empty((B, N, E), dtype=torch.float32, device=dev)
        lib().dclip_encoder_last_layer_output(self._handle, B, _ptr_array(self._params()), self.wcache.data_ptr(),
                                              self.workspace.data_ptr(), self.workspace.numel(), 1 if training else 0,
                                              scratch.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out

    def backward(self, x, d_out, d_reps=None, d_emb=None, grad_flat=None):
        """grad_flat: None = accumulate into p.grad (views of the tower's flat gradient buffer: the product path); a zeroed flat
        buffer = write this backward's gradients there and leave p.grad to autograd (_TowerFn hands the views back as the
        gradients of its parameter inputs, so AccumulateGrad hooks — a DistributedDataParallel reducer — see them)."""
        B = x.shape[0]
        if self._saved_batch != B:
            raise RuntimeError('backward without a matching training-mode forward (activations are kept in the workspace '
                               'of the most recent forward)')
        ps = self._params()
        if grad_flat is None:
            self.attach_grads()
            self._grad_clean = False              # the kernels below accumulate into the flat gradient buffer
            gs = [None if (p is None or not p.requires_grad) else p.grad for p in ps]
        else:
            live_off = iter(self._offsets)
            gs = []
            for p in ps:
                if p is None:
                    gs.append(None)
                    continue
                off = next(live_off)
                gs.append(grad_flat[off:off + p.numel()].view(p.shape) if p.requires_grad else None)
        if d_out is None:
            d_out = torch.zeros((B, self.cfg.out_dim), dtype=torch.float32, device=x.device)
        d_out = d_out.contiguous().float()
        keep = [None if g is None else g.contiguous().float() for g in (d_reps or [])]
        d_emb = None if d_emb is None else d_emb.contiguous().float()
        # data-parallel run: every gradient bucket is handed to the exchange as soon as its last writer is enqueued, so the
        # reduce-scatter of block l travels under the backward GEMMs of block l - 1
        cb, failed = None, []
        if grad_flat is None and self.dp is not None and self.sync is not None and self.sync.enabled and self.sync.armed:
            def _ready(_user, bucket):
                try:
                    self.sync.bucket_ready(self, bucket)
                except BaseException as exc:       # an exception cannot cross the C frame: re-raised below
                    failed.append(exc)
            cb = _BUCKET_CB(_ready)
            self.dp_released = 0
        rows = getattr(self, '_patch_rows', None)
        if rows is not None:
            rows.record_stream(torch.cuda.current_stream())
        call = lib().dclip_encoder_backward if rows is None else lib().dclip_encoder_backward_patches
        call(self._handle, x.data_ptr() if rows is None else rows.data_ptr(), B, _ptr_array(ps), _ptr_array(gs), self.wcache.data_ptr(),
             self.workspace.data_ptr(), self.workspace.numel(), d_out.data_ptr(),
             _ptr_array(keep) if any(g is not None for g in keep) else None,
             None if d_emb is None else d_emb.data_ptr(),
             ctypes.cast(cb, ctypes.c_void_p) if cb is not None else None, None,
             torch.cuda.current_stream().cuda_stream)
        if failed:
            raise failed[0]
        self._saved_batch = None
        self._patch_rows = None
        # gradient exchange may start as soon as THIS tower's backward is done (GradSync waits on this event, not on the
        # whole backward pass): the image tower's all-reduce overlaps the (longer) text tower backward
        if self.bwd_done is None:
            self.bwd_done = torch.cuda.Event()
        self.bwd_stream = torch.cuda.current_stream()
        self.bwd_done.record(self.bwd_stream)
        self.grads_ready = None                    # set by the gradient exchange (event after this tower's all-reduce)
        return gs


def autograd_params_mode(tower):
    """Who sees the parameter gradients.  False (default): the backward accumulates straight into p.grad — views of the tower's flat
    buffer, which the fused optimizer and the built-in exchange consume; nothing passes through autograd's AccumulateGrad nodes.
    True: the parameters are inputs of the autograd Function and their gradients are returned to autograd, so hooks on them fire
    — what torch.nn.parallel.DistributedDataParallel (Lightning's ddp strategy, reference l_clip.yaml:56) needs to find, bucket
    and all-reduce them; costs one zero-fill of a gradient buffer per backward plus autograd's own accumulation pass.
    Chosen by DCLIP_DP_MODE=off (an outer wrapper owns the exchange) or per tower through `tower.autograd_params`."""
    forced = getattr(tower, 'autograd_params', None)
    if forced is not None:
        return bool(forced)
    return os.environ.get('DCLIP_DP_MODE', '') == 'off'


class _TowerFn(torch.autograd.Function):
    """autograd edge of a student tower: forward / backward are one C-ABI call each.  Safe under torch.autocast (Lightning
    `precision: 16`, reference l_clip.yaml:64): floating inputs arrive as fp32, autocast is off inside, and the backward runs in
    the forward's autocast state; a GradScaler's scale reaches the kernels through d_out and leaves through unscale_ on p.grad."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, anchor, x, tower, need_rep, need_emb, *params):
        out, xin, reps, emb = tower.forward(x, training=True, need_rep=need_rep, need_emb=need_emb)
        ctx.tower = tower
        ctx.x = xin
        ctx.n_rep = len(reps)
        ctx.has_emb = emb is not None
        ctx.n_params = len(params)
        return (out,) + tuple(reps) + ((emb,) if emb is not None else ())

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, d_out, *rest):
        d_reps = list(rest[:ctx.n_rep])
        d_emb = rest[ctx.n_rep] if ctx.has_emb else None
        tower = ctx.tower
        if not ctx.n_params:
            tower.backward(ctx.x, d_out, d_reps, d_emb)
            return None, None, None, None, None
        # a buffer of its own per backward: autograd may keep the returned views as p.grad (no copy), and the next backward must not
        # write into them
        grad_flat = torch.zeros_like(tower.flat)
        gs = tower.backward(ctx.x, d_out, d_reps, d_emb, grad_flat=grad_flat)
        return (None, None, None, None, None) + tuple(g for p, g in zip(tower._params(), gs) if p is not None)


def run_tower(tower, x, anchor, need_rep=False, need_emb=False):
    """-> (last_representation, [hidden state per block execution], embedding or None)"""
    if torch.is_grad_enabled() and any(p.requires_grad for p in tower.module.parameters()):
        params = ()
        if autograd_params_mode(tower):
            tower.materialize(x.device)            # the parameters must already be the views of the flat buffer they will stay
            params = tuple(p for p in tower._params() if p is not None)
        res = _TowerFn.apply(anchor, x, tower, need_rep, need_emb, *params)
        nex = tower.cfg.layers * tower.cfg.repeats if need_rep else 0
        return res[0], list(res[1:1 + nex]), (res[1 + nex] if need_emb else None)
    out, _, reps, emb = tower.forward(x, training=False, need_rep=need_rep, need_emb=need_emb)
    return out, reps, emb
