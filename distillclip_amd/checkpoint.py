"""Checkpoints in the reference's on-disk layout.

The reference trains under Lightning, whose `ModelCheckpoint` writes `{'state_dict': {'student.<key>': ..,
'teacher.<key>': ..}, 'optimizer_states': [AdamW.state_dict()], 'lr_schedulers': [..], 'epoch', 'global_step',
'hyper_parameters'}`; stage 2 (`l_clip.yaml` load_path) reads the stage-1 files back through `load_weight`
(reference model/dual_distill_model.py:22-38: keep `student.*`, strip the prefix).  `save_checkpoint` writes that layout from
the mirror modules (whose parameter names equal the reference's), `load_checkpoint` restores model + fused optimizer +
schedule.  Guarantee: the weights (`state_dict`, both directions, incl. the reference's `load_weight`) and the optimizer state
(`optimizer_states[0]` loads into `torch.optim.AdamW` and back).  `pytorch-lightning_version` carries a parseable 1.x version
(the reference does not pin Lightning; `pytorch_lightning.cli` + `find_usable_cuda_devices` + `validation_epoch_end` +
`save_config_overwrite` bound it to 1.8-1.9) so Lightning's checkpoint migration can compare it; the writer's own tag lives
under `distillclip_amd_format`.  `lr_schedulers[0]` carries the keys a `LambdaLR.load_state_dict` expects.
"""
import torch

FORMAT = 'distillclip_amd/lightning-layout-2'
LIGHTNING_VERSION = '1.9.5'


def _plain(v):
    if isinstance(v, (int, float, str, bool, type(None))):
        return v
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in v.items()}
    return repr(v)


def trainable_parameters(model):
    """the iteration the reference builds AdamW from (distil_model.py:161, dual_distill_model.py:195)"""
    return [p for p in model.parameters() if p.requires_grad]


def checkpoint_dict(model, optimizer=None, scheduler=None, epoch=0, global_step=0):
    if optimizer is not None and hasattr(optimizer, 'join'):
        optimizer.join()                         # un-joined per-tower updates must land before the weights are copied
    ckpt = {'epoch': int(epoch), 'global_step': int(global_step), 'pytorch-lightning_version': LIGHTNING_VERSION,
            'distillclip_amd_format': FORMAT,
            'state_dict': {k: v.detach().to('cpu', copy=True) for k, v in model.state_dict().items()},
            'hyper_parameters': _plain(dict(getattr(model, 'hparams', {})))}
    if optimizer is not None:
        sd = optimizer.state_dict(trainable_parameters(model))
        sd['state'] = {i: {k: (t.cpu() if torch.is_tensor(t) else t) for k, t in st.items()} for i, st in sd['state'].items()}
        ckpt['optimizer_states'] = [sd]
    if scheduler is not None:
        ckpt['lr_schedulers'] = [scheduler.state_dict()]
    return ckpt


def save_checkpoint(path, model, optimizer=None, scheduler=None, epoch=0, global_step=0):
    """Data-parallel runs: call on EVERY rank.  With the sharded optimizer each rank holds 1/W of the Adam moments, so building
    the dict is a collective (FusedAdamW.state_dict all-gathers them); only rank 0 writes the file, and all ranks leave
    together.  (Lightning's rank-0-only ModelCheckpoint convention would dead-lock in that all-gather: keep the call
    collective, or run DCLIP_DP_MODE=allreduce, where every rank holds the full state and rank 0 may save alone.)"""
    import torch.distributed as dist
    ckpt = checkpoint_dict(model, optimizer, scheduler, epoch, global_step)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not multi or dist.get_rank() == 0:
        torch.save(ckpt, path)
    if multi:
        dist.barrier()


def load_checkpoint(path, model, optimizer=None, scheduler=None, strict=True):
    """-> (epoch, global_step).  The optimizer must come from model.configure_optimizers() (towers materialised)."""
    ckpt = torch.load(path, map_location='cpu') if isinstance(path, (str, bytes)) or hasattr(path, 'read') else path
    if 'state_dict' not in ckpt:
        raise ValueError('not a Lightning-layout checkpoint: no "state_dict" entry')
    model.load_state_dict(ckpt['state_dict'], strict=strict)
    if optimizer is not None and ckpt.get('optimizer_states'):
        optimizer.load_state_dict(ckpt['optimizer_states'][0], trainable_parameters(model))
    if scheduler is not None and ckpt.get('lr_schedulers'):
        scheduler.load_state_dict(ckpt['lr_schedulers'][0])
    if hasattr(model, 'current_epoch'):
        model.current_epoch = int(ckpt.get('epoch', 0))
    return int(ckpt.get('epoch', 0)), int(ckpt.get('global_step', 0))


def student_state_dict(ckpt):
    """what the reference's load_weight extracts from a stage-1 checkpoint (dual_distill_model.py:29-31)"""
    return {k.replace('student.', ''): v for k, v in ckpt['state_dict'].items() if k.startswith('student')}
