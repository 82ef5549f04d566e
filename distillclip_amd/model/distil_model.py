"""DistillModel (reference model/distil_model.py:19-221): one-tower distillation.

The reference class is a pytorch_lightning.LightningModule; Lightning is not installed here, so the same constructor,
`forward`, `training_step`, `configure_optimizers`, `freeze_image_embedding` are provided on a plain nn.Module (the
methods Lightning would call).  `validation_step` / `validation_epoch_end` return the scalars the reference logs, computed
by the HIP retrieval kernel (distillclip_amd/metrics.py) instead of torchmetrics; wandb / heat-map logging is not mirrored.
"""
from typing import Dict, List

import torch
from torch import nn

from ._loss import LossCalculator
from .utils import teacher_load
from .component.weight_share_model import RepeatVisionTransformer
from .component.image_encoder import ImageEncoder
from .component._tower import shared_image_patches
from ..optim import FusedAdamW, EpochCosineSchedule
from ..parallel import GradSync
from ..metrics import retrieval_metrics, gather_rows


class _HParams(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class DistillModel(nn.Module):
    def __init__(self, student_encoder: torch.nn.Module, loss_control_para: Dict, download_root: str,
                 teacher_name: str = 'ViT-B/32', freeze_embed: bool = False, teacher_need_layers: List = None,
                 model_type: str = 'image', warm_steps=10, total_steps=200, weight_decay=1e-3, lr: float = 1e-3,
                 norm: bool = False, unfreeze_epoch=None, teacher_state_dict=None):
        super().__init__()
        if model_type not in ['text', 'image']:
            raise ValueError(f"the model_type should in ['text', 'image'], bug got {model_type}")     # reference :44-45
        self.hparams = _HParams(loss_control_para=loss_control_para, download_root=download_root, teacher_name=teacher_name,
                                freeze_embed=freeze_embed, teacher_need_layers=teacher_need_layers, model_type=model_type,
                                warm_steps=warm_steps, total_steps=total_steps, weight_decay=weight_decay, lr=lr, norm=norm,
                                unfreeze_epoch=unfreeze_epoch)
        self.student = student_encoder
        self.teacher_name = teacher_name
        self.teacher = teacher_load(teacher_name, download_root, model_type, need_layers=teacher_need_layers,
                                    state_dict=teacher_state_dict)
        self.loss_control = LossCalculator(**loss_control_para)
        self.need_return_para = self.loss_control.get_control_output()
        for p in self.teacher.parameters():
            p.requires_grad = False                                                       # reference :59-60
        if model_type == 'image' and freeze_embed:
            self.freeze_image_embedding()
        self.k_list = [1, 3, 5, 10, 20, 50]
        self.current_epoch = 0
        self._sync = None

    def forward(self, inputs):
        # reference :81-89
        # (image.yaml: teacher and student unfold the same images — one im2row for both when they cut them the same way; no-op for text)
        with shared_image_patches(inputs, [getattr(self.student, '_tower', None), getattr(self.teacher, '_tower', None)]):
            student_outs = self.student(inputs, self.need_return_para)
            with torch.no_grad():
                teacher_outs = self.teacher(inputs, self.need_return_para)
        if self.hparams.norm:
            # reference :86-88 (in place there; same values here, autograd-safe)
            for o in (student_outs, teacher_outs):
                o.last_representation = o.last_representation / o.last_representation.norm(dim=-1, keepdim=True)
        return student_outs, teacher_outs

    def training_step(self, inputs, batch_idx=0):
        # reference :97-102 (logging left to the caller: cal_res holds every scalar self.log would receive)
        self.teacher.eval()
        student_outs, teacher_outs = self.forward(inputs)
        loss, cal_res = self.loss_control(student_outs, teacher_outs, self.hparams.model_type)
        self.last_cal_res = cal_res
        return loss

    def towers(self):
        return [self.student._tower]

    def _ensure_sync(self):
        """data-parallel plumbing of the student towers (lazy: torch.distributed may be initialised after __init__)"""
        self._sync = GradSync.current(self._sync).attach(self.towers())
        return self._sync

    def backward_and_sync(self, loss, defer_wait=False):
        """loss.backward() + the data-parallel gradient exchange (reference strategy ddp_find_unused_parameters_false):
        bucketed reduce-scatter released from inside the backward + sharded AdamW + parameter all-gather (parallel.py)."""
        sync = self._ensure_sync()
        if loss is not None:
            sync.armed = sync.enabled                          # per-bucket release from inside the tower's backward: only here
            try:
                loss.backward()
            finally:
                sync.armed = False
        if not sync.enabled:
            return
        tw = self.student._tower
        if tw.dp is not None:
            tw.grads_ready = sync.finish(tw)
            tw._grad_clean = True
        else:
            tw.grads_ready = sync.launch(tw.flat_grad, after=tw.bwd_done)
        if not defer_wait:
            sync.wait()
        else:
            sync.forget()

    def _acc(self, log, rows, cols, section, prefix, acc=True, score=False):
        # reference norm_and_logits :224-231 builds stu_logits = stu_encode @ encode.T : rows = this tower, cols = the other
        m = retrieval_metrics(rows, cols, self.k_list)
        if acc:
            for k in self.k_list:                                          # reference log_acc :187-191
                log[f'{section}/{prefix}_acc_top{k}'] = m[f'acc_top{k}']
        if score:                                                          # reference log_diag_score :171-179
            log[f'{section}/{prefix}_softmax_mean_score'] = m['softmax_mean_score']
            log[f'{section}/{prefix}_mean_score'] = m['mean_score']

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        """reference :104-126.  batch = (imgs, texts, _) where the modality that is not distilled arrives as a
        pre-computed [B, E] teacher representation (`contrary_rep`)."""
        imgs, texts = batch[0], batch[1]
        inputs, contrary_rep = (texts, imgs) if self.hparams.model_type == 'text' else (imgs, texts)
        student_outs, teacher_outs = self.forward(inputs)
        loss, cal_res = self.loss_control(student_outs, teacher_outs, self.hparams.model_type)
        log = {'val_loss/loss': loss.detach()}
        log.update({f'val_loss/{k}': v for k, v in cal_res.items()})
        s, t = student_outs.last_representation, teacher_outs.last_representation
        self._acc(log, s, contrary_rep, 'val_step', 'stu', score=True)
        self._acc(log, t, contrary_rep, 'val_step', 'tea')
        return {'student': gather_rows(s), 'teacher': gather_rows(t), 'contrary_rep': gather_rows(contrary_rep)}, log

    @torch.no_grad()
    def validation_epoch_end(self, outputs):
        """reference :131-152"""
        cat = {k: torch.cat([o[k].reshape(-1, o[k].shape[-1]) for o in outputs], dim=0).float()
               for k in ('student', 'teacher', 'contrary_rep')}
        log = {}
        self._acc(log, cat['student'], cat['contrary_rep'], 'val_stu_acc', 'stu')
        self._acc(log, cat['student'], cat['contrary_rep'], 'val_stu_score', 'stu', acc=False, score=True)
        if self.current_epoch == 0:
            self._acc(log, cat['teacher'], cat['contrary_rep'], 'val_tea_score', 'tea', acc=False, score=True)
            self._acc(log, cat['teacher'], cat['contrary_rep'], 'val_tea_acc', 'tea')
        return log

    def configure_optimizers(self):
        # reference :160-169: AdamW over every requires_grad parameter (one group) + cosine schedule stepped per epoch
        self.student._tower.materialize(next(self.student.parameters()).device)
        extras = getattr(self.student, 'extra_parameters', lambda: [])()       # a plain CLIP encoder's projection linears
        opt = FusedAdamW([self.student._tower], lr=self.hparams.lr, weight_decay=self.hparams.weight_decay, extra_params=extras)
        sched = EpochCosineSchedule(opt, self.hparams.warm_steps, self.hparams.total_steps)
        self._ensure_sync()          # data-parallel run: shard plan over the same trainable set the optimizer was built with
        return [opt], [sched]

    def on_train_epoch_start(self):
        if self.hparams.unfreeze_epoch and self.current_epoch >= self.hparams.unfreeze_epoch:
            self.unfreeze_embed()
            self.hparams.unfreeze_epoch = False

    def unfreeze_embed(self):
        for _, p in self.student.named_parameters():
            p.requires_grad = True

    def freeze_image_embedding(self):
        # reference :197-213
        if isinstance(self.student, ImageEncoder):             # reference :211-219
            freeze_key = ['visual.conv1.weight', 'visual.class_embedding', 'visual.positional_embedding']
            sw, tw = self.student.state_dict(), self.teacher.state_dict()
            for k in freeze_key:
                sw[k] = tw[k]
            self.student.load_state_dict(sw)
            for n, p in self.student.named_parameters():
                if n in freeze_key:
                    p.requires_grad = False
            return
        if not isinstance(self.student, RepeatVisionTransformer):
            return                                             # (the reference does nothing for other student classes)
        stu_keys = ['patch_embed.proj.weight', 'cls_token', 'pos_embed']
        tea_keys = ['visual.conv1.weight', 'visual.class_embedding', 'visual.positional_embedding']
        sw, tw = self.student.state_dict(), self.teacher.state_dict()
        for s_k, t_k in zip(stu_keys, tea_keys):
            w = tw[t_k]
            if 'cls_token' in s_k:
                w = w.unsqueeze(0).unsqueeze(0)
            if 'pos_embed' in s_k:
                w = w.unsqueeze(0)
            sw[s_k] = w
        self.student.load_state_dict(sw)
        for n, p in self.student.named_parameters():
            if n in stu_keys:
                p.requires_grad = False
