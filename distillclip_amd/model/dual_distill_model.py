"""DualDistillModel (reference model/dual_distill_model.py:41-268): two-tower (image + text) distillation.

Same constructor / forward / training_step / configure_optimizers / load_weight / freeze_with_prefix as the reference's
LightningModule, on a plain nn.Module (Lightning, wandb and torchmetrics are absent here).  `validation_step` /
`validation_epoch_end` return the dictionary of scalars the reference hands to `self.log`, computed by the HIP retrieval
kernel (distillclip_amd/metrics.py) instead of torchmetrics.
"""
import os
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from ._loss import LossCalculator
from .utils import teacher_load
from .component.clip_model import CLIPModel
from .component.output import CLIPOutput
from .component.weight_share_model import RepeatVisionTransformer
from .component.image_encoder import ImageEncoder
from .component._tower import shared_image_patches
from .distil_model import _HParams
from ..optim import FusedAdamW, EpochCosineSchedule
from ..parallel import GradSync
from ..metrics import retrieval_metrics, gather_rows


def load_weight(image_student, text_student, load_path):
    # reference dual_distill_model.py:22-38: Lightning checkpoint -> strip the 'student.' prefix
    def load_one_model(model: nn.Module, cpk: Optional[str]):
        if cpk is None:
            raise ValueError('the cpk is None! if you set the load_path parameter in model,'
                             ' you should give the image and text checkpoint path')
        save_res = torch.load(cpk, map_location='cpu')
        state_dict = {k.replace('student.', ''): v for k, v in save_res['state_dict'].items() if k.startswith('student')}
        model.load_state_dict(state_dict)
        return model

    return load_one_model(image_student, load_path['image']), load_one_model(text_student, load_path['text'])


class DualDistillModel(nn.Module):
    def __init__(self, image_student: nn.Module, text_student: nn.Module, loss_control_para: Dict, warm_steps, total_steps,
                 weight_decay, lr: float, download_root: str, norm=False, teacher_name: str = 'ViT-B/32',
                 freeze_embed: bool = False, unfreeze_epoch: int = None, load_path: Dict = None,
                 teacher_need_layers: List = None, freeze_prefix: List = None, teacher_state_dict=None):
        super().__init__()
        self.hparams = _HParams(loss_control_para=loss_control_para, warm_steps=warm_steps, total_steps=total_steps,
                                weight_decay=weight_decay, lr=lr, download_root=download_root, norm=norm,
                                teacher_name=teacher_name, freeze_embed=freeze_embed, unfreeze_epoch=unfreeze_epoch,
                                load_path=load_path, teacher_need_layers=teacher_need_layers, freeze_prefix=freeze_prefix)
        if load_path:
            image_student, text_student = load_weight(image_student, text_student, load_path)
        self.student = CLIPModel(True, image_student, text_student, norm)
        self.teacher = teacher_load(teacher_name, download_root, 'all', need_layers=teacher_need_layers,
                                    state_dict=teacher_state_dict)
        for p in self.teacher.parameters():
            p.requires_grad = False                                                       # reference :76-77
        self.loss_control = LossCalculator(**loss_control_para)
        self.need_return_para = self.loss_control.get_control_output()
        if freeze_embed:
            self.freeze_image_embedding()
        self.unfreeze_epoch = unfreeze_epoch
        self.freeze_with_prefix(prefix_list=freeze_prefix)
        self.k_list = [1, 3, 5, 10, 20, 50]
        self.current_epoch = 0
        self._sync = None
        self._streams = None
        self.multi_stream = os.environ.get('DCLIP_MULTI_STREAM', '1') != '0'

    def set_text_length_hint(self, max_tokens):
        """Host-side knowledge of the batch (the tokenizer knows it): every caption's EOT lies in the first `max_tokens`
        positions.  Only the frozen, causal teacher text tower uses it (the students attend over the padding)."""
        self.teacher.text_encoder.max_tokens = max_tokens

    def towers(self):
        return [self.student.image_encoder._tower, self.student.text_encoder._tower]

    def _tower_streams(self):
        if self._streams is None:
            self._streams = [torch.cuda.Stream() for _ in range(4)]
        return self._streams

    @torch.no_grad()
    def teacher_forward_async(self, inputs):
        """Issue the frozen teacher towers for a batch on their two streams WITHOUT joining the current stream, and return a
        handle for `forward(inputs, teacher=handle)`.  The teacher does not depend on the student, so a training loop may call this
        for batch t+1 right after `loss.backward()` of batch t: the teacher's forward then runs under the students' backward (the
        reference computes it inside the same forward, dual_distill_model.py:109; the values are identical)."""
        image, text = inputs
        main = torch.cuda.current_stream()
        streams = self._tower_streams()
        co = self.need_return_para
        outs, events = [], []
        for st, enc, x in ((streams[0], self.teacher.image_encoder, image), (streams[1], self.teacher.text_encoder, text)):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                o = enc(x, co)
                ev = torch.cuda.Event()
                ev.record(st)
            outs.append(o)
            events.append(ev)
        return CLIPOutput(visual_output=outs[0], text_output=outs[1]), events

    def _forward_towers(self, image, text, teacher):
        if not (self.multi_stream and image.is_cuda):
            student_outs = self.student(text, image, self.need_return_para)
            if teacher is not None:
                teacher_outs, events = teacher
                for ev in events:
                    torch.cuda.current_stream().wait_event(ev)
            else:
                with torch.no_grad():  # the reference only relies on requires_grad=False; values are identical
                    teacher_outs = self.teacher(text, image, self.need_return_para)
        else:
            # The four towers are independent until the loss: issue each on its own HIP stream so their kernels fill each
            # other's tail waves and hide the latency-bound attention kernels.  autograd replays each tower's backward on
            # the stream its forward ran on, so the two student backwards overlap as well.
            main = torch.cuda.current_stream()
            streams = self._tower_streams()
            co = self.need_return_para
            jobs = [(streams[2], self.student.image_encoder, image, True), (streams[3], self.student.text_encoder, text, True)]
            if teacher is None:
                jobs = [(streams[0], self.teacher.image_encoder, image, False), (streams[1], self.teacher.text_encoder, text, False)] + jobs
            outs = []
            for st, enc, x, grad in jobs:
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    if grad:
                        o = enc(x, co)
                    else:
                        with torch.no_grad():
                            o = enc(x, co)
                outs.append(o)
            for st, *_ in jobs:
                main.wait_stream(st)
            if teacher is not None:
                teacher_outs, events = teacher
                for ev in events:
                    main.wait_event(ev)
                outs = [teacher_outs.visual_output, teacher_outs.text_output] + outs
            for o in outs:
                for t_ in [o.last_representation, o.embedding] + list(o.representations or []):
                    if t_ is not None:
                        t_.record_stream(main)
            teacher_outs = CLIPOutput(visual_output=outs[0], text_output=outs[1])
            student_outs = CLIPOutput(visual_output=outs[2], text_output=outs[3])
        return student_outs, teacher_outs

    def forward(self, inputs, teacher=None) -> Tuple[CLIPOutput, CLIPOutput]:
        # reference :106-112.  The batch is (image, text) while CLIPModel.forward takes (text, image).
        image, text = inputs
        # teacher and student image towers unfold the same images: one im2row for both when they cut them the same way
        towers = [getattr(self.student.image_encoder, '_tower', None)]
        if teacher is None:
            towers.append(getattr(self.teacher.image_encoder, '_tower', None))
        with shared_image_patches(image, towers):
            student_outs, teacher_outs = self._forward_towers(image, text, teacher)
        if self.hparams.norm:
            # reference :110-111 + norm_last_representation :278-284 (in place there; same values here, autograd-safe)
            for outs in (student_outs, teacher_outs):
                for o in (outs.visual_output, outs.text_output):
                    o.last_representation = o.last_representation / o.last_representation.norm(dim=-1, keepdim=True)
        return student_outs, teacher_outs

    def training_step(self, inputs, batch_idx=0, teacher=None):
        # reference :120-127 ; `teacher`: optional handle from teacher_forward_async(inputs) issued earlier
        self.teacher.eval()
        student_outs, teacher_outs = self.forward(inputs, teacher)
        loss, cal_res = self.loss_control(student_outs, teacher_outs, 'all')
        self.last_cal_res = cal_res
        return loss

    def _ensure_sync(self):
        """data-parallel plumbing of the student towers (lazy: torch.distributed may be initialised after __init__)"""
        self._sync = GradSync.current(self._sync).attach(self.towers())
        return self._sync

    def backward_and_sync(self, loss, defer_wait=False):
        """loss.backward() + the data-parallel gradient exchange (reference strategy ddp_find_unused_parameters_false,
        l_clip.yaml:56).  Sharded mode: every gradient bucket is reduce-scattered from inside its tower's backward (per-block
        release, reverse layer order) and FusedAdamW.step() updates the owned shards and all-gathers the parameters; fallback:
        each tower's flat buffer is all-reduced on a side stream right after its backward has been enqueued."""
        sync = self._ensure_sync()
        if loss is not None:                                   # None: the caller already ran loss.backward()
            sync.armed = sync.enabled                          # per-bucket release from inside the towers' backward: only here
            try:
                loss.backward()
            finally:
                sync.armed = False
        if not sync.enabled:
            return
        for tw in self.towers():
            if tw.dp is not None:
                tw.grads_ready = sync.finish(tw)
                tw._grad_clean = True                          # exchanged buckets were cleared behind their reduce-scatter
            else:
                tw.grads_ready = sync.launch(tw.flat_grad, after=tw.bwd_done)
        if not defer_wait:       # defer_wait: FusedAdamW.step waits per tower on `grads_ready` / runs on the exchange stream
            sync.wait()
        else:
            sync.forget()

    def _acc(self, log, img, txt, section, prefix, acc=True, score=False):
        m = retrieval_metrics(img, txt, self.k_list)
        if acc:
            for k in self.k_list:                                          # reference log_acc :220-224
                log[f'{section}/{prefix}_acc_top{k}'] = m[f'acc_top{k}']
        if score:                                                          # reference log_diag_score :204-212
            log[f'{section}/{prefix}_softmax_mean_score'] = m['softmax_mean_score']
            log[f'{section}/{prefix}_mean_score'] = m['mean_score']

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        """reference :129-147.  -> (gathered representations for validation_epoch_end, {log key: 0-dim tensor})"""
        student_outs, teacher_outs = self.forward(batch)
        loss, cal_res = self.loss_control(student_outs, teacher_outs, 'all')
        si, st = student_outs.visual_output.last_representation, student_outs.text_output.last_representation
        ti, tt = teacher_outs.visual_output.last_representation, teacher_outs.text_output.last_representation
        log = {'val_loss/loss': loss.detach()}
        log.update({f'val_loss/{k}': v for k, v in cal_res.items()})
        self._acc(log, si, st, 'val_step', 'stu', score=True)
        self._acc(log, ti, tt, 'val_step', 'tea')
        out = {'stu_image_outs': gather_rows(si), 'stu_text_outs': gather_rows(st),
               'tea_image_outs': gather_rows(ti), 'tea_text_outs': gather_rows(tt)}
        return out, log

    @torch.no_grad()
    def validation_epoch_end(self, outputs):
        """reference :152-187: metrics over the whole validation set (5000 COCO pairs for the shipped configs)"""
        cat = {k: torch.cat([o[k].reshape(-1, o[k].shape[-1]) for o in outputs], dim=0).float()
               for k in ('stu_image_outs', 'stu_text_outs', 'tea_image_outs', 'tea_text_outs')}
        log = {}
        self._acc(log, cat['stu_image_outs'], cat['stu_text_outs'], 'val_stu_acc', 'stu')
        self._acc(log, cat['stu_image_outs'], cat['tea_text_outs'], 'val_stu_image_tea_text', 'stu_image_tea_text')
        self._acc(log, cat['tea_image_outs'], cat['stu_text_outs'], 'val_stu_text_tea_image', 'stu_text_tea_image')
        self._acc(log, cat['stu_image_outs'], cat['stu_text_outs'], 'val_stu_score', 'stu', acc=False, score=True)
        if self.current_epoch == 0:
            self._acc(log, cat['tea_image_outs'], cat['tea_text_outs'], 'val_tea_score', 'tea', acc=False, score=True)
            self._acc(log, cat['tea_image_outs'], cat['tea_text_outs'], 'val_tea_acc', 'tea')
        return log

    def configure_optimizers(self):
        # reference :194-202
        dev = next(self.student.parameters()).device
        for tw in self.towers():
            tw.materialize(dev)
        extras = [p for enc in (self.student.image_encoder, self.student.text_encoder)
                  for p in getattr(enc, 'extra_parameters', lambda: [])()]     # plain CLIP encoders' projection linears
        opt = FusedAdamW(self.towers(), lr=self.hparams.lr, weight_decay=self.hparams.weight_decay, extra_params=extras)
        sched = EpochCosineSchedule(opt, self.hparams.warm_steps, self.hparams.total_steps)
        self._ensure_sync()          # data-parallel run: shard plan over the same trainable set the optimizer was built with
        return [opt], [sched]

    def on_train_epoch_start(self):
        if self.unfreeze_epoch and self.current_epoch >= self.unfreeze_epoch:
            self.unfreeze_embed()
            self.unfreeze_epoch = False

    def freeze_with_prefix(self, prefix_list):
        # reference :230-238
        if prefix_list is None:
            return
        for n, p in self.student.named_parameters():
            if any(n.startswith(prefix) for prefix in prefix_list):
                p.requires_grad = False

    def unfreeze_embed(self):
        for _, p in self.student.named_parameters():
            p.requires_grad = True

    def freeze_image_embedding(self):
        # reference :240-268: teacher patch / class / positional embeddings copied into the image student and frozen
        enc = self.student.image_encoder
        if isinstance(enc, ImageEncoder):                      # reference :258-266: same keys on both sides
            freeze_key = ['visual.conv1.weight', 'visual.class_embedding', 'visual.positional_embedding']
            sw, tw = enc.state_dict(), self.teacher.state_dict()
            for k in freeze_key:
                sw[k] = tw['image_encoder.' + k]
            enc.load_state_dict(sw)
            for n, p in enc.named_parameters():
                if n in freeze_key:
                    p.requires_grad = False
            return
        if not isinstance(enc, RepeatVisionTransformer):
            return                                             # (the reference does nothing for other student classes)
        keys = {'patch_embed.proj.weight': 'image_encoder.visual.conv1.weight',
                'cls_token': 'image_encoder.visual.class_embedding', 'pos_embed': 'image_encoder.visual.positional_embedding'}
        sw, tw = enc.state_dict(), self.teacher.state_dict()
        for s_k, t_k in keys.items():
            w = tw[t_k]
            if s_k == 'cls_token':
                w = w.unsqueeze(0).unsqueeze(0)
            if s_k == 'pos_embed':
                w = w.unsqueeze(0)
            sw[s_k] = w
        enc.load_state_dict(sw)
        for n, p in enc.named_parameters():
            if n in keys:
                p.requires_grad = False
