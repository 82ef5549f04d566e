"""The reference's shipped configurations bind to the mirror classes unchanged: tests/golden/yaml_init_args.json holds the
class_path / init_args of the `model:` section of config/final_config/{l_clip,image,text}.yaml (extracted by
tools/golden/gen_yaml_init_args.py in the build container; data, not the YAML text) and the models are instantiated from exactly
those keyword arguments, the way jsonargparse does for `python main.py fit --conf <yaml>` (reference l_clip.yaml:4-39,
image.yaml:5-35, text.yaml:6-21)."""
import json
import os

import numpy as np
import pytest
import torch

from distillclip_amd import synth
from distillclip_amd.model.from_config import instantiate, resolve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'yaml_init_args.json')))


def T(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


@pytest.fixture(scope='module')
def teacher_sd():
    tsd = synth.teacher_image_state(5)
    tsd.update(synth.teacher_text_state(5))
    return T(tsd)


def test_class_paths_resolve_to_the_mirror_package():
    from distillclip_amd.model import DistillModel, DualDistillModel
    from distillclip_amd.model.component.weight_share_model import RepeatTextTransformer, RepeatVisionTransformer
    assert resolve(CFG['l_clip']['model']['class_path']) is DualDistillModel
    assert resolve(CFG['image']['model']['class_path']) is DistillModel
    assert resolve(CFG['l_clip']['model']['init_args']['image_student']['class_path']) is RepeatVisionTransformer
    assert resolve(CFG['text']['model']['init_args']['student_encoder']['class_path']) is RepeatTextTransformer


def test_l_clip_yaml_binds(teacher_sd):
    spec = CFG['l_clip']['model']
    # the stage-1 checkpoints the YAML names are placeholders: with them the constructor fails in torch.load like the reference
    # (dual_distill_model.py:28) ...
    with pytest.raises(FileNotFoundError):
        instantiate(spec, teacher_state_dict=teacher_sd)
    # ... without them every other keyword argument binds as written
    m = instantiate(spec, overrides={'load_path': None}, teacher_state_dict=teacher_sd)
    ia = spec['init_args']
    assert (m.hparams['lr'], m.hparams['weight_decay'], m.hparams['warm_steps'], m.hparams['total_steps']) == (
        ia['lr'], ia['weight_decay'], ia['warm_steps'], ia['total_steps'])
    assert m.loss_control.loss_name == ia['loss_control_para']['loss_name']
    sd = m.state_dict()
    stu_i = {k[len('student.image_encoder.'):] for k in sd if k.startswith('student.image_encoder.')}
    stu_t = {k[len('student.text_encoder.'):] for k in sd if k.startswith('student.text_encoder.')}
    # key sets = the reference's (68 / 44 tensors), i.e. what synth generates by the reference's names
    assert stu_i == set(synth.student_image_state(1, **ia['image_student']['init_args'])) and len(stu_i) == 68
    assert stu_t == set(synth.student_text_state(1, **ia['text_student']['init_args'])) and len(stu_t) == 44
    assert any(k.startswith('teacher.image_encoder.') for k in sd) and any(k.startswith('teacher.text_encoder.') for k in sd)
    assert not any(p.requires_grad for p in m.teacher.parameters())


def test_image_yaml_binds(teacher_sd):
    spec = CFG['image']['model']
    m = instantiate(spec, teacher_state_dict=teacher_sd)
    ia = spec['init_args']
    assert m.hparams['model_type'] == 'image' and m.hparams['freeze_embed'] is True
    keys = {k[len('student.'):] for k in m.state_dict() if k.startswith('student.')}
    assert keys == set(synth.student_image_state(1, **{k: v for k, v in ia['student_encoder']['init_args'].items()
                                                       if k in ('img_size', 'patch_size', 'in_chans', 'out_dim', 'embed_dim', 'depth',
                                                                'num_heads', 'mlp_ratio', 'qkv_bias', 'repeated_times', 'use_transform')}))
    # freeze_embed: the patch embedding / class token / positional embedding of the student are copied from the teacher and frozen
    frozen = [n for n, p in m.student.named_parameters() if not p.requires_grad]
    assert frozen and all(('patch_embed' in n) or ('cls_token' in n) or ('pos_embed' in n) for n in frozen)
    assert CFG['image']['train_batch_size'] == 1024          # image.yaml:46 (BASELINE.json quotes the configuration at 256)


def test_text_yaml_binds(teacher_sd):
    spec = CFG['text']['model']
    m = instantiate(spec, teacher_state_dict=teacher_sd)
    assert m.hparams['model_type'] == 'text'
    keys = {k[len('student.'):] for k in m.state_dict() if k.startswith('student.')}
    assert keys == set(synth.student_text_state(1, **spec['init_args']['student_encoder']['init_args']))
    assert 'patch_embed.0.weight' in keys and 'patch_embed.1.weight' in keys        # compression_embedding: True (text.yaml:10)
