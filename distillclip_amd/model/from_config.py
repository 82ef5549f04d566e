"""Instantiate the mirror classes from the reference's YAML `model:` section the way its LightningCLI (jsonargparse) does:
`{'class_path': .., 'init_args': {..}}` dictionaries, nested ones (the student encoders) instantiated first and passed as keyword
arguments (reference main.py + config/final_config/*.yaml: `python main.py fit --conf l_clip.yaml`).

class_path resolution follows the reference's import layout: a bare name (`DualDistillModel`, `DistillModel`) is looked up in the
`model` package, a dotted one (`model.component.weight_share_model.RepeatVisionTransformer`) is a module path under it -- here the
`model` package is `distillclip_amd.model`, which mirrors those module paths one to one."""
import importlib


def resolve(class_path):
    if '.' not in class_path:
        mod = importlib.import_module('distillclip_amd.model')
        return getattr(mod, class_path)
    module, name = class_path.rsplit('.', 1)
    if module == 'model' or module.startswith('model.'):
        module = 'distillclip_amd.' + module
    return getattr(importlib.import_module(module), name)


def instantiate(spec, overrides=None, **extra):
    """spec: {'class_path', 'init_args'}.  overrides: init_args to replace at the top level (e.g. load_path=None when the stage-1
    checkpoints the YAML names do not exist); extra: keyword arguments the reference does not have (teacher_state_dict)."""
    kwargs = {}
    for k, v in dict(spec.get('init_args') or {}).items():
        kwargs[k] = instantiate(v) if isinstance(v, dict) and 'class_path' in v else v
    kwargs.update(overrides or {})
    kwargs.update(extra)
    return resolve(spec['class_path'])(**kwargs)
