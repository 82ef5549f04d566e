"""bench.py's CPU leg (the `cpu_baseline` object of the JSON line) is the oracle, built from the same pinned YAML values as the HIP
model.  One oracle training step of every workload at B = 2: the factory resolves every name it needs (the leg only runs on the
GPU box otherwise, where a NameError would surface at the end of a bench run) and the loss is finite."""
import math
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.parametrize('name', ['lclip', 'image', 'text'])
def test_cpu_leg_runs_one_oracle_step(name):
    import torch
    import bench
    torch.set_num_threads(4)
    wl = dict(bench.WORKLOADS[name])
    step = bench.cpu_step_factory(wl, 2022)(2)      # (B = 1 has no in-batch negatives: cos_diff is NaN there, as in the reference)
    loss = step()
    assert math.isfinite(loss) and loss > 0.0


def test_every_workload_names_a_pinned_yaml():
    import bench
    for name, wl in bench.WORKLOADS.items():
        assert wl['yaml'] in bench.YAML_ARGS, name
        assert wl['kind'] in ('dual', 'image', 'text')
