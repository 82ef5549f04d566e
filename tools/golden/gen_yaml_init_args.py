"""Build-container only: extract the `model:` section (class_path / init_args values, recursively) of the reference's three shipped
configurations into tests/golden/yaml_init_args.json.  The JSON is DATA -- the constructor keyword arguments jsonargparse would pass,
nothing else of the YAML files (no comments, no data / trainer sections, no text).  tests/test_yaml_binding_cpu.py instantiates the
mirror classes from exactly these kwargs, and bench.py reads its workload kwargs from the same file.

    python tools/golden/gen_yaml_init_args.py            (reads /root/reference/config/final_config/*.yaml)
"""
import json
import os
import sys

import yaml

REF = os.environ.get('DCLIP_REFERENCE', '/root/reference')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tests', 'golden', 'yaml_init_args.json')


def main():
    out = {}
    for name in ('l_clip', 'image', 'text'):
        path = os.path.join(REF, 'config', 'final_config', name + '.yaml')
        cfg = yaml.safe_load(open(path))
        model = cfg['model']
        assert set(model) == {'class_path', 'init_args'}, sorted(model)
        out[name] = {'source': f'config/final_config/{name}.yaml', 'model': model,
                     # the two scalars of the other sections that the benchmark configurations quote (BASELINE.json)
                     'train_batch_size': cfg.get('data', {}).get('init_args', {}).get('train_batch_size'),
                     'strategy': cfg.get('trainer', {}).get('strategy')}
    with open(OUT, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write('\n')
    print('wrote', OUT, file=sys.stderr)


if __name__ == '__main__':
    main()
