"""CPU tier: the per-wave code of the head-mixed attention kernels (distillclip_amd/csrc/attn_mix_wave.h) on the 64-lane host
emulation (tools/emu/): lane / register index maps, LDS-DMA ring and wait placement, tail handling, against an f64 loop nest of the
reference arithmetic (model/component/weight_share_model.py:101-125).  Test infrastructure only: the product has no CPU path."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host_compiler():
    for c in ('/opt/rocm/lib/llvm/bin/clang++', shutil.which('clang++')):
        if c and os.path.exists(c):
            return c
    return None


@pytest.mark.skipif(_host_compiler() is None, reason='needs a host clang++ (ext_vector_type, __bf16)')
def test_attn_mix_wave_code_on_the_host_emulation(tmp_path):
    exe = str(tmp_path / 'emu_attn_mix')
    r = subprocess.run([_host_compiler(), '-O1', '-std=c++17', os.path.join(ROOT, 'tools', 'emu', 'emu_attn_mix.cpp'), '-o', exe, '-lpthread'],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0 and 'all cases ok' in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
