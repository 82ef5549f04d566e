"""CPU-only checks of the drop-in boundary: the shared library loads without a GPU, exports every symbol include/dclip.h
declares, rejects bad arguments before touching the device, and the product path fails loudly without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from distillclip_amd._lib import lib, _HEADER
    l = lib()
    declared = set(re.findall(r'\b(dclip_\w+)\s*\(', open(_HEADER).read()))
    declared -= {'dclip_encoder_cfg'}
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(l._dll, name), name
    assert l.dclip_version() == 4 and l.dclip_arch() == b'gfx950'


def test_loading_the_c_abi_before_torch_is_reported_in_words():
    """include/dclip.h "Load order": PyTorch-ROCm ships its own libamdhip64 and asks for it by another name than this library's DT_NEEDED
    entry, so a process that maps libdistillclip_hip.so FIRST and torch second holds two HIP runtimes; round 4's symptom was HIP's
    "no ROCm-capable device is detected" from every kernel of the second one.  dclip_runtime_check() (and every failed launch) now says
    what happened.  Child processes, no GPU needed: (a) the library alone: one runtime, and without a GPU the message names it and the
    missing device; (b) library first, torch second: the message names both runtimes and the fix; (c) torch first (what _lib.py does):
    one runtime."""
    import subprocess
    import sys
    so = os.path.join(ROOT, 'distillclip_amd', 'libdistillclip_hip.so')
    prog = (
        'import ctypes, sys\n'
        'order = sys.argv[1]\n'
        'if order == "torch_first": import torch\n'
        f'dll = ctypes.CDLL({so!r})\n'
        'if order == "lib_first": import torch\n'
        'dll.dclip_last_error_string.restype = ctypes.c_char_p\n'
        'rc = dll.dclip_runtime_check()\n'
        'n = sum(1 for p in {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})\n'
        'print(rc, n, dll.dclip_last_error_string().decode())\n')
    out = {}
    for order in ('lib_only', 'lib_first', 'torch_first'):
        r = subprocess.run([sys.executable, '-c', prog, order], capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        rc, n, msg = r.stdout.strip().split(' ', 2) if r.stdout.strip().count(' ') >= 2 else (r.stdout.strip().split(' ') + [''])[:3]
        out[order] = (int(rc), int(n), msg)
    assert out['lib_only'][1] == 1 and out['torch_first'][1] == 1, out
    rc, n, msg = out['lib_first']
    assert n == 2 and rc == -2, out
    assert 'HIP runtimes are mapped' in msg and 'import torch' in msg and 'BEFORE' in msg, msg
    if not torch.cuda.is_available():                  # no GPU in this container: the single-runtime cases report the missing device, by name
        for order in ('lib_only', 'torch_first'):
            rc, n, msg = out[order]
            assert rc == -2 and 'sees no usable device' in msg and 'libamdhip64' in msg, out


def test_argument_validation_happens_on_host():
    from distillclip_amd._lib import lib
    l = lib()
    with pytest.raises(ValueError, match='multiple of 64'):
        l.dclip_gemm_nt(16, 8, 16, 8, 16, 8, 4, 8, 48, 1.0, None, 0, None, None, None, 0, 0, 0, None, None, None)
    with pytest.raises(ValueError, match='null'):
        l.dclip_gemm_nt(None, 8, 16, 8, 16, 8, 4, 8, 64, 1.0, None, 0, None, None, None, 0, 0, 0, None, None, None)
    with pytest.raises(ValueError):
        l.dclip_layernorm_fwd(16, 4, None, 16, 16, 16, 4, 0, None, None, 4, 2048, 1e-5, None)     # D > 1024
    with pytest.raises(ValueError, match='head dim'):
        l.dclip_attn_nt(16, 8, 16, 8, 16, 1, 1, 1, 8, 8, 48, 1.0, None)
    import ctypes
    one = lambda a: (ctypes.c_void_p * 1)(a)
    with pytest.raises(ValueError, match='1..24'):
        l.dclip_adamw_multi(one(16), one(16), one(16), one(16), (ctypes.c_int64 * 1)(64), 25, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 0, None)
    with pytest.raises(ValueError, match='multiple of 4'):
        l.dclip_adamw_multi(one(16), one(16), one(16), one(16), (ctypes.c_int64 * 1)(66), 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 0, None)
    with pytest.raises(ValueError, match='16-byte aligned'):
        l.dclip_adamw_multi(one(16), one(24), one(16), one(16), (ctypes.c_int64 * 1)(64), 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 0, None)
    assert l.dclip_distill_loss_workspace(512, 512) > 6 * 512 * 512 * 4


def test_encoder_plan_is_host_side():
    """dclip_encoder_create / workspace sizing never touch the device: usable here to size buffers."""
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    from distillclip_amd._lib import lib
    v = RepeatVisionTransformer(img_size=224, patch_size=32, out_dim=512, embed_dim=768, depth=6, num_heads=24, qkv_bias=True,
                                repeated_times=2, use_transform=True)
    t = RepeatTextTransformer(depth=4, repeated_times=2, use_transform=True, compression_embedding=True)
    for m, n_expected in ((v, 4 + 3 * (8 + 2 * 6) + 4), (t, 4 + 2 * (8 + 2 * 6) + 4)):
        h = m._tower._handle
        assert lib().dclip_encoder_num_params(h) == n_expected
        infer = lib().dclip_encoder_workspace_bytes(h, 512, 0)
        train = lib().dclip_encoder_workspace_bytes(h, 512, 1)
        assert 0 < infer < train < 40 * 2 ** 30
        assert lib().dclip_encoder_wcache_bytes(h) > 0
    with pytest.raises(ValueError, match='head dim'):
        RepeatVisionTransformer(img_size=224, patch_size=32, out_dim=512, embed_dim=768, depth=2, num_heads=8,
                                repeated_times=2, use_transform=True)


def test_state_dict_keys_match_reference_layout():
    from distillclip_amd import synth
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer, ImageEncoder, TextEncoder
    v = RepeatVisionTransformer(img_size=224, patch_size=32, out_dim=512, embed_dim=768, depth=6, num_heads=24, qkv_bias=True,
                                repeated_times=2, use_transform=True)
    assert set(v.state_dict()) == set(synth.student_image_state(0)) and len(v.state_dict()) == 68
    t = RepeatTextTransformer(depth=4, repeated_times=2, use_transform=True)
    assert set(t.state_dict()) == set(synth.student_text_state(0)) and len(t.state_dict()) == 44
    ti = ImageEncoder(False, dict(input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512))
    assert set(ti.state_dict()) == set(synth.teacher_image_state(0, layers=12))
    tt = TextEncoder(512, 12, 8, 77, None, 49408, 512, is_student=False)
    assert set(tt.state_dict()) == set(synth.teacher_text_state(0))


def test_no_cpu_fallback():
    from distillclip_amd import ops
    from distillclip_amd.model.component import RepeatTextTransformer
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.gemm_nt(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16))
    t = RepeatTextTransformer(vocab_size=97, context_length=13, out_dim=64, embed_dim=128, depth=2, num_heads=2,
                              repeated_times=2, use_transform=True)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        t(torch.zeros(2, 13, dtype=torch.long))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'distillclip_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle\b', src, flags=re.M), os.path.join(dirpath, f)
