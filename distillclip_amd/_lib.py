"""ctypes binding of libdistillclip_hip.so (include/dclip.h).  Fails loudly: there is no CPU fallback.

Signatures are parsed from include/dclip.h at import so the binding cannot drift from the header.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libdistillclip_hip.so')
_HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'dclip.h')

_CT = {
    "int32_t": ctypes.c_int32,
    'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float, 'size_t': ctypes.c_size_t,
    'double': ctypes.c_double, 'dclip_bucket_cb': ctypes.c_void_p,
}


def _parse_header(path=_HEADER):
    """-> {name: (restype, [argtypes])} for every `dclip_*` prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    src = re.sub(r'^\s*#[^\n]*$', '', src, flags=re.M)
    src = src.replace('extern "C" {', '')
    src = re.sub(r'typedef\s+struct[^;{]*\{.*?\}[^;]*;', '', src, flags=re.S)
    src = re.sub(r'typedef[^;]*;', '', src)
    protos = {}
    for m in re.finditer(r'((?:const\s+)?(?:int|void|char|float|size_t|int64_t|int32_t|dclip_encoder)[\s\*]*?)\b(dclip_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        protos[name] = (_ctype(ret), [] if args in ('', 'void') else [_ctype(a) for a in args.split(',')])
    return protos


def _ctype(decl):
    decl = decl.strip()
    if '*' in decl:
        return ctypes.c_char_p if re.match(r'const\s+char\s*\*', decl) and decl.count('*') == 1 and \
            not re.search(r'\*\s*\w+$', decl) else ctypes.c_void_p
    toks = [t for t in re.split(r'\s+', decl) if t not in ('const', 'unsigned', 'struct')]
    return None if toks[0] == 'void' else _CT[toks[0]]


class DclipError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f'{_LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `make -C distillclip_amd/csrc`).  distillclip_amd has no CPU fallback.')
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so): it must be in the process BEFORE this library is mapped, or the
        # loader resolves our libamdhip64 dependency to /opt/rocm's copy and the process ends up with two runtimes — kernels of this
        # library then fail with "no ROCm-capable device is detected" (seen with `python __graft_entry__.py smoke`, which loaded the
        # library in build() before anything had imported torch)
        import torch                                # noqa: F401
        self._dll = ctypes.CDLL(_LIB_PATH)
        self.protos = _parse_header()
        for name, (res, args) in self.protos.items():
            fn = getattr(self._dll, name)       # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if self._dll.dclip_arch() != b'gfx950':
            raise ImportError('libdistillclip_hip.so was not built for gfx950')

    def __getattr__(self, name):
        fn = getattr(self._dll, name)
        if self.protos.get(name, (None,))[0] is not ctypes.c_int or name in ('dclip_version', 'dclip_encoder_num_grad_buckets') or name.endswith('_supported'):   # plain values, not status codes
            return fn

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                msg = self._dll.dclip_last_error_string().decode()
                if rc == -1:
                    raise ValueError(msg)
                raise DclipError(f'{name} failed ({rc}): {msg}')
        call.__name__ = name
        setattr(self, name, call)
        return call


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
