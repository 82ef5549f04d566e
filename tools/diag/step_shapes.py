"""per-shape table of one single-stream step: every traced launch (GEMM M,N,K,variant / attention / LayerNorm) with its HIP-event
time, grouped by shape.  python tools/diag/step_shapes.py [config]"""
import os, sys, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from distillclip_amd._lib import lib
cfg = sys.argv[1] if len(sys.argv) > 1 else 'lclip'
wl = bench.WORKLOADS[cfg]
dev = torch.device('cuda')
model = bench.build_model(wl, 2022, dev)
(opt,), _ = model.configure_optimizers()
image, text, _ = bench.make_inputs(wl, 2022, wl['batch'])
batch = [image.cuda(), text.cuda()] if wl['kind'] == 'dual' else (image.cuda() if wl['kind'] == 'image' else text.cuda())
if hasattr(model, 'multi_stream'):
    model.multi_stream = False
def step():
    loss = model.training_step(batch); opt.zero_grad(); model.backward_and_sync(loss); opt.step(zero_grad=True)
for _ in range(3): step()
torch.cuda.synchronize()
cap = 20000
lib().dclip_trace_begin(cap)
NP = 3
for _ in range(NP): step()
torch.cuda.synchronize()
dims = (ctypes.c_int32 * (4 * cap))()
n = lib().dclip_trace_dims(ctypes.cast(dims, ctypes.c_void_p), cap)
kind = (ctypes.c_int32 * cap)(); ms = (ctypes.c_float * cap)(); fl = (ctypes.c_double * cap)(); by = (ctypes.c_double * cap)()
lib().dclip_trace_end(*(ctypes.cast(a, ctypes.c_void_p) for a in (kind, ms, fl, by)), cap)
agg = collections.OrderedDict()
for i in range(min(n, cap)):
    key = (kind[i], dims[4 * i], dims[4 * i + 1], dims[4 * i + 2], dims[4 * i + 3])
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += ms[i]; a[2] += fl[i]; a[3] += by[i]
names = {0: 'gemm_nt', 1: 'gemm_tn', 2: 'ln_fwd', 3: 'loss', 4: 'attn', 5: 'ln_bwd'}
tot = collections.defaultdict(float)
print(f'{"kind":8} {"d0":>7} {"d1":>6} {"d2":>6} {"var":>4} {"n/step":>6} {"us/call":>8} {"ms/step":>8} {"TF/s":>7} {"GB/s":>7}')
for (k, a, b, c, d), (cnt, t, f, y) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot[k] += t / NP
    print(f'{names.get(k, k):8} {a:7d} {b:6d} {c:6d} {d:4d} {cnt / NP:6.1f} {t / cnt * 1e3:8.1f} {t / NP:8.3f} {f / t / 1e9 if t else 0:7.1f} {y / t / 1e6 if t else 0:7.1f}')
print({names.get(k, k): round(v, 3) for k, v in tot.items()}, 'sum', round(sum(tot.values()), 3))
