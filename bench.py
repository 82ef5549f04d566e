#!/usr/bin/env python3
"""bench.py — image-text pairs/sec of one full distill step on N x MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[3], the config the pairs/sec metric is quoted on): l_clip.yaml dual distillation,
224 px / 77 tokens, B = 512 pairs per GPU, frozen ViT-B/32 CLIP teacher (image + text towers), weight-shared students
(RepeatVisionTransformer depth 6 / 24 heads / R=2, RepeatTextTransformer depth 4 / 12 heads / R=2), losses
out_l1 + out_cos + 0.1 * cos_diff, fused AdamW.  One step = student fwd + teacher fwd + fused loss + student bwd +
data-parallel gradient average (RCCL, N > 1) + optimizer step, on synthetic inputs already resident in HBM.
Weak scaling: per-GPU batch fixed, local negatives like the reference's training_step.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# one hardware queue per tower stream (+ main + RCCL): read at HIP initialisation, see distillclip_amd/__init__.py
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np   # noqa: E402
import torch         # noqa: E402

STEP_GFLOP_PER_PAIR = 42.17      # teacher fwd 14.86 + student fwd 9.10 + student bwd 18.20 GFLOP
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0

S_IMG = dict(img_size=224, patch_size=32, in_chans=3, out_dim=512, embed_dim=768, depth=6, num_heads=24, mlp_ratio=4.0,
             qkv_bias=True, repeated_times=2, use_transform=True)            # l_clip.yaml:4-17
S_TXT = dict(depth=4, repeated_times=2, use_transform=True)                  # l_clip.yaml:18-23
LOSS = dict(loss_name=['out_l1', 'out_cos', 'cos_diff'], loss_scale={'cos_diff': 0.1})     # l_clip.yaml:29-32


def T(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


def build_model(seed, device):
    from distillclip_amd import synth
    from distillclip_amd.model import DualDistillModel
    from distillclip_amd.model.component import RepeatVisionTransformer, RepeatTextTransformer
    s_img, s_txt = RepeatVisionTransformer(**S_IMG), RepeatTextTransformer(**S_TXT)
    s_img.load_state_dict(T(synth.student_image_state(seed, **S_IMG)))
    s_txt.load_state_dict(T(synth.student_text_state(seed, **S_TXT)))
    tsd = synth.teacher_image_state(seed)
    tsd.update(synth.teacher_text_state(seed))
    model = DualDistillModel(s_img, s_txt, LOSS, warm_steps=15, total_steps=300, weight_decay=1e-3, lr=1e-4,
                             download_root='./.cache', teacher_state_dict=T(tsd))       # l_clip.yaml:35-39
    return model.to(device)


def cpu_baseline(seed, target_seconds=20.0):
    """The oracle (CPU fp32 restatement of the reference) timed on this host: same model, same step contents
    (fwd, loss, backward, AdamW), bounded sample."""
    import oracle
    from distillclip_amd import synth
    nthreads = torch.get_num_threads()
    B = 8
    sd_i = {k: v.requires_grad_(True) for k, v in T(synth.student_image_state(seed, **S_IMG)).items()}
    sd_t = {k: v.requires_grad_(True) for k, v in T(synth.student_text_state(seed, **S_TXT)).items()}
    t_i, t_t = T(synth.teacher_image_state(seed)), T(synth.teacher_text_state(seed))
    image = torch.from_numpy(synth.images(seed, B))
    text = torch.from_numpy(synth.captions(seed, B))
    opt = torch.optim.AdamW(list(sd_i.values()) + list(sd_t.values()), lr=1e-4, weight_decay=1e-3)
    lc = oracle.LossOracle(LOSS['loss_name'], LOSS['loss_scale'])

    def step():
        so = oracle.clip_forward(oracle.student_image_forward(sd_i, image, 24), oracle.student_text_forward(sd_t, text, 12))
        with torch.no_grad():
            to = oracle.clip_forward(oracle.teacher_image_forward(t_i, image), oracle.teacher_text_forward(t_t, text))
        loss, _ = lc(so, to, 'all')
        opt.zero_grad()
        loss.backward()
        opt.step()
    step()                                  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or n >= 16:
            break
    return {'value': round(n * B / dt, 3), 'unit': 'pairs/s', 'cores': nthreads, 'kind': 'port',
            'sample': f'{n} steps of the same l_clip dual step at batch {B} (fp32, torch CPU, {nthreads} threads)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=512, help='pairs per GPU (BASELINE.json configs[3])')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--global-negatives', action='store_true',
                    help='opt-in north-star mode: in-batch negatives over all ranks (all-gather of the embeddings over RCCL)')
    ap.add_argument('--teacher-text-prefix', action='store_true',
                    help='opt-in: run the causal text teacher only on the prefix that contains every EOT (identical output, '
                         'less work; NOT used for the headline number, which processes all 77 positions)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    use_dist = world > 1 or os.environ.get('DCLIP_FORCE_DIST') == '1'      # the latter: exercise the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    from distillclip_amd import synth
    from distillclip_amd._lib import lib
    seed = 2022                                             # main.py:24 seed_everything_default
    model = build_model(seed, device)
    (opt,), _ = model.configure_optimizers()
    model.loss_control.global_negatives = args.global_negatives
    if use_dist:
        from distillclip_amd.parallel import GradSync
        model._sync = GradSync()
        model._sync.enabled = True          # also at world size 1 (DCLIP_FORCE_DIST): the collectives still run
    B = args.batch
    image = torch.from_numpy(synth.images(seed + rank, B)).to(device)      # per-rank shard, resident in HBM
    caps = synth.captions(seed + rank, B)
    text = torch.from_numpy(caps).to(device)
    # the tokenizer knows caption lengths on the host: the causal teacher text tower only needs the prefix holding every EOT
    tt_tokens = int((caps != 0).sum(1).max()) if args.teacher_text_prefix else 77
    model.set_text_length_hint(tt_tokens if args.teacher_text_prefix else None)

    FUSED_ZERO = os.environ.get('DCLIP_BENCH_FUSED_ZERO', '1') != '0'
    OVERLAP_OPT = os.environ.get('DCLIP_BENCH_OVERLAP_OPT', '1') != '0'   # per-tower optimizer step on the tower's own stream

    PIPE_TEACHER = os.environ.get('DCLIP_BENCH_PIPELINE_TEACHER', '0') == '1'
    pending = {'teacher': None}

    def step():
        if PIPE_TEACHER:
            # the frozen teacher of the NEXT batch runs under this batch's student backward (same work per step, same values)
            handle = pending['teacher'] or model.teacher_forward_async([image, text])
            loss = model.training_step([image, text], teacher=handle)
            opt.zero_grad()
            loss.backward()
            pending['teacher'] = model.teacher_forward_async([image, text])
            model.backward_and_sync(None, defer_wait=OVERLAP_OPT)
        else:
            loss = model.training_step([image, text])
            opt.zero_grad()
            model.backward_and_sync(loss, defer_wait=OVERLAP_OPT)
        opt.step(zero_grad=FUSED_ZERO, overlap=OVERLAP_OPT, join=not OVERLAP_OPT)   # the fused kernel clears each gradient element as it consumes it: the next zero_grad() is free
        return loss

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()

    roofline = None
    if not args.no_roofline:
        # dominant kernel = gemm_nt (forward linears + dgrad): HIP events around every launch of 2 further steps, recorded on the
        # launch stream by the library's trace hooks (include/dclip.h: dclip_trace_*).  EVERY rank runs these steps (they contain
        # the gradient collectives); only rank 0 traces.
        import ctypes
        cap = 20000
        nprobe = 2
        multi, model.multi_stream = model.multi_stream, False      # one stream: event intervals then bracket one kernel each
        step()
        torch.cuda.synchronize()
        if rank == 0:
            lib().dclip_trace_begin(cap)
        for _ in range(nprobe):
            step()
        torch.cuda.synchronize()
        model.multi_stream = multi
    if rank == 0 and not args.no_roofline:
        kind = (ctypes.c_int32 * cap)()
        ms = (ctypes.c_float * cap)()
        fl = (ctypes.c_double * cap)()
        by = (ctypes.c_double * cap)()
        n = lib().dclip_trace_end(ctypes.cast(kind, ctypes.c_void_p), ctypes.cast(ms, ctypes.c_void_p),
                                  ctypes.cast(fl, ctypes.c_void_p), ctypes.cast(by, ctypes.c_void_p), cap)
        n = min(n, cap)
        agg = {}
        for i in range(n):
            a = agg.setdefault(kind[i], [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += ms[i]; a[2] += fl[i]; a[3] += by[i]
        names = {0: 'gemm_nt_kernel', 1: 'gemm_tn_kernel', 2: 'ln_fwd_kernel', 3: 'distill_loss'}
        g = agg.get(0, [1, 1.0, 0.0, 0.0])
        achieved = g[2] / (g[1] * 1e-3) / 1e12
        traffic = None                                          # HBM bytes / launch from the committed PMC passes of this command
        import glob
        tj = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_traffic.json')))
        if tj:
            traffic = json.load(open(tj[-1])).get('gemm_nt', {}).get('hbm_bytes_per_launch')
        roofline = {'kernel': 'gemm_nt (gemm_nt_kernel 128x128 + gemm_nt256_kernel 256x256)', 'bound': 'mfma',
                    'achieved': round(achieved, 2), 'peak': PEAK_BF16_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_BF16_TFLOPS, 4), 'traffic': traffic,
                    'algorithmic_bytes_per_launch': g[3] / g[0],
                    'launches_per_step': g[0] // nprobe, 'avg_launch_us': round(g[1] / g[0] * 1e3, 2),
                    'flop_per_launch': g[2] / g[0],
                    'others': {names.get(k, str(k)): {'launches_per_step': v[0] // nprobe, 'ms_per_step': round(v[1] / nprobe, 3),
                                                      'TFLOP/s': round(v[2] / (v[1] * 1e-3) / 1e12, 2) if v[2] else None,
                                                      'GB/s': round(v[3] / (v[1] * 1e-3) / 1e9, 1)} for k, v in agg.items()}}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(seed)

    if rank == 0:
        pairs = B * world * args.steps
        value = pairs / dt
        out = {
            'metric': 'image-text pairs/sec (distill step)', 'value': round(value, 2), 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'l_clip.yaml dual distill: ViT-B/32 CLIP teacher -> weight-shared ViT(6x768,24h,R2) + '
                                   'text(4x768,12h,R2) students, 224px/77tok, losses out_l1+out_cos+0.1*cos_diff, fwd+loss+bwd+AdamW',
                       'global_batch': B * world, 'batch_per_gpu': B, 'parallelism': f'dp{world}',
                       'negatives': 'global (all-gather over RCCL)' if args.global_negatives else 'local (reference training_step semantics)', 'optimizer_in_step': True,
                       'teacher_text_tokens_processed': tt_tokens},
            'step_gflop_per_pair': STEP_GFLOP_PER_PAIR,
            'mfma_roofline_frac_whole_step': round(value / world * STEP_GFLOP_PER_PAIR * 1e9 / (PEAK_BF16_TFLOPS * 1e12), 4),
            'final_loss': round(final_loss, 6),
        }
        if roofline is not None:
            out['roofline'] = roofline
        if cpu is not None:
            out['cpu_baseline'] = cpu
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
