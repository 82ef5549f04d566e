#!/usr/bin/env python3
"""bench.py — distill-step throughput on N x MI355X (BASELINE.json metric: image-text pairs/sec of one full distill step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config lclip|image|text|lclip336]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts N fresh rank processes (one per GPU, RCCL over
xGMI) BEFORE it touches the GPU itself, relays rank 0's JSON line and exits non-zero if any rank fails.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the ranks are taken from the environment instead.

Default workload (BASELINE.json configs[3], the config the pairs/sec metric is quoted on): l_clip.yaml dual distillation,
224 px / 77 tokens, B = 512 pairs per GPU, frozen ViT-B/32 CLIP teacher (image + text towers), weight-shared students
(RepeatVisionTransformer depth 6 / 24 heads / R=2, RepeatTextTransformer depth 4 / 12 heads / R=2), losses
out_l1 + out_cos + 0.1 * cos_diff, fused AdamW.  One step = student fwd + teacher fwd + fused loss + student bwd +
data-parallel exchange (N > 1: bucketed reduce-scatter under the backward -> sharded AdamW -> parameter all-gather) +
optimizer step, on synthetic inputs already resident in HBM.  Weak scaling: per-GPU batch fixed, local negatives like the
reference's training_step.  `--config` selects the other BASELINE configurations (image.yaml B=256,
text.yaml B=1024, l_clip at 336 px B=512), each with its own unit and FLOP count.
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

PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0

# model keyword arguments: the `model:` section of the reference's shipped YAMLs, extracted (values only) into
# tests/golden/yaml_init_args.json by tools/golden/gen_yaml_init_args.py and bound through distillclip_amd.model.from_config,
# the way `python main.py fit --conf <yaml>` binds them (l_clip.yaml:4-39, image.yaml:5-35, text.yaml:6-21)
YAML_ARGS = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'yaml_init_args.json')))

# step FLOP per unit: SURVEY.md §8d (teacher fwd + student fwd + student bwd = 2 x fwd)
WORKLOADS = {
    'lclip': dict(kind='dual', yaml='l_clip', res=224, batch=512, unit='pairs', gflop=42.17,
                  metric='image-text pairs/sec (distill step)',
                  desc='l_clip.yaml dual distill: ViT-B/32 CLIP teacher -> weight-shared ViT(6x768,24h,R2) + text(4x768,12h,R2) '
                       'students, 224px/77tok, losses out_l1+out_cos+0.1*cos_diff, fwd+loss+bwd+AdamW'),
    'lclip336': dict(kind='dual', yaml='l_clip', res=336, batch=512, unit='pairs', gflop=65.98,
                     metric='image-text pairs/sec (distill step, 336 px)',
                     desc='l_clip.yaml dual distill at 336x336 (101 image tokens; BASELINE configs[4] per-GPU share): same towers, '
                          'losses out_l1+out_cos+0.1*cos_diff, fwd+loss+bwd+AdamW'),
    'image': dict(kind='image', yaml='image', res=224, batch=256, unit='images', gflop=22.65,
                  metric='images/sec (image.yaml distill step)',
                  desc='image.yaml: ViT-B/32 image teacher -> weight-shared ViT(6x768,24h,R2) student, freeze_embed, losses '
                       'out_l1+out_cos, fwd+loss+bwd+AdamW'),
    'text': dict(kind='text', yaml='text', res=224, batch=1024, unit='captions', gflop=19.61,
                 metric='captions/sec (text.yaml distill step)',
                 desc='text.yaml: CLIP text teacher (12x512, causal) -> weight-shared text(4x768,12h,R2, compressed embedding) '
                      'student, 77 tok, losses out_l1+out_cos, fwd+loss+bwd+AdamW'),
}


def T(d):
    import numpy as np
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


def build_model(wl, seed, device):
    import copy
    from distillclip_amd import synth
    from distillclip_amd.model.from_config import instantiate
    tsd = synth.teacher_image_state(seed, resolution=wl['res'])
    tsd.update(synth.teacher_text_state(seed))
    tsd = T(tsd)
    spec = copy.deepcopy(YAML_ARGS[wl['yaml']]['model'])
    ia = spec['init_args']
    if wl['kind'] == 'dual':
        ia['image_student']['init_args']['img_size'] = wl['res']          # configs[4]: the same YAML at 336 px
        # load_path names the stage-1 checkpoints of a real run; the benchmark uses seeded synthetic weights instead
        model = instantiate(spec, overrides={'load_path': None}, teacher_state_dict=tsd)
        model.student.image_encoder.load_state_dict(T(synth.student_image_state(seed, **ia['image_student']['init_args'])))
        model.student.text_encoder.load_state_dict(T(synth.student_text_state(seed, **ia['text_student']['init_args'])))
    else:
        model = instantiate(spec, teacher_state_dict=tsd)
        enc_args = {k: v for k, v in ia['student_encoder']['init_args'].items() if v is not None}
        gen = synth.student_image_state if wl['kind'] == 'image' else synth.student_text_state
        sd = T(gen(seed, **enc_args))
        if wl['kind'] == 'image':
            # freeze_embed copied the teacher's patch embedding into the student at construction (distil_model.py:200-213): keep it
            frozen = {n for n, p in model.student.named_parameters() if not p.requires_grad}
            sd = {k: v for k, v in sd.items() if k not in frozen}
            model.student.load_state_dict(sd, strict=False)
        else:
            model.student.load_state_dict(sd)
    return model.to(device)


def make_inputs(wl, seed, B):
    import torch
    from distillclip_amd import synth
    image = torch.from_numpy(synth.images(seed, B, wl['res'])) if wl['kind'] != 'text' else None
    caps = synth.captions(seed, B) if wl['kind'] != 'image' else None
    text = torch.from_numpy(caps) if caps is not None else None
    return image, text, caps


def log(msg):
    """progress on stderr (stdout carries only the JSON line)"""
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def effective_cpus():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota (a GPU box hands each job a share of
    the host: os.cpu_count() reports the whole machine and oversubscribing it makes the torch CPU path slower, not faster)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_step_factory(wl, seed):
    """-> make_step(B): one oracle training step (fwd, loss, zero_grad, backward, AdamW) of the workload on the CPU, built from
    the same YAML values as the HIP model (tests/test_bench_cpu.py runs one step of every workload at B = 2)"""
    import torch
    import oracle
    from distillclip_amd import synth
    kind = wl['kind']
    res = wl['res']
    # the same YAML values the HIP model is built from (tests/golden/yaml_init_args.json)
    ia = YAML_ARGS[wl['yaml']]['model']['init_args']
    enc = lambda key: {k: v for k, v in ia[key]['init_args'].items() if v is not None}
    if kind == 'dual':
        cfg_i, tcfg = dict(enc('image_student'), img_size=res), enc('text_student')
    else:
        cfg_i = dict(enc('student_encoder'), img_size=res) if kind == 'image' else None
        tcfg = enc('student_encoder') if kind == 'text' else None
    heads_i = (cfg_i or {}).get('num_heads', 24)
    heads_t = (tcfg or {}).get('num_heads', 12)
    loss_para = dict(ia['loss_control_para'])
    sd_i = sd_t = None
    params = []
    tsd_i = T(synth.teacher_image_state(seed, resolution=res)) if kind != 'text' else None
    tsd_t = T(synth.teacher_text_state(seed)) if kind != 'image' else None
    if kind != 'text':
        sd_i = {k: v.requires_grad_(True) for k, v in T(synth.student_image_state(seed, **cfg_i)).items()}
        if kind == 'image':      # freeze_embed (distil_model.py:197-213): teacher patch / class / positional embeddings, frozen
            sd_i['patch_embed.proj.weight'] = tsd_i['visual.conv1.weight'].clone()
            sd_i['cls_token'] = tsd_i['visual.class_embedding'].view(1, 1, -1).clone()
            sd_i['pos_embed'] = tsd_i['visual.positional_embedding'].unsqueeze(0).clone()
        params += [v for v in sd_i.values() if v.requires_grad]
    if kind != 'image':
        sd_t = {k: v.requires_grad_(True) for k, v in T(synth.student_text_state(seed, **tcfg)).items()}
        params += list(sd_t.values())
    lr, wd = ia['lr'], ia['weight_decay']
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=wd)
    lc = oracle.LossOracle(**loss_para)

    def make_step(B):
        image, text, _ = make_inputs(wl, seed, B)

        def step():
            if kind == 'dual':
                so = oracle.clip_forward(oracle.student_image_forward(sd_i, image, heads_i), oracle.student_text_forward(sd_t, text, heads_t))
                with torch.no_grad():
                    to = oracle.clip_forward(oracle.teacher_image_forward(tsd_i, image), oracle.teacher_text_forward(tsd_t, text))
                loss, _ = lc(so, to, 'all')
            elif kind == 'image':
                so = oracle.student_image_forward(sd_i, image, heads_i)
                with torch.no_grad():
                    to = oracle.teacher_image_forward(tsd_i, image)
                loss, _ = lc(so, to, 'image')
            else:
                so = oracle.student_text_forward(sd_t, text, heads_t)
                with torch.no_grad():
                    to = oracle.teacher_text_forward(tsd_t, text)
                loss, _ = lc(so, to, 'text')
            opt.zero_grad()
            loss.backward()
            opt.step()
            return float(loss.detach())
        return step

    return make_step


def cpu_baseline(wl, seed, budget_s=40.0):
    """The oracle (CPU fp32 restatement of the reference, pinned by reference-run goldens) timed on this host: same model, same
    step contents (fwd, loss, zero_grad, backward, AdamW).  BASELINE.md §2: B in {4, 32}, thread sweep, 1 warm-up then best of
    >= 3 steps; bounded to ~budget_s of CPU work (configurations are tried most-promising first and the sweep stops when the
    budget is spent).  Reports the best (batch, threads)."""
    import torch
    ncpu = effective_cpus()
    make_step = cpu_step_factory(wl, seed)
    threads = sorted({max(1, ncpu // 4), max(1, ncpu // 2), ncpu}, reverse=True)
    plan = [(32, t) for t in threads] + [(4, t) for t in reversed(threads)]
    t_start = time.perf_counter()
    tried, best = [], None
    for B, nt in plan:
        if tried and time.perf_counter() - t_start > budget_s:
            break
        torch.set_num_threads(nt)
        step = make_step(B)
        log(f'cpu baseline: batch {B}, {nt} threads ...')
        t0 = time.perf_counter()
        step()                                      # warm-up
        t_warm = time.perf_counter() - t0
        times = []
        for _ in range(3):
            if times and time.perf_counter() - t_start + t_warm > 2.0 * budget_s:
                break                               # slow host: fewer than 3 timed steps, said so in `sample`
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
        rate = B / min(times)
        tried.append({'batch': B, 'threads': nt, 'best_step_s': round(min(times), 4), 'steps_timed': len(times),
                      f'{wl["unit"]}_per_s': round(rate, 3)})
        if best is None or rate > best[0]:
            best = (rate, B, nt, len(times))
    torch.set_num_threads(ncpu)
    rate, B, nt, n = best
    return {'value': round(rate, 3), 'unit': f'{wl["unit"]}/s', 'cores': nt, 'kind': 'port', 'cpu_model': cpu_model_name(),
            'usable_cpus': ncpu, 'host_logical_cpus': os.cpu_count(),
            'sample': f'best of {n} timed steps (after 1 warm-up) of the same {wl["kind"]} distill step at batch {B}, fp32 torch CPU, '
                      f'{nt} threads; sweep over batch {{32, 4}} x threads {threads} bounded to ~{int(budget_s)} s',
            'sweep': tried}


# ---------------------------------------------------------------------------------------------------------------------
# launcher: N fresh rank processes, started before this process makes any GPU call
# ---------------------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv, timeout_s=1800.0, grace_s=120.0):
    """Start n fresh rank processes (before this process makes any GPU call) and wait for them.

    Rendezvous is a torch.distributed FileStore in a private temporary directory (DCLIP_RDZV_FILE): there is no port to pick, so
    no bind-then-close race.  Deadlines: every rank must finish within timeout_s, and once the first rank has exited the others get
    grace_s to follow (a rank stuck in a collective would otherwise block the launcher until an outside kill).  On any failure the
    remaining children are terminated (then killed) and the exit code is non-zero.  Children are always NEW processes: a process
    that has touched the GPU is never re-executed."""
    import shutil
    import subprocess
    import tempfile
    rdzv_dir = tempfile.mkdtemp(prefix='dclip_rdzv_')
    rdzv = os.path.join(rdzv_dir, 'store')
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), DCLIP_RDZV_FILE=rdzv)
        env.pop('MASTER_PORT', None)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # rank 0 inherits stdout (its JSON line is this command's output); the other ranks' stdout goes to stderr
        out = None if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    rc = 0
    alive = list(procs)
    t_start = time.monotonic()
    t_first_exit = None
    why = None
    while alive:
        time.sleep(0.1)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if t_first_exit is None:
                t_first_exit = time.monotonic()
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                why = f'rank {procs.index(p)} exited with code {code}'
        now = time.monotonic()
        if rc == 0 and alive and now - t_start > timeout_s:
            rc, why = 124, f'{len(alive)} rank(s) still running after the {timeout_s:.0f} s deadline'
        if rc == 0 and alive and t_first_exit is not None and now - t_first_exit > grace_s:
            rc, why = 124, f'{len(alive)} rank(s) still running {grace_s:.0f} s after the first rank exited'
        if rc != 0 and alive:
            # a rank failed or hung: the others would wait in a collective forever
            print(f'bench.py launcher: {why}; terminating {len(alive)} remaining rank(s)', file=sys.stderr, flush=True)
            for q in alive:
                q.terminate()
            t_kill = time.monotonic() + 10.0
            while any(q.poll() is None for q in alive) and time.monotonic() < t_kill:
                time.sleep(0.1)
            for q in alive:
                if q.poll() is None:
                    q.kill()
            for q in alive:
                q.wait()
            alive = []
    if rc == 0 and any(p.returncode != 0 for p in procs):
        rc = 1
    shutil.rmtree(rdzv_dir, ignore_errors=True)
    return rc


def init_dist(backend, rank, world, device=None):
    """Join the process group: FileStore named by the launcher (DCLIP_RDZV_FILE), else env:// (torch.distributed.run)."""
    import torch.distributed as dist
    kw = {'device_id': device} if device is not None else {}
    rdzv = os.environ.get('DCLIP_RDZV_FILE')
    if rdzv:
        dist.init_process_group(backend, init_method='file://' + rdzv, rank=rank, world_size=world, **kw)
    else:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def dry_launch():
    """launcher rehearsal without a GPU: every rank joins a gloo group, rank 0 reports the world size the ranks observed.
    DCLIP_DRY_HANG_RANK=r (or 'all') makes rank r sleep forever after the collective (launcher deadline test)."""
    import torch
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    dist = init_dist('gloo', rank, world)
    t = torch.ones(1)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({'dry_launch': True, 'n_gpus': dist.get_world_size(), 'ranks_seen': int(t.item()), 'backend': 'gloo'}), flush=True)
    dist.barrier()
    if os.environ.get('DCLIP_DRY_HANG_RANK') in (str(rank), 'all'):
        time.sleep(3600)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', choices=sorted(WORKLOADS), default='lclip',
                    help='lclip = BASELINE.json configs[3] (the headline metric); image / text / lclip336 = configs[1] / [2] / [4]')
    ap.add_argument('--batch', type=int, default=0, help='units per GPU (default: the configuration\'s own batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-clock-probe', action='store_true')
    ap.add_argument('--dry-launch', action='store_true', help='spawn the ranks, join a gloo group on the CPU, exit (launcher test)')
    ap.add_argument('--launch-timeout', type=float, default=1800.0, help='launcher: seconds every rank has to finish')
    ap.add_argument('--launch-grace', type=float, default=120.0, help='launcher: seconds the other ranks get after the first one exited')
    ap.add_argument('--global-negatives', action='store_true',
                    help='opt-in north-star mode: in-batch negatives over all ranks (all-gather of the embeddings over RCCL)')
    ap.add_argument('--teacher-text-prefix', action='store_true',
                    help='opt-in: run the causal text teacher only on the prefix that contains every EOT (identical output, '
                         'less work; NOT used for the headline number, which processes all 77 positions)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # nothing above touched the GPU (torch is not even imported)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout, args.launch_grace))
    if args.dry_launch:
        return dry_launch()

    # stdout carries exactly ONE JSON line: libraries that print to fd 1 (RCCL's version banner at communicator creation does)
    # are sent to stderr; the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    import torch

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    use_dist = world > 1 or os.environ.get('DCLIP_FORCE_DIST') == '1'      # the latter: exercise the RCCL path on one GPU
    ranks_seen = 1
    if use_dist:
        dist = init_dist('nccl', rank, world, device)
        world = dist.get_world_size()                       # n_gpus reported = the RCCL world size the ranks observed
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)                               # census on the device: every rank really took part in an RCCL collective
        ranks_seen = int(ones.item())
        if ranks_seen != world:
            raise SystemExit(f'RCCL census: all-reduce of ones gave {ranks_seen}, world size {world}')

    from distillclip_amd._lib import lib
    wl = WORKLOADS[args.config]
    dual = wl['kind'] == 'dual'
    seed = 2022                                             # main.py:24 seed_everything_default
    model = build_model(wl, seed, device)
    if use_dist:
        from distillclip_amd.parallel import GradSync
        model._sync = GradSync()
        model._sync.enabled = True          # also at world size 1 (DCLIP_FORCE_DIST): the collectives still run
    (opt,), _ = model.configure_optimizers()
    model.loss_control.global_negatives = args.global_negatives
    B = args.batch or wl['batch']
    image, text, caps = make_inputs(wl, seed + rank, B)     # per-rank shard, resident in HBM
    image = image.to(device) if image is not None else None
    text = text.to(device) if text is not None else None
    batch = [image, text] if dual else (image if wl['kind'] == 'image' else text)
    tt_tokens = 77
    if dual:
        # the tokenizer knows caption lengths on the host: the causal teacher text tower only needs the prefix holding every EOT
        tt_tokens = int((caps != 0).sum(1).max()) if args.teacher_text_prefix else 77
        model.set_text_length_hint(tt_tokens if args.teacher_text_prefix else None)

    FUSED_ZERO = os.environ.get('DCLIP_BENCH_FUSED_ZERO', '1') != '0'
    OVERLAP_OPT = os.environ.get('DCLIP_BENCH_OVERLAP_OPT', '1') != '0'   # per-tower optimizer step on the tower's own stream
    PIPE_TEACHER = dual and os.environ.get('DCLIP_BENCH_PIPELINE_TEACHER', '0') == '1'
    pending = {'teacher': None}

    def step():
        if PIPE_TEACHER:
            # the frozen teacher of the NEXT batch runs under this batch's student backward (same work per step, same values)
            handle = pending['teacher'] or model.teacher_forward_async(batch)
            loss = model.training_step(batch, teacher=handle)
            opt.zero_grad()
            loss.backward()
            pending['teacher'] = model.teacher_forward_async(batch)
            model.backward_and_sync(None, defer_wait=OVERLAP_OPT)
        else:
            loss = model.training_step(batch)
            opt.zero_grad()
            model.backward_and_sync(loss, defer_wait=OVERLAP_OPT)
        # the fused kernel clears each gradient element as it consumes it: the next zero_grad() is free
        opt.step(zero_grad=FUSED_ZERO, overlap=OVERLAP_OPT, join=not OVERLAP_OPT)
        return loss

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    log(f'{args.config}: model built, batch {B} / GPU, world {world}; warm-up')
    for _ in range(args.warmup):
        loss = step()
    barrier()
    # shader clock actually held inside the dominant kernel during the timed steps: workgroup 0 of every 256- / 320-row gemm_nt launch
    # stamps (s_memtime, s_memrealtime) at its start and end (include/dclip.h: dclip_trace_gemm_clock; one thread per launch, no extra
    # kernel).  Roofline fractions are priced at the nominal 2.4 GHz peak; this says what the chip held.
    clock_buf = None
    if rank == 0 and not args.no_clock_probe:
        clock_cap = 16384
        clock_buf = torch.zeros(4 * clock_cap, dtype=torch.int64, device=device)
        lib().dclip_trace_gemm_clock(clock_buf.data_ptr(), clock_cap)
    log('timed steps')
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_issue = time.perf_counter() - t0                      # host time to enqueue the steps (the GPU runs behind it)
    barrier()
    dt = time.perf_counter() - t0
    clock = None
    if clock_buf is not None:
        n_stamped = min(int(lib().dclip_trace_gemm_clock(None, 0)), clock_cap)
        raw = clock_buf.view(-1, 4)[:n_stamped].cpu().double()
        ok = (raw[:, 3] > raw[:, 1]) & (raw[:, 2] > raw[:, 0])
        if int(ok.sum()) >= 8:
            mhz = ((raw[ok, 2] - raw[ok, 0]) / (raw[ok, 3] - raw[ok, 1]) * 100.0).sort().values
            q = lambda f: round(float(mhz[min(len(mhz) - 1, int(f * len(mhz)))]), 1)
            clock = {'median': q(0.5), 'p10': q(0.1), 'p90': q(0.9), 'launches': int(ok.sum()), 'nominal': 2400.0,
                     'how': 'workgroup 0 of every 256-/320-row gemm_nt launch of the timed steps: d(s_memtime) / d(s_memrealtime) x 100 MHz '
                            'over the workgroup\'s lifetime'}
    if use_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()
    log(f'{args.steps} steps in {dt:.3f} s (host enqueue {t_issue:.3f} s); loss {final_loss:.5f}')

    roofline = None
    if not args.no_roofline:
        # dominant kernel = gemm_nt (forward linears + dgrad): HIP events around every launch of 2 further steps, recorded on the
        # launch stream by the library's trace hooks (include/dclip.h: dclip_trace_*).  EVERY rank runs these steps (they contain
        # the gradient collectives); only rank 0 traces.
        import ctypes
        cap = 20000
        nprobe = 2
        multi = getattr(model, 'multi_stream', None)
        if multi is not None:
            model.multi_stream = False          # one stream: event intervals then bracket one kernel each
        step()
        torch.cuda.synchronize()
        if rank == 0:
            lib().dclip_trace_begin(cap)
        for _ in range(nprobe):
            step()
        torch.cuda.synchronize()
        if multi is not None:
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
        names = {0: 'gemm_nt_kernel', 1: 'gemm_tn_kernel', 2: 'ln_fwd_kernel', 3: 'distill_loss', 4: 'attention', 5: 'ln_bwd_kernel'}
        g = agg.get(0, [1, 1.0, 0.0, 0.0])
        achieved = g[2] / (g[1] * 1e-3) / 1e12
        # HBM bytes / launch: from the committed rocprofv3 PMC passes of this same command (separate --pmc runs, FETCH_SIZE x 2 +
        # WRITE_SIZE as MI355X_MICROARCH.md prescribes) — counters cannot be read from inside the process
        traffic, traffic_source = None, None
        import glob
        tag = '' if args.config == 'lclip' else f'{args.config}_'
        tj = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_{tag}traffic.json')))
        if tj:
            traffic = json.load(open(tj[-1])).get('gemm_nt', {}).get('hbm_bytes_per_launch')
            traffic_source = f'profiles/{os.path.basename(tj[-1])} (committed rocprofv3 --pmc passes of this command; not measured in this run)'
        roofline = {'kernel': 'gemm_nt (gemm_nt_kernel 128x128 + gemm_nt256_kernel 256x256)', 'bound': 'mfma',
                    'achieved': round(achieved, 2), 'peak': PEAK_BF16_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_BF16_TFLOPS, 4), 'traffic': traffic,
                    'traffic_source': traffic_source,
                    'algorithmic_bytes_per_launch': g[3] / g[0],
                    'launches_per_step': g[0] // nprobe, 'avg_launch_us': round(g[1] / g[0] * 1e3, 2),
                    'flop_per_launch': g[2] / g[0],
                    'others': {names.get(k, str(k)): {'launches_per_step': v[0] // nprobe, 'ms_per_step': round(v[1] / nprobe, 3),
                                                      'TFLOP/s': round(v[2] / (v[1] * 1e-3) / 1e12, 2) if v[2] else None,
                                                      'GB/s': round(v[3] / (v[1] * 1e-3) / 1e9, 1),
                                                      'hbm_frac': round(v[3] / (v[1] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
                               for k, v in agg.items()}}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('roofline probe done; CPU baseline')
        cpu = cpu_baseline(wl, seed)

    if rank == 0:
        units = B * world * args.steps
        value = units / dt
        dp = 'none'
        if use_dist:
            dp = 'reduce-scatter -> sharded AdamW -> all-gather' if getattr(model._sync, 'sharded', False) else 'all-reduce'
        out = {
            'metric': wl['metric'], 'value': round(value, 2), 'unit': f'{wl["unit"]}/s',
            'n_gpus': world, 'rccl_ranks_seen': ranks_seen if use_dist else None, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': wl['desc'], 'name': args.config,
                       'global_batch': B * world, 'batch_per_gpu': B, 'parallelism': f'dp{world}', 'gradient_exchange': dp,
                       'negatives': 'global (all-gather over RCCL)' if args.global_negatives else 'local (reference training_step semantics)',
                       'optimizer_in_step': True, 'teacher_text_tokens_processed': tt_tokens,
                       'teacher_residual': 'fp16', 'saved_gelu_derivative': 'u8 fixed point', 'clock_probe': not args.no_clock_probe},
            f'step_gflop_per_{wl["unit"][:-1]}': wl['gflop'],
            'mfma_roofline_frac_whole_step': round(value / world * wl['gflop'] * 1e9 / (PEAK_BF16_TFLOPS * 1e12), 4),
            'final_loss': round(final_loss, 6),
            'host_enqueue_ms_per_step': round(t_issue / args.steps * 1e3, 3),
            'clock_mhz_during_timed_steps': clock,
        }
        if roofline is not None:
            out['roofline'] = roofline
        if cpu is not None:
            out['cpu_baseline'] = cpu
        json_out.write(json.dumps(out) + '\n')
        json_out.flush()
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
