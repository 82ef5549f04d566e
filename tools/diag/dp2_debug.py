"""debug the two-rank single-GPU path: python tools/diag/dp2_debug.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.multiprocessing as mp
import test_parallel_gpu as T

def rank_fn(rank, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=2)
    loss = dict(loss_name=['out_cos', 'out_kl', 'cos_diff'], loss_scale={'cos_diff': 0.1}, temperature=2.0)
    model = T._build(loss)
    (opt,), _ = model.configure_optimizers()
    image, text = T._data()
    B = T.B
    img, txt = image[rank * B:(rank + 1) * B].cuda(), text[rank * B:(rank + 1) * B].cuda()
    for step in range(2):
        l = model.training_step([img, txt])
        opt.zero_grad()
        model.backward_and_sync(l, defer_wait=True)
        torch.cuda.synchronize()
        for tw in model.towers():
            print(f'[r{rank} s{step}] loss {l.item():.5f} gshard finite {torch.isfinite(tw.gshard).all().item()} absmax {tw.gshard.abs().max().item():.3e} '
                  f'flat finite {torch.isfinite(tw.flat).all().item()} grad left {tw.flat_grad.abs().max().item():.2e} shard_elems {tw.dp.shard_elems} of {tw.flat.numel()}', flush=True)
        # instrumented copy of FusedAdamW._step_sharded
        from distillclip_amd.parallel import all_gather_flat
        opt.step_count += 1
        for tw in model.towers():
            sync = tw.sync
            m, v = opt._moments(tw)
            s_ = sync.stream_for(tw.flat)
            with sync._On(s_):
                st = s_.cuda_stream
                for i, b in enumerate(tw.dp.buckets):
                    if b is None: continue
                    b0, b1, o0, o1, off, own_tr = b
                    for a, e in own_tr:
                        lo, hi = off + a - o0, off + e - o0
                        pre = torch.isfinite(tw.flat[a:e]).all().item() and torch.isfinite(tw.gshard[lo:hi]).all().item() and torch.isfinite(m[lo:hi]).all().item() and torch.isfinite(v[lo:hi]).all().item()
                        vneg = (v[lo:hi] < 0).sum().item()
                        opt._adamw(tw.flat[a:e], tw.gshard[lo:hi], m[lo:hi], v[lo:hi], False, st)
                        torch.cuda.synchronize()
                        print(f'[r{rank} s{step}] tower {tw.flat.numel()} bucket {i} range ({a},{e}) shard ({lo},{hi}) pre-finite {pre} vneg {vneg} post-finite {torch.isfinite(tw.flat[a:e]).all().item()} lr {opt.lr} step {opt.step_count}', flush=True)
                    all_gather_flat(tw.flat[b0:b1], tw.flat[o0:o1])
                    torch.cuda.synchronize()
                    print(f'[r{rank} s{step}]   after gather bucket {i}: finite {torch.isfinite(tw.flat[b0:b1]).all().item()}', flush=True)
            tw.wcache_dirty = True
        torch.cuda.synchronize()
        for tw in model.towers():
            m, v = opt._state[id(tw)]
            print(f'[r{rank} s{step}] after step: flat finite {torch.isfinite(tw.flat).all().item()} nan count {(~torch.isfinite(tw.flat)).sum().item()} m finite {torch.isfinite(m).all().item()} v min {v.min().item():.3e}', flush=True)
            bad = (~torch.isfinite(tw.flat)).nonzero().flatten()
            if bad.numel():
                print(f'   first bad {bad[:4].tolist()} last bad {bad[-4:].tolist()} buckets {[(b[0], b[1], b[2], b[3]) if b else None for b in tw.dp.buckets]}', flush=True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == '__main__':
    port = T._free_port()
    ctx = mp.get_context('spawn')
    ps = [ctx.Process(target=rank_fn, args=(r, port)) for r in range(2)]
    [p.start() for p in ps]; [p.join() for p in ps]
