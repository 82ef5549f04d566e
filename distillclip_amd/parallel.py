"""Data-parallel gradient exchange over RCCL (torch.distributed backend "nccl" on ROCm), one process per GPU.

Reference behaviour: Lightning DDP all-reduces the student's f32 gradients.  Here every tower
already keeps its gradients in ONE flat f32 buffer, so the exchange is a handful of large collectives issued on a
side HIP stream; the image tower's exchange overlaps the text tower's backward (and vice versa) by construction of
`GradSync.launch(tower)` being called as soon as a tower's backward call has been enqueued.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, bucket_elems=32 * 1024 * 1024):
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.enabled else 1
        self.bucket = bucket_elems
        self._stream = None
        self._pending = []

    def launch(self, flat_grad, after=None):
        """average `flat_grad` across ranks, asynchronously with respect to the compute stream.  `after`: CUDA event that
        marks the buffer complete (a tower's end-of-backward event); default = everything enqueued on the current stream."""
        if not self.enabled or flat_grad is None:
            return None
        if flat_grad.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            if after is not None:
                self._stream.wait_event(after)
            else:
                self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                for b in range(0, flat_grad.numel(), self.bucket):
                    chunk = flat_grad[b:b + self.bucket]
                    dist.all_reduce(chunk, op=dist.ReduceOp.AVG)      # RCCL averages in the collective itself
            self._pending.append(flat_grad)
            done = torch.cuda.Event()
            done.record(self._stream)              # this buffer's average is complete: a per-tower optimizer step may wait on it
            return done
        else:   # gloo / CPU rehearsal of the same call pattern
            for b in range(0, flat_grad.numel(), self.bucket):
                chunk = flat_grad[b:b + self.bucket]
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
                chunk.mul_(1.0 / self.world)

    def wait(self):
        if self._stream is not None and self._pending:
            torch.cuda.current_stream().wait_stream(self._stream)
        self._pending = []

    def forget(self):
        """the caller carries the dependency itself (per-buffer events returned by launch)"""
        self._pending = []


def gather_embeddings(tensors):
    """All-gather a list of [B, E] tensors across ranks with ONE collective: -> ([W*B, E] tensors, rank, world).

    Global-negative mode (not in the reference's training_step): every rank evaluates the fused loss on the
    gathered batch and keeps the gradient rows of its own shard, so no reduce-scatter of embedding gradients is needed; the
    gradient is scaled by `world` so that the DDP *average* of parameter gradients equals the single-process gradient of the
    loss on the concatenated batch."""
    if not (dist.is_available() and dist.is_initialized()):
        return list(tensors), 0, 1
    world, rank = dist.get_world_size(), dist.get_rank()
    packed = torch.cat([t.detach().float() for t in tensors], dim=1).contiguous()          # [B, k*E]: one fused gather
    out = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out.view(-1, packed.shape[1]), packed)
    full = out.view(world * packed.shape[0], packed.shape[1])
    widths = [t.shape[1] for t in tensors]
    return [c.contiguous() for c in torch.split(full, widths, dim=1)], rank, world
