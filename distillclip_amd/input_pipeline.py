"""Host -> HBM staging for the training loop: pinned double buffers + a copy stream, so that batch k+1
crosses PCIe and is augmented / normalised (distillclip_amd/augment.py) while step k computes.

The reference's MainDataModule (data/main_datamodule.py:9-98) hands float32 image tensors from 12 PIL workers to Lightning,
which copies them synchronously.  Here the loader yields what the CPU can produce cheaply — decoded, resized, cropped uint8
images [B,H,W,3] and int64 token ids [B,77] — and everything after that runs on the GPU.
"""
import torch

from .augment import RandAugmentGPU, EvalTransformGPU


class DevicePrefetcher:
    """Iterates `(image float32 [B,3,H,W] normalised, text int64 [B,L])` CUDA batches from a loader of
    `(uint8 [B,H,W,3], int64 [B,L])` CPU batches.  Two pinned staging buffers; copy + augmentation run on a side stream;
    the consumer's stream waits on the batch's event and the tensors are recorded on it before being handed out."""

    def __init__(self, loader, device=None, train=True, num_ops=4, generator=None):
        if not torch.cuda.is_available():
            raise RuntimeError('DevicePrefetcher stages batches into HBM: it needs a CUDA (HIP) device')
        self.loader = loader
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self.transform = RandAugmentGPU(num_ops=num_ops) if train else EvalTransformGPU()
        self.train = train
        self.generator = generator
        self.stream = torch.cuda.Stream(self.device)
        self._pin = [None, None]

    def _stage(self, slot, image, text):
        pin = self._pin[slot]
        if pin is None or pin[0].shape != image.shape or pin[1].shape != text.shape:
            pin = (torch.empty(image.shape, dtype=torch.uint8).pin_memory(), torch.empty(text.shape, dtype=torch.int64).pin_memory())
            self._pin[slot] = pin
        pin[0].copy_(image)
        pin[1].copy_(text)
        with torch.cuda.stream(self.stream):
            img_u8 = pin[0].to(self.device, non_blocking=True)
            txt = pin[1].to(self.device, non_blocking=True)
            img = self.transform(img_u8, generator=self.generator) if self.train else self.transform(img_u8)
            done = torch.cuda.Event()
            done.record(self.stream)
        return img, txt, done

    def __iter__(self):
        it = iter(self.loader)
        slot = 0
        pending = None
        reuse = [None, None]                      # event after which a pinned slot may be overwritten
        for image, text in it:
            if image.dtype != torch.uint8 or image.dim() != 4 or image.shape[-1] != 3:
                raise ValueError(f'loader must yield uint8 [B,H,W,3] images, got {image.dtype} {tuple(image.shape)}')
            if reuse[slot] is not None:
                reuse[slot].synchronize()
            nxt = self._stage(slot, image, text)
            reuse[slot] = nxt[2]
            slot ^= 1
            if pending is not None:
                yield self._hand_out(pending)
            pending = nxt
        if pending is not None:
            yield self._hand_out(pending)

    def _hand_out(self, item):
        img, txt, done = item
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)
        img.record_stream(cur)
        txt.record_stream(cur)
        return img, txt
