"""Host -> device feed of the raw boundary (uint8 aligned faces), double-buffered against compute.

The reference feeds its GPU per batch from DataLoader workers: 64 preprocessed fp32 images per ResNet50 batch
(api/resnet50_extractor.py:53-72: 38.5 MB over PCIe per 64 frames) and one snippet batch per head call
(api/tester.py:65-73,84-85).  Here the boundary is the uint8 frame itself (37.6 KB), preprocessing runs on the GPU, and the
upload of the NEXT chunk runs on a dedicated copy stream while the current chunk computes:

    copy stream :  [upload c+1 -> slot B]            [upload c+2 -> slot A] ...
    compute     :  [wait ready(A)] chunk c on slot A [record free(A)] [wait ready(B)] chunk c+1 ...

A slot is re-filled only after the compute stream has recorded that it is done with it (`free` event); compute only starts on a
slot after its upload has been recorded (`ready` event).  No host synchronisation anywhere: the host thread only enqueues.
Sources should be pinned (`pin()`): a pageable source makes the copy synchronous with the host and nothing overlaps.
"""
import numpy as np
import torch


def pin(array):
    """uint8 frames (numpy array or CPU tensor) -> pinned CPU tensor (page-locked: asynchronous DMA over PCIe)."""
    t = torch.from_numpy(np.ascontiguousarray(array)) if isinstance(array, np.ndarray) else array.contiguous()
    return t if t.is_pinned() else t.pin_memory()


class FrameStream(object):
    def __init__(self, device, frames_per_slot, frame_shape=(112, 112, 3), depth=2, dtype=torch.uint8):
        self.device = torch.device(device)
        self.depth = int(depth)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = [torch.empty((int(frames_per_slot),) + tuple(frame_shape), dtype=dtype, device=self.device)
                      for _ in range(self.depth)]
        self.ready = [torch.cuda.Event() for _ in range(self.depth)]
        self.free = [torch.cuda.Event() for _ in range(self.depth)]
        self.count = [0] * self.depth
        self.tag = [None] * self.depth        # what the slot holds (caller's key), None = nothing
        self.bytes_uploaded = 0
        cur = torch.cuda.current_stream(self.device)
        for e in self.free:
            e.record(cur)                     # every slot starts free

    def upload(self, slot, pieces, tag=None):
        """Enqueue the copy of `pieces` (CPU tensors, concatenated along dim 0) into slot `slot` on the copy stream."""
        n = sum(int(p.shape[0]) for p in pieces)
        buf = self.slots[slot]
        if n > buf.shape[0]:
            raise ValueError("%d frames do not fit a slot of %d" % (n, buf.shape[0]))
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.free[slot])      # compute is done with what the slot held
            off = 0
            for p in pieces:
                k = int(p.shape[0])
                buf[off:off + k].copy_(p, non_blocking=True)
                off += k
            self.ready[slot].record(self.copy_stream)
        # the slot was allocated on the constructing stream but is written on the copy stream: tell the caching allocator, so a
        # FrameStream dropped with an upload in flight does not hand the bytes to a new tensor while the copy engine still writes them
        buf.record_stream(self.copy_stream)
        self.count[slot], self.tag[slot] = n, tag
        self.bytes_uploaded += n * buf[0].numel() * buf.element_size()
        return n

    def acquire(self, slot):
        """The current stream waits for the slot's upload; returns the filled part of the slot."""
        torch.cuda.current_stream(self.device).wait_event(self.ready[slot])
        return self.slots[slot][:self.count[slot]]

    def release(self, slot):
        """Everything enqueued so far on the current stream has to finish before the slot is overwritten."""
        self.free[slot].record(torch.cuda.current_stream(self.device))
        self.tag[slot] = None

    def close(self):
        """Wait for in-flight uploads; call before dropping or replacing a FrameStream."""
        self.copy_stream.synchronize()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def find(self, tag):
        for i, t in enumerate(self.tag):
            if t is not None and t == tag:
                return i
        return None
