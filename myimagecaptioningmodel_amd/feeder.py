"""Device feeder: the host -> HBM half of the reference's data path.

The reference pulls batches from `DataReader.get_reader(batch_size, mode)` (/root/reference/ImageCaptioning/
reader.py:41-76: a generator of lists of `(image float32 [3,S,S], caption int64 [L])` samples built by
`fluid.io.batch`, fixed sample order, the last batch may be short; the HDF5 store holds fp16 pixels that the reader
casts to fp32, :45-47) through a Paddle DataLoader with `data_loader_capacity` batches of look-ahead (train.py:46,129).
Here the same contract is served by pinned host buffers and a copy stream: batch t+1 is staged and copied while
batch t trains, the compute stream only waits on the copy's event.  fp16 images travel as fp16 (half the PCIe
bytes) and are widened on the device.  Only torch's memory / stream plumbing is used; no arithmetic happens here.

Staging runs on one worker thread (reader order is kept) and fills the pinned buffers with NumPy copies, which
release the GIL.  Not with `Tensor.copy_`: torch's intra-op pool starts one spinning OpenMP thread per visible core
for a 38 MB copy, and under a CPU quota that burns the whole process's time slice - measured on the GPU box as a
70 ms stall every 100 ms (37 ms/step instead of 10.5).
"""
import queue
import threading

import numpy as np
import torch


def collate(batch):
    """One reader item -> (image [B,3,S,S] float16/float32, caption [B,L] int64) NumPy arrays.  Accepts a list of
    (image, caption) samples (fluid.io.batch form) or an already stacked (images, captions) pair."""
    if isinstance(batch, (tuple, list)) and len(batch) == 2 and hasattr(batch[0], 'ndim') and batch[0].ndim == 4:
        img, cap = np.asarray(batch[0]), np.asarray(batch[1])
    else:
        img = np.stack([np.asarray(s[0]) for s in batch])
        cap = np.stack([np.asarray(s[1], dtype=np.int64) for s in batch])
    if img.dtype not in (np.float16, np.float32):
        img = img.astype(np.float32)
    return np.ascontiguousarray(img), np.ascontiguousarray(cap.astype(np.int64))


class DeviceFeeder:
    """Iterates device batches `(image float32 [B,3,S,S], caption int64 [B,L])` in reader order.

    depth = number of batches in flight ahead of the consumer (the reference's `data_loader_capacity`); each slot
    owns a pinned staging buffer and device buffers, re-allocated only when a batch is larger than any before (the
    short last batch reuses them).  A handed-out batch aliases its slot: it stays valid until the next batch is
    requested (the engine copies its feeds at once).  An exception raised by the reader is re-raised to the consumer
    at the position where it occurred."""

    _END = object()

    def __init__(self, batches, device='cuda:0', depth=2):
        self.it = iter(batches)
        self.device = torch.device(device)
        self.cuda = self.device.type == 'cuda'
        self.depth = max(1, int(depth))
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.free_q = queue.Queue()
        self.ready_q = queue.Queue()
        for _ in range(self.depth + 1):          # `depth` staged ahead + the one the consumer holds
            self.free_q.put(dict(h_img=None, h_cap=None, d_img=None, d_cap=None, ready=None, free=None))
        self.held = None
        self.finished = False
        self.stop = threading.Event()
        self.worker = threading.Thread(target=self._work, name='capmi-feeder', daemon=True)
        self.worker.start()

    def _buffers(self, slot, img, cap):
        def fit(t, shape, dtype, pinned, dev):
            n = int(np.prod(shape))
            if t is None or t.dtype != dtype or t.numel() < n:
                t = torch.empty(max(n, 1), dtype=dtype, device=dev, pin_memory=pinned)
            return t
        idt = torch.float16 if img.dtype == np.float16 else torch.float32
        grows = slot['d_img'] is not None and (slot['d_img'].dtype != idt or slot['d_img'].numel() < img.size or slot['d_cap'].numel() < cap.size)
        if grows and slot['free'] is not None:
            # the device buffers are about to be dropped: the step that still reads them (on the engine's streams, which
            # the caching allocator knows nothing about) has to be over first
            slot['free'].synchronize()
        slot['h_img'] = fit(slot['h_img'], img.shape, idt, self.cuda, 'cpu')
        slot['h_cap'] = fit(slot['h_cap'], cap.shape, torch.int64, self.cuda, 'cpu')
        slot['d_img'] = fit(slot['d_img'], img.shape, idt, False, self.device)
        slot['d_cap'] = fit(slot['d_cap'], cap.shape, torch.int64, False, self.device)

    def _take_free(self):
        while not self.stop.is_set():
            try:
                return self.free_q.get(timeout=0.1)
            except queue.Empty:
                pass
        return None

    def _work(self):
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            for batch in self.it:
                slot = self._take_free()
                if slot is None:
                    return
                self._stage(slot, batch)
                self.ready_q.put(slot)
            self.ready_q.put(self._END)
        except BaseException as e:               # handed to the consumer, in order
            self.ready_q.put(e)

    def _stage(self, slot, batch):
        img, cap = collate(batch)
        if slot['ready'] is not None:             # the old copy out of this slot's pinned buffer has finished (long ago)
            slot['ready'].synchronize()
        self._buffers(slot, img, cap)
        n_i, n_c = img.size, cap.size
        np.copyto(slot['h_img'].numpy()[:n_i], img.reshape(-1))
        np.copyto(slot['h_cap'].numpy()[:n_c], cap.reshape(-1))
        if self.cuda:
            if slot['free'] is not None:          # ... and the step that read its device buffers runs before the new copy
                self.copy_stream.wait_event(slot['free'])
            with torch.cuda.stream(self.copy_stream):
                slot['d_img'][:n_i].copy_(slot['h_img'][:n_i], non_blocking=True)
                slot['d_cap'][:n_c].copy_(slot['h_cap'][:n_c], non_blocking=True)
                slot['ready'] = torch.cuda.Event()
                slot['ready'].record(self.copy_stream)
        else:
            slot['d_img'][:n_i].copy_(slot['h_img'][:n_i])
            slot['d_cap'][:n_c].copy_(slot['h_cap'][:n_c])
        slot['shape'] = (img.shape, cap.shape)

    def __iter__(self):
        return self

    def __next__(self):
        self.release()           # whatever was enqueued since the last batch was handed out is what reads it
        if self.finished:
            raise StopIteration
        slot = self.ready_q.get()
        if slot is self._END:
            self.finished = True
            raise StopIteration
        if isinstance(slot, BaseException):
            self.finished = True
            raise slot
        ishape, cshape = slot['shape']
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(slot['ready'])
        image = slot['d_img'][:int(np.prod(ishape))].view(ishape)
        if image.dtype != torch.float32:
            image = image.to(torch.float32)               # the reader's astype('float32'), on the device
        caption = slot['d_cap'][:int(np.prod(cshape))].view(cshape)
        self.held = slot
        return image, caption

    def release(self):
        """Mark the batch handed out last as consumed by everything enqueued on the current stream so far and give
        its buffers back to the worker (done automatically when the next batch is requested)."""
        slot, self.held = self.held, None
        if slot is not None:
            if self.cuda:
                slot['free'] = torch.cuda.Event()
                slot['free'].record(torch.cuda.current_stream(self.device))
            self.free_q.put(slot)

    def close(self):
        """Stop staging early (the worker is a daemon thread; an exhausted feeder needs no close)."""
        self.stop.set()
        self.finished = True
