"""Host -> device ingestion for the episodic test loop (reference runs/test_net_few_shot.py:59-62: `.cuda(non_blocking=True)` on
the pinned tensors its DataLoader yields, reference datasets/base/builder.py:83-92: DATA_LOADER.PIN_MEMORY / NUM_WORKERS).

A cfg2 step of 16 episodes carries 771 MB of fp32 frames: 16 ms of PCIe at ~48 GB/s next to ~50 ms of compute.  Uploaded on the
compute stream that is 24 % of the step (240 instead of 317 episodes/s, tools/pcie_probe.py); uploaded on a COPY stream into a second
device buffer while the previous step computes it disappears (312 of 317).  `DevicePrefetcher` is that double buffer:

    for task_dict in DevicePrefetcher(loader, device, pre_upload=check_on_host):
        model(task_dict)          # tensors are device tensors; the compute stream already waits for their copy

Two device buffer sets, allocated once per shape; the copy of step i + 1 is issued before step i is handed out and waits (on the
copy stream) for the event that marks the end of the compute work enqueued on the buffer it overwrites (step i - 1).  Host tensors
should be pinned (DataLoader(pin_memory=True)); pageable ones still work, their copies just do not overlap.
"""
import torch


class DevicePrefetcher:
    def __init__(self, loader, device, pre_upload=None):
        self.loader = loader
        self.device = torch.device(device)
        self.pre_upload = pre_upload          # callable(host_batch) -> dict of extra (non-tensor) entries, run BEFORE the upload
        self.copy_stream = torch.cuda.Stream(self.device)
        self._bufs = [None, None]
        self._ready = [None, None]
        self._done = [None, None]

    def __len__(self):
        return len(self.loader)

    def _stage(self, i, host):
        extra = self.pre_upload(host) if self.pre_upload is not None else None
        tens = {k: v for k, v in host.items() if isinstance(v, torch.Tensor)}
        if tens and all(v.is_cuda for v in tens.values()):              # already resident: hand it through
            out = dict(host)
            if extra:
                out.update(extra)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            self._ready[i] = ev
            return out
        buf = self._bufs[i]
        if buf is None or any(k not in buf or buf[k].shape[1:] != v.shape[1:] or buf[k].dtype != v.dtype or buf[k].shape[0] < v.shape[0]
                              for k, v in tens.items()):
            buf = {k: torch.empty(v.shape, dtype=v.dtype, device=self.device) for k, v in tens.items()}
            self._bufs[i] = buf
            # the caching allocator may hand out a block whose previous owner still has kernels queued on the compute stream: the copy
            # stream must not write it before they ran
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
            self._done[i] = None
        out = {}
        with torch.cuda.stream(self.copy_stream):
            if self._done[i] is not None:
                self.copy_stream.wait_event(self._done[i])          # the compute work that read this buffer two steps ago
            for k, v in tens.items():
                dst = buf[k][:v.shape[0]]                            # ragged last step: a prefix view of the same buffer
                dst.copy_(v, non_blocking=True)
                out[k] = dst
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._ready[i] = ev
        for k, v in host.items():
            if not isinstance(v, torch.Tensor):
                out[k] = v
        if extra:
            out.update(extra)
        return out

    def __iter__(self):
        comp = torch.cuda.current_stream(self.device)
        it = iter(self.loader)
        host = next(it, None)
        if host is None:
            return
        i = 0
        cur = self._stage(0, host)
        while True:
            host = next(it, None)
            nxt = self._stage(1 - i, host) if host is not None else None     # in flight while the consumer computes on `cur`
            comp.wait_event(self._ready[i])
            yield cur
            ev = torch.cuda.Event()
            ev.record(comp)                                          # everything the consumer enqueued on `cur` precedes this
            self._done[i] = ev
            if nxt is None:
                return
            cur, i = nxt, 1 - i
