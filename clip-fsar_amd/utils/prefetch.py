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
    def __init__(self, loader, device, pre_upload=None, collate=1):
        self.loader = loader
        self.device = torch.device(device)
        self.pre_upload = pre_upload          # callable(host_batch) -> dict of extra (non-tensor) entries, run BEFORE the upload
        # collate = k > 1: k consecutive loader items are concatenated along dim 0 into ONE step (a reference-shaped loader yields one
        # episode per item, runs/test_net_few_shot.py:57-64; the tower wants ~16): host items are copied straight into their slice of
        # the device buffer, resident items are concatenated on the device
        self.collate = max(1, int(collate))
        self.copy_stream = torch.cuda.Stream(self.device)
        self._bufs = [None, None]
        self._ready = [None, None]
        self._done = [None, None]

    def __len__(self):
        return (len(self.loader) + self.collate - 1) // self.collate

    def _stage(self, i, hosts):
        extra = None
        if self.pre_upload is not None:
            for h in hosts:
                e = self.pre_upload(h)
                if e is not None and extra is not None and e != extra:
                    raise ValueError("DevicePrefetcher: the collated episodes disagree on %r vs %r" % (e, extra))
                extra = e if e is not None else extra
        host = hosts[0]
        keys = [k for k, v in host.items() if isinstance(v, torch.Tensor)]
        for j, h in enumerate(hosts[1:], 1):                        # collated items must agree (a mismatch would fail inside copy_ or drop keys silently)
            hk = [k for k, v in h.items() if isinstance(v, torch.Tensor)]
            if sorted(hk) != sorted(keys):                          # (dict insertion order is not part of the contract: ADVICE r5)
                raise ValueError("DevicePrefetcher: collated item %d has tensor keys %s, item 0 has %s" % (j, hk, keys))
            for k in keys:
                if h[k].shape[1:] != host[k].shape[1:] or h[k].dtype != host[k].dtype:
                    raise ValueError("DevicePrefetcher: collated item %d disagrees on %r: %s %s vs %s %s" % (
                        j, k, tuple(h[k].shape), h[k].dtype, tuple(host[k].shape), host[k].dtype))
        if keys and all(h[k].is_cuda for h in hosts for k in keys):      # already resident: hand it through / concatenate on the device
            out = dict(host)
            if len(hosts) > 1:
                for k in keys:
                    out[k] = torch.cat([h[k] for h in hosts], 0)
            if extra:
                out.update(extra)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            self._ready[i] = ev
            return out
        rows = {k: sum(int(h[k].shape[0]) for h in hosts) for k in keys}
        buf = self._bufs[i]
        if buf is None or any(k not in buf or buf[k].shape[1:] != host[k].shape[1:] or buf[k].dtype != host[k].dtype or buf[k].shape[0] < rows[k]
                              for k in keys):
            if buf is not None:
                # the old (smaller) buffer may still be read by the step in flight and written by a pending copy: keep it alive until the
                # compute work that used it has finished instead of returning it to the allocator with compute-stream ordering only
                self._retired = [(b, e) for b, e in getattr(self, "_retired", []) if e is not None and not e.query()]
                self._retired.append((buf, self._done[i]))
            buf = {k: torch.empty((rows[k],) + tuple(host[k].shape[1:]), dtype=host[k].dtype, device=self.device) for k in keys}
            for t in buf.values():
                t.record_stream(self.copy_stream)                    # allocated on the compute stream, written on the copy stream
            self._bufs[i] = buf
            # the caching allocator may hand out a block whose previous owner still has kernels queued on the compute stream: the copy
            # stream must not write it before they ran
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
            self._done[i] = None
        out = {}
        with torch.cuda.stream(self.copy_stream):
            if self._done[i] is not None:
                self.copy_stream.wait_event(self._done[i])          # the compute work that read this buffer two steps ago
            for k in keys:
                r = 0
                for h in hosts:
                    v = h[k]
                    buf[k][r:r + v.shape[0]].copy_(v, non_blocking=True)
                    r += int(v.shape[0])
                out[k] = buf[k][:r]                                  # ragged last step: a prefix view of the same buffer
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._ready[i] = ev
        for k, v in host.items():
            if not isinstance(v, torch.Tensor):
                out[k] = v
        if extra:
            out.update(extra)
        return out

    def _groups(self):
        it = iter(self.loader)
        while True:
            g = []
            for _ in range(self.collate):
                h = next(it, None)
                if h is None:
                    break
                g.append(h)
            if not g:
                return
            yield g

    def __iter__(self):
        comp = torch.cuda.current_stream(self.device)
        it = self._groups()
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
