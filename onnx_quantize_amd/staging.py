"""Host <-> HBM staging for the plugin seam (reference: qrules/_common.py:126-142, which hands the algorithm a NumPy
weight and takes NumPy results back).

The fused RTN kernel needs 42 us for a 4096 x 11008 weight; getting the same 180 MB over PCIe is what the seam costs.
Measured on the MI355X host (`scripts/quick_staging.py`, `quick_staging2.py`):

* memory the GPU has never touched moves at 7 GB/s (26 ms for that weight: the driver maps 45 000 pages on the way);
  every weight of a model is such memory, it is uploaded exactly once.  The same copy from memory that is already
  mapped -- a page-locked buffer that is reused -- takes 3.2 ms (56 GB/s), and the CPU copy into that buffer 0.7-2 ms.
* a device-to-host copy into a fresh NumPy array is the mirror image (2.4 ms for 22.5 MB against 0.4 ms into page-locked
  memory + 0.3-0.7 ms for the CPU copy out), and one 45 MB pageable download runs at 8.5 GB/s where three of 15 MB run
  at 43 GB/s.

So every transfer goes through a small ring of page-locked bounce buffers that live as long as the process, and
:meth:`WeightStager.prefetch` takes the whole list of weights a model is going to hand to the seam -- known as soon as
the pre-passes are done -- and uploads them from a worker thread on a side stream while the main thread quantizes:
weight i+1 travels while weight i is computed and downloaded.  288 GB of HBM hold every weight of a 7B model (26 GB);
the in-flight budget is bounded by ``max_ahead_bytes`` anyway.

Nothing here computes anything: torch is used for page-locked memory, device memory, streams and events only.
"""
from __future__ import annotations

import threading
import time
from collections import OrderedDict

import numpy as np

__all__ = ["WeightStager", "default_stager", "upload", "download"]

_CHUNK = 16 << 20


def _identity(a: np.ndarray):
    """What has to be unchanged for a prefetched copy to still stand for the array: shape, dtype and a strided sample of
    512 elements (a pass that rescales or replaces a weight between `prefetch` and `take` changes the sample; the sample
    costs microseconds, a full comparison would cost more than the upload it saves)."""
    flat = a.reshape(-1) if a.flags.c_contiguous else np.ascontiguousarray(a).reshape(-1)
    step = max(1, flat.size // 512)
    return (a.shape, a.dtype.str, flat[::step][:512].tobytes(), flat[-1:].tobytes())


class _PinnedRing:
    """`slots` page-locked byte buffers, grown on demand and touched once when they are made (a fresh page-locked buffer
    pays for its page faults on the first write); a slot is reusable once the copy that last used it has completed."""

    def __init__(self, slots: int):
        self.slots = [None] * slots
        self.events = [None] * slots
        self.next = 0
        self.lock = threading.Lock()
        self.init_ms = 0.0

    def acquire(self, nbytes: int):
        import torch

        i = self.next
        self.next = (i + 1) % len(self.slots)
        if self.events[i] is not None:
            self.events[i].synchronize()
            self.events[i] = None
        buf = self.slots[i]
        if buf is None or buf.numel() < nbytes:
            t0 = time.perf_counter()
            self.slots[i] = buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
            buf.zero_()
            self.init_ms += (time.perf_counter() - t0) * 1e3
        return i, buf


class WeightStager:
    """Uploads of fp32 weights through page-locked bounce buffers, optionally ahead of time from a worker thread, and the
    matching downloads."""

    def __init__(self, device=None, slots: int = 3, max_ahead_bytes: int = 16 << 30):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("WeightStager needs a GPU: the HIP path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_ahead_bytes = int(max_ahead_bytes)
        self._up = _PinnedRing(slots)
        self._down = _PinnedRing(2)
        self._stream = torch.cuda.Stream(device=self.device)
        self._ready: "OrderedDict[str, tuple]" = OrderedDict()      # name -> (identity, device tensor, event)
        self._cv = threading.Condition()
        self._ahead = 0
        self._pending: set = set()                                   # names the worker has not uploaded yet
        self._worker = None
        self._stop = False
        self.stats = {"prefetched": 0, "hits": 0, "misses": 0, "stale": 0}

    @property
    def init_ms(self) -> float:
        """One-time cost of the page-locked buffers so far."""
        return self._up.init_ms + self._down.init_ms

    def warm(self, nbytes: int) -> None:
        """Make every upload buffer at least `nbytes` large (and the download pieces) now; otherwise they grow on first use."""
        for ring, size in ((self._up, nbytes), (self._down, _CHUNK)):
            with ring.lock:
                for _ in ring.slots:
                    ring.acquire(size)

    # ------------------------------------------------------------------ one upload
    def _upload(self, a: np.ndarray, stream):
        """fp32 C-contiguous copy of `a` in HBM, asynchronous on `stream`; the event marks its arrival (and frees the slot)."""
        import torch

        src = np.ascontiguousarray(a, dtype=np.float32)
        n = src.size
        if n * 4 < (1 << 20):                                            # small: the bounce is not worth two copies
            with torch.cuda.stream(stream):
                dev = torch.from_numpy(src).to(self.device)
                ev = torch.cuda.Event()
                ev.record(stream)
            return dev, ev
        with self._up.lock:
            i, buf = self._up.acquire(n * 4)
            host = buf[: n * 4].view(torch.float32)
            host.copy_(torch.from_numpy(src.reshape(-1)))            # CPU copy into page-locked memory (multi-threaded)
            with torch.cuda.stream(stream):
                dev = torch.empty(src.shape, dtype=torch.float32, device=self.device)
                dev.view(-1).copy_(host, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(stream)
            self._up.events[i] = ev
        return dev, ev

    def upload(self, a: np.ndarray):
        """Upload now, ordered on the CURRENT stream (no prefetch)."""
        import torch

        dev, _ = self._upload(a, torch.cuda.current_stream(self.device))
        return dev

    def download(self, t, dtype=None) -> np.ndarray:
        """Device tensor -> fresh NumPy array through the page-locked ring, in pieces of 16 MB: piece i+1 travels while
        piece i is copied out."""
        import torch

        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        if nbytes < (1 << 20):                                           # small: the bounce is not worth two copies
            out = t.cpu().numpy()
            return out if dtype is None else out.astype(dtype, copy=False)
        out = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
        dst = torch.from_numpy(out).view(-1).view(torch.uint8)
        src = t.view(-1).view(torch.uint8)
        cur = torch.cuda.current_stream(self.device)
        with self._down.lock:
            pend = None
            for o in range(0, nbytes, _CHUNK):
                m = min(_CHUNK, nbytes - o)
                i, buf = self._down.acquire(_CHUNK)
                buf[:m].copy_(src[o:o + m], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(cur)
                self._down.events[i] = ev
                if pend is not None:
                    po, pm, pbuf, pev = pend
                    pev.synchronize()
                    dst[po:po + pm].copy_(pbuf[:pm])
                pend = (o, m, buf, ev)
            po, pm, pbuf, pev = pend
            pev.synchronize()
            dst[po:po + pm].copy_(pbuf[:pm])
        return out if dtype is None else out.astype(dtype, copy=False)

    # ------------------------------------------------------------------ prefetch
    def prefetch(self, named_arrays) -> None:
        """Start uploading ``[(name, ndarray), ...]`` in order from a worker thread.  Returns immediately."""
        items = [(str(k), v) for k, v in named_arrays]
        self.cancel()
        self._stop = False
        self._pending = {k for k, _ in items}
        self._worker = threading.Thread(target=self._run, args=(items,), name="oq-weight-stager", daemon=True)
        self._worker.start()

    def _run(self, items) -> None:
        import torch

        torch.cuda.set_device(self.device)
        for name, a in items:
            nbytes = int(np.prod(a.shape)) * 4
            with self._cv:
                while not self._stop and self._ahead > 0 and self._ahead + nbytes > self.max_ahead_bytes:
                    self._cv.wait(0.05)
                if self._stop:
                    return
            ident = _identity(a)
            dev, ev = self._upload(a, self._stream)
            with self._cv:
                self._ready[name] = (ident, dev, ev)
                self._pending.discard(name)
                self._ahead += nbytes
                self.stats["prefetched"] += 1
                self._cv.notify_all()

    def take(self, name: str, a: np.ndarray):
        """The HBM copy of weight `name`: the prefetched one when it is (still) about the same array, else a fresh
        upload.  The returned tensor is ready on the current stream."""
        import torch

        hit = None
        with self._cv:
            # a prefetch that is still on its way to this name: wait for the worker instead of uploading twice
            while name in self._pending and name not in self._ready and self._worker is not None and self._worker.is_alive():
                self._cv.wait(0.05)
            hit = self._ready.pop(name, None)
            if hit is not None:
                self._ahead -= hit[1].numel() * 4
                self._cv.notify_all()
        if hit is not None:
            ident, dev, ev = hit
            if ident == _identity(a):
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                dev.record_stream(cur)                   # allocated on the worker's stream, consumed on this one
                self.stats["hits"] += 1
                return dev
            self.stats["stale"] += 1
        self.stats["misses"] += 1
        return self.upload(a)

    def cancel(self) -> None:
        """Stop a running prefetch and drop what it uploaded."""
        w = self._worker
        if w is not None and w.is_alive():
            with self._cv:
                self._stop = True
                self._cv.notify_all()
            w.join()
        self._worker = None
        with self._cv:
            self._ready.clear()
            self._pending = set()
            self._ahead = 0


_DEFAULT: dict = {}
_DEFAULT_LOCK = threading.Lock()


def default_stager() -> WeightStager:
    """Process-wide stager of the current device (created on first use)."""
    import torch

    idx = torch.cuda.current_device()
    with _DEFAULT_LOCK:
        st = _DEFAULT.get(idx)
        if st is None:
            st = _DEFAULT[idx] = WeightStager(torch.device("cuda", idx))
        return st


def upload(a: np.ndarray):
    """fp32 C-contiguous copy of `a` in HBM on the current stream."""
    return default_stager().upload(a)


def download(t, dtype=None) -> np.ndarray:
    """Device tensor -> NumPy array through the stager of its device."""
    return default_stager().download(t, dtype)
