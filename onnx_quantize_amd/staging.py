"""Host <-> HBM staging for the plugin seam (reference: qrules/_common.py:126-142, which hands the algorithm a NumPy
weight and takes NumPy results back).

The fused RTN kernel needs 42 us for a 4096 x 11008 weight; moving the same 180 MB over PCIe is what the seam costs.
Measured on the MI355X host (`scripts/quick_staging*.py`): a pageable upload of that weight takes 3.7 ms the first time
and 3.2 ms from memory the GPU has seen before (56 GB/s: pageable memory already moves at the PCIe rate there); a
download into a fresh NumPy array 1.5 ms for the 22.5 MB blob and 2.9 ms for 45 MB (`np.empty` + one copy; `Tensor.cpu()`
needs 4.1 / 8.5 ms for the same bytes, transparent huge pages make it worse, a page-locked bounce buffer adds a CPU copy
that costs more than it saves).  What is left to remove is the SERIALISATION of upload, kernels and download:
:meth:`WeightStager.prefetch` takes the whole list of weights a model is going to hand to the seam -- known as soon as
the pre-passes are done -- and uploads them from a worker thread on a side stream while the main thread quantizes and
downloads, so weight i+1 travels while weight i is in the kernels or on its way back.  The worker page-locks each source
array in place for the time of its copy (`hipHostRegister`, microseconds on that host), which is what lets the upload run
beside the main thread's download: 3.7 ms per weight for upload + download against 5.6 ms without it and 8.4 ms in
sequence.  288 GB of HBM hold every weight of a 7B model (26 GB); the in-flight budget is bounded by ``max_ahead_bytes``.

Nothing here computes anything: torch is used for device memory, streams and events only.
"""
from __future__ import annotations

import os
import threading
from collections import OrderedDict

import numpy as np

__all__ = ["WeightStager", "default_stager", "upload", "download"]

def _identity(a: np.ndarray):
    """What has to be unchanged for a prefetched copy to still stand for the array: shape, dtype and a strided sample of
    512 elements (a pass that rescales or replaces a weight between `prefetch` and `take` changes the sample; the sample
    costs microseconds, a full comparison would cost more than the upload it saves)."""
    flat = a.reshape(-1) if a.flags.c_contiguous else np.ascontiguousarray(a).reshape(-1)
    step = max(1, flat.size // 512)
    return (a.shape, a.dtype.str, flat[::step][:512].tobytes(), flat[-1:].tobytes())


class WeightStager:
    """Uploads of fp32 weights, optionally ahead of time from a worker thread on a side stream, and the matching
    downloads."""

    def __init__(self, device=None, max_ahead_bytes: int = 16 << 30):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("WeightStager needs a GPU: the HIP path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_ahead_bytes = int(max_ahead_bytes)
        self._registered: list = []                                  # (event, array) page-locked until the copy is done
        self.lock_pages = os.environ.get("OQ_STAGER_LOCK_PAGES", "1") != "0"
        self._stream = torch.cuda.Stream(device=self.device)
        self._ready: "OrderedDict[str, tuple]" = OrderedDict()      # name -> (identity, device tensor, event)
        self._cv = threading.Condition()
        self._ahead = 0
        self._pending: set = set()                                   # names the worker has not uploaded yet
        self._worker = None
        self._stop = False
        self.stats = {"prefetched": 0, "hits": 0, "misses": 0, "stale": 0}

    # ------------------------------------------------------------------ transfers
    def _sweep(self, wait: bool = False) -> None:
        """Release the page locks of source arrays whose copy has completed."""
        import torch

        rt = torch.cuda.cudart()
        keep = []
        for ev, src in self._registered:
            if wait:
                ev.synchronize()
            if ev.query():
                rt.cudaHostUnregister(src.ctypes.data)
            else:
                keep.append((ev, src))
        self._registered = keep

    def _upload(self, a: np.ndarray, stream, lock_pages: bool = False):
        """fp32 C-contiguous copy of `a` in HBM issued on `stream`; the event marks its arrival."""
        import torch

        src = np.ascontiguousarray(a, dtype=np.float32)
        locked = False
        if lock_pages and src.nbytes >= (1 << 20):
            self._sweep()
            locked = int(torch.cuda.cudart().cudaHostRegister(src.ctypes.data, src.nbytes, 0)) == 0
        with torch.cuda.stream(stream):
            # asynchronous only from page-locked memory that `_registered` keeps alive until the event has passed: an
            # asynchronous copy from pageable memory may still be reading `src` after this function has returned (and a
            # temporary `src` has been freed): a GPU page fault.  The blocking form returns when the bytes have left.
            dev = torch.from_numpy(src).to(self.device, non_blocking=locked)
            ev = torch.cuda.Event()
            ev.record(stream)
        if locked:
            self._registered.append((ev, src))
        return dev, ev

    def upload(self, a: np.ndarray):
        """Upload now, ordered on the CURRENT stream (no prefetch)."""
        import torch

        dev, _ = self._upload(a, torch.cuda.current_stream(self.device))
        return dev

    def download(self, t, dtype=None) -> np.ndarray:
        """Device tensor -> fresh NumPy array (`np.empty` + one copy, see the module docstring)."""
        import torch

        t = t.contiguous()
        out = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
        if t.numel():
            torch.from_numpy(out.reshape(-1) if out.ndim else out.reshape(1)).copy_(t.reshape(-1))
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
            dev, ev = self._upload(a, self._stream, lock_pages=self.lock_pages)
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
        self._sweep(wait=True)


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
