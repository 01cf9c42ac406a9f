"""Host <-> HBM transfers for the plugin seam (reference: qrules/_common.py:126-142, which hands the algorithm a NumPy
weight and takes NumPy results back).

The fused RTN kernel needs 40 us for a 4096 x 11008 weight; moving the same 180 MB over PCIe is what the seam costs.
Measured on the MI355X host: a pageable upload of that weight takes 3.2-3.7 ms (56 GB/s: pageable memory already moves
at the PCIe rate there, a page-locked bounce buffer only adds a CPU copy); a download into a fresh NumPy array 1.5 ms for
the 22.5 MB blob and 2.9 ms for 45 MB (`np.empty` + one copy; `Tensor.cpu()` needs 4.1 / 8.5 ms for the same bytes).

Both directions are BLOCKING copies issued on the current stream: the source / destination array is never touched by the
GPU after the call returns, so no lifetime rule exists that a caller could break.  Rounds 1-2 also carried a
worker-thread prefetcher (page-locked sources, side stream); in the driver-run record it was slower than the on-demand
route (6.10 against 4.59 ms per weight, BENCH_r02) and it could dead-lock on its in-flight budget, so it was removed in
round 3.

A download's cost is the FIRST TOUCH of its fresh destination (every result of a model is held until the file is written, so
there is nothing to reuse): 9.6-14.8 GB/s into `np.empty` against 51 GB/s into memory touched before.  Destinations of 4 MiB
and more are therefore faulted in by four threads at once (`madvise(MADV_POPULATE_WRITE)` on a quarter each) before the one
blocking copy: 23.5 GB/s (`scripts/lab_upload_paths.py`; one thread, huge pages or a page-locked destination gain nothing).
A destination whose first and last page are resident already (`mincore`: memory the allocator hands out again) is left alone.

Nothing here computes anything: torch is used for device memory and copies only.
"""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np

__all__ = ["upload", "download"]

_PREFAULT_MIN_BYTES = 4 << 20
_PREFAULT_THREADS = 4
_MADV_POPULATE_WRITE = 23                                  # Linux >= 5.14; refused (EINVAL) by older kernels: then the copy faults as before
_prefault_lock = threading.Lock()
_prefault_state: dict = {}                                 # {"pid", "pool", "madvise", "page"} of THIS process (a forked child makes its own)

# THE INVARIANT OF THIS MODULE (tests/test_seam.py holds it): no GPU call, copy or stream is ever issued from a helper
# thread.  The helper threads run `_populate` and nothing else -- one `madvise` system call on host memory that the calling
# thread allocated and still owns; they see no tensor, no stream, no device pointer.  Both copies are issued by the CALLING
# thread, blocking, after the helpers have returned (`pool.map` is consumed before the copy starts).


def _populate(madvise, addr: int, length: int) -> int:
    """What a helper thread runs: fault in `length` bytes of host memory at `addr`.  Host memory only."""
    return madvise(addr, length, _MADV_POPULATE_WRITE)


def _prefault_reset_in_child() -> None:
    """A forked child inherits neither the parent's helper threads nor a lock some other parent thread may have held."""
    global _prefault_lock
    _prefault_lock = threading.Lock()
    _prefault_state.clear()


os.register_at_fork(after_in_child=_prefault_reset_in_child)


def _prefault_setup(st: dict) -> None:
    if st.get("pid") == os.getpid():                       # another thread got here first
        return
    from concurrent.futures import ThreadPoolExecutor
    try:
        madvise = ctypes.CDLL(None, use_errno=True).madvise
        madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        madvise.restype = ctypes.c_int
    except (OSError, AttributeError):
        madvise = None
    try:
        mincore = ctypes.CDLL(None, use_errno=True).mincore
        mincore.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p]
        mincore.restype = ctypes.c_int
    except (OSError, AttributeError):
        mincore = None
    st.update(pool=ThreadPoolExecutor(_PREFAULT_THREADS, thread_name_prefix="oq-prefault"), madvise=madvise, mincore=mincore,
              page=os.sysconf("SC_PAGE_SIZE"), works=madvise is not None)
    st["pid"] = os.getpid()                                # last: the state is complete when a reader sees its own pid


def _prefault(out: np.ndarray) -> None:
    """Touch the pages of a fresh C-contiguous array from several threads so that the copy into it does not fault one page at a time."""
    st = _prefault_state
    if st.get("pid") != os.getpid():
        with _prefault_lock:
            _prefault_setup(st)
    if not st["works"]:
        return
    page, addr = st["page"], out.ctypes.data
    lo, hi = addr + (-addr % page), addr + out.nbytes - (addr + out.nbytes) % page
    if hi - lo < _PREFAULT_MIN_BYTES:
        return
    if st["mincore"] is not None:
        # memory the allocator hands out AGAIN (a caller that frees every result before the next one: the single-weight seam) is
        # resident already; populating it once more cost that loop 0.18 ms per 22 MB result (scripts/lab_seam_prefault.py).  Two
        # system calls on one page each tell: fresh memory has neither its first nor its last page
        first, last = ctypes.create_string_buffer(1), ctypes.create_string_buffer(1)
        if (st["mincore"](lo, page, first) == 0 and st["mincore"](hi - page, page, last) == 0
                and (first.raw[0] & 1) and (last.raw[0] & 1)):
            return
    step = -(-(hi - lo) // _PREFAULT_THREADS)
    step += -step % page
    starts = range(lo, hi, step)
    rcs = list(st["pool"].map(_populate, [st["madvise"]] * len(starts), starts, [min(step, hi - a) for a in starts]))
    if any(rcs):
        st["works"] = False                                # this kernel does not know the advice: stop asking


def _device():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("the seam needs a GPU: the HIP path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def upload(a: np.ndarray):
    """fp32 C-contiguous copy of ``a`` in HBM, ordered on the current stream.  The copy is blocking (pageable source):
    when this returns the bytes have left ``a`` (and any temporary made of it)."""
    import torch

    import warnings

    src = np.ascontiguousarray(a, dtype=np.float32)
    with warnings.catch_warnings():
        # a read-only source (a view into the bytes of a model file, model_quantize.py) is only read here
        warnings.filterwarnings("ignore", message="The given NumPy array is not writable")
        return torch.from_numpy(src).to(_device(), non_blocking=False)


def download(t, dtype=None) -> np.ndarray:
    """Device tensor -> fresh NumPy array (`np.empty` + one blocking copy, see the module docstring)."""
    import torch

    t = t.contiguous()
    out = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
    if out.nbytes >= _PREFAULT_MIN_BYTES:
        _prefault(out)
    if t.numel():
        torch.from_numpy(out.reshape(-1) if out.ndim else out.reshape(1)).copy_(t.reshape(-1))
    return out if dtype is None else out.astype(dtype, copy=False)
