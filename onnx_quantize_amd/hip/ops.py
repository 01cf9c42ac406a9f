"""Device-level operators: torch tensors in HBM in, torch tensors in HBM out.

torch is plumbing here (device memory, streams); every computation is a call through the
C ABI of liboq_hip.so on the current torch stream.  No function in this module has a CPU
implementation: a missing library or a non-CUDA tensor is an error.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np
import torch

from . import _lib as L

_CONTAINER = {"int4": torch.int8, "uint4": torch.uint8, "int8": torch.int8, "uint8": torch.uint8,
              "int32": torch.int32, "uint32": torch.uint32}
BITS = {"int4": 4, "uint4": 4, "int8": 8, "uint8": 8, "int32": 32, "uint32": 32}


def container_dtype(qtype: str) -> torch.dtype:
    return _CONTAINER[qtype]


_LAYOUTS = {"kn": L.OQ_LAYOUT_KN, "nbits": L.OQ_LAYOUT_NBITS, "kn_packed4": L.OQ_LAYOUT_KN_PACKED4}


def _layout_code(layout: str) -> int:
    if layout not in _LAYOUTS:
        raise ValueError(f"layout must be one of {sorted(_LAYOUTS)}, got {layout!r}")
    return _LAYOUTS[layout]


def _q_buffer(layout: str, lead: tuple, k: int, n: int, g: int, qtype: str, device):
    """The integer output of one matrix (or a stack: ``lead``) in the requested layout: [K, N] one value per byte, the
    MatMulNBits blob [N, K/g, g*bits/8], or [K, N/2] packed nibbles (core/_pack.py:8-22 order)."""
    _layout_code(layout)
    if layout == "kn":
        return torch.empty((*lead, k, n), dtype=container_dtype(qtype), device=device)
    if layout == "kn_packed4":
        if BITS[qtype] != 4 or n % 2:
            raise ValueError("layout 'kn_packed4' takes a 4-bit type and an even number of columns")
        return torch.empty((*lead, k, n // 2), dtype=torch.uint8, device=device)
    return torch.empty((*lead, n, k // g, g * BITS[qtype] // 8), dtype=torch.uint8, device=device)


def _ptr(t: torch.Tensor | None) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_device(t: torch.Tensor, name: str, dtype=None) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a torch tensor in GPU memory (the HIP path has no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        # the library launches on the CURRENT device's current stream and never switches devices itself
        raise RuntimeError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                           f"call inside `with torch.cuda.device({t.device.index}):`")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")


_ARENA_MIN_BYTES = 32 << 20
_ARENA_MAX_STREAMS = 4
_ARENA: dict = {}     # (device index, stream handle) -> scratch tensor, grow-only; most recently used last


def _workspace(nbytes: int, device) -> torch.Tensor | None:
    """Scratch memory for ONE library call on the current stream.  Nothing a call returns lives in it.

    Small requests come from torch's allocator.  Requests of 32 MiB and more (the factor chain of a stack of eight
    11008 x 11008 Hessians asks for 15.5 GB) are served from one grow-only buffer per (device, stream): calls on a stream run
    in order, so the next call may overwrite what the previous one left; at most four streams keep one (least recently used
    out).  Why not `torch.empty` every time: when the
    caching allocator has split its one block of that size for a smaller request in between, the next call maps a new one,
    and `hipMalloc` of 15.5 GB takes 0.36 s on the MI355X host (measured, scripts/lab_alloc_trace.py) -- with the kernels of
    the call waiting behind it.  `release_workspaces()` hands the buffers back."""
    nbytes = max(int(nbytes), 256)
    if nbytes < _ARENA_MIN_BYTES:
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    dev = torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (index, torch.cuda.current_stream(index).cuda_stream)
    buf = _ARENA.pop(key, None)
    if buf is None or buf.numel() < nbytes:
        del buf
        buf = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", index))
    _ARENA[key] = buf                                   # (re-)inserted last: dicts keep insertion order
    while len(_ARENA) > _ARENA_MAX_STREAMS:             # a caller cycling through many streams must not pin one buffer per stream:
        _ARENA.pop(next(iter(_ARENA)))                  # the least recently used one goes back to torch's allocator (stream-ordered)
    return buf[:nbytes]       # exactly what was asked for: some calls size their split-K slabs by the room they are given


def release_workspaces() -> None:
    """Drop the per-stream scratch buffers `_workspace` keeps (they return to torch's caching allocator)."""
    _ARENA.clear()
    _MINMAX_WS.clear()
    _RTN_STATE.clear()


_RTN_STATE: dict = {}     # (device index, stream handle) -> zero-filled state of the one-read channel / tensor RTN kernels


def _rtn_state(nbytes: int, device) -> torch.Tensor | None:
    """The caller-kept state of `oq_rtn_quantize_stateful_f32`: zero before its first use and left zero by every call (the
    kernels clean up after themselves), so per-channel / per-tensor RTN needs no clear launch.  One buffer per (device,
    stream): calls on a stream run in order; at most four streams keep one.  A larger request takes a NEW zero-filled buffer."""
    if nbytes <= 0:
        return None
    dev = torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (index, torch.cuda.current_stream(index).cuda_stream)
    buf = _RTN_STATE.pop(key, None)
    if buf is None or buf.numel() < nbytes:
        buf = torch.zeros(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=torch.device("cuda", index))
    _RTN_STATE[key] = buf
    while len(_RTN_STATE) > _ARENA_MAX_STREAMS:
        _RTN_STATE.pop(next(iter(_RTN_STATE)))
    return buf


def _row_major(t: torch.Tensor) -> tuple[torch.Tensor, int]:
    """Return (tensor, leading dimension) for a 2-D tensor whose rows are contiguous."""
    assert t.dim() == 2
    if t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t, t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


def resolve_group(strategy: str, k: int, group_size) -> int:
    """Rows per (scale, zp): utils.py:16-22 for groups, K for channel / tensor."""
    if strategy != "group":
        return k
    g = min(group_size, k)
    return k if g == -1 else g


# ----------------------------------------------------------------------------- A1 / Q2
def rtn_quantize(w: torch.Tensor, qtype: str, strategy: str, group_size=-1, symmetric=False,
                 reduce_range=False, clip_ratio=1.0, mse=False, layout: str = "kn", emit_q: bool = True,
                 out=None):
    """rtn.py:54-109 on the GPU.  ``w`` [K, N] fp32 in HBM.

    Returns (q, scale, zp) -- q [K, N] (layout "kn"), the MatMulNBits blob [N, K/g, g*bits/8] (layout "nbits") or
    [K, N/2] nibble pairs in core/_pack.py:8-22 order (layout "kn_packed4": 4-bit types, group strategy);
    scale/zp 0-d | [N] | [N*K/g, 1].  With ``emit_q=False`` q is None
    (utils.py:302-348 only).
    """
    _require_device(w, "w", torch.float32)
    if w.dim() != 2:
        raise ValueError(f"weights must be 2-D [K, N], got shape {tuple(w.shape)}")
    w, ldw = _row_major(w)
    k, n = w.shape
    lib = L.load()
    g = resolve_group(strategy, k, group_size if group_size is not None else -1)
    if strategy == "group":
        if g <= 0 or (k * n) % g:
            raise ValueError(f"cannot reshape array of size {k * n} into shape (-1, {g})")
        count, shape = (k * n) // g, ((k * n) // g, 1)
    elif strategy == "channel":
        count, shape = n, (n,)
    else:
        count, shape = 1, ()
    cdt = container_dtype(qtype)
    dev = w.device
    if out is not None:
        q, scale, zp = out
    else:
        if not emit_q:
            q = None
        else:
            q = _q_buffer(layout, (), k, n, g, qtype, dev)
        scale = torch.empty(count, dtype=torch.float32, device=dev)
        zp = torch.empty(count, dtype=cdt, device=dev)
    gs = -1 if group_size is None else int(group_size)
    ws_bytes = lib.oq_rtn_workspace_bytes(k, n, L.STRATEGY_CODE[strategy], gs, int(mse))
    ws = _workspace(ws_bytes, dev)
    if emit_q:
        state = None if mse else _rtn_state(lib.oq_rtn_state_bytes(k, n, L.STRATEGY_CODE[strategy], gs), dev)
        st = lib.oq_rtn_quantize_stateful_f32(_ptr(w), k, n, ldw, L.QTYPE_CODE[qtype], L.STRATEGY_CODE[strategy], gs,
                                              int(symmetric), int(reduce_range), float(clip_ratio), int(mse), _ptr(q),
                                              _ptr(scale), _ptr(zp), _layout_code(layout),
                                              _ptr(ws), ws.numel(), _ptr(state), 0 if state is None else state.numel(), _stream())
    else:
        st = lib.oq_rtn_qparams_f32(_ptr(w), k, n, ldw, L.QTYPE_CODE[qtype], L.STRATEGY_CODE[strategy], gs,
                                    int(symmetric), int(reduce_range), float(clip_ratio), int(mse), _ptr(scale),
                                    _ptr(zp), _ptr(ws), ws.numel(), _stream())
    L.check(st)
    return q, scale.reshape(shape), zp.reshape(shape)


def fingerprint64(t: torch.Tensor) -> int:
    """64-bit content fingerprint of a contiguous device tensor's bytes (`oq_fingerprint64`, one pass at the HBM rate).  Synchronises
    (eight bytes come back): it decides what the host does next (seam.py: reuse a cached Hessian or compute it)."""
    _require_device(t, "t")
    if not t.is_contiguous():
        raise ValueError("fingerprint64 needs a contiguous tensor")
    nbytes = t.numel() * t.element_size()
    if nbytes == 0:
        raise ValueError("fingerprint64 of an empty tensor")
    if t.data_ptr() % 16:
        t = t.clone()
    out = torch.empty(1, dtype=torch.int64, device=t.device)
    L.check(L.load().oq_fingerprint64(_ptr(t), nbytes, _ptr(out), _stream()))
    return int(out.item()) & 0xFFFFFFFFFFFFFFFF


def hqq_quantize(w: torch.Tensor, group_size: int, reduce_range=False, clip_ratio=1.0, mse=False, lp_norm=0.7, beta=1e1,
                 kappa=1.01, iters=20, early_stop=True, emit_q: bool = True, layout: str = "kn", per_round_launches: bool = False):
    """hqq.py:147-213 on the GPU: uint4 / asymmetric / group with float zero points.  ``w`` [K, N] fp32 in HBM.
    Returns (q [K, N] uint8 | MatMulNBits blob [N, K/g, g/2] for layout="nbits" | None, scale [N*K/g, 1] fp32,
    zero_point [N*K/g, 1] fp32, rounds int32[1] on device)."""
    _require_device(w, "w", torch.float32)
    if w.dim() != 2:
        raise ValueError(f"weights must be 2-D [K, N], got shape {tuple(w.shape)}")
    w, ldw = _row_major(w)
    k, n = w.shape
    g = resolve_group("group", k, group_size if group_size is not None else -1)
    if g <= 0 or (k * n) % g:
        raise ValueError(f"cannot reshape array of size {k * n} into shape (-1, {g})")
    # hqq.py:181-192: initial parameters = the RTN ones with float zero points (integral values)
    _, scale, zp0 = rtn_quantize(w, "uint4", "group", group_size, False, reduce_range, clip_ratio, mse, emit_q=False)
    lib = L.load()
    dev = w.device
    rows = (k * n) // g
    zp_in = zp0.reshape(-1).to(torch.float32)
    zp = torch.empty(rows, dtype=torch.float32, device=dev)
    if not emit_q:
        q = None
    elif layout == "kn":
        q = torch.empty((k, n), dtype=torch.uint8, device=dev)
    else:
        q = torch.empty((n, k // g, g // 2), dtype=torch.uint8, device=dev)
    rounds = torch.zeros(1, dtype=torch.int32, device=dev)
    gs = -1 if group_size is None else int(group_size)
    ws = _workspace(lib.oq_hqq_workspace_bytes(k, n, gs), dev)
    L.check(lib.oq_hqq_optimize_f32(_ptr(w), k, n, ldw, gs, int(reduce_range), _ptr(scale), _ptr(zp_in), float(lp_norm),
                                    float(beta), float(kappa), int(iters), int(early_stop), int(bool(per_round_launches)), _ptr(q),
                                    L.OQ_LAYOUT_KN if layout == "kn" else L.OQ_LAYOUT_NBITS, _ptr(zp), _ptr(rounds),
                                    _ptr(ws), ws.numel(), _stream()))
    return q, scale.reshape(rows, 1), zp.reshape(rows, 1), rounds


def rtn_quantize_tensor_many(ws, qtype: str, symmetric=False, reduce_range=False, clip_ratio=1.0):
    """rtn.py:54-109 with strategy "tensor" for a list of contiguous fp32 weights of any shapes in THREE launches
    (oq_rtn_tensor_many_f32): a model of many small matrices makes a per-matrix loop launch-bound.  4- / 8-bit types.
    Returns [(q like w in the container dtype, scale 0-d fp32, zp 0-d)], views of three shared buffers."""
    if not ws:
        return []
    if BITS[qtype] > 8:
        raise NotImplementedError("rtn_quantize_tensor_many: 4- and 8-bit types only")
    flats = []
    for w in ws:
        _require_device(w, "w", torch.float32)
        flats.append(w if w.is_contiguous() else w.contiguous())
    dev = flats[0].device
    cdt = container_dtype(qtype)
    counts = [f.numel() for f in flats]
    if min(counts) == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + (c + 15) // 16 * 16)                 # every q slice starts 16-byte aligned
    q_all = torch.empty(offs[-1], dtype=cdt, device=dev)
    s_all = torch.empty(len(flats), dtype=torch.float32, device=dev)
    z_all = torch.empty(len(flats), dtype=cdt, device=dev)
    qb, sb, zb = q_all.data_ptr(), s_all.data_ptr(), z_all.data_ptr()
    table = torch.tensor([[f.data_ptr(), c, qb + o, sb + 4 * i, zb + i] for i, (f, c, o) in enumerate(zip(flats, counts, offs))],
                         dtype=torch.int64).to(dev)
    lib = L.load()
    wsb = _workspace(lib.oq_rtn_tensor_many_workspace_bytes(len(flats)), dev)
    L.check(lib.oq_rtn_tensor_many_f32(_ptr(table), len(flats), L.QTYPE_CODE[qtype], int(symmetric), int(reduce_range),
                                       float(clip_ratio), _ptr(wsb), wsb.numel(), _stream()))
    return [(q_all[o:o + c].view(f.shape), s_all[i], z_all[i]) for i, (f, c, o) in enumerate(zip(flats, counts, offs))]


def rtn_quantize_batched(w: torch.Tensor, qtype: str, group_size: int, symmetric=False, reduce_range=False,
                         clip_ratio=1.0, layout: str = "kn", out=None):
    """rtn.py:54-109 for a stack of equally shaped weights ``w`` [B, K, N] in ONE launch (group strategy).
    Returns (q [B, K, N] | [B, N, K/g, g*bits/8], scale [B, N*K/g, 1], zp [B, N*K/g, 1])."""
    _require_device(w, "w", torch.float32)
    if w.dim() != 3 or not w.is_contiguous():
        raise ValueError("w must be a contiguous [B, K, N] tensor")
    b, k, n = w.shape
    g = resolve_group("group", k, group_size)
    if k % g:
        raise ValueError("batched RTN needs K % group_size == 0")
    lib = L.load()
    cdt = container_dtype(qtype)
    if out is not None:
        q, scale, zp = out
    else:
        q = _q_buffer(layout, (b,), k, n, g, qtype, w.device)
        scale = torch.empty((b, n * k // g, 1), dtype=torch.float32, device=w.device)
        zp = torch.empty((b, n * k // g, 1), dtype=cdt, device=w.device)
    ws = _workspace(lib.oq_rtn_batched_workspace_bytes(b, k, n, int(group_size)), w.device)
    L.check(lib.oq_rtn_quantize_batched_f32(_ptr(w), b, k * n, k, n, n, L.QTYPE_CODE[qtype], int(group_size), int(symmetric),
                                            int(reduce_range), float(clip_ratio), _ptr(q), _ptr(scale), _ptr(zp),
                                            _layout_code(layout), _ptr(ws), ws.numel(), _stream()))
    return q, scale, zp


_TABLE_STAGE: dict = {}     # device index -> (pinned int64 [cap, 4], side stream): the pointer tables of rtn_quantize_many


def _table_stage(dev, rows: int):
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    pin, side = _TABLE_STAGE.get(index, (None, None))
    if side is None:
        side = torch.cuda.Stream(device=dev)
    if pin is None or pin.shape[0] < rows:
        pin = torch.empty((max(rows, 256), 4), dtype=torch.int64).pin_memory()
    _TABLE_STAGE[index] = (pin, side)
    side.synchronize()          # an earlier call's copies have left the pinned rows (they ran long ago: this returns at once)
    return pin, side


def rtn_quantize_many(ws, qtype: str, group_size: int, symmetric=False, reduce_range=False, clip_ratio=1.0, layout: str = "kn"):
    """rtn.py:54-109 (group strategy) for a LIST of [K, N] fp32 weights of any shapes living anywhere in HBM -- the MatMul
    weights of a model, which the reference quantizes node by node (qrules/_common.py:126-142).  Weights of one shape go
    through ONE C call (oq_rtn_quantize_ptrs_f32), which puts about 1.6e8 parameters into a launch (a device table of
    pointers, blockIdx.y = entry): rounds of waves merge across matrices and small matrices are no longer launch-bound
    (gemma-3-270m's 126 weights: 1.0 ms instead of 2.6 ms in a per-matrix loop; 4096 x 4096 at a model's footprint: 0.72 of
    the HBM peak instead of 0.54).  Per matrix the bits are those of `rtn_quantize`.  Returns [(q, scale, zp)] in input order; q [K, N], the
    MatMulNBits blob [N, K/g, g*bits/8] or [K, N/2] nibble pairs (layout "kn_packed4"), scale / zp [N*K/g, 1]; the outputs of
    one C call (a shape, or a piece of the first shapes) are views of three shared buffers.

    Host work runs beside the kernels (see the comments in the body): the pointer tables go through a page-locked staging
    buffer and a side stream, and the calls are ordered and cut so that only ~50 us of head have the GPU waiting (with all
    tables built and uploaded before the first launch a 224-matrix model had 0.4 ms of idle GPU in front of 5.7 ms of
    kernels; a blocking upload per shape on the launch stream was worse: it queues behind the previous shape's kernels)."""
    if not ws:
        return []
    lib = L.load()
    cdt = container_dtype(qtype)
    lay = _layout_code(layout)

    def key_of(w):
        return (w.shape, w.stride())           # two attribute reads per weight: the scan of 224 weights is 0.1 ms of head otherwise

    dev = ws[0].device
    out = [None] * len(ws)
    pin, side = _table_stage(dev, len(ws))
    table_dev = torch.empty((len(ws), 4), dtype=torch.int64, device=dev)
    table_dev.record_stream(side)
    cur = torch.cuda.current_stream()
    state = {"wsb": None, "row": 0}
    made = []

    pin_np = pin.numpy()                       # the page-locked rows as a NumPy view: filling them costs microseconds, not torch dispatches
    cur_dev = torch.cuda.current_device()

    def prepare(key, idx):
        """Check the members of one C call, allocate their outputs, write their table rows (host side only)."""
        (k, n), _strides = key
        if k == 0 or n == 0:
            raise ValueError(f"rtn_quantize_many: weight {idx[0]} is empty ({k} x {n})")
        g = resolve_group("group", k, group_size)
        if g <= 0 or k % g:
            raise ValueError("rtn_quantize_many needs K % group_size == 0 for every weight")
        mats, ldw = [], None
        st0 = _strides[0]                         # every member of a call shares shape and strides (the grouping key)
        plain = _strides[1] == 1 and st0 >= n and k > 1
        for i in idx:
            w = ws[i]
            if not (w.is_cuda and w.dtype is torch.float32 and w.device.index == cur_dev):
                _require_device(w, "w", torch.float32)      # raises with the full message
            if plain:                              # rows contiguous: the tensor as it is (the per-weight helper calls were a third of
                w2, ldw = w, st0                   # this function's time on a 126-weight model)
            else:
                w2, ldw = _row_major(w)
            mats.append(w2)
        cnt, row = len(idx), state["row"]
        q = _q_buffer(layout, (cnt,), k, n, g, qtype, dev)
        sc = torch.empty((cnt, n * k // g, 1), dtype=torch.float32, device=dev)
        zp = torch.empty((cnt, n * k // g, 1), dtype=cdt, device=dev)
        need = lib.oq_rtn_batched_workspace_bytes(cnt, k, n, int(group_size))
        if state["wsb"] is None or state["wsb"].numel() < need:
            state["wsb"] = _workspace(need, dev)     # calls on one stream run in order: a later, larger shape may take a new buffer
        # output pointers by arithmetic (the per-matrix views the caller gets back are made AFTER the launches)
        rows = pin_np[row:row + cnt]
        rows[:, 0] = [m.data_ptr() for m in mats]
        steps = np.arange(cnt, dtype=np.int64)
        rows[:, 1] = q.data_ptr() + steps * (q[0].numel() * q.element_size())
        rows[:, 2] = sc.data_ptr() + steps * (sc[0].numel() * 4)
        rows[:, 3] = zp.data_ptr() + steps * (zp[0].numel() * zp.element_size())
        made.append((idx, q, sc, zp, mats))      # operands stay referenced until the views are made
        state["row"] = row + cnt
        return (k, n, ldw, cnt, row, state["wsb"])

    def upload(row, cnt):
        """Rows [row, row + cnt) of the table to the device on the side stream; the launch stream waits for them."""
        with torch.cuda.stream(side):
            table_dev[row:row + cnt].copy_(pin[row:row + cnt], non_blocking=True)
            ready = side.record_event()
        cur.wait_event(ready)

    def launch(call):
        k, n, ldw, cnt, row, wsb = call
        L.check(lib.oq_rtn_quantize_ptrs_f32(C.c_void_p(pin[row:row + cnt].data_ptr()), C.c_void_p(table_dev[row:row + cnt].data_ptr() if cnt > 1 else 0), cnt,
                                             k, n, ldw, L.QTYPE_CODE[qtype], int(group_size), int(symmetric), int(reduce_range), float(clip_ratio), lay,
                                             _ptr(wsb), wsb.numel(), _stream()))

    def run(key, idx):
        """One C call: prepare, copy its table rows, launch."""
        call = prepare(key, idx)
        if call[3] > 1:
            upload(call[4], call[3])
        launch(call)

    for w in ws[:1]:
        if not isinstance(w, torch.Tensor) or w.dim() != 2:
            _require_device(w, "w", torch.float32)
            raise ValueError(f"weights must be 2-D [K, N], got shape {tuple(w.shape)}")
    # Host work beside the kernels.  (1) Look at the first 24 weights only and launch up to 8 of the shape that carries the
    # most parameters among them: the GPU starts after ~50 us of head.  (2) Group everything else by shape while those
    # kernels run.  (3) Shape by shape, fewest matrices first; the first shape in pieces (4, 8, rest) so that every call is
    # prepared in less time than the kernels in front of it take.
    # A small model (gemma-3-270m: 126 weights, 1e8 parameters, 0.2 ms of kernels) is bound by this function, not by its kernels:
    # everything is prepared first, the table goes up in ONE copy and the C calls follow back to back (0.90 -> see the bench's
    # `model_rtn.small_matrices`).  The interleaving below pays from a few hundred million parameters on.
    if sum(w.shape[0] * w.shape[1] for w in ws if isinstance(w, torch.Tensor) and w.dim() == 2) < (1 << 29):
        small: dict = {}
        for i, w in enumerate(ws):
            if not isinstance(w, torch.Tensor) or w.dim() != 2:
                _require_device(w, "w", torch.float32)
                raise ValueError(f"weights must be 2-D [K, N], got shape {tuple(w.shape)}")
            small.setdefault(key_of(w), []).append(i)
        calls = [prepare(key, idx) for key, idx in small.items()]
        if any(c[3] > 1 for c in calls):
            upload(0, state["row"])
        for c in calls:
            launch(c)
        for idx, q, sc, zp, _mats in made:
            for j, i in enumerate(idx):
                out[i] = (q[j], sc[j], zp[j])
        return out
    early: dict = {}
    for i, w in enumerate(ws[:24]):
        if not isinstance(w, torch.Tensor) or w.dim() != 2:
            _require_device(w, "w", torch.float32)
            raise ValueError(f"weights must be 2-D [K, N], got shape {tuple(w.shape)}")
        early.setdefault(key_of(w), []).append(i)
    taken = set()
    if len(ws) > 24:
        key0, idx0 = max(early.items(), key=lambda kv: len(kv[1]) * kv[0][0][0] * kv[0][0][1])
        run(key0, idx0[:8])
        taken = set(idx0[:8])
    groups: dict = {}
    for i, w in enumerate(ws):
        if i in taken:
            continue
        if not isinstance(w, torch.Tensor) or w.dim() != 2:
            _require_device(w, "w", torch.float32)
            raise ValueError(f"weights must be 2-D [K, N], got shape {tuple(w.shape)}")
        groups.setdefault(key_of(w), []).append(i)
    order = sorted(groups.items(), key=lambda kv: len(kv[1]))
    if order and len(order[0][1]) > 12:
        key1, idx1 = order[0]
        order = [(key1, idx1[:4]), (key1, idx1[4:12]), (key1, idx1[12:])] + order[1:]
    for key, idx in order:
        run(key, idx)
    for idx, q, sc, zp, _mats in made:
        for j, i in enumerate(idx):
            out[i] = (q[j], sc[j], zp[j])
    return out


# ----------------------------------------------------------------------------- Q1
def qparams(rmin: torch.Tensor, rmax: torch.Tensor, qtype: str, symmetric: bool, reduce_range: bool):
    """utils.py:242-299 on device ranges (any shape); returns (scale fp32, zp int32) of that shape."""
    _require_device(rmin, "rmin", torch.float32)
    _require_device(rmax, "rmax", torch.float32)
    a, b = rmin.contiguous().reshape(-1), rmax.contiguous().reshape(-1)
    scale = torch.empty_like(a)
    zp = torch.empty(a.numel(), dtype=torch.int32, device=a.device)
    L.check(L.load().oq_qparams_f32(_ptr(a), _ptr(b), a.numel(), L.QTYPE_CODE[qtype], int(symmetric),
                                    int(reduce_range), _ptr(scale), _ptr(zp), _stream()))
    return scale.reshape(rmin.shape), zp.reshape(rmin.shape)


def qparams_f64(rmin: torch.Tensor, rmax: torch.Tensor, qtype: str, symmetric: bool, reduce_range: bool):
    """utils.py:242-299 for float64 ranges (arithmetic in double, fp32 scale out)."""
    _require_device(rmin, "rmin", torch.float64)
    _require_device(rmax, "rmax", torch.float64)
    a, b = rmin.contiguous().reshape(-1), rmax.contiguous().reshape(-1)
    scale = torch.empty(a.numel(), dtype=torch.float32, device=a.device)
    zp = torch.empty(a.numel(), dtype=torch.int32, device=a.device)
    L.check(L.load().oq_qparams_f64(_ptr(a), _ptr(b), a.numel(), L.QTYPE_CODE[qtype], int(symmetric),
                                    int(reduce_range), _ptr(scale), _ptr(zp), _stream()))
    return scale.reshape(rmin.shape), zp.reshape(rmin.shape)


def minmax_rows(x: torch.Tensor):
    """utils.py:60-61 (axis=1): raw per-row (min, max) of a 2-D fp32 tensor."""
    _require_device(x, "x", torch.float32)
    x2, ldx = _row_major(x)
    r, c = x2.shape
    mn = torch.empty(r, dtype=torch.float32, device=x.device)
    mx = torch.empty(r, dtype=torch.float32, device=x.device)
    L.check(L.load().oq_minmax_rows_f32(_ptr(x2), r, c, ldx, _ptr(mn), _ptr(mx), _stream()))
    return mn, mx


# ----------------------------------------------------------------------------- K1 / K2
def _param_index(mode: str, r: int, c: int, group: int = 1):
    if mode == "tensor":
        return 1, 0, 0
    if mode == "row":
        return 1, 1, 0
    if mode == "col":
        return 1, 0, 1
    if mode == "group":        # groups of `group` rows on a [K, N] matrix, entry n*(K/g)+kg
        return group, 1, r // group
    raise ValueError(mode)


def _ragged_group(r: int, c: int, mode: str, group: int) -> bool:
    """Groups of `group` rows that do not divide K: the reference then forms them on `W.T.reshape(-1, g)` (utils.py:24),
    i.e. element (k, n) belongs to group (n*K + k) // g and groups straddle columns."""
    if mode != "group" or r % group == 0:
        return False
    if (r * c) % group:
        raise ValueError(f"cannot reshape array of size {r * c} into shape (-1, {group})")
    return True


def quantize(x: torch.Tensor, scale: torch.Tensor, zp: torch.Tensor, qtype: str, symmetric: bool,
             reduce_range: bool, mode: str = "tensor", group: int = 1) -> torch.Tensor:
    """utils.py:72-79.  x [R, C] fp32; (scale, zp) indexed per `mode` (see oq_quantize_f32)."""
    _require_device(x, "x", torch.float32)
    x2 = x.reshape(1, -1) if x.dim() != 2 else x
    if _ragged_group(x2.shape[0], x2.shape[1], mode, group):      # the reference's own row layout, one parameter per row
        rows = x2.t().contiguous().reshape(-1, group)
        return quantize(rows, scale, zp, qtype, symmetric, reduce_range, mode="row").reshape(x2.shape[1], x2.shape[0]).t().contiguous()
    x2, ldx = _row_major(x2)
    r, c = x2.shape
    s = scale.to(torch.float32).contiguous().reshape(-1)
    z = zp.to(torch.int32).contiguous().reshape(-1)
    q = torch.empty((r, c), dtype=container_dtype(qtype), device=x.device)
    rd, rs, cs = _param_index(mode, r, c, group)
    L.check(L.load().oq_quantize_f32(_ptr(x2), r, c, ldx, _ptr(s), _ptr(z), rd, rs, cs, L.QTYPE_CODE[qtype],
                                     int(symmetric), int(reduce_range), _ptr(q), _stream()))
    return q.reshape(x.shape)


def dequantize(q: torch.Tensor, scale: torch.Tensor, zp: torch.Tensor, qtype: str, mode: str = "tensor",
               group: int = 1) -> torch.Tensor:
    """utils.py:102-137 (without the layout shuffles: `mode` addresses the parameters in place).  Floating-point zero
    points (HQQ, hqq.py:77-78) are subtracted as they are, like `zero_point.astype(float32)` in utils.py:131."""
    _require_device(q, "q", container_dtype(qtype))
    q2 = q.reshape(1, -1) if q.dim() != 2 else q
    if _ragged_group(q2.shape[0], q2.shape[1], mode, group):
        rows = q2.t().contiguous().reshape(-1, group)
        return dequantize(rows, scale, zp, qtype, mode="row").reshape(q2.shape[1], q2.shape[0]).t().contiguous()
    q2 = q2.contiguous()
    r, c = q2.shape
    s = scale.to(torch.float32).contiguous().reshape(-1)
    out = torch.empty((r, c), dtype=torch.float32, device=q.device)
    rd, rs, cs = _param_index(mode, r, c, group)
    if zp.dtype.is_floating_point:
        z = zp.to(torch.float32).contiguous().reshape(-1)
        L.check(L.load().oq_dequantize_fzp_f32(_ptr(q2), r, c, L.QTYPE_CODE[qtype], _ptr(s), _ptr(z), rd, rs, cs,
                                               _ptr(out), c, _stream()))
    else:
        z = zp.to(torch.int32).contiguous().reshape(-1)
        L.check(L.load().oq_dequantize_f32(_ptr(q2), r, c, L.QTYPE_CODE[qtype], _ptr(s), _ptr(z), rd, rs, cs,
                                           _ptr(out), c, _stream()))
    return out.reshape(q.shape)


def quantize_bias(bias: torch.Tensor, x_scale: float, w_scale: torch.Tensor):
    """rtn.py:112-138 -> (int32 bias, fp32 bias scale)."""
    _require_device(bias, "bias", torch.float32)
    _require_device(w_scale, "weight_scale", torch.float32)
    b = bias.contiguous()
    ws = w_scale.contiguous().reshape(-1)
    q = torch.empty(b.numel(), dtype=torch.int32, device=b.device)
    bs = torch.empty(b.numel(), dtype=torch.float32, device=b.device)
    L.check(L.load().oq_quantize_bias_f32(_ptr(b), b.numel(), _ptr(ws), ws.numel(), float(x_scale), _ptr(q),
                                          _ptr(bs), _stream()))
    return q, bs


# ----------------------------------------------------------------------------- C1 / S1
def minmax_state(device, dtype=torch.float32) -> torch.Tensor:
    """Fresh device-resident calibrator state {min, max, seen, -} (see oq_minmax_collect_f32)."""
    return torch.zeros(4, dtype=dtype, device=device)


_MINMAX_WS: dict = {}   # (device index, stream) -> workspace reused by every collect on that stream (calls are ordered)


def minmax_collect(x: torch.Tensor, state: torch.Tensor, momentum: float = 0.0) -> None:
    """minmax.py:40-64: fold one activation batch into `state`, entirely on the device.  This sits in the
    calibration loop (thousands of tensors per run), so the host side is kept to one ctypes call."""
    if not x.is_cuda or x.dtype not in (torch.float32, torch.float64):
        raise TypeError("activations must be fp32/fp64 tensors in GPU memory")
    if state.dtype != x.dtype:
        raise TypeError("state dtype must match the activation dtype")
    flat = x if x.is_contiguous() else x.contiguous()
    n = flat.numel()
    if n == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    lib = L.load()
    stream = torch.cuda.current_stream(x.device).cuda_stream
    key = (x.device.index, stream)
    ws = _MINMAX_WS.get(key)
    if ws is None:
        ws = _MINMAX_WS[key] = _workspace(lib.oq_minmax_workspace_bytes(n), x.device)
    fn = lib.oq_minmax_collect_f32 if x.dtype == torch.float32 else lib.oq_minmax_collect_f64
    st = fn(flat.data_ptr(), n, state.data_ptr(), float(momentum), ws.data_ptr(), ws.numel(), stream)
    if st:
        L.check(st)


def minmax_collect_many(xs, states, momentum: float = 0.0) -> None:
    """minmax.py:40-64 for a list of fp32 tensors in ONE launch pair (oq_minmax_collect_many_f32): ``xs[i]`` is folded
    into ``states[i]``.  The 24-byte descriptors travel in one small host-to-device copy per call."""
    if len(xs) != len(states) or not xs:
        raise ValueError("minmax_collect_many needs equally long, non-empty lists")
    flats = []
    for x, st in zip(xs, states):
        if not x.is_cuda or x.dtype != torch.float32 or st.dtype != torch.float32:
            raise TypeError("minmax_collect_many takes fp32 tensors in GPU memory (use minmax_collect for fp64)")
        f = x if x.is_contiguous() else x.contiguous()
        if f.numel() == 0:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")
        flats.append(f)
    dev = flats[0].device
    table = torch.tensor([[f.data_ptr(), f.numel(), st.data_ptr()] for f, st in zip(flats, states)], dtype=torch.int64)
    desc = table.to(dev, non_blocking=False)
    lib = L.load()
    ws = _workspace(lib.oq_minmax_many_workspace_bytes(len(flats)), dev)
    L.check(lib.oq_minmax_collect_many_f32(_ptr(desc), len(flats), float(momentum), _ptr(ws), ws.numel(), _stream()))


def absmax(x: torch.Tensor, per_row: bool = False) -> torch.Tensor:
    """smooth_quant.py:62-74: max |x| per last-axis channel ([..., C] -> [C]); per_row=True gives the
    per-row absmax of a 2-D matrix (weights [K, N] -> [K])."""
    _require_device(x, "x", torch.float32)
    x2 = x.reshape(-1, x.shape[-1])
    x2, ldx = _row_major(x2)
    r, c = x2.shape
    lib = L.load()
    out = torch.empty(r if per_row else c, dtype=torch.float32, device=x.device)
    ws = _workspace(lib.oq_absmax_workspace_bytes(r, c, int(per_row)), x.device)
    L.check(lib.oq_absmax_f32(_ptr(x2), r, c, ldx, int(per_row), _ptr(out), _ptr(ws), ws.numel(), _stream()))
    return out


# ----------------------------------------------------------------------------- N2: AWQ / SmoothQuant searches
def _flat_inputs(x: torch.Tensor) -> torch.Tensor:
    _require_device(x, "inputs", torch.float32)
    return x.reshape(-1, x.shape[-1])


def _search_args(x, w, qtype, strategy, group_size):
    x2, ldx = _row_major(_flat_inputs(x))
    _require_device(w, "w", torch.float32)
    if w.dim() != 2 or w.shape[0] != x2.shape[1]:
        raise ValueError(f"weights must be [K, N] with K = {x2.shape[1]}, got {tuple(w.shape)}")
    w2, ldw = _row_major(w)
    if BITS[qtype] > 8:
        raise NotImplementedError("the AWQ searches take 4- and 8-bit types")
    gs = -1 if group_size is None else int(group_size)
    return x2, ldx, w2, ldw, gs


def awq_scale_search(x: torch.Tensor, w: torch.Tensor, qtype: str, strategy: str, group_size, symmetric=False,
                     reduce_range=False, n_grid: int = 20):
    """pre_passes/awq.py:114-184 on the device, one C call (oq_awq_scale_search_f32): activation / weight statistics, the
    n_grid candidate scales, and per candidate RTN -> dequantize -> X (W - W^) on the matrix cores with the squared error
    reduced in the GEMM's epilogue.  Nothing leaves the GPU until the losses and the winning scale are read.
    Returns (best_scale [K] on device, losses float64[n_grid] on host)."""
    x2, ldx, w2, ldw, gs = _search_args(x, w, qtype, strategy, group_size)
    t, k = x2.shape
    n = w2.shape[1]
    lib = L.load()
    dev = w2.device
    scales = torch.empty((n_grid, k), dtype=torch.float32, device=dev)
    losses = torch.empty(n_grid, dtype=torch.float32, device=dev)
    best = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = _workspace(lib.oq_awq_workspace_bytes(t, k, n) + 256, dev)
    off = (-ws.data_ptr()) % 256
    L.check(lib.oq_awq_scale_search_f32(_ptr(x2), t, k, ldx, _ptr(w2), n, ldw, L.QTYPE_CODE[qtype], L.STRATEGY_CODE[strategy], gs,
                                        int(symmetric), int(reduce_range), int(n_grid), _ptr(scales), _ptr(losses), _ptr(best),
                                        C.c_void_p(ws.data_ptr() + off), ws.numel() - off, _stream()))
    lv = losses.double().cpu().numpy()
    return scales[int(best.item())], lv


def awq_clip_search(x: torch.Tensor, w: torch.Tensor, qtype: str, strategy: str, group_size, symmetric=False,
                    reduce_range=False):
    """pre_passes/awq.py:207-259 (oq_awq_clip_search_f32): (best clip_ratio, losses[10])."""
    x2, ldx, w2, ldw, gs = _search_args(x, w, qtype, strategy, group_size)
    t, k = x2.shape
    n = w2.shape[1]
    lib = L.load()
    dev = w2.device
    losses = torch.empty(10, dtype=torch.float32, device=dev)
    best = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = _workspace(lib.oq_awq_workspace_bytes(t, k, n) + 256, dev)
    off = (-ws.data_ptr()) % 256
    L.check(lib.oq_awq_clip_search_f32(_ptr(x2), t, k, ldx, _ptr(w2), n, ldw, L.QTYPE_CODE[qtype], L.STRATEGY_CODE[strategy], gs,
                                       int(symmetric), int(reduce_range), _ptr(losses), _ptr(best), C.c_void_p(ws.data_ptr() + off),
                                       ws.numel() - off, _stream()))
    return 1 - int(best.item()) / 100, losses.double().cpu().numpy()


class SearchStatistics:
    """What the AWQ / SmoothQuant searches need of a calibration input, as running statistics over its batches (no batch is
    held): `gram` = (2 / rows) X^T X (the Hessian update rule of gptq.py:246-260 with n counting ROWS), `abs_sum[k]` = sum_t |x[t, k]|
    (awq.py:47-50 divides by the rows), `absmax[k]` = max_t |x[t, k]| (smooth_quant.py:62-69), `rows`.  `divide(scale)` is the
    reference's in-place rescale of the stored input (`node.meta["input"] /= scale`, awq.py:191 / smooth_quant.py:121) in these
    terms: the next consumer of the same value searches on what the previous one left."""

    def __init__(self, k: int, device):
        self.gram = torch.zeros((k, k), dtype=torch.float32, device=device)
        self.abs_sum = torch.zeros(k, dtype=torch.float32, device=device)
        self.absmax = torch.zeros(k, dtype=torch.float32, device=device)
        self.rows = 0

    def add(self, x: torch.Tensor) -> None:
        x2 = _flat_inputs(x if x.dtype == torch.float32 else x.to(torch.float32))
        x2, ldx = _row_major(x2)
        t, k = x2.shape
        lib = L.load()
        ws = _workspace(lib.oq_abs_sum_cols_workspace_bytes(k), x2.device)
        L.check(lib.oq_abs_sum_cols_f32(_ptr(x2), t, k, ldx, _ptr(self.abs_sum), 1, _ptr(ws), ws.numel(), _stream()))
        self.absmax = torch.maximum(self.absmax, absmax(x2))
        self.rows = hessian_accumulate(x2, self.gram, self.rows)            # rows as samples: gram = (2 / rows) X^T X

    @staticmethod
    def add_many(stats: "list[SearchStatistics]", xs) -> None:
        """`add` for several values of one batch: the Gram updates in ONE grouped launch chain (`hessian_accumulate_many`: a batch of
        a small model is dozens of small products, launch-bound one at a time)."""
        flat = []
        for x in xs:
            x2 = _flat_inputs(x if x.dtype == torch.float32 else x.to(torch.float32))
            flat.append(_row_major(x2)[0])
        lib = L.load()
        for st, x2 in zip(stats, flat):
            t, k = x2.shape
            ws = _workspace(lib.oq_abs_sum_cols_workspace_bytes(k), x2.device)
            L.check(lib.oq_abs_sum_cols_f32(_ptr(x2), t, k, x2.stride(0), _ptr(st.abs_sum), 1, _ptr(ws), ws.numel(), _stream()))
            st.absmax = torch.maximum(st.absmax, absmax(x2))
        if hessian_method() in ("auto", "f16x3"):
            for st, n in zip(stats, hessian_accumulate_many(flat, [st.gram for st in stats], [st.rows for st in stats])):
                st.rows = n
        else:
            for st, x2 in zip(stats, flat):
                st.rows = hessian_accumulate(x2, st.gram, st.rows)

    def divide(self, scale: torch.Tensor) -> None:
        s = scale.to(self.gram.device, torch.float32).reshape(-1)
        self.abs_sum /= s
        self.absmax /= s
        self.gram /= s.reshape(-1, 1)
        self.gram /= s.reshape(1, -1)


def _stats_args(stats: "SearchStatistics", w, qtype, group_size):
    _require_device(w, "w", torch.float32)
    k = stats.gram.shape[0]
    if w.dim() != 2 or w.shape[0] != k:
        raise ValueError(f"weights must be [K, N] with K = {k}, got {tuple(w.shape)}")
    if stats.rows <= 0:
        raise ValueError("the statistics hold no calibration rows")
    if BITS[qtype] > 8:
        raise NotImplementedError("the AWQ searches take 4- and 8-bit types")
    w2, ldw = _row_major(w)
    return w2, ldw, k, w2.shape[1], (-1 if group_size is None else int(group_size))


def awq_scale_search_stats(stats: "SearchStatistics", w: torch.Tensor, qtype: str, strategy: str, group_size, symmetric=False,
                           reduce_range=False, n_grid: int = 20):
    """`awq_scale_search` from running statistics (oq_awq_scale_search_stats_f32): (best_scale [K] on device, losses[n_grid])."""
    w2, ldw, k, n, gs = _stats_args(stats, w, qtype, group_size)
    lib = L.load()
    dev = w2.device
    scales = torch.empty((n_grid, k), dtype=torch.float32, device=dev)
    losses = torch.empty(n_grid, dtype=torch.float32, device=dev)
    best = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = _workspace(lib.oq_awq_stats_workspace_bytes(k, n) + 256, dev)
    off = (-ws.data_ptr()) % 256
    L.check(lib.oq_awq_scale_search_stats_f32(_ptr(stats.abs_sum), _ptr(stats.gram), stats.rows, k, _ptr(w2), n, ldw, L.QTYPE_CODE[qtype],
                                              L.STRATEGY_CODE[strategy], gs, int(symmetric), int(reduce_range), int(n_grid), _ptr(scales),
                                              _ptr(losses), _ptr(best), C.c_void_p(ws.data_ptr() + off), ws.numel() - off, _stream()))
    return scales[int(best.item())], losses.double().cpu().numpy()


def awq_clip_search_stats(stats: "SearchStatistics", w: torch.Tensor, qtype: str, strategy: str, group_size, symmetric=False,
                          reduce_range=False):
    """`awq_clip_search` from running statistics (oq_awq_clip_search_stats_f32): (best clip_ratio, losses[10])."""
    w2, ldw, k, n, gs = _stats_args(stats, w, qtype, group_size)
    lib = L.load()
    dev = w2.device
    losses = torch.empty(10, dtype=torch.float32, device=dev)
    best = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = _workspace(lib.oq_awq_stats_workspace_bytes(k, n) + 256, dev)
    off = (-ws.data_ptr()) % 256
    L.check(lib.oq_awq_clip_search_stats_f32(_ptr(stats.gram), stats.rows, k, _ptr(w2), n, ldw, L.QTYPE_CODE[qtype], L.STRATEGY_CODE[strategy],
                                             gs, int(symmetric), int(reduce_range), _ptr(losses), _ptr(best), C.c_void_p(ws.data_ptr() + off),
                                             ws.numel() - off, _stream()))
    return 1 - int(best.item()) / 100, losses.double().cpu().numpy()


def smooth_quant_scale_stats(stats: "SearchStatistics", w: torch.Tensor, alpha: float) -> torch.Tensor:
    """`smooth_quant_scale` from the running per-channel absmax: the kernel's column absmax of a one-row matrix is that row."""
    return smooth_quant_scale(stats.absmax.reshape(1, -1), w, alpha)


def smooth_quant_scale(x: torch.Tensor, w: torch.Tensor, alpha: float) -> torch.Tensor:
    """pre_passes/smooth_quant.py:62-74, :111-113 (oq_smooth_quant_scale_f32): the smoothing scale [K] on the device."""
    x2, ldx = _row_major(_flat_inputs(x))
    _require_device(w, "w", torch.float32)
    t, k = x2.shape
    if w.dim() != 2 or w.shape[0] != k:           # the kernel reads K rows of W: a shorter W would be read out of bounds
        raise ValueError(f"weights must be [K, N] with K = {k}, got {tuple(w.shape)}")
    if w.device != x2.device:
        raise RuntimeError(f"inputs live on {x2.device} and weights on {w.device}")
    w2, ldw = _row_major(w)
    lib = L.load()
    out = torch.empty(k, dtype=torch.float32, device=w2.device)
    ws = _workspace(lib.oq_smooth_quant_workspace_bytes(k), w2.device)
    L.check(lib.oq_smooth_quant_scale_f32(_ptr(x2), t, k, ldx, _ptr(w2), w2.shape[1], ldw, float(alpha), _ptr(out), _ptr(ws), ws.numel(),
                                          _stream()))
    return out


# ----------------------------------------------------------------------------- N3
def pack_zero_points_u4(zp: torch.Tensor, n: int, blocks: int) -> torch.Tensor:
    """_common.py:96-121: [N*blocks] 4-bit zero points -> [N, ceil(blocks/2)] (pad nibble 0x8)."""
    _require_device(zp, "zp")
    z = zp.contiguous().reshape(-1).view(torch.uint8)
    out = torch.empty((n, (blocks + 1) // 2), dtype=torch.uint8, device=zp.device)
    L.check(L.load().oq_pack_zero_points_u4(_ptr(z), n, blocks, _ptr(out), _stream()))
    return out


def pack_matmul_nbits(q: torch.Tensor, group_size: int, bits: int) -> torch.Tensor:
    """_common.py:72-87: [K, N] 4- / 8-bit values (one per byte) -> the MatMulNBits blob [N, K/g, g*bits/8]."""
    _require_device(q, "q")
    if q.dim() != 2:
        raise ValueError("q must be [K, N]")
    k, n = q.shape
    v = q.contiguous().view(torch.uint8)
    out = torch.empty((n, k // group_size, group_size * bits // 8), dtype=torch.uint8, device=q.device)
    L.check(L.load().oq_pack_matmul_nbits(_ptr(v), k, n, int(group_size), int(bits), _ptr(out), _stream()))
    return out


def pack_nibbles(values: torch.Tensor) -> torch.Tensor:
    """_pack.py:8-22: flat 4-bit packing, element 2j in the low nibble."""
    _require_device(values, "values")
    v = values.contiguous().reshape(-1).view(torch.uint8)
    out = torch.empty((v.numel() + 1) // 2, dtype=torch.uint8, device=values.device)
    L.check(L.load().oq_pack_nibbles(_ptr(v), v.numel(), _ptr(out), _stream()))
    return out


# ----------------------------------------------------------------------------- N1: the calibration walk's large products
class MatmulOperand:
    """One operand of `matmul_pieces` split into fp16 pieces on the device (`oq_matmul_prepare_f32`).  A constant weight is
    prepared once and reused for every batch of the calibration walk."""

    __slots__ = ("pieces", "kd", "cols", "per_row")

    def __init__(self, pieces, kd, cols, per_row=False):
        self.pieces, self.kd, self.cols, self.per_row = pieces, kd, cols, per_row


def _aligned_bytes(nbytes: int, device) -> torch.Tensor:
    buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    off = (-buf.data_ptr()) % 256
    return buf[off:off + nbytes]


def matmul_prepare(t: torch.Tensor, contraction_is_fast_axis: bool, per_row: bool = False) -> MatmulOperand:
    """``t`` [cols, Kd] (activations: contraction on the fast axis) or [Kd, cols] (a weight) -> its fp16 pieces.  ``per_row``
    (activations only): one power-of-two scale per row instead of one for the operand."""
    _require_device(t, "operand", torch.float32)
    if t.dim() != 2:
        raise ValueError(f"matmul_prepare takes a matrix, got {tuple(t.shape)}")
    t2, ld = _row_major(t)
    cols, kd = (t2.shape[0], t2.shape[1]) if contraction_is_fast_axis else (t2.shape[1], t2.shape[0])
    lib = L.load()
    need = lib.oq_matmul_pieces_bytes(kd, cols)
    if need == 0:
        raise ValueError(f"matmul_prepare: an operand of {kd} x {cols} is too large for the piece kernels")
    pieces = _aligned_bytes(need, t2.device)
    L.check(lib.oq_matmul_prepare_f32(_ptr(t2), kd, cols, ld, 1 if contraction_is_fast_axis else 0, 1 if per_row else 0, _ptr(pieces), need,
                                      _stream()))
    return MatmulOperand(pieces, kd, cols, bool(per_row))


def matmul_pieces(x: torch.Tensor, w: MatmulOperand | torch.Tensor) -> torch.Tensor:
    """``x`` [..., Kd] fp32 times a weight [Kd, N] (a tensor, or its `matmul_prepare(w, False)` pieces) on the fp16 matrix
    cores with two-piece operands: 22 significand bits per operand, fp32 accumulate -- the arithmetic of the Hessian
    kernels.  Returns [..., N] fp32.  The weight is scaled as a whole and every ROW of ``x`` on its own by a power of two (any
    magnitude fp32 can hold): the 22 bits hold for elements within ~2^18 of their row's (the weight's) largest, whatever the other
    rows of the batch hold."""
    _require_device(x, "x", torch.float32)
    if not isinstance(w, MatmulOperand):
        w = matmul_prepare(w, False)
    x2 = x.reshape(-1, x.shape[-1])
    if x2.shape[1] != w.kd:
        raise ValueError(f"matmul_pieces: x has {x2.shape[1]} columns, the weight {w.kd} rows")
    a = matmul_prepare(x2, True, per_row=True)               # every row (token) keeps its own 22 bits, whatever its neighbours hold
    out = torch.empty((x2.shape[0], w.cols), dtype=torch.float32, device=x.device)
    L.check(L.load().oq_matmul_pieces_f32(_ptr(a.pieces), _ptr(w.pieces), x2.shape[0], w.cols, w.kd, 1.0, 0.0, _ptr(out), w.cols, 1, _stream()))
    return out.reshape(*x.shape[:-1], w.cols)


# ----------------------------------------------------------------------------- G1 - G4
def hessian_accumulate(x: torch.Tensor, h: torch.Tensor, n_seen: int, method: str | None = None) -> int:
    """gptq.py:246-260, in place on ``h`` [K, K]; ``x`` [n_add, ..., K] fp32.  Returns the new sample
    count.  The activations are streamed through the MFMA TN GEMM; nothing is concatenated.  ``method``: the X^T X kernel
    of this call (None: the calling thread's default, `hessian_set_method`)."""
    _require_device(x, "x", torch.float32)
    _require_device(h, "H", torch.float32)
    n_add = int(x.shape[0])
    x2 = x.reshape(-1, x.shape[-1])
    x2, ldx = _row_major(x2)
    t, k = x2.shape
    if h.shape != (k, k) or not h.is_contiguous():
        raise ValueError(f"H must be a contiguous [{k}, {k}] tensor")
    lib = L.load()
    ws = _workspace(lib.oq_hessian_workspace_bytes(t, k), x.device)
    L.check(lib.oq_hessian_accumulate_f32(_ptr(x2), t, k, ldx, int(n_seen), n_add, _ptr(h), _method_code(method), _ptr(ws), ws.numel(),
                                          _stream()))
    return int(n_seen) + n_add


_MANY_MIN_K, _MANY_MIN_ROWS, _MANY_MAX_ROWS = 512, 512, 32768


def hessian_accumulate_many(xs, hs, n_seen) -> list[int]:
    """gptq.py:246-260 for a list of (input, Hessian) pairs -- the tensors one calibration batch taps -- in ONE launch chain
    (`oq_hessian_accumulate_many_f32`).  ``xs[i]`` [n_add, ..., K_i] fp32, ``hs[i]`` [K_i, K_i] updated in place, ``n_seen[i]``
    the samples already in it; returns the new sample counts.  Items narrower than 512 columns or shorter than 512 rows
    (their padding to 256-wide tiles would cost more than the launches save), items longer than 32 768 rows (the grouped
    product sums an item's rows in ONE fp32 chain; the per-tensor call slices long inputs, which also keeps the rounding error
    at 1e-6 of max |H|), and every item when another Hessian method than the fp16 pieces is selected, go through
    `hessian_accumulate` one by one."""
    xs, hs, n_seen = list(xs), list(hs), [int(n) for n in n_seen]
    if not (len(xs) == len(hs) == len(n_seen)):
        raise ValueError("hessian_accumulate_many: xs, hs and n_seen must have one entry per item")
    out = [0] * len(xs)
    rows, keep = [], []
    grouped = hessian_method() in ("auto", "f16x3")
    for i, (x, h) in enumerate(zip(xs, hs)):
        _require_device(x, "x", torch.float32)
        _require_device(h, "H", torch.float32)
        x2, ldx = _row_major(x.reshape(-1, x.shape[-1]))
        t, k = x2.shape
        if h.shape != (k, k) or not h.is_contiguous():
            raise ValueError(f"H[{i}] must be a contiguous [{k}, {k}] tensor")
        if not grouped or k < _MANY_MIN_K or t < _MANY_MIN_ROWS or t > _MANY_MAX_ROWS:
            out[i] = hessian_accumulate(x, h, n_seen[i])
            continue
        rows.append((x2.data_ptr(), h.data_ptr(), t, k, ldx, n_seen[i], int(x.shape[0]), 0))
        keep.append(x2)
        out[i] = n_seen[i] + int(x.shape[0])
    if rows:
        import numpy as np

        lib = L.load()
        host = np.asarray(rows, dtype=np.int64)
        dev = torch.from_numpy(host).to(keep[0].device)                   # 64 bytes per item, one blocking copy
        hp = C.c_void_p(host.ctypes.data)
        ws = _workspace(lib.oq_hessian_many_workspace_bytes(hp, len(rows)), keep[0].device)
        L.check(lib.oq_hessian_accumulate_many_f32(hp, _ptr(dev), len(rows), _ptr(ws), ws.numel(), _stream()))
    return out


class HessianPipeline:
    """gptq.py:246-260 for a SEQUENCE of batches, possibly of many inputs, with the two halves of every batch on two
    streams: the HBM-bound preparation (max |x|, scale, fp16 pieces: oq_hessian_prepare_f32) of batch i + 1 runs on a side
    stream while the matrix-core bound product of batch i (oq_hessian_accumulate_prepared_f32) runs on the caller's
    stream.  Two piece buffers, one slab buffer.  `accumulate(x, h, n_seen, next_x)` returns the new sample count like
    `hessian_accumulate`; `next_x` (the batch that will be accumulated next, for any H, or None) is prepared ahead.
    The results are bit-identical to `hessian_accumulate` with the fp16-piece method."""

    def __init__(self, device):
        self.device = device
        self.side = torch.cuda.Stream(device=device)
        self.bufs = [None, None]        # piece buffers
        self.ready = [None, None]       # event: pieces of the buffer are complete
        self.free = [None, None]        # event: the product that read the buffer is done
        self.tag = [None, None]         # (the prepared tensor itself, its ._version, shape, n_total): identity, not address
        self.slab = None
        self.slot = 0

    def reserve(self, rows: int, k: int) -> None:
        """Allocate the two piece buffers and the slab buffer for batches of up to `rows` x `k` now (otherwise on first use)."""
        lib = L.load()
        need = lib.oq_hessian_pieces_bytes(int(rows), int(k))
        for i in (0, 1):
            if self.bufs[i] is None or self.bufs[i].numel() < need:
                self.bufs[i] = torch.empty(need, dtype=torch.uint8, device=self.device)
        sl = lib.oq_hessian_slab_bytes(int(k))
        if self.slab is None or self.slab.numel() < sl:
            self.slab = torch.empty(sl, dtype=torch.uint8, device=self.device)

    @staticmethod
    def _flat(x):
        _require_device(x, "x", torch.float32)
        x2 = x.reshape(-1, x.shape[-1])
        return _row_major(x2)

    def _prepare(self, i, x, n_total):
        lib = L.load()
        x2, ldx = self._flat(x)
        t, k = x2.shape
        need = lib.oq_hessian_pieces_bytes(t, k)
        cur = torch.cuda.current_stream(self.device)
        if self.bufs[i] is None or self.bufs[i].numel() < need:
            old = self.bufs[i]
            if old is not None:
                # a prepare-ahead that was never consumed may still be writing the old buffer on the side stream, and a
                # product may still be reading it on the caller's: the block must not go back to the allocator (which
                # hands it out in the CALLER's stream order) before both are done
                old.record_stream(self.side)
                if self.ready[i] is not None:
                    cur.wait_event(self.ready[i])
            self.bufs[i] = torch.empty(need, dtype=torch.uint8, device=self.device)
            self.ready[i] = self.free[i] = self.tag[i] = None
        buf = self.bufs[i]
        off = (-buf.data_ptr()) % 256
        ev = torch.cuda.Event()
        ev.record(cur)                                  # X is ready on the caller's stream at this point
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            if self.free[i] is not None:
                self.side.wait_event(self.free[i])      # the product that read this buffer last
            L.check(lib.oq_hessian_prepare_f32(_ptr(x2), t, k, ldx, int(n_total), C.c_void_p(buf.data_ptr() + off), buf.numel() - off,
                                               C.c_void_p(self.side.cuda_stream)))
            done = torch.cuda.Event()
            done.record(self.side)
        x2.record_stream(self.side)
        self.ready[i] = done
        # the tag holds the tensor: while it is held its address cannot be handed to another batch by the allocator, and
        # `_version` catches an in-place edit between the preparation and the product
        self.tag[i] = (x2, x2._version, tuple(x2.shape), int(n_total))

    def accumulate(self, x, h, n_seen: int, next_x=None, next_n_total: int | None = None) -> int:
        lib = L.load()
        _require_device(h, "H", torch.float32)
        n_add = int(x.shape[0])
        x2, ldx = self._flat(x)
        t, k = x2.shape
        if h.shape != (k, k) or not h.is_contiguous():
            raise ValueError(f"H must be a contiguous [{k}, {k}] tensor")
        n_total = int(n_seen) + n_add
        i = self.slot
        tag = self.tag[i]
        # same storage (held by the tag, so its address cannot have been handed to another batch), same view, unedited
        prepared = (tag is not None and tag[0].untyped_storage().data_ptr() == x2.untyped_storage().data_ptr() and
                    tag[0].data_ptr() == x2.data_ptr() and tag[0].stride() == x2.stride() and tag[1] == x2._version and
                    tag[2] == tuple(x2.shape) and tag[3] == n_total)
        if not prepared:
            self._prepare(i, x, n_total)                # nothing (valid) was prepared ahead for this batch
        if next_x is not None:
            self._prepare(1 - i, next_x, next_n_total if next_n_total is not None else int(next_x.shape[0]))
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.ready[i])
        need = lib.oq_hessian_slab_bytes(k)
        if self.slab is None or self.slab.numel() < need:
            self.slab = torch.empty(need, dtype=torch.uint8, device=self.device)
        buf = self.bufs[i]
        off = (-buf.data_ptr()) % 256
        L.check(lib.oq_hessian_accumulate_prepared_f32(C.c_void_p(buf.data_ptr() + off), t, k, int(n_seen), n_add, _ptr(h), _ptr(self.slab),
                                                       self.slab.numel(), C.c_void_p(cur.cuda_stream)))
        fr = torch.cuda.Event()
        fr.record(cur)
        self.free[i] = fr
        self.tag[i] = None
        self.slot = 1 - i
        return n_total


HESSIAN_METHODS = {"auto": 0, "f32": 1, "bf16x6": 2, "bf16x9": 3, "f16x3": 4}
_METHOD_DEFAULT = threading.local()


def hessian_set_method(method: str) -> None:
    """The calling THREAD's default for the matrix-core products of the GPTQ path (include/oq_hip.h, G1): "auto" | "f32" |
    "bf16x6" | "bf16x9" | "f16x3".  A default provider only: the C library has no such state (ABI 2: the method is an
    argument of every entry point that uses it), and every function below also takes ``method=`` per call."""
    if method not in HESSIAN_METHODS:
        raise ValueError(f"unknown Hessian method {method!r}: one of {sorted(HESSIAN_METHODS)}")
    _METHOD_DEFAULT.value = method


def hessian_method() -> str:
    return getattr(_METHOD_DEFAULT, "value", "auto")


def _method_code(method: str | None) -> int:
    m = hessian_method() if method is None else method
    if m not in HESSIAN_METHODS:
        raise ValueError(f"unknown Hessian method {m!r}: one of {sorted(HESSIAN_METHODS)}")
    return HESSIAN_METHODS[m]


def gptq_prepare(w: torch.Tensor, h: torch.Tensor, actorder: bool):
    """gptq.py:118-127 in place on the working copies ``w`` [K, N] and ``h`` [K, K]; returns perm or None."""
    _require_device(w, "W", torch.float32)
    _require_device(h, "H", torch.float32)
    assert w.is_contiguous() and h.is_contiguous()
    k, n = w.shape
    lib = L.load()
    perm = torch.empty(k, dtype=torch.int32, device=w.device) if actorder else None
    ws = _workspace(lib.oq_gptq_prepare_workspace_bytes(k, n, int(actorder)), w.device)
    L.check(lib.oq_gptq_prepare_f32(_ptr(w), k, n, _ptr(h), int(actorder), _ptr(perm), _ptr(ws), ws.numel(), _stream()))
    return perm


def gptq_factor(h: torch.Tensor, percdamp: float, method: str | None = None):
    """gptq.py:134-150: returns (U [K, K] upper with inv(H + damp I) = U^T U, info int32[1] on device)."""
    _require_device(h, "H", torch.float32)
    assert h.is_contiguous() and h.dim() == 2 and h.shape[0] == h.shape[1]
    k = h.shape[0]
    lib = L.load()
    u = torch.empty((k, k), dtype=torch.float32, device=h.device)
    info = torch.zeros(1, dtype=torch.int32, device=h.device)
    ws = _workspace(lib.oq_gptq_factor_workspace_bytes(k), h.device)
    L.check(lib.oq_gptq_factor_f32(_ptr(h), k, float(percdamp), _ptr(u), _ptr(info), _method_code(method), _ptr(ws), ws.numel(), _stream()))
    return u, info


def gptq_factor_batched(h: torch.Tensor, percdamp: float, fix_dead: bool = False, method: str | None = None):
    """gptq.py:134-150 for a stack ``h`` [B, K, K] of Hessians of one width, factored in lock-step (one chain of
    launches for all of them).  Returns (U [B, K, K], info int32[B] on the device); matrix by matrix bit-identical to
    `gptq_factor`.  ``fix_dead``: zero diagonal entries count as 1 (gptq.py:119-120)."""
    _require_device(h, "H", torch.float32)
    if h.dim() != 3 or h.shape[1] != h.shape[2] or not h.is_contiguous():
        raise ValueError("H must be a contiguous [B, K, K] tensor")
    b, k = int(h.shape[0]), int(h.shape[1])
    lib = L.load()
    u = torch.empty((b, k, k), dtype=torch.float32, device=h.device)
    info = torch.zeros(b, dtype=torch.int32, device=h.device)
    ws = _workspace(lib.oq_gptq_factor_batched_workspace_bytes(k, b), h.device)
    L.check(lib.oq_gptq_factor_batched_f32(_ptr(h), k, k * k, b, float(percdamp), int(bool(fix_dead)), _ptr(u), k * k, _ptr(info),
                                           _method_code(method), _ptr(ws), ws.numel(), _stream()))
    return u, info


def gptq_shared_factors(h: torch.Tensor, percdamp: float, method: str | None = None):
    """`gptq_shared_factor` (no actorder) for a stack of Hessians [B, K, K]: one batched factor chain; returns the list
    of per-input dictionaries ``gptq_quantize(..., shared=...)`` takes (views into the batched results)."""
    u, info = gptq_factor_batched(h, percdamp, fix_dead=True, method=method)
    dead = torch.diagonal(h, dim1=1, dim2=2) == 0
    return [{"u": u[i], "info": info[i:i + 1], "perm": None, "dead": dead[i]} for i in range(h.shape[0])]


_LOOP_MODES = {"parity": L.OQ_GPTQ_PARITY, "corrected": L.OQ_GPTQ_CORRECTED, "corrected_columns": L.OQ_GPTQ_CORRECTED_COLUMNS}


def gptq_loop(w: torch.Tensor, u: torch.Tensor, qtype: str, group_size, symmetric: bool, reduce_range: bool,
              clip_ratio: float, mse: bool, block_size: int, mode: str, init_scale: torch.Tensor,
              init_zp: torch.Tensor, want_used: bool = False, layout: str = "kn", method: str | None = None):
    """gptq.py:153-216 on the working copy ``w`` (modified in place in corrected mode).
    Returns (q_int [K, N] or, with layout "kn_packed4", [K, N/2] nibble pairs; q_deq [K, N]; used_scale | None; used_zp | None).
    ``mode`` "corrected_columns" = "corrected" on the one-column-per-lane kernel (same bytes)."""
    _require_device(w, "W", torch.float32)
    _require_device(u, "U", torch.float32)
    assert w.is_contiguous() and u.is_contiguous()
    k, n = w.shape
    lib = L.load()
    g = int(group_size) if group_size else 0
    if layout not in ("kn", "kn_packed4"):
        raise ValueError("the GPTQ loop writes layout 'kn' or 'kn_packed4'")
    q_int = _q_buffer(layout, (), k, n, 0, qtype, w.device)
    q_deq = torch.empty((k, n), dtype=torch.float32, device=w.device)
    used_s = used_z = None
    if want_used and g > 0:
        ng = (k + g - 1) // g
        used_s = torch.empty((ng, n), dtype=torch.float32, device=w.device)
        used_z = torch.empty((ng, n), dtype=torch.int32, device=w.device)
    s0 = init_scale.to(torch.float32).contiguous().reshape(-1)
    z0 = init_zp.to(torch.int32).contiguous().reshape(-1)
    ws = _workspace(lib.oq_gptq_loop_workspace_bytes(k, n, int(block_size)), w.device)
    L.check(lib.oq_gptq_loop_f32(_ptr(w), k, n, _ptr(u), L.QTYPE_CODE[qtype], g, int(symmetric), int(reduce_range),
                                 float(clip_ratio), int(mse), int(block_size), _LOOP_MODES[mode], _method_code(method), _ptr(s0), _ptr(z0),
                                 s0.numel(), _ptr(q_int), _layout_code(layout), _ptr(q_deq), _ptr(used_s), _ptr(used_z), _ptr(ws),
                                 ws.numel(), _stream()))
    return q_int, q_deq, used_s, used_z


def gptq_shared_factor(h: torch.Tensor, percdamp: float, actorder: bool, method: str | None = None):
    """Everything of `_gptq` that depends on H only (gptq.py:118-150): the dead-channel mask, the optional
    actorder permutation and the inverse factor.  The reference recomputes this for every node; nodes that
    share an input (q/k/v, gate/up) share H, so one factor serves them all -- pass the result to
    ``gptq_quantize(..., shared=...)``."""
    _require_device(h, "H", torch.float32)
    k = h.shape[0]
    h1 = h.contiguous().clone()
    dummy = torch.zeros((k, 4), dtype=torch.float32, device=h.device)
    perm = gptq_prepare(dummy, h1, actorder)
    u, info = gptq_factor(h1, percdamp, method)
    dead = torch.diagonal(h) == 0
    return {"u": u, "info": info, "perm": perm, "dead": dead}


def gptq_quantize(w: torch.Tensor, h: torch.Tensor, qtype: str, strategy: str, group_size, symmetric=False,
                  reduce_range=False, clip_ratio=1.0, block_size=128, percdamp=0.01, actorder=False, mse=False,
                  mode: str = "parity", shared=None, layout: str = "kn", method: str | None = None):
    """gptq.py:76-243 (`_gptq`) on device tensors: ``w`` [K, N] weights, ``h`` [K, K] accumulated Hessian.
    Neither input is modified.  Returns (q_int, scale, zp, info) with the reference's output shapes; ``layout``
    "kn_packed4": q_int [K, N/2] as core/_pack.py:8-22 serialises the 4-bit result, written by the loop kernels."""
    if mode not in _LOOP_MODES:
        raise ValueError("mode must be 'parity' or 'corrected'")
    _require_device(w, "W", torch.float32)
    _require_device(h, "H", torch.float32)
    k, n = w.shape
    used = "channel" if strategy == "group" else strategy                      # gptq.py:92-96
    w1 = w.contiguous().clone()                                                  # :99-100
    h1 = h.contiguous().clone() if shared is None else None
    # :104-116 initial parameters: per out-channel over all of K (or global)
    _, s0, z0 = rtn_quantize(w1, qtype, used, -1, symmetric, reduce_range, clip_ratio, mse, emit_q=False)
    if shared is None:
        perm = gptq_prepare(w1, h1, actorder)                                    # :118-127
        u, info = gptq_factor(h1, percdamp, method)                              # :134-150
    else:   # H-only work done once for all layers with this input
        u, info, perm = shared["u"], shared["info"], shared["perm"]
        w1.masked_fill_(shared["dead"].unsqueeze(1), 0.0)                        # :121 (no host round trip, unlike w1[mask] = 0)
        if perm is not None:
            w1 = w1.index_select(0, perm.to(torch.int64)).contiguous()          # :126
    loop_g = group_size if (group_size and group_size != -1) else 0
    corrected_own = mode != "parity" and not actorder
    q_int, q_deq, us, uz = gptq_loop(w1, u, qtype, loop_g, symmetric, reduce_range, clip_ratio, mse, block_size,
                                     mode, s0, z0, want_used=corrected_own and strategy == "group", layout=layout, method=method)
    if actorder:                                                                 # :210-213
        inv = torch.argsort(perm.to(torch.int64))
        q_int = q_int.index_select(0, inv)
        q_deq = q_deq.index_select(0, inv)
    # :219-231 final parameters re-derived from the dequantized Q in the user's layout
    _, scale, zp = rtn_quantize(q_deq, qtype, strategy, group_size if group_size is not None else -1, symmetric,
                                reduce_range, clip_ratio, mse, emit_q=False)
    if corrected_own:
        g_eff = resolve_group("group", k, loop_g) if loop_g else 0
        if strategy == "group" and loop_g and k % g_eff == 0 and us is not None:
            # corrected mode returns the parameters the integers were produced with (see oracle/oq_oracle.py)
            scale = us.t().contiguous().reshape(-1, 1)
            zp = uz.t().contiguous().reshape(-1, 1).to(container_dtype(qtype))
        elif not loop_g and strategy in ("tensor", "channel"):
            scale, zp = s0.reshape(scale.shape), z0.reshape(zp.shape)
    return q_int, scale, zp, info
