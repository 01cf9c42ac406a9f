"""Node-level sharding of a model's MatMul weights over the GPUs of one node.

Every MatMul / Gemm weight is quantized independently of every other one (the reference's rewriter calls
the numeric seam once per node, qrules/_common.py:126-142), so the path shards by *objects*: one process
per GPU, each quantizes the layers a cost-balanced plan assigns to it, no data-path collective while
quantizing, and one exchange at the end that brings (q, scale, zero point) of every layer to rank 0 --
an RCCL gather over xGMI on a GPU node (backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

import dataclasses
from collections.abc import Callable, Sequence

import numpy as np


@dataclasses.dataclass(frozen=True)
class LayerSpec:
    """One weight to quantize: [K, N] = (in, out) and, for GPTQ, the number of calibration tokens."""

    name: str
    k: int
    n: int
    tokens: int = 0          # 0: RTN (no Hessian)
    hessian_key: str = ""    # layers sharing the same input share one Hessian (q/k/v, gate/up)


def layer_cost(spec: LayerSpec, hessian_cached: bool = False) -> float:
    """Relative cost in flop-equivalents: Hessian SYRK (T K^2), inverse factor (~K^3), loop + lazy
    updates (K^2 N / 2 for corrected GPTQ, ~K N otherwise); RTN is byte-bound: ~K N."""
    k, n, t = spec.k, spec.n, spec.tokens
    if t == 0:
        return 5.0 * k * n
    cost = 0.5 * k * k * n + 5.0 * k * n
    if not hessian_cached:
        cost += float(t) * k * k + (2.0 / 3.0) * k ** 3
    return cost


def plan_lpt(specs: Sequence[LayerSpec], world_size: int) -> list[list[int]]:
    """Longest-processing-time greedy plan.  Layers with the same ``hessian_key`` stay on one rank so the
    Hessian and its factor are computed once.  Returns, per rank, the indices of its layers (ascending)."""
    assert world_size >= 1
    bundles: dict[str, list[int]] = {}
    for i, s in enumerate(specs):
        bundles.setdefault(s.hessian_key or f"__solo_{i}", []).append(i)
    weighted = []
    for key, idx in bundles.items():
        c = sum(layer_cost(specs[i], hessian_cached=(j > 0)) for j, i in enumerate(idx))
        weighted.append((c, key, idx))
    weighted.sort(key=lambda t: (-t[0], t[1]))
    loads = [0.0] * world_size
    plan: list[list[int]] = [[] for _ in range(world_size)]
    for c, _, idx in weighted:
        r = min(range(world_size), key=lambda j: (loads[j], j))
        loads[r] += c
        plan[r].extend(idx)
    return [sorted(p) for p in plan]


def _pack(result) -> np.ndarray:
    """(q, scale, zp) -> one contiguous byte buffer with a small header."""
    q, s, z = (np.ascontiguousarray(a) for a in result)
    header = np.array([q.nbytes, s.nbytes, z.nbytes], dtype=np.int64).view(np.uint8)
    return np.concatenate([header, q.view(np.uint8).reshape(-1), s.view(np.uint8).reshape(-1), z.view(np.uint8).reshape(-1)])


def _unpack(buf: np.ndarray, like) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    nq, ns, nz = buf[:24].view(np.int64)
    o = 24
    q_dtype, q_shape, s_shape, z_dtype, z_shape = like
    q = buf[o:o + nq].view(q_dtype).reshape(q_shape); o += nq
    s = buf[o:o + ns].view(np.float32).reshape(s_shape); o += ns
    z = buf[o:o + nz].view(z_dtype).reshape(z_shape)
    return q, s, z


def quantize_sharded(specs: Sequence[LayerSpec], quantize_fn: Callable[[int, LayerSpec], tuple], *, device=None,
                     group=None):
    """Run ``quantize_fn(index, spec) -> (q, scale, zp)`` (NumPy arrays) for this rank's share of ``specs``
    and gather every result on rank 0.  Returns ``{name: (q, scale, zp)}`` on rank 0, ``None`` elsewhere.

    Works without an initialised process group (single process: everything is local).
    """
    import torch
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    plan = plan_lpt(specs, world)
    mine = {i: quantize_fn(i, specs[i]) for i in plan[rank]}
    if world == 1:
        return {specs[i].name: mine[i] for i in sorted(mine)}

    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                             if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    # 1) everybody learns every buffer size and result signature (tiny, object all-gather)
    meta_local = {i: (len(_pack(r)), r[0].dtype.str, r[0].shape, r[1].shape, r[2].dtype.str, r[2].shape) for i, r in mine.items()}
    metas = [None] * world
    dist.all_gather_object(metas, meta_local, group=group)
    # 2) payload: one flat byte tensor per rank, gathered on rank 0 (padded to the largest)
    flat = np.concatenate([_pack(mine[i]) for i in plan[rank]]) if plan[rank] else np.zeros(0, np.uint8)
    sizes = [sum(m[i][0] for i in plan[r]) for r, m in enumerate(metas)]
    cap = max(max(sizes), 1)
    send = torch.zeros(cap, dtype=torch.uint8, device=dev)
    send[: flat.size] = torch.from_numpy(flat).to(dev)
    recv = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0, group=group)
    if rank != 0:
        return None
    out = {}
    for r in range(world):
        buf = recv[r].cpu().numpy()
        o = 0
        for i in plan[r]:
            nbytes, qd, qs, ss, zd, zs = metas[r][i]
            out[specs[i].name] = _unpack(buf[o:o + nbytes], (np.dtype(qd), qs, ss, np.dtype(zd), zs))
            o += nbytes
    return {s.name: out[s.name] for s in specs}


def llama2_7b_specs(tokens: int = 0, layers: int = 32, hidden: int = 4096, ffn: int = 11008) -> list[LayerSpec]:
    """MatMul weights of Llama-2-7B in [K, N] layout (BASELINE.json configs 4/5); lm_head excluded like the
    reference's examples do."""
    specs = []
    for l in range(layers):
        p = f"model.layers.{l}"
        for nm in ("q_proj", "k_proj", "v_proj"):
            specs.append(LayerSpec(f"{p}.self_attn.{nm}", hidden, hidden, tokens, f"{p}.attn_in"))
        specs.append(LayerSpec(f"{p}.self_attn.o_proj", hidden, hidden, tokens, f"{p}.attn_out"))
        for nm in ("gate_proj", "up_proj"):
            specs.append(LayerSpec(f"{p}.mlp.{nm}", hidden, ffn, tokens, f"{p}.mlp_in"))
        specs.append(LayerSpec(f"{p}.mlp.down_proj", ffn, hidden, tokens, f"{p}.mlp_mid"))
    return specs


def gather_device_results(specs: Sequence[LayerSpec], plan: list[list[int]], mine: dict, *, group=None, always_exchange: bool = False):
    """End-of-run exchange for results that live in HBM: ``mine[i] = (q, scale, zp)`` torch tensors on this
    rank's GPU.  Each rank flattens its results into ONE byte tensor, sizes are exchanged with a tiny
    all_gather, and a single padded ``gather`` (RCCL over xGMI with backend nccl) brings everything to rank 0.
    Returns ``({name: (q, scale, zp)} on rank 0 | None, bytes_gathered)``.  ``always_exchange``: run the collectives even
    in a process group of ONE rank (`collectives_selftest`: the packing, the size exchange, the padded gather and the
    unpacking meet the communicator on a one-GPU box)."""
    import torch
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    if world == 1 and not (always_exchange and distributed):
        return {specs[i].name: mine[i] for i in sorted(mine)}, 0
    order = plan[rank]
    parts, meta = [], []
    pad16 = lambda nbytes: (nbytes + 15) // 16 * 16   # noqa: E731  every tensor starts 16-byte aligned in the flat buffer
    for i in order:
        q, s, z = mine[i]
        for t in (q, s, z):
            b = t.contiguous().view(-1).view(torch.uint8)
            parts.append(b)
            if pad16(b.numel()) != b.numel():
                parts.append(torch.zeros(pad16(b.numel()) - b.numel(), dtype=torch.uint8, device=b.device))
        meta.append((i, str(q.dtype), tuple(q.shape), tuple(s.shape), str(z.dtype), tuple(z.shape),
                     q.numel() * q.element_size(), s.numel() * 4, z.numel() * z.element_size()))
    if parts:
        dev = parts[0].device
    else:   # a rank without work still takes part in the exchange, on the backend's device
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    flat = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    sizes[rank] = flat.numel()
    dist.all_reduce(sizes, group=group)
    cap = int(sizes.max().item())
    send = torch.zeros(max(cap, 1), dtype=torch.uint8, device=dev)
    send[: flat.numel()] = flat
    recv = [torch.empty(max(cap, 1), dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
    metas = [None] * world
    dist.all_gather_object(metas, meta, group=group)
    if dist.get_backend(group) != "nccl" and send.is_cuda:
        # gloo has no gather for device tensors (tests and the one-GPU rehearsal of bench.py): stage through the host
        send = send.cpu()
        recv = [torch.empty(max(cap, 1), dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0, group=group)
    total = int(sizes.sum().item())
    if rank != 0:
        return None, total
    out = {}
    for r in range(world):
        o = 0
        for (i, qd, qs, ss, zd, zs, nq, ns, nz) in metas[r]:
            buf = recv[r]
            q = buf[o:o + nq].view(getattr(torch, qd.split(".")[-1])).reshape(qs); o += (nq + 15) // 16 * 16
            s = buf[o:o + ns].view(torch.float32).reshape(ss); o += (ns + 15) // 16 * 16
            z = buf[o:o + nz].view(getattr(torch, zd.split(".")[-1])).reshape(zs); o += (nz + 15) // 16 * 16
            out[specs[i].name] = (q, s, z)
    return {s.name: out[s.name] for s in specs}, total


def rtn_quantize_model_sharded(specs: Sequence[LayerSpec], weights: dict, qtype: str, group_size: int, symmetric: bool = False,
                               reduce_range: bool = False, clip_ratio: float = 1.0, layout: str = "kn", *, group=None):
    """SURVEY.md 8e (1) for an RTN-configured model whose weights already live in HBM: the node-by-node walk of
    `qrules/_common.py:126-142` becomes, per rank, ONE `ops.rtn_quantize_many` call over this rank's share of the LPT plan (a
    table of pointers per shape, a few launches for the whole share) and one gather of (q, scale, zp) to rank 0.
    ``weights[i]``: the [K, N] fp32 device tensor of ``specs[i]`` (only this rank's indices are read).  Per matrix the bits are
    those of `ops.rtn_quantize`.  Returns ``({name: (q, scale, zp)} on rank 0 | None, bytes gathered)``."""
    import torch.distributed as dist

    from .hip import ops

    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    plan = plan_lpt(specs, world)
    order = plan[rank]
    results = ops.rtn_quantize_many([weights[i] for i in order], qtype, group_size, symmetric, reduce_range, clip_ratio, layout) if order else []
    return gather_device_results(specs, plan, dict(zip(order, results)), group=group)


class StreamedGather:
    """Results to rank 0 WHILE the ranks still compute, instead of one padded gather behind all of it.

    Every rank's layers are cut into bundles (`bundles[r][b]` = the layer indices rank r finishes b-th; all ranks pass the
    same structure, e.g. `wave_bundles`), and the byte layout of a layer's result is a function of its spec
    (`layout_fn(spec) -> [(dtype, shape)] * 3` for q, scale, zp), so nobody exchanges sizes or metadata:
      * rank 0 posts the `irecv`s of round b (bundle b of every other rank, buffers of exactly the bundles' sizes) when its OWN
        bundle b is done -- the peers' bundle b is due about then -- and whatever is left in `finish()`; its own results stay
        where they are;
      * rank r calls `push(b, results)` when its bundle b is done: the tensors are packed into ONE flat byte tensor (one copy)
        and `isend`-ed (NCCL over xGMI with backend nccl: asynchronous on the communicator's stream, so the next bundle's
        kernels run beside the transfer; gloo: staged through the host, for the CPU tests and the one-GPU rehearsal);
      * `finish()` waits for what is still in flight and returns ({name: (q, scale, zp)} on rank 0 | None, bytes received).
    Round 2's `gather_device_results` made three full copies of a rank's results (cat, padded send, world x cap receive
    buffers) and started after the last kernel."""

    def __init__(self, specs: Sequence[LayerSpec], bundles: list[list[list[int]]], layout_fn, *, device=None, group=None):
        import torch
        import torch.distributed as dist

        self.specs, self.bundles, self.layout_fn, self.group = specs, bundles, layout_fn, group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.nccl = self.distributed and dist.get_backend(group) == "nccl"
        self.device = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if self.nccl else torch.device("cpu"))
        self.own: dict = {}
        self.sent: list = []          # (flat tensor, work) kept alive until finish()
        self.recv: list = []          # rank 0: (rank, bundle, buffer, work)
        self.nbytes = 0
        if len(bundles) != self.world:
            raise ValueError(f"bundles for {len(bundles)} ranks, world size {self.world}")
        self.rounds = max(len(b) for b in bundles) if bundles else 0
        self.posted = 0               # rank 0: rounds whose receives have been posted

    def _post_round(self, b: int) -> None:
        """Rank 0: the receives of round `b` (bundle b of every other rank), as one grouped launch on the communicator that
        already exists (a bare irecv would create a two-rank communicator per peer on first use).  Posted when rank 0 has
        finished its OWN bundle b, not at construction: a receive kernel that spins on the GPU for a whole wave can hold the
        hardware queue a compute stream shares, and the peers' bundle b is due about now anyway."""
        import torch
        import torch.distributed as dist

        ops, bufs = [], []
        for r in range(1, self.world):
            if b < len(self.bundles[r]) and self.bundles[r][b]:
                n = self._bundle_bytes(self.bundles[r][b])
                buf = torch.empty(n, dtype=torch.uint8, device=self.device if self.nccl else torch.device("cpu"))
                ops.append(dist.P2POp(dist.irecv, buf, r, self.group))
                bufs.append((r, b, buf))
                self.nbytes += n
        if ops:
            works = dist.batch_isend_irecv(ops)
            for j, (r, bb, buf) in enumerate(bufs):
                self.recv.append((r, bb, buf, works[j] if j < len(works) else works[-1]))

    def _post_through(self, b: int) -> None:
        while self.posted <= b and self.posted < self.rounds:
            self._post_round(self.posted)
            self.posted += 1

    @staticmethod
    def _pad16(n: int) -> int:
        return (n + 15) // 16 * 16

    def _layer_layout(self, i: int):
        import torch

        out = []
        for dtype, shape in self.layout_fn(self.specs[i]):
            n = int(np.prod(shape)) if len(shape) else 1
            out.append((dtype, tuple(shape), n * torch.empty(0, dtype=dtype).element_size()))
        return out

    def _bundle_bytes(self, idx) -> int:
        return sum(self._pad16(nb) for i in idx for (_, _, nb) in self._layer_layout(i))

    def push(self, b: int, results: dict) -> None:
        """This rank's bundle `b` is complete: `results[i] = (q, scale, zp)` for i in bundles[rank][b]."""
        import torch
        import torch.distributed as dist

        idx = self.bundles[self.rank][b]
        if self.rank == 0 or self.world == 1:
            for i in idx:
                self.own[i] = results[i]
            if self.world > 1:
                self._post_through(b)
            return
        if not idx:
            return
        parts = []
        for i in idx:
            for t, (dtype, shape, nb) in zip(results[i], self._layer_layout(i)):
                if t.dtype != dtype or tuple(t.shape) != shape:
                    raise ValueError(f"{self.specs[i].name}: result {t.dtype} {tuple(t.shape)} does not match the layout {dtype} {shape}")
                flat = t.contiguous().view(-1).view(torch.uint8)
                parts.append(flat)
                if self._pad16(nb) != nb:
                    parts.append(torch.zeros(self._pad16(nb) - nb, dtype=torch.uint8, device=flat.device))
        flat = torch.cat(parts)                           # the one copy on the sending side
        if not self.nccl and flat.is_cuda:
            flat = flat.cpu()
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend, flat, 0, self.group)])
        self.sent.append((flat, works[-1]))

    def finish(self):
        """Wait for the transfers still in flight.  Returns ({name: (q, scale, zp)} on rank 0 | None, bytes received)."""
        import torch

        for _, w in self.sent:
            w.wait()
        if self.nccl and self.sent:
            torch.cuda.current_stream().synchronize()   # NCCL's wait() orders streams; the flat buffers are freed below
        self.sent.clear()
        if self.rank != 0:
            return None, self.nbytes
        if self.world > 1:
            self._post_through(self.rounds - 1)         # rounds rank 0 has no bundle of its own for
        out = {i: v for i, v in self.own.items()}
        for r, b, buf, w in self.recv:
            w.wait()
            o = 0
            for i in self.bundles[r][b]:
                got = []
                for dtype, shape, nb in self._layer_layout(i):
                    got.append(buf[o:o + nb].view(dtype).reshape(shape))
                    o += self._pad16(nb)
                out[i] = tuple(got)
        if self.nccl:
            torch.cuda.current_stream().synchronize()
        return {s.name: out[i] for i, s in enumerate(self.specs) if i in out}, self.nbytes


class PeersMissing(RuntimeError):
    """A rank did not reach the rendezvous in time; `.missing` lists the ranks that were not seen."""

    def __init__(self, msg: str, missing):
        super().__init__(msg)
        self.missing = list(missing)


def await_all_ranks(tag: str = "oq", *, timeout_s: float = 120.0, group=None, always_exchange: bool = False) -> int:
    """Bounded "is everybody here?" that does not touch the data-path communicator: every rank writes one key into the
    process group's key-value store and waits -- at most ``timeout_s`` -- for the keys of all ranks.  Returns the number of
    ranks seen (= world size) or raises `PeersMissing` naming the absent ranks.  First contact with a real multi-GPU node
    must end in a line of output or a readable error, never in a silent hang until the driver's limit:
    an RCCL collective with a missing peer blocks for as long as the watchdog allows."""
    import datetime

    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1 and not always_exchange:
        return 1
    from torch.distributed import distributed_c10d as c10d

    store = c10d._get_default_store()
    store.set(f"{tag}/here/{rank}", b"1")
    keys = [f"{tag}/here/{r}" for r in range(world)]
    try:
        store.wait(keys, datetime.timedelta(seconds=float(timeout_s)))
    except Exception as e:      # the store's timeout error type differs between back ends
        missing = []
        for r, k in enumerate(keys):
            try:
                store.wait([k], datetime.timedelta(milliseconds=50))
            except Exception:
                missing.append(r)
        raise PeersMissing(f"rank {rank}: {len(missing)} of {world} ranks did not arrive within {timeout_s:.0f} s: missing {missing}", missing) from e
    return world


def connect_to_rank0(device=None, *, group=None, timeout_s: float = 120.0) -> None:
    """One tiny grouped send / receive between rank 0 and every other rank, waited for: with NCCL the first point-to-point
    operation between two ranks sets up their connection, and the host thread that issues it blocks until the peer issues its
    side.  `StreamedGather` posts rank 0's receives before rank 0 computes anything while a peer sends only when its first
    bundle is done -- without this call rank 0 would sit in that handshake for as long as the slowest peer's first bundle
    takes.  Call it once per process group, outside any timed region (every rank must call it).

    Bounded: the ranks first meet in the key-value store (`await_all_ranks`, which names missing peers), then the
    point-to-point handshake runs on a helper thread that the caller joins for at most ``timeout_s``; if the transport
    itself does not come up the call raises (the helper thread is a daemon: the process can exit non-zero) instead of
    blocking for the communicator's own watchdog time.  Nothing is retried and nothing is re-executed."""
    import threading

    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return
    await_all_ranks("oq/connect", timeout_s=timeout_s, group=group)
    nccl = dist.get_backend(group) == "nccl"
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu"))
    if not nccl:
        dev = torch.device("cpu")
    failure: list = []

    def handshake():
        try:
            if nccl:
                torch.cuda.set_device(dev)
            if rank == 0:
                bufs = [torch.zeros(16, dtype=torch.uint8, device=dev) for _ in range(world - 1)]
                works = dist.batch_isend_irecv([dist.P2POp(dist.irecv, b, r, group) for r, b in zip(range(1, world), bufs)])
                for w in works:
                    w.wait()
                if nccl:
                    torch.cuda.current_stream().synchronize()
                bad = [r for r, b in zip(range(1, world), bufs) if int(b[0]) != r % 256]
                if bad:
                    failure.append(RuntimeError(f"connect_to_rank0: the greetings of ranks {bad} did not arrive intact"))
            else:
                hello = torch.full((16,), rank % 256, dtype=torch.uint8, device=dev)
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, hello, 0, group)]):
                    w.wait()
                if nccl:
                    torch.cuda.current_stream().synchronize()
        except Exception as e:      # surfaced on the calling thread below
            failure.append(e)

    t = threading.Thread(target=handshake, name="oq-connect", daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        peers = list(range(1, world)) if rank == 0 else [0]
        raise PeersMissing(f"rank {rank}: the point-to-point handshake with ranks {peers} over {dist.get_backend(group)} did not complete "
                           f"within {timeout_s:.0f} s (all ranks had reached the rendezvous): the transport did not come up", peers)
    if failure:
        raise failure[0]


def wave_bundles(specs: Sequence[LayerSpec], plan: list[list[int]], groups_per_wave: int) -> list[list[list[int]]]:
    """The bundles `StreamedGather` works with, for the wave structure of bench_gptq.py: a rank's layers are grouped by the
    input they share (plan order), `groups_per_wave` such groups form a wave, and a wave is one bundle."""
    out = []
    for mine in plan:
        groups, seen = [], {}
        for i in mine:
            key = specs[i].hessian_key
            if key not in seen:
                seen[key] = len(groups)
                groups.append([])
            groups[seen[key]].append(i)
        per = max(1, groups_per_wave)
        out.append([[i for g in groups[w0:w0 + per] for i in g] for w0 in range(0, len(groups), per)])
    return out


# ------------------------------------------------------------------------------------------------ inside one matrix
# groups run along K for a fixed output channel, so a matrix also shards by COLUMNS: group / channel RTN
# and the GPTQ loop need no exchange at all (every output channel is independent given the inverse factor); per-tensor RTN
# needs the global range -- one all_reduce of two floats (utils.py:42-69 over the whole array); the Hessian of one input
# can be accumulated over disjoint SAMPLES on different ranks and summed with one all_reduce of [K, K] floats
# (gptq.py:246-260: H = (2 / n) sum X^T X, n = samples).  This is what lets ONE wide layer use several GPUs.

def column_ranges(n: int, world_size: int, align: int = 32) -> list[tuple[int, int]]:
    """Contiguous column ranges [n0, n1) per rank, boundaries on multiples of `align` (32 = one wave strip of the fused
    kernels, 2 = a zero-point byte of the MatMulNBits layout); trailing ranks may be empty for narrow matrices."""
    units = -(-n // align)
    out, start = [], 0
    for r in range(world_size):
        cnt = units // world_size + (1 if r < units % world_size else 0)
        end = min(n, start + cnt * align)
        out.append((start, end))
        start = end
    return out


class HipKernels:
    """The device kernels the column-sharded drivers call (torch tensors in HBM)."""

    @staticmethod
    def rtn(w, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio):
        from .hip import ops
        return ops.rtn_quantize(w, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio)

    @staticmethod
    def minmax(w):
        """Raw (min, max) of a tensor as a 2-element fp32 tensor on its device (the calibration reduction kernel)."""
        from .hip import ops
        st = ops.minmax_state(w.device)
        ops.minmax_collect(w, st)
        return st[:2].clone()

    @staticmethod
    def quantize_tensor(w, lo, hi, qtype, symmetric, reduce_range):
        from .hip import ops
        scale, zp = ops.qparams(lo.reshape(1), hi.reshape(1), qtype, symmetric, reduce_range)
        q = ops.quantize(w, scale, zp, qtype, symmetric, reduce_range, mode="tensor")
        return q, scale.reshape(()), zp.reshape(()).to(ops.container_dtype(qtype))

    @staticmethod
    def hessian(x, h, n_seen):
        from .hip import ops
        return ops.hessian_accumulate(x, h, n_seen)

    @staticmethod
    def gptq(w, h, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio, block_size, percdamp, actorder, mode):
        from .hip import ops
        q, s, z, _ = ops.gptq_quantize(w, h, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio, block_size, percdamp,
                                       actorder, False, mode=mode)
        return q, s, z


def _dist(group):
    import torch.distributed as dist

    on = dist.is_available() and dist.is_initialized()
    return dist, on, (dist.get_world_size(group) if on else 1), (dist.get_rank(group) if on else 0)


def rtn_quantize_column_shard(w_cols, qtype: str, strategy: str, group_size=-1, symmetric=False, reduce_range=False,
                              clip_ratio: float = 1.0, *, group=None, kernels=HipKernels):
    """rtn.py:54-109 for this rank's COLUMNS ``w_cols`` [K, n_local] of a matrix whose other columns live on other ranks.
    group / channel: purely local.  tensor: the raw (min, max) are combined with ONE all_reduce of two floats
    ((-min, max) under MAX), then every rank derives the same (scale, zp) and quantizes its columns.
    Returns the local (q [K, n_local], scale, zp); scale / zp are the global 0-d ones for "tensor"."""
    import torch

    dist, on, world, _ = _dist(group)
    if strategy != "tensor":
        return kernels.rtn(w_cols, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio)
    mm = kernels.minmax(w_cols) if w_cols.numel() else torch.tensor([float("inf"), float("-inf")], dtype=torch.float32, device=w_cols.device)
    both = torch.stack([-mm[0], mm[1]])
    if on and world > 1:
        dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)
    zero = torch.zeros((), dtype=torch.float32, device=both.device)
    lo = torch.minimum(-both[0] * clip_ratio, zero)                       # utils.py:63-69
    hi = torch.maximum(both[1] * clip_ratio, zero)
    return kernels.quantize_tensor(w_cols, lo, hi, qtype, symmetric, reduce_range)


def hessian_all_reduce(h_local, n_local: int, *, group=None, always_exchange: bool = False):
    """Hessians accumulated on disjoint samples (gptq.py:246-260 on each rank's batches: H_r = (2 / n_r) sum X^T X) ->
    the Hessian of all samples, H = sum_r (n_r / n) H_r, on every rank: one all_reduce(sum) of [K, K] floats (plus one of
    the sample counts).  Returns (H, n).  ``always_exchange``: as in `gather_device_results`."""
    import torch

    dist, on, world, _ = _dist(group)
    if not on or (world == 1 and not always_exchange):
        return h_local, int(n_local)
    n = torch.tensor([float(n_local)], dtype=torch.float64, device=h_local.device)
    dist.all_reduce(n, group=group)
    total = int(n.item())
    h = h_local * (float(n_local) / total) if total else h_local.clone()
    dist.all_reduce(h, group=group)
    return h, total


def gptq_quantize_column_shard(w_cols, x_batches, qtype: str, strategy: str, group_size, symmetric=False, reduce_range=False,
                               clip_ratio: float = 1.0, block_size: int = 128, percdamp: float = 0.01, actorder: bool = False,
                               mode: str = "parity", *, group=None, kernels=HipKernels):
    """gptq.py:263-324 with the layer spread over the ranks of `group`: every rank holds the columns ``w_cols``
    [K, n_local] and accumulates the Hessian of ITS calibration batches ``x_batches`` (disjoint samples, each
    [n_b, ..., K]); one all_reduce gives every rank the full Hessian, the factor is computed redundantly (it is a serial
    chain) and the loop runs on the local columns without any exchange.  Strategies channel / group (tensor would need
    the global range twice: use `rtn_quantize_column_shard` for per-tensor weights)."""
    import torch

    if strategy == "tensor":
        raise NotImplementedError("column-sharded GPTQ supports the channel and group strategies")
    k = w_cols.shape[0]
    h = torch.zeros((k, k), dtype=torch.float32, device=w_cols.device)
    n = 0
    for x in x_batches:
        n = kernels.hessian(x, h, n)
    h, _ = hessian_all_reduce(h, n, group=group)
    if w_cols.shape[1] == 0:
        return None
    return kernels.gptq(w_cols, h, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio, block_size, percdamp, actorder, mode)


def gather_column_shards(local, ranges, strategy: str, *, group=None):
    """Bring the column shards of ONE matrix to rank 0 and put them together: q [K, N] along the columns, channel
    parameters [N] and group parameters [N*K/g, 1] (out-channel major, rtn.py:98-109) along their first axis, per-tensor
    parameters as they are.  ``local`` = (q, scale, zp) of this rank or None for an empty range.  Returns (q, scale, zp) on
    rank 0, None elsewhere.  One padded gather (RCCL with backend nccl), like `gather_device_results`."""
    import torch

    dist, on, world, rank = _dist(group)
    if world == 1:
        return local
    nonempty = [r for r in range(world) if ranges[r][1] > ranges[r][0]]
    index = {r: i for i, r in enumerate(nonempty)}
    specs = [LayerSpec(f"cols[{ranges[r][0]}:{ranges[r][1]}]", 0, ranges[r][1] - ranges[r][0]) for r in nonempty]
    plan = [[index[r]] if r in index else [] for r in range(world)]
    got, _ = gather_device_results(specs, plan, {index[rank]: local} if rank in index else {}, group=group)
    if rank != 0:
        return None
    parts = [got[s.name] for s in specs]
    q = torch.cat([p[0] for p in parts], dim=1)
    if strategy == "tensor":
        return q, parts[0][1], parts[0][2]
    return q, torch.cat([p[1] for p in parts], dim=0), torch.cat([p[2] for p in parts], dim=0)


def collectives_selftest(device, *, group=None, timeout_s: float = 90.0, self_p2p: bool = True) -> dict:
    """Every kind of exchange the multi-rank path issues, once, on DEVICE tensors, in the process group that is up -- which
    may have ONE rank: with backend "nccl" that is a real RCCL communicator on a one-GPU box (before this,
    the first `nccl` call of this code base would have happened on the driver's 8-GPU node).  Steps, each checked and timed:
    the key-value rendezvous (`await_all_ranks`), the all_reduces the benches use (int32 SUM, float64 MAX, int64 sizes),
    `all_gather_object`, the padded gather of real kernel results (`gather_device_results`: pack, size exchange, gather, unpack),
    the Hessian all_reduce (`hessian_all_reduce`), `StreamedGather.push / finish`, the rank-0 handshake (`connect_to_rank0`; in
    a group of one nccl rank a grouped send + receive of rank 0 with itself, the same `batch_isend_irecv` call), a barrier.
    Runs on a helper thread joined for at most ``timeout_s`` (a transport that does not come up must end in a message).
    Returns {"ok": bool, "backend", "world", "steps": {name: ms}, "error": str | None}; never raises for a failing step."""
    import threading
    import time

    import torch
    import torch.distributed as dist

    rec = {"ok": False, "backend": None, "world": 1, "steps": {}, "error": None}
    if not (dist.is_available() and dist.is_initialized()):
        rec["error"] = "no process group is up"
        return rec
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rec["backend"], rec["world"] = dist.get_backend(group), world
    on_device = device is not None and torch.device(device).type == "cuda"
    comm_dev = torch.device(device) if rec["backend"] == "nccl" else torch.device("cpu")

    def timed(name, fn):
        t0 = time.perf_counter()
        fn()
        if on_device:
            torch.cuda.synchronize(device)
        rec["steps"][name] = round((time.perf_counter() - t0) * 1e3, 2)

    def body():
        if on_device:
            torch.cuda.set_device(device)

        def reduces():
            a = torch.ones(1, dtype=torch.int32, device=comm_dev)
            dist.all_reduce(a, group=group)
            b = torch.tensor([1.5 + rank, 2.5], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(b, op=dist.ReduceOp.MAX, group=group)
            c = torch.zeros(world, dtype=torch.int64, device=comm_dev)
            c[rank] = 1000 + rank
            dist.all_reduce(c, group=group)
            assert int(a.item()) == world and float(b[0]) == 0.5 + world and c.tolist() == [1000 + r for r in range(world)]

        def objects():
            got = [None] * world
            dist.all_gather_object(got, ("rank", rank, (1, 2)), group=group)
            assert got == [("rank", r, (1, 2)) for r in range(world)]

        def padded_gather():
            gen = torch.Generator(device="cpu").manual_seed(100 + rank)
            specs = [LayerSpec(f"selftest.r{r}.w{j}", 64 * (j + 1), 96) for r in range(world) for j in range(2)]
            plan = [[2 * r, 2 * r + 1] for r in range(world)]
            mine = {}
            for i in plan[rank]:
                k, n = specs[i].k, specs[i].n
                mine[i] = (torch.randint(0, 255, (k, n), generator=gen, dtype=torch.uint8).to(device if on_device else "cpu"),
                           torch.rand((n * k // 32, 1), generator=gen).to(device if on_device else "cpu"),
                           torch.randint(0, 15, (n * k // 32 + 3,), generator=gen, dtype=torch.uint8).to(device if on_device else "cpu"))   # odd length: the 16-byte padding
            got, nbytes = gather_device_results(specs, plan, mine, group=group, always_exchange=True)
            if rank == 0:
                assert nbytes > 0 and list(got) == [s_.name for s_ in specs]
                for i in plan[0]:
                    assert all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(got[specs[i].name], mine[i]))

        def hessian():
            h = torch.full((128, 128), float(rank + 1), dtype=torch.float32, device=comm_dev)
            out, n = hessian_all_reduce(h, 4 * (rank + 1), group=group, always_exchange=True)
            total = sum(4 * (r + 1) for r in range(world))
            want = sum((4 * (r + 1) / total) * (r + 1) for r in range(world))
            assert n == total and abs(float(out[5, 7]) - want) < 1e-5

        def streamed():
            specs = [LayerSpec(f"selftest.s{r}", 64, 32) for r in range(world)]
            bundles = [[[r]] for r in range(world)]
            layout = lambda sp: [(torch.uint8, (sp.k, sp.n)), (torch.float32, (sp.n, 1)), (torch.uint8, (sp.n,))]   # noqa: E731
            sg = StreamedGather(specs, bundles, layout, device=comm_dev, group=group)
            res = (torch.full((64, 32), rank, dtype=torch.uint8, device=comm_dev), torch.full((32, 1), float(rank), device=comm_dev),
                   torch.full((32,), rank, dtype=torch.uint8, device=comm_dev))
            sg.push(0, {rank: res})
            got, _ = sg.finish()
            if rank == 0:
                assert [int(got[f"selftest.s{r}"][0][0, 0]) for r in range(world)] == list(range(world))

        def handshake():
            if world > 1:
                connect_to_rank0(comm_dev, group=group, timeout_s=timeout_s)
                return
            src = torch.arange(4096, dtype=torch.uint8, device=comm_dev)
            dst = torch.zeros(4096, dtype=torch.uint8, device=comm_dev)
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0, group), dist.P2POp(dist.irecv, dst, 0, group)]):
                w.wait()
            if on_device:
                torch.cuda.synchronize(device)
            assert torch.equal(src, dst)

        try:
            timed("store_rendezvous", lambda: await_all_ranks("oq/selftest", timeout_s=timeout_s, group=group, always_exchange=True))
            timed("all_reduce_x3", reduces)
            timed("all_gather_object", objects)
            timed("padded_gather", padded_gather)
            timed("hessian_all_reduce", hessian)
            timed("streamed_gather", streamed)
            if world > 1:
                timed("rank0_handshake", handshake)
            elif self_p2p and rec["backend"] == "nccl":
                # no rank of a real run sends to itself: a transport that refuses this is noted, not counted as a failure
                try:
                    timed("self_send_recv", handshake)
                except Exception as e:      # noqa: BLE001
                    rec["self_send_recv_error"] = f"{type(e).__name__}: {e}"
            timed("barrier", lambda: dist.barrier(group=group))
            rec["ok"] = True
        except BaseException as e:      # noqa: BLE001 -- reported in the record
            import traceback
            rec["error"] = f"{type(e).__name__}: {e}"
            rec["traceback"] = traceback.format_exc()[-1500:]

    t = threading.Thread(target=body, name="oq-selftest", daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        rec["error"] = f"no answer within {timeout_s:.0f} s after steps {list(rec['steps'])}"
    return rec
