"""Multi-GPU row: the cost-balanced plan and the end-of-run gather, exercised with two
gloo ranks on the CPU (the per-layer worker here is the oracle: the orchestration is what is under test)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT
from onnx_quantize_amd.sharding import LayerSpec, layer_cost, llama2_7b_specs, plan_lpt


def test_plan_covers_everything_once_and_balances():
    specs = llama2_7b_specs(tokens=2048 * 128)
    assert len(specs) == 32 * 7
    assert sum(s.k * s.n for s in specs) == 6_476_005_376          # 6.476 B params
    for world in (1, 2, 4, 8):
        plan = plan_lpt(specs, world)
        flat = sorted(i for p in plan for i in p)
        assert flat == list(range(len(specs)))
        loads = []
        for p in plan:
            seen, c = set(), 0.0
            for i in p:
                c += layer_cost(specs[i], specs[i].hessian_key in seen)
                seen.add(specs[i].hessian_key)
            loads.append(c)
        assert max(loads) / (sum(loads) / world) < 1.05
        for p in plan:                      # layers sharing a Hessian stay together
            keys = {specs[i].hessian_key for i in p}
            for k in keys:
                assert all(i in p for i, s in enumerate(specs) if s.hessian_key == k)


def test_plan_more_ranks_than_layers():
    specs = [LayerSpec("a", 64, 64), LayerSpec("b", 64, 128)]
    plan = plan_lpt(specs, 4)
    assert sorted(i for p in plan for i in p) == [0, 1] and sum(1 for p in plan if p) == 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist

    import oq_oracle as O
    from onnx_quantize_amd.sharding import LayerSpec, plan_lpt, quantize_sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = [LayerSpec(f"l{i}", 32 * (1 + i % 3), 16 * (1 + i % 4)) for i in range(7)]
    done = []

    def fn(i, spec):
        done.append(i)
        w = np.random.default_rng(i).standard_normal((spec.k, spec.n), dtype=np.float32)
        return O.rtn_quantize(w, "uint4", "group", 16)
    res = quantize_sharded(specs, fn)
    assert sorted(done) == plan_lpt(specs, world)[rank]
    if rank == 0:
        ok = list(res) == [s.name for s in specs]
        for i, s in enumerate(specs):
            w = np.random.default_rng(i).standard_normal((s.k, s.n), dtype=np.float32)
            eq, es, ez = O.rtn_quantize(w, "uint4", "group", 16)
            gq, gs, gz = res[s.name]
            ok = ok and np.array_equal(gq, eq) and gs.tobytes() == es.tobytes() and np.array_equal(gz, ez)
            ok = ok and gq.dtype == eq.dtype and gs.shape == es.shape and gz.shape == ez.shape
        q.put(bool(ok))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gather_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker_device_gather(rank, world, port, q):
    """gather_device_results (the exchange bench_gptq.py uses with RCCL) over gloo with CPU tensors: odd byte counts,
    a rank with nothing to send, dtypes and shapes restored on rank 0."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from onnx_quantize_amd.sharding import LayerSpec, gather_device_results

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = [LayerSpec(f"m{i}", 8 + i, 5 + 2 * i) for i in range(5)]
    plan = [[0, 2, 3, 4], [1]] if world == 2 else [[0, 1, 2, 3, 4], [], []][:world]

    def result(i):
        g = torch.Generator().manual_seed(100 + i)
        k, n = specs[i].k, specs[i].n
        return (torch.randint(0, 255, ((k * n + 1) // 2,), generator=g, dtype=torch.uint8),     # odd byte counts
                torch.rand((n * 3, 1), generator=g), torch.randint(-8, 7, (n * 3, 1), generator=g, dtype=torch.int8))
    mine = {i: result(i) for i in plan[rank]}
    out, nbytes = gather_device_results(specs, plan, mine)
    if rank == 0:
        ok = list(out) == [s.name for s in specs] and nbytes > 0
        for i, s in enumerate(specs):
            e = result(i)
            ok = ok and all(torch.equal(a, b) and a.dtype == b.dtype and a.shape == b.shape for a, b in zip(out[s.name], e))
        q.put(bool(ok))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world", [2, 3])
def test_device_result_gather_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_device_gather, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_single_process_path():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oq_oracle as O
    from onnx_quantize_amd.sharding import quantize_sharded
    specs = [LayerSpec("x", 32, 16), LayerSpec("y", 64, 8)]
    res = quantize_sharded(specs, lambda i, s: O.rtn_quantize(np.ones((s.k, s.n), np.float32) * (i + 1), "int8", "channel"))
    assert list(res) == ["x", "y"] and res["y"][1].shape == (8,)


# ------------------------------------------------------------------------------------------------ inside one matrix
def test_column_ranges_are_aligned_contiguous_and_complete():
    from onnx_quantize_amd.sharding import column_ranges
    for n, world, align in ((11008, 8, 32), (4096, 3, 32), (100, 4, 32), (24, 8, 2), (33, 2, 32)):
        r = column_ranges(n, world, align)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(r, r[1:])) and all(a % align == 0 for a, _ in r if a < n)
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(s for s in sizes if s or True) <= align or min(sizes) == 0


class _OracleKernels:
    """The kernels interface of sharding.HipKernels on CPU torch tensors through the oracle: what is under test is the
    orchestration (who reduces what, when), exactly as with `quantize_sharded` above."""

    @staticmethod
    def _t(*arrs):
        import torch
        return tuple(torch.from_numpy(np.array(a)) for a in arrs)        # np.array keeps 0-d parameters 0-d

    @classmethod
    def rtn(cls, w, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio):
        import oq_oracle as O
        return cls._t(*O.rtn_quantize(w.numpy(), qtype, strategy, group_size, symmetric, reduce_range, clip_ratio))

    @staticmethod
    def minmax(w):
        import torch
        return torch.stack([w.min(), w.max()])

    @classmethod
    def quantize_tensor(cls, w, lo, hi, qtype, symmetric, reduce_range):
        import oq_oracle as O
        s, z = O.qparams(lo.numpy(), hi.numpy(), qtype, symmetric, reduce_range)
        q = O.quantize(w.numpy(), s, z, qtype, symmetric, reduce_range)
        return cls._t(q, np.asarray(s, np.float32), np.asarray(z))

    @staticmethod
    def hessian(x, h, n_seen):
        import torch
        import oq_oracle as O
        hn, n = O.accumulate_hessian(x.numpy(), h.numpy().copy(), n_seen)
        h.copy_(torch.from_numpy(hn))
        return n

    @classmethod
    def gptq(cls, w, h, qtype, strategy, group_size, symmetric, reduce_range, clip_ratio, block_size, percdamp, actorder, mode):
        import oq_oracle as O
        return cls._t(*O.gptq(w.numpy(), h.numpy(), qtype, strategy, group_size, symmetric, reduce_range, clip_ratio, block_size, percdamp,
                              actorder, False, mode=mode))


def _worker_columns(rank, world, port, q):
    """One 96 x 80 matrix spread by columns over the ranks: RTN group / channel (no exchange), RTN tensor (one all_reduce of
    two floats) and GPTQ with the Hessian accumulated on disjoint samples per rank (one all_reduce of K x K floats),
    gathered on rank 0 and compared with the UNSHARDED oracle."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import oq_oracle as O
    from onnx_quantize_amd import sharding as S
    from test_sharding import _OracleKernels as KER

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, n = 96, 80
    rng = np.random.default_rng(11)
    w = rng.standard_normal((k, n)).astype(np.float32)
    w[:, 70:] *= 5                                                  # the global range lives on the last rank only
    x = (rng.standard_normal((6, 7, k)) * rng.uniform(0.3, 3, size=k)).astype(np.float32)
    ranges = S.column_ranges(n, world, 32)
    a, b = ranges[rank]
    w_cols = torch.from_numpy(np.ascontiguousarray(w[:, a:b]))
    ok = True
    for qtype, strategy, g, sym in (("uint4", "group", 32, False), ("int8", "channel", -1, True), ("uint8", "tensor", -1, False),
                                    ("int8", "tensor", -1, True)):
        local = S.rtn_quantize_column_shard(w_cols, qtype, strategy, g, sym, False, 0.9, kernels=KER) if b > a or strategy == "tensor" else None
        whole = S.gather_column_shards(local if b > a else None, ranges, strategy)
        if rank == 0:
            eq, es, ez = O.rtn_quantize(w, qtype, strategy, g, sym, False, 0.9)
            ok = ok and np.array_equal(whole[0].numpy(), eq) and whole[1].numpy().tobytes() == np.asarray(es).tobytes()
            ok = ok and np.array_equal(whole[2].numpy(), ez) and whole[1].shape == np.shape(es)
    # GPTQ: samples 0..5 dealt round robin, columns as above
    mine = [torch.from_numpy(x[i:i + 1]) for i in range(rank, 6, world)]
    local = S.gptq_quantize_column_shard(w_cols, mine, "int4", "group", 32, block_size=32, kernels=KER)
    whole = S.gather_column_shards(local, ranges, "group")
    h_all, n_all = S.hessian_all_reduce(torch.zeros((k, k)), 0)      # ranks without samples still take part
    if rank == 0:
        eq, es, ez = O.gptq_quantize(w, x, "int4", "group", 32, block_size=32)
        ok = ok and np.array_equal(whole[0].numpy(), eq) and np.array_equal(whole[2].numpy(), ez)
        ok = ok and np.allclose(whole[1].numpy(), es, rtol=1e-6) and n_all == 0
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 3, 4])
def test_one_matrix_sharded_by_columns_gloo(world):
    """SURVEY.md 8e (2).  World 4 leaves the last rank without columns (80 columns = 3 strips of 32)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_columns, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker_streamed_gather(rank, world, port, q):
    """StreamedGather over gloo: bundles pushed as they finish, odd byte counts, a rank without work, an empty bundle; what
    rank 0 assembles equals what every rank produced (and what the end-of-run gather_device_results returns)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from onnx_quantize_amd.sharding import LayerSpec, StreamedGather, connect_to_rank0, gather_device_results, wave_bundles

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    connect_to_rank0()                      # the handshake bench_gptq does before its clock starts; every rank takes part
    specs = [LayerSpec(f"m{i}", 8 + i, 5 + 2 * i, hessian_key=f"h{i // 2}") for i in range(9)]
    plan = [[0, 1, 4, 5, 8], [2, 3, 6, 7]] if world == 2 else [[0, 1, 6, 7], [2, 3, 4, 5, 8], []]
    bundles = wave_bundles(specs, plan, groups_per_wave=1)
    assert [len(b) for b in bundles] == ([3, 2] if world == 2 else [2, 3, 0])

    def layout(sp):
        return [(torch.uint8, ((sp.k * sp.n + 1) // 2,)), (torch.float32, (sp.n * 3, 1)), (torch.int8, (sp.n * 3, 1))]

    def result(i):
        g = torch.Generator().manual_seed(100 + i)
        k, n = specs[i].k, specs[i].n
        return (torch.randint(0, 255, ((k * n + 1) // 2,), generator=g, dtype=torch.uint8),
                torch.rand((n * 3, 1), generator=g), torch.randint(-8, 7, (n * 3, 1), generator=g, dtype=torch.int8))
    sg = StreamedGather(specs, bundles, layout)
    mine = {}
    for b, idx in enumerate(bundles[rank]):
        res = {i: result(i) for i in idx}
        mine.update(res)
        sg.push(b, res)
    out, nbytes = sg.finish()
    ref, _ = gather_device_results(specs, plan, mine)
    if rank == 0:
        ok = list(out) == [s.name for s in specs] and nbytes > 0
        for i, s in enumerate(specs):
            e = result(i)
            ok = ok and all(torch.equal(a, b) and a.dtype == b.dtype and a.shape == b.shape for a, b in zip(out[s.name], e))
            ok = ok and all(torch.equal(a, b) for a, b in zip(out[s.name], ref[s.name]))
        q.put(bool(ok))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world", [2, 3])
def test_streamed_gather_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_streamed_gather, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_streamed_gather_single_rank_and_layout_check():
    import torch
    from onnx_quantize_amd.sharding import StreamedGather, wave_bundles
    specs = [LayerSpec("a", 8, 4), LayerSpec("b", 8, 6)]
    layout = lambda sp: [(torch.uint8, (sp.k * sp.n,)), (torch.float32, (sp.n,)), (torch.int8, (sp.n,))]  # noqa: E731
    bundles = wave_bundles(specs, [[0, 1]], 1)
    sg = StreamedGather(specs, bundles, layout)
    res = {i: (torch.zeros(sp.k * sp.n, dtype=torch.uint8), torch.ones(sp.n), torch.zeros(sp.n, dtype=torch.int8)) for i, sp in enumerate(specs)}
    for b, idx in enumerate(bundles[0]):
        sg.push(b, {i: res[i] for i in idx})
    out, nbytes = sg.finish()
    assert list(out) == ["a", "b"] and nbytes == 0 and out["b"][1].shape == (6,)


def _worker_missing_peer(rank, world, port, q):
    """Rank 1 never reaches the rendezvous: rank 0's bounded wait names it instead of hanging."""
    sys.path.insert(0, ROOT)
    import time

    import torch.distributed as dist

    from onnx_quantize_amd.sharding import PeersMissing, await_all_ranks, connect_to_rank0

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:
        t0 = time.time()
        try:
            await_all_ranks("t/missing", timeout_s=2.0)
            q.put(("no error", None, 0.0))
        except PeersMissing as e:
            q.put(("missing", e.missing, time.time() - t0))
        try:
            connect_to_rank0(timeout_s=2.0)          # the same bounded wait in front of the point-to-point handshake
            q.put(("no error", None, 0.0))
        except PeersMissing as e:
            q.put(("missing", e.missing, 0.0))
        await_all_ranks("t/late", timeout_s=60.0)    # everybody arrives eventually: returns the world size
        q.put(("seen", await_all_ranks("t/again", timeout_s=60.0), 0.0))
    else:
        time.sleep(8.0)
        await_all_ranks("t/late", timeout_s=60.0)
        await_all_ranks("t/again", timeout_s=60.0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bounded_rendezvous_names_the_missing_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_missing_peer, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    first = q.get(timeout=60)
    assert first[0] == "missing" and first[1] == [1] and first[2] < 6.0, first
    second = q.get(timeout=60)
    assert second[0] == "missing" and second[1] == [1], second
    assert q.get(timeout=90) == ("seen", 2, 0.0)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
