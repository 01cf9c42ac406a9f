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
