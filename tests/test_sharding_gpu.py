"""SURVEY.md 8e (2) with the product's own kernels (sharding.HipKernels) and more than one rank: the ranks share cuda:0 and
talk over gloo (a one-GPU box; the gather is staged through the host there), the results are compared on rank 0 with the
UNSHARDED GPU path -- integers and parameters bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from onnx_quantize_amd import sharding as S
    from onnx_quantize_amd.hip import ops

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, n = 1024, 352                                                  # 11 strips of 32 columns: uneven over 2 / 3 ranks
    rng = np.random.default_rng(23)
    w_np = rng.standard_normal((k, n)).astype(np.float32)
    w_np[:, 330:] *= 7                                                # the global range lives on the last rank only
    x_np = (rng.standard_normal((6, 96, k)) * rng.uniform(0.3, 3, size=k)).astype(np.float32)
    w = torch.from_numpy(w_np).cuda()
    ranges = S.column_ranges(n, world, 32)
    a, b = ranges[rank]
    w_cols = w[:, a:b].contiguous()
    ok = []
    for qtype, strategy, g, sym in (("uint4", "group", 128, False), ("int8", "channel", -1, True), ("uint8", "tensor", -1, False),
                                    ("int8", "tensor", -1, True)):
        local = S.rtn_quantize_column_shard(w_cols, qtype, strategy, g, sym, False, 0.9)
        whole = S.gather_column_shards(local, ranges, strategy)
        if rank == 0:
            eq, es, ez = ops.rtn_quantize(w, qtype, strategy, g, sym, False, 0.9)
            ok.append(torch.equal(whole[0].cuda(), eq) and torch.equal(whole[1].cuda().reshape(es.shape), es)
                      and torch.equal(whole[2].cuda().reshape(ez.shape), ez))
    mine = [torch.from_numpy(x_np[i:i + 1]).cuda() for i in range(rank, 6, world)]      # samples dealt round robin
    local = S.gptq_quantize_column_shard(w_cols, mine, "int4", "group", 128)
    whole = S.gather_column_shards(local, ranges, "group")
    if rank == 0:
        h = torch.zeros((k, k), device="cuda")
        ops.hessian_accumulate(torch.from_numpy(x_np).cuda(), h, 0)
        eq, es, ez, _ = ops.gptq_quantize(w, h, "int4", "group", 128)
        ok.append(torch.equal(whole[0].cuda(), eq) and torch.equal(whole[2].cuda().reshape(ez.shape), ez))
        ok.append(bool(torch.allclose(whole[1].cuda().reshape(es.shape), es, rtol=1e-6, atol=0)))
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_one_matrix_sharded_by_columns_on_the_gpu(world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = q.get(timeout=5)
    assert res == [True] * len(res) and len(res) == 6, res


def _model_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from onnx_quantize_amd import sharding as S
    from onnx_quantize_amd.hip import ops

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [(512, 768), (512, 768), (768, 512), (256, 1024), (512, 768), (768, 512), (256, 1024)]
    specs = [S.LayerSpec(name=f"l{i}", k=k, n=n, tokens=0, hessian_key=f"l{i}") for i, (k, n) in enumerate(shapes)]
    g = torch.Generator(device="cuda").manual_seed(31)
    weights = {i: torch.randn(kn, generator=g, device="cuda") * (0.5 + i) for i, kn in enumerate(shapes)}   # every rank: the same model
    out, nbytes = S.rtn_quantize_model_sharded(specs, weights, "uint4", 128, layout="nbits")
    if rank == 0:
        ok = []
        for i, sp in enumerate(specs):
            eq, es, ez = ops.rtn_quantize(weights[i], "uint4", "group", 128, layout="nbits")
            gq, gs, gz = out[sp.name]
            ok.append(bool(torch.equal(gq.cuda().reshape(eq.shape), eq) and torch.equal(gs.cuda().reshape(es.shape), es)
                           and torch.equal(gz.cuda().reshape(ez.shape), ez)))
        q.put((ok, nbytes > 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_rtn_model_sharded_by_layers_uses_one_list_call_per_rank(world):
    """sharding.rtn_quantize_model_sharded: every rank quantizes its LPT share with one ops.rtn_quantize_many call, rank 0
    receives all layers -- bit-equal to the per-matrix kernel."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    ok, moved = q.get(timeout=5)
    assert ok == [True] * 7 and moved


@pytest.mark.gpu
def test_rccl_one_rank_communicator_runs_every_exchange_of_the_multi_rank_path():
    """First contact with RCCL: a child process brings up `init_process_group("nccl", world_size=1,
    device_id=cuda:0)` before any other GPU call and runs `sharding.collectives_selftest` on device tensors: the rendezvous,
    the all_reduces, `all_gather_object`, the padded gather of `gather_device_results` (forced through the collectives),
    the Hessian all_reduce, `StreamedGather.push / finish`, a barrier; then tears the group down.  Exit code 0 and ok = true."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--nccl-selftest"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"rccl_selftest"')][0])["rccl_selftest"]
    assert rec["ok"] and rec["backend"] == "nccl" and rec["world"] == 1 and rec["error"] is None, rec
    assert {"all_reduce_x3", "all_gather_object", "padded_gather", "hessian_all_reduce", "streamed_gather", "barrier"} <= set(rec["steps"])
