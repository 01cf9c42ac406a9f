"""SURVEY.md 8e on the file path (`model_quantize.quantize_model_sharded`): the weights of one ONNX model over the ranks of a process
group, one gather at the end, rank 0 emits.  The file must be the single-process file, byte for byte.

* CPU: two and three gloo ranks, the oracle as numeric provider (the orchestration is what is under test).
* GPU: two ranks sharing cuda:0 over gloo with the product's providers, weight-only and GPTQ (every rank calibrates for itself,
  nodes that read one value stay on one rank).
"""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _configs():
    from onnx_quantize_amd import AwqConfig, GPTQConfig, QActivationArgs, QConfig, QuantType, QWeightArgs
    import numpy as np
    data = np.random.default_rng(5).standard_normal((16, 6, 64)).astype(np.float32)
    return {
        "uint4_g32": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=32)),
        "int8_channel": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, group_size=-1), ignore=["/v/"]),
        "static_int8": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8), input_activations=QActivationArgs(dtype=QuantType.QInt8),
                                       output_activations=QActivationArgs(dtype=QuantType.QInt8), calibration_data=data,
                                       calibration_params={"num_samples": 16, "batch_size": 4}),
        # the rescaled weights live in HBM only on every rank (`_Graph.pending_host`); int4: the grouped QDQ route, packed nibbles
        "awq_clip_int4_g32": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt4, group_size=32), preprocessors=[AwqConfig(clip_search=True)],
                                             calibration_data=data, calibration_params={"num_samples": 16, "batch_size": 4}),
        "gptq_int4_g32": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt4, group_size=32, algorithm=GPTQConfig(block_size=16)),
                                         calibration_data=data, calibration_params={"num_samples": 16, "batch_size": 4}),
    }


def _worker(rank, world, port, q, device):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from onnx_quantize_amd import onnx_proto as P
    from onnx_quantize_amd.model_quantize import quantize_model, quantize_model_sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if device == "cuda":
        torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = P.load_model(os.path.join(ROOT, "tests", "golden", "onnx", "block.onnx"))
    providers = {}
    if device == "cpu":
        import oq_oracle as O
        from onnx_model_helpers import OracleSearches, oracle_calibrate, oracle_weight_arrays
        providers = dict(weight_arrays=oracle_weight_arrays, quantize_bias=O.quantize_bias, calibrate=oracle_calibrate("cpu"), searches=OracleSearches())
    seen = []
    if "weight_arrays" in providers:
        inner = providers["weight_arrays"]

        def counting(value, cfg, out, nbits):
            seen.append(value.name)
            return inner(value, cfg, out, nbits)
        providers["weight_arrays"] = counting
    results = {}
    for name, make in _configs().items():
        del seen[:]
        out = quantize_model_sharded(model, make(), device=device, **providers)
        mine = len(seen)
        if rank == 0:
            single = quantize_model(model, make(), device=device, **({k: v for k, v in providers.items()} if device == "cpu" else {}))
            same = P.serialize(out) == P.serialize(single)
            if not same and name.startswith("gptq") and device == "cuda":
                a = {t.name: P.tensor_to_numpy(t) for t in out.graph.initializer}
                b = {t.name: P.tensor_to_numpy(t) for t in single.graph.initializer}
                same = set(a) == set(b) and all((a[k] != b[k]).mean() < 0.01 if a[k].dtype.kind == "i" else True for k in a)
            results[name] = (bool(same), mine)
        else:
            assert out is None
            results[name] = (None, mine)
    gathered = [None] * world
    dist.all_gather_object(gathered, results)
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, device):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, device)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
        assert p.exitcode == 0
    return q.get(timeout=10)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_file_equals_the_single_process_file_gloo(world):
    gathered = _run(world, "cpu")
    for name in gathered[0]:
        assert gathered[0][name][0] is True, name
        counts = [g[name][1] for g in gathered]
        # every rank ran the seam for its share only (rank 0 also runs the single-process reference afterwards: not counted here)
        assert all(c > 0 for c in counts) and sum(counts) >= 5, (name, counts)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_sharded_file_equals_the_single_process_file_two_ranks_on_one_gpu():
    gathered = _run(2, "cuda")
    for name in gathered[0]:
        assert gathered[0][name][0] is True, name
