#!/usr/bin/env python3
"""End-to-end file path: an ONNX file of Llama-2-7B-shaped decoder MLP / projection weights
-> `quantize_file` -> an ONNX file, on one GPU, with the phases timed apart.

This is NOT the headline bench (bench.py times the kernels with inputs resident in HBM).  It is the PCIe- and file-
inclusive rate of the path a user of the reference's `quantize(model, qconfig)` takes: parse + memory-map the source, one
upload per weight, the kernels, one download of the wire-format arrays, serialisation.  The model is synthetic (random
weights, `--layers` decoder layers of 4096 / 11008 width: q, k, v, o, gate, up, down as MatMuls joined by elementwise
operators so that the graph runs), written by this package's own writer with its tensors in a side file.

    python bench_model.py --layers 4 --config uint4_g128          # weight-only, MatMulNBits (BASELINE config 2's rule)
    python bench_model.py --layers 2 --config static_int8         # calibrated on the GPU (config 3's rule), random data
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 bench_model.py --layers 32 --config gptq_int4_g128
                                                                  # one rank per GPU: `quantize_model_sharded` (BASELINE config 5's shape:
                                                                  # every rank maps the file and walks it, the weights are spread, rank 0
                                                                  # gathers and writes).  OQ_BENCH_REHEARSAL=1: the ranks share cuda:0 over
                                                                  # gloo (the code path on a one-GPU box, not a scaling measurement)
"""
import argparse
import json
import os
import shutil
import tempfile
import time

import numpy as np
import torch

from onnx_quantize_amd import AwqConfig, GPTQConfig, QActivationArgs, QConfig, QuantType, QWeightArgs
from onnx_quantize_amd import onnx_proto as P
from onnx_quantize_amd.model_quantize import quantize_model


def build_model(layers: int, hidden: int, ffn: int, seed: int = 0) -> P.Message:
    gen = torch.Generator().manual_seed(seed)
    nodes, inits = [], []

    def weight(name, k, n):
        w = (torch.randn(k, n, generator=gen) * (1.0 / k ** 0.5)).numpy()
        inits.append(P.numpy_to_tensor(name, w))
        return name

    x = "x"
    for i in range(layers):
        p = f"/layers.{i}"
        for proj in ("q", "k", "v"):
            nodes.append(P.make_node("MatMul", [x, weight(f"layers.{i}.{proj}.weight", hidden, hidden)], [f"{p}/{proj}/out"], name=f"{p}/{proj}/MatMul"))
        nodes.append(P.make_node("Add", [f"{p}/q/out", f"{p}/k/out"], [f"{p}/qk"], name=f"{p}/Add_qk"))
        nodes.append(P.make_node("Add", [f"{p}/qk", f"{p}/v/out"], [f"{p}/qkv"], name=f"{p}/Add_qkv"))
        nodes.append(P.make_node("Tanh", [f"{p}/qkv"], [f"{p}/mix"], name=f"{p}/Tanh"))
        nodes.append(P.make_node("MatMul", [f"{p}/mix", weight(f"layers.{i}.o.weight", hidden, hidden)], [f"{p}/o/out"], name=f"{p}/o/MatMul"))
        nodes.append(P.make_node("Add", [x, f"{p}/o/out"], [f"{p}/h"], name=f"{p}/Add_h"))
        nodes.append(P.make_node("MatMul", [f"{p}/h", weight(f"layers.{i}.gate.weight", hidden, ffn)], [f"{p}/gate/out"], name=f"{p}/gate/MatMul"))
        nodes.append(P.make_node("MatMul", [f"{p}/h", weight(f"layers.{i}.up.weight", hidden, ffn)], [f"{p}/up/out"], name=f"{p}/up/MatMul"))
        nodes.append(P.make_node("Sigmoid", [f"{p}/gate/out"], [f"{p}/gate/act"], name=f"{p}/Sigmoid"))
        nodes.append(P.make_node("Mul", [f"{p}/gate/act", f"{p}/up/out"], [f"{p}/ffn"], name=f"{p}/Mul"))
        nodes.append(P.make_node("MatMul", [f"{p}/ffn", weight(f"layers.{i}.down.weight", ffn, hidden)], [f"{p}/down/out"], name=f"{p}/down/MatMul"))
        nodes.append(P.make_node("Add", [f"{p}/h", f"{p}/down/out"], [f"{p}/out"], name=f"{p}/Add_out"))
        x = f"{p}/out"
    nodes.append(P.make_node("Identity", [x], ["y"], name="/Identity"))
    graph = P.Message("GraphProto", name="decoder_weights", node=nodes, initializer=inits,
                      input=[P.make_value_info("x", P.DataType.FLOAT, ["batch", "seq", hidden])],
                      output=[P.make_value_info("y", P.DataType.FLOAT, ["batch", "seq", hidden])])
    return P.Message("ModelProto", ir_version=10, producer_name="onnx_quantize_amd.bench_model", graph=graph,
                     opset_import=[P.Message("OperatorSetIdProto", domain="", version=21)])


def configs(name, data):
    act = lambda dt: QActivationArgs(dtype=QuantType.from_string(dt), is_static=True)      # noqa: E731
    cal = {"num_samples": data.shape[0], "batch_size": max(1, data.shape[0] // 4)}
    return {
        "int8_tensor": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, symmetric=True)),
        "uint4_g128": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=128)),
        "int4_g128": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt4, group_size=128)),
        "static_int8": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8), input_activations=act("int8"), output_activations=act("int8"),
                                       calibration_data=data, calibration_params=cal),
        "awq_uint4_g128": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=128), preprocessors=[AwqConfig()],
                                          calibration_data=data, calibration_params=cal),
        "awq_static_int8": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt8, strategy="channel"), input_activations=act("int8"),
                                           preprocessors=[AwqConfig()], calibration_data=data, calibration_params=cal),     # two walks: the ranges come from the second
        "gptq_int4_g128": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt4, group_size=128, algorithm=GPTQConfig(mode="corrected")),
                                          calibration_data=data, calibration_params=cal),
        # the reference's loop AS WRITTEN (gptq.py:199,208: its error feedback is zero) -- the default mode of this package
        "gptq_int4_g128_parity": lambda: QConfig(weights=QWeightArgs(dtype=QuantType.QInt4, group_size=128, algorithm=GPTQConfig()),
                                                 calibration_data=data, calibration_params=cal),
    }[name]()


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--ffn", type=int, default=11008)
    ap.add_argument("--config", default="uint4_g128", help="one configuration, or several separated by commas: they share ONE source file")
    ap.add_argument("--samples", type=int, default=8, help="calibration sequences (calibrated configurations)")
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--dir", default=None, help="where the files go (default: a temporary directory, removed afterwards)")
    ap.add_argument("--repeat", type=int, default=2, help="quantize passes over the same source (the first one pages the file in)")
    ap.add_argument("--phases", action="store_true", help="also report the wall time of the calibration walk(s), the searches and the emission")
    return ap


def run(args) -> dict:
    """One file -> file measurement (also the `model_file` object of bench.py's line).  The result is verified: one weight's
    MatMulNBits blob / grouped integers in the emitted file against the kernels' own output on the same weight.  The temporary
    directory with the (multi-GB) source goes away whatever happens."""
    made: list = []
    try:
        return _run(args, made)
    finally:
        if args.dir is None and int(os.environ.get("RANK", "0")) == 0:
            for d in made:
                shutil.rmtree(d, ignore_errors=True)


def _run(args, made) -> dict:
    phases: dict = {}
    if args.phases:
        import onnx_quantize_amd.model_quantize as MQ

        def timed(name, fn):
            def wrapper(*a, **kw):
                torch.cuda.synchronize()
                t = time.perf_counter()
                out = fn(*a, **kw)
                torch.cuda.synchronize()
                phases[name] = round(phases.get(name, 0.0) + time.perf_counter() - t, 4)
                return out
            return wrapper
        MQ._calibrate, MQ._preprocess, MQ.plan_node = timed("calibrate_s", MQ._calibrate), timed("searches_s", MQ._preprocess), timed("seam_s", MQ.plan_node)
        import onnx_quantize_amd.staging as ST                 # the seam's share that is PCIe: looked up at call time by seam.py
        ST.upload, ST.download = timed("upload_s", ST.upload), timed("download_s", ST.download)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    rehearsal = os.environ.get("OQ_BENCH_REHEARSAL", "0") == "1"
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(0 if rehearsal else local)
        dist.init_process_group("gloo" if rehearsal else "nccl")
    work = args.dir or (tempfile.mkdtemp(prefix="oq_bench_model_") if world == 1 else os.path.join(tempfile.gettempdir(), "oq_bench_model_ranks"))
    os.makedirs(work, exist_ok=True)
    made.append(work)
    src, dst = os.path.join(work, "model.onnx"), os.path.join(work, "model_q.onnx")
    t0 = time.perf_counter()
    params = 0
    if rank == 0:                                              # one source file for all ranks (they memory-map it)
        model = build_model(args.layers, args.hidden, args.ffn)
        params = sum(int(np.prod(t.dims)) for t in model.graph.initializer)
        P.save_model(model, src, external_data="model.onnx.data")
        del model
    t_build = time.perf_counter() - t0
    data = torch.randn(args.samples, args.seq, args.hidden, generator=torch.Generator().manual_seed(1)).numpy()
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    if world > 1:
        dist.barrier()
    names = args.config.split(",")
    lines = {}
    for name in names:
        line = _run_config(args, name, data, src, dst, phases, params, t_build, world, rank, rehearsal)
        if line is not None:
            lines[name] = line
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return None
    if len(names) == 1:
        return lines[names[0]]
    first = lines[names[0]]
    return {"bench": "model_file", "layers": args.layers, "weights": first["weights"], "params": first["params"], "source_bytes": first["source_bytes"],
            "build_source_s": first["build_source_s"], "n_gpus": world, "configs": lines,
            "verified": all(ln.get("verified", False) for ln in lines.values())}


def _run_config(args, config, data, src, dst, phases, params, t_build, world, rank, rehearsal):
    if world > 1:
        import torch.distributed as dist
    runs = []
    for r in range(args.repeat):
        t0 = time.perf_counter()
        loaded = P.load_model(src)
        t1 = time.perf_counter()
        if world > 1:
            from onnx_quantize_amd.model_quantize import quantize_model_sharded
            out = quantize_model_sharded(loaded, configs(config, data))
        else:
            out = quantize_model(loaded, configs(config, data))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t2 = time.perf_counter()
        if rank == 0:
            P.save_model(out, dst, external_data="model_q.onnx.data")
        t3 = time.perf_counter()
        runs.append({"load_s": round(t1 - t0, 4), "quantize_s": round(t2 - t1, 4), "save_s": round(t3 - t2, 4), "total_s": round(t3 - t0, 4), **phases})
        phases.clear()
        del loaded
    if world > 1:
        times = [None] * world
        dist.all_gather_object(times, runs)
        runs = [{k: max(t[i][k] for t in times) for k in runs[i]} for i in range(len(runs))]        # the slowest rank's clock
        if rank != 0:
            return None
    out_bytes = os.path.getsize(dst) + os.path.getsize(dst + ".data")
    calls = sorted({(n.op_type, n.domain) for n in out.graph.node if n.domain})
    best = min(runs, key=lambda r: r["total_s"])
    line = {"bench": "model_file", "config": config, "layers": args.layers, "weights": 7 * args.layers, "params": params,
            "source_bytes": os.path.getsize(src) + os.path.getsize(src + ".data"), "result_bytes": out_bytes, "calls": calls,
            "build_source_s": round(t_build, 2), "runs": runs, "best": best,
            "mparam_per_s_file_to_file": round(params / best["total_s"] / 1e6, 1),
            "source_gb_per_s_quantize_phase": round(params * 4 / best["quantize_s"] / 1e9, 2), "n_gpus": world}
    if world > 1 and rehearsal:
        line["rehearsal"] = "ranks share one GPU, collectives on gloo: the multi-rank code path executed, NOT a scaling measurement"
    if config in ("uint4_g128", "int4_g128", "int8_tensor"):              # the file holds what the kernels produce on that weight
        from onnx_quantize_amd.hip import ops
        source = P.load_model(src)
        name = "layers.0.down.weight"
        w = torch.from_numpy(np.array(P.tensor_to_numpy(next(t for t in source.graph.initializer if t.name == name)))).cuda()
        got = torch.from_numpy(np.array(P.tensor_to_numpy(next(t for t in out.graph.initializer if t.name == name))))
        if config == "uint4_g128":
            want, _, _ = ops.rtn_quantize(w, "uint4", "group", 128, layout="nbits")
        elif config == "int4_g128":
            want, _, _ = ops.rtn_quantize(w, "int4", "group", 128)
        else:
            want, _, _ = ops.rtn_quantize(w, "int8", "tensor", -1, True)
        line["verified"] = bool(torch.equal(got.view(torch.uint8).reshape(-1), want.cpu().view(torch.uint8).reshape(-1)))
    if config.startswith("gptq_int4_g128"):
        # against the per-layer device path on the same bytes.  The first layer's q projection reads the model input, i.e. the
        # calibration data itself: its Hessian is accumulated here from the same batches in the same order (calibrate.py:296-307
        # walks them one by one), factored, and the loop run on the weight from the file.  parity: the integers in the emitted
        # file must be those, byte for byte (and, the reference's loop being what it is, those of RTN with per-group parameters);
        # corrected: the share of equal integers is reported (a Hessian built by the grouped kernels of the walk may differ in
        # its last bits, and a flipped rounding is fed back down the column: DESIGN.md 4.5) and must be >= 0.99
        from onnx_quantize_amd.hip import ops
        source = P.load_model(src)
        name = "layers.0.q.weight"
        w = torch.from_numpy(np.array(P.tensor_to_numpy(next(t for t in source.graph.initializer if t.name == name)))).cuda()
        got = torch.from_numpy(np.array(P.tensor_to_numpy(next(t for t in out.graph.initializer if t.name == name)))).reshape(w.shape)
        h = torch.zeros((w.shape[0], w.shape[0]), dtype=torch.float32, device="cuda")
        n, bs = 0, max(1, data.shape[0] // 4)
        for b0 in range(0, data.shape[0] - data.shape[0] % bs, bs):
            n = ops.hessian_accumulate(torch.from_numpy(data[b0:b0 + bs]).cuda(), h, n)
        mode = "parity" if config.endswith("parity") else "corrected"
        want, _, _, _ = ops.gptq_quantize(w, h, "int4", "group", 128, False, False, 1.0, 128, 0.01, False, False, mode=mode)
        same = float((got.to(torch.int16) == want.cpu().to(torch.int16)).float().mean())
        line["tokens_per_input"] = int(data.shape[0] * data.shape[1])
        line["mode"] = mode
        line["integers_equal_to_per_layer_device_path"] = round(same, 6)
        if mode == "parity":
            rtn, _, _ = ops.rtn_quantize(w, "int4", "group", 128)
            line["integers_equal_to_rtn"] = bool(torch.equal(got.to(torch.int16), rtn.cpu().to(torch.int16)))
            line["verified"] = bool(same == 1.0)
        else:
            line["verified"] = bool(same >= 0.99)
    return line


def main():
    line = run(build_parser().parse_args())
    if line is not None:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
