#!/usr/bin/env python3
"""GPTQ benchmark (BASELINE.json configs 4 and 5): all MatMul weights of a Llama-2-7B-shaped model,
QInt4 group 128, on 1..8 MI355X.

    python bench_gptq.py [--layers 32 --tokens 262144 --mode parity]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench_gptq.py --gpus N

Synthetic data (no network): weights normal(0, 0.02), activations normal * per-channel scale, generated on
the device batch by batch and streamed into the MFMA Hessian kernel (never concatenated).  Weight matrices
are sharded over the ranks by `onnx_quantize_amd.sharding.plan_lpt` (layers sharing an input stay together:
one Hessian and one inverse factor serve q/k/v resp. gate/up); no collective while quantizing; one RCCL
gather of (packed int4, scales, zero points) to rank 0 at the end.  Strong scaling: the model is fixed.
Prints one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=128 * 2048, help="calibration tokens per layer input (128 seqs x 2048)")
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--batch-seqs", type=int, default=32, help="sequences per Hessian call (T = 32 x 2048 = 65536 rows: one X^T X launch without T-slabs at K = 11008)")
    ap.add_argument("--mode", choices=["parity", "corrected"], default="parity")
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--ffn", type=int, default=11008)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from onnx_quantize_amd.hip import ops
    from onnx_quantize_amd.sharding import gather_device_results, llama2_7b_specs, plan_lpt

    specs = llama2_7b_specs(tokens=args.tokens, layers=args.layers, hidden=args.hidden, ffn=args.ffn)
    plan = plan_lpt(specs, world)
    my = plan[rank]
    n_seqs = args.tokens // args.seq
    t_h = t_f = t_l = 0.0
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up (library load, first-touch allocations)
    wtmp = torch.randn((256, 256), device=dev)
    htmp = torch.zeros((256, 256), device=dev)
    ops.hessian_accumulate(torch.randn((4, 64, 256), device=dev), htmp, 0)
    ops.gptq_quantize(wtmp, htmp, "int4", "group", 128)
    fence()

    t0 = time.perf_counter()
    results, shared_cache, timings = {}, {}, []
    for i in my:
        sp = specs[i]
        key = sp.hessian_key
        if key not in shared_cache:
            shared_cache.clear()                                   # only the current input's Hessian is kept
            gen = torch.Generator(device=dev).manual_seed(zlib.crc32(key.encode()) % (2**31))
            chan = 0.1 + 3.9 * torch.rand(sp.k, generator=gen, device=dev)
            h = torch.zeros((sp.k, sp.k), device=dev)
            n = 0
            for b in range(0, n_seqs, args.batch_seqs):
                nb = min(args.batch_seqs, n_seqs - b)
                # synthetic activations stand in for the calibration forward pass (not part of the path): generated
                # outside the timed events, one batch at a time (never concatenated)
                x = torch.randn((nb, args.seq, sp.k), generator=gen, device=dev) * chan
                e0, e1 = ev(), ev()
                e0.record()
                n = ops.hessian_accumulate(x, h, n)
                e1.record()
                timings.append(("h", e0, e1))
            e1, e2 = ev(), ev()
            e1.record()
            shared = ops.gptq_shared_factor(h, 0.01, False)
            e2.record()
            shared_cache[key] = (h, shared)
            timings.append(("f", e1, e2))
        h, shared = shared_cache[key]
        genw = torch.Generator(device=dev).manual_seed(1000 + i)
        w = torch.randn((sp.k, sp.n), generator=genw, device=dev) * 0.02
        e0, e1 = ev(), ev()
        e0.record()
        q, s, z, info = ops.gptq_quantize(w, h, "int4", "group", 128, mode=args.mode, shared=shared)
        results[i] = (ops.pack_nibbles(q), s, z)                   # 0.5 B / param on the wire
        e1.record()
        timings.append(("l", e0, e1))
    torch.cuda.synchronize()
    t_quant = time.perf_counter() - t0
    for t in timings:
        if t[0] == "h":
            t_h += t[1].elapsed_time(t[2])
        elif t[0] == "f":
            t_f += t[1].elapsed_time(t[2])
        else:
            t_l += t[1].elapsed_time(t[2])
    fence()
    t1 = time.perf_counter()
    gathered, nbytes = gather_device_results(specs, plan, results)
    fence()
    t_gather = time.perf_counter() - t1
    wall_all = time.perf_counter() - t0
    wall = (t_h + t_f + t_l) * 1e-3 + t_gather        # device time of the path + the gather; excludes the synthetic data generation

    stats = torch.tensor([wall, t_quant, t_gather, t_h, t_f, t_l, wall_all], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    if rank == 0:
        params = sum(s.k * s.n for s in specs)
        assert gathered is not None and len(gathered) == len(specs)
        wall = float(stats[0])
        print(json.dumps({
            "metric": "M-params quantized/sec, GPTQ QInt4 group-128, Llama-2-7B MatMul weights",
            "value": round(params / wall / 1e6, 2), "unit": "M-param/s", "n_gpus": world,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"gptq_qint4_g128_llama2_7b_{args.layers}layers", "params": params,
                       "tokens_per_input": args.tokens, "mode": args.mode, "block_size": 128, "percdamp": 0.01},
            "seconds": {"path": round(wall, 3), "wall_incl_synthetic_data_generation": round(float(stats[6]), 3),
                        "quantize_incl_datagen_max_rank": round(float(stats[1]), 3),
                        "gather": round(float(stats[2]), 4), "hessian_ms_max_rank": round(float(stats[3]), 1),
                        "factor_ms_max_rank": round(float(stats[4]), 1), "loop_ms_max_rank": round(float(stats[5]), 1)},
            "gather_bytes": nbytes,
            "hessian_flops_2TK2": float(sum(2.0 * args.tokens * specs[i].k ** 2 for i in
                                            {specs[j].hessian_key: j for j in range(len(specs))}.values())),
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
