#!/usr/bin/env python3
"""GPTQ benchmark (BASELINE.json configs 4 and 5): all MatMul weights of a Llama-2-7B-shaped model,
QInt4 group 128, on 1..8 MI355X.

    python bench_gptq.py [--layers 32 --tokens 262144 --mode parity]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench_gptq.py --gpus N

Synthetic data (no network): weights normal(0, 0.02); activations normal * per-channel scale, one set per
distinct input width generated BEFORE the timed region and reused for every layer (they stand in for the
calibration forward pass, which is not part of the path), streamed batch by batch into the MFMA Hessian kernel
(never concatenated).  Weight matrices are sharded over the ranks by `onnx_quantize_amd.sharding.plan_lpt`
(layers sharing an input stay together: one Hessian and one inverse factor serve q/k/v resp. gate/up); no
collective while quantizing; one RCCL gather of (packed int4, scales, zero points) to rank 0 at the end.
Strong scaling: the model is fixed.

HIP streams per rank: one for the Hessians (chip-filling MFMA GEMMs) and `--factor-streams` (4) that take the inverse
factor + loop of successive inputs in turn (latency-bound chains of small launches, side by side and next to the
Hessian of the following inputs); `--no-overlap` serialises everything on one stream.  `value` is
parameters / wall time of the timed region (first Hessian launch to the end of the gather, max over ranks).
Prints one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=128 * 2048, help="calibration tokens per layer input (128 seqs x 2048)")
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--batch-seqs", type=int, default=32,
                    help="sequences per Hessian call (T = 32 x 2048 = 65536 rows per X^T X launch)")
    ap.add_argument("--mode", choices=["parity", "corrected"], default="parity")
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--ffn", type=int, default=11008)
    ap.add_argument("--no-overlap", action="store_true", help="one stream: Hessian, factor and loop strictly in sequence")
    ap.add_argument("--factor-streams", type=int, default=4,
                    help="streams that take the factor + loop chains of successive inputs in turn (chains of small launches: "
                         "several of them side by side hide each other's launch and diagonal-block latency)")
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
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- synthetic calibration activations, one set per input width (outside the timed region)
    acts = {}
    for k in sorted({specs[i].k for i in my}):
        gen = torch.Generator(device=dev).manual_seed(1234 + k)
        chan = 0.1 + 3.9 * torch.rand(k, generator=gen, device=dev)
        acts[k] = [torch.randn((min(args.batch_seqs, n_seqs - b), args.seq, k), generator=gen, device=dev) * chan
                   for b in range(0, n_seqs, args.batch_seqs)]
    # weights: a few distinct random matrices per shape, reused round robin (generation is not part of the path either)
    wpool = {}
    for i in my:
        sp = specs[i]
        if (sp.k, sp.n) not in wpool:
            genw = torch.Generator(device=dev).manual_seed(1000 + sp.k + sp.n)
            wpool[(sp.k, sp.n)] = [torch.randn((sp.k, sp.n), generator=genw, device=dev) * 0.02 for _ in range(3)]

    # warm-up (library load, first-touch allocations)
    htmp = torch.zeros((256, 256), device=dev)
    ops.hessian_accumulate(torch.randn((4, 64, 256), device=dev), htmp, 0)
    ops.gptq_quantize(torch.randn((256, 256), device=dev), htmp, "int4", "group", 128)

    # group this rank's weights by the input they share, in plan order
    groups, seen = [], {}
    for i in my:
        key = specs[i].hessian_key
        if key not in seen:
            seen[key] = len(groups)
            groups.append((key, []))
        groups[seen[key]][1].append(i)

    s_h = torch.cuda.Stream(device=dev)
    q_streams = [s_h] if args.no_overlap else [torch.cuda.Stream(device=dev) for _ in range(max(1, args.factor_streams))]
    results, timings = {}, []
    fence()
    t0 = time.perf_counter()
    for gi, (key, members) in enumerate(groups):
        s_q = q_streams[gi % len(q_streams)]
        k = specs[members[0]].k
        with torch.cuda.stream(s_h):
            h = torch.zeros((k, k), device=dev)
            e0, e1 = ev(), ev()
            e0.record()
            n = 0
            for x in acts[k]:
                n = ops.hessian_accumulate(x, h, n)
            e1.record()
            timings.append(("h", e0, e1))
        with torch.cuda.stream(s_q):
            s_q.wait_event(e1)
            h.record_stream(s_q)
            e2, e3 = ev(), ev()
            e2.record()
            shared = ops.gptq_shared_factor(h, 0.01, False)
            e3.record()
            timings.append(("f", e2, e3))
            for j, i in enumerate(members):
                sp = specs[i]
                w = wpool[(sp.k, sp.n)][(i + j) % 3]
                e4, e5 = ev(), ev()
                e4.record()
                q, s, z, info = ops.gptq_quantize(w, h, "int4", "group", 128, mode=args.mode, shared=shared)
                results[i] = (ops.pack_nibbles(q), s, z)               # 0.5 B / param on the wire
                e5.record()
                timings.append(("l", e4, e5))
    torch.cuda.synchronize()
    t_quant = time.perf_counter() - t0
    fence()
    t1 = time.perf_counter()
    gathered, nbytes = gather_device_results(specs, plan, results)
    fence()
    t_gather = time.perf_counter() - t1
    wall = t_quant + t_gather
    t_h = sum(a.elapsed_time(b) for tag, a, b in timings if tag == "h")
    t_f = sum(a.elapsed_time(b) for tag, a, b in timings if tag == "f")
    t_l = sum(a.elapsed_time(b) for tag, a, b in timings if tag == "l")

    # outside the timed region: the Hessian kernel that just ran against float64 (a 256-column strip of one input)
    hcheck = None
    if rank == 0:
        kc = min(acts)
        hc = torch.zeros((kc, kc), device=dev)
        ref = torch.zeros((256, kc), dtype=torch.float64, device=dev)
        nc = 0
        for x in acts[kc]:
            nc = ops.hessian_accumulate(x, hc, nc)
            x2 = x.reshape(-1, kc)
            for i in range(0, x2.shape[0], 16384):
                blk = x2[i:i + 16384].double()
                ref += blk[:, :256].t() @ blk
        ref *= 2.0 / nc
        hcheck = {"k": kc, "rows": int(sum(x.shape[0] * x.shape[1] for x in acts[kc])),
                  "max_abs_err_over_max_abs_h": float((hc[:256].double() - ref).abs().max() / ref.abs().max()),
                  "exactly_symmetric": bool(torch.equal(hc, hc.T))}
        del hc, ref
    # and its rate alone (no other stream active): one batch of the widest input, events on the current stream
    roof = None
    if rank == 0:
        kw = max(acts)
        xw = acts[kw][0]
        hw = torch.zeros((kw, kw), device=dev)
        ops.hessian_accumulate(xw, hw, 0)
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        reps = 3
        for _ in range(reps):
            ops.hessian_accumulate(xw, hw, xw.shape[0])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        t_rows = xw.shape[0] * xw.shape[1]
        tiles = (kw + 255) // 256
        fp32_equiv = 2.0 * t_rows * (tiles * (tiles + 1) // 2) * 256 * 256          # upper 256 x 256 tiles, as executed
        method = ops.hessian_method()
        split = method != "f32" and kw >= 1024
        terms = 9 if method == "bf16x9" else 6
        roof = {"bound": "mfma", "kernel": "oq::syrk_pieces_kernel<%d>" % terms if split else "oq::gemm_tn_kernel",
                "achieved": round(fp32_equiv * (terms if split else 1) / ms / 1e9, 1), "peak": 2500.0 if split else 157.3,
                "unit": "TFLOP/s", "dtype": "bf16 pieces, fp32 accumulate" if split else "f32",
                "fp32_equivalent_TFLOPs": round(fp32_equiv / ms / 1e9, 1), "call_ms": round(ms, 2), "k": kw, "rows": t_rows,
                "traffic": None}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        del hw

    stats = torch.tensor([wall, t_quant, t_gather, t_h, t_f, t_l], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    if rank == 0:
        params = sum(s.k * s.n for s in specs)
        assert gathered is not None and len(gathered) == len(specs)
        wall = float(stats[0])
        flops_exec = float(sum(args.tokens * specs[i].k ** 2 for i in {specs[j].hessian_key: j for j in range(len(specs))}.values()))
        print(json.dumps({
            "metric": "M-params quantized/sec, GPTQ QInt4 group-128, Llama-2-7B MatMul weights",
            "value": round(params / wall / 1e6, 2), "unit": "M-param/s", "n_gpus": world,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"gptq_qint4_g128_llama2_7b_{args.layers}layers", "params": params,
                       "tokens_per_input": args.tokens, "mode": args.mode, "block_size": 128, "percdamp": 0.01,
                       "streams": 1 if args.no_overlap else 1 + len(q_streams), "hessian_method": ops.hessian_method()},
            "seconds": {"wall": round(wall, 3), "quantize_max_rank": round(float(stats[1]), 3),
                        "gather": round(float(stats[2]), 4),
                        # per-phase device time (with two streams the phases overlap: their sum exceeds the wall time)
                        "hessian_ms_max_rank": round(float(stats[3]), 1),
                        "factor_ms_max_rank": round(float(stats[4]), 1), "loop_ms_max_rank": round(float(stats[5]), 1)},
            "gather_bytes": nbytes,
            "hessian_flops_executed": flops_exec,
            "hessian_check_vs_float64": hcheck,
            "roofline": roof,
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
