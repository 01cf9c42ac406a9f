#!/usr/bin/env python3
"""Headline benchmark: group-128 uint4 RTN quantization of a 4096x11008 fp32 weight on MI355X.

    python bench.py [--gpus N --steps K --warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment bench.py launches the N ranks ITSELF (a
`python -m torch.distributed.run` child, started before this process touches a GPU); under torch.distributed.run it is
one of the ranks.  Either way `n_gpus` in the JSON line is the number of ranks that joined, and a mismatch with --gpus is
an error, never a silent one-GPU run.

One "step" = one pass of the hot path over one 4096x11008 matrix already resident in HBM: one call of
oq_rtn_quantize_f32 through the C ABI.  Default output layout "nbits" = reference rtn.py:54-109
(`_rtn_quantize`) fused with qrules/_common.py:65-123 (`_prepare_for_matmul_nbits`), i.e. the packed int4
MatMulNBits blob + scales + zero points the reference emits for this configuration and the layout the
algorithmic byte count (4.539 B/param) is defined on; "kn" = the [K, N] one-value-per-byte array
`_rtn_quantize` itself returns (timed too and reported under "other_layout").  Inputs rotate over `--rotate` distinct
HBM buffers (default 4 x 180 MB, more than the 256 MiB Infinity Cache) so every step streams from HBM, not from cache.

N > 1 is weak scaling: every rank quantizes its own matrices (independent MatMul weights shard with
no data-path collective); `value` = params of all ranks / max-over-ranks time.  After the timed steps
every rank's last result travels to rank 0 through `sharding.gather_device_results` (one padded RCCL gather over xGMI),
reported as the `gather` object.

Prints ONE JSON line (contract in the task description) with these extra objects:
  roofline      algorithmic bytes per launch / measured launch duration vs the 8 TB/s HBM peak
  cpu_baseline  the NumPy oracle (oracle/oq_oracle.py) timed on the host, rank 0 at N=1 only
  model_rtn     a whole model's weights in one call: Llama-2-7B's 224 MatMul weights resident in HBM through
                oq_rtn_quantize_ptrs_f32 (one launch per distinct shape), HIP events, fraction of the HBM peak
  seam          host -> host throughput through the plugin seam (qrules/_common.py:126-142) on the same configuration:
                round-1 route ([K,N] kernel, 45 MB download, second round trip for the packer) vs the device-resident
                seam (seam.py: one upload, fused blob kernel, one download), digest-checked
  calibration   BASELINE config 3 stand-in from the same run (bench_calib.run): 51 batches x 72 activation tensors through
                MinMaxCalibrator.collect_many, `roofline` of oq::minmax_partial, a CPU baseline, `verified`
  model_file    next rows N4 / N1: an ONNX file (14 Llama-2-7B-width weights, 1.6 GB, tensors in a side file) -> quantize_file
                (uint4 g128 -> MatMulNBits) -> an ONNX file (bench_model.run): load / quantize / save seconds, `verified`
  awq           the AWQ scale / clip searches of one 4096 x 4096 layer on own kernels (next row N2), verified in float64
  searches      the MSE range search (row M1) and HQQ's zero-point optimisation (next row N2) on the headline matrix, each
                verified through the error it is meant to lower
  gptq          BASELINE configs 4 / 5 from the same run: GPTQ QInt4 g128 of all Llama-2-7B MatMul weights
                (bench_gptq.run): wall, M-param/s, the Hessian kernels' MFMA rooflines, a CPU baseline, `verified`;
                `corrected` = the same model with the error-correcting loop, `wall_by_hessian_method` = the whole-model
                wall with the Hessian on the fp32 MFMA kernel next to the default split-operand one
  model_file_7b north_star's sentence, file to file (round 6): a 7B-shaped ONNX file (224 weights, 25.9 GB) -> GPTQ QInt4 g128 of all of
                them -> an ONNX file, in the reference's default behaviour (`parity`) and `corrected`, 65 536 calibration tokens through
                the graph, phases apart, one weight of each file verified against the per-layer device path (~1 min, N = 1 only)
  rccl_selftest N = 1: a child process brings up a one-rank `nccl` communicator before any other GPU call and runs every kind of
                exchange the N-rank path issues (`bench.py --nccl-selftest`, sharding.collectives_selftest): ms per step
  gather        N > 1: seconds, bytes and ranks of the end-of-run RCCL gather

`python bench.py --gpus N --nccl-selftest` (N >= 1) runs only that self-test on N ranks and exits 0 / 1.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_DIM, N_DIM, GROUP = 4096, 11008, 128
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)


def algorithmic_bytes(layout: str) -> int:
    """read W once + packed int4 + fp32 scales + 1-byte zero points = 4.539 B/param."""
    params = K_DIM * N_DIM
    groups = params // GROUP
    return params * 4 + params // 2 + groups * 4 + groups


def moved_bytes(layout: str) -> int:
    params = K_DIM * N_DIM
    groups = params // GROUP
    return params * 4 + (params if layout == "kn" else params // 2) + groups * 5


def sha16(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


# ------------------------------------------------------------------------------------------------ launch plumbing
def self_launch(script: str, gpus: int, argv: list[str]) -> int:
    """Start `gpus` ranks of `script` under torch.distributed.run and relay their output.  Runs in a process that has not
    touched a GPU (nothing before this imports torch.cuda state), and never replaces the running program: a child process
    is started and its exit code returned."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script, *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def init_ranks(gpus: int, backend: str = "nccl"):
    """(device, rank, world) of this rank; the process group is up when world > 1.  --gpus must equal WORLD_SIZE."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != gpus:
        raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus {gpus}` (it spawns the ranks) or "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {gpus} --master-addr 127.0.0.1 bench.py --gpus {gpus}`")
    # OQ_BENCH_REHEARSAL=1: the N-rank flow on ONE GPU (ranks share cuda:0, collectives on gloo with host staging): a way to
    # execute the multi-rank code path -- plans, sharded GPTQ, the gather, max-over-ranks -- on a 1-GPU box.  Its line says so;
    # it is not a scaling measurement.
    rehearsal = backend == "nccl" and os.environ.get("OQ_BENCH_REHEARSAL", "0") == "1"
    if backend == "gloo":                       # CPU plumbing test only (tests/test_bench_launch.py)
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        if rehearsal:
            local_rank = local_rank % torch.cuda.device_count()
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} are visible")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a collective that a dead peer never joins fails after this many seconds instead of the communicator's ten minutes
        import datetime
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("OQ_BENCH_PG_TIMEOUT", "240")))
        if backend == "gloo" or rehearsal:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
    return dev, rank, world


def selftest_main(args) -> None:
    """`--nccl-selftest`: the process group comes up first -- `init_process_group("nccl", device_id=...)` is the first GPU call of
    this process --, then `sharding.collectives_selftest`.  One JSON line; exit code 1 with the message when a step fails.  The
    program is never replaced and nothing is retried."""
    import datetime

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    timeout = datetime.timedelta(seconds=int(os.environ.get("OQ_BENCH_PG_TIMEOUT", "120")))
    t0 = time.perf_counter()
    if args.selftest_backend == "gloo":
        dev = None
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timeout)
    else:
        if torch.cuda.device_count() <= local_rank:       # counting devices does not initialise the GPU
            raise SystemExit(f"rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} are visible")
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=timeout)
        torch.cuda.set_device(dev)
    init_ms = (time.perf_counter() - t0) * 1e3
    from onnx_quantize_amd.sharding import collectives_selftest

    rec = collectives_selftest(dev, timeout_s=90.0)
    rec["init_process_group_ms"] = round(init_ms, 1)
    rec["device"] = None if dev is None else torch.cuda.get_device_name(dev)
    if rank == 0:
        print(json.dumps({"rccl_selftest": rec}), flush=True)
    if rec["ok"]:
        dist.destroy_process_group()
        return
    sys.stderr.write(f"[bench] rank {rank}: collectives self-test failed: {rec['error']}\n")
    sys.stderr.flush()
    os._exit(1)         # a stuck communicator must not keep the interpreter's shutdown waiting


def run_selftest_child(timeout_s: float = 150.0) -> dict:
    """`bench.py --gpus 1 --nccl-selftest` as a CHILD process (started, waited for, never exec'ed) and its record."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--nccl-selftest"]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"ok": False, "error": f"the self-test child gave no answer within {timeout_s:.0f} s", "command": " ".join(cmd[1:])}
    rec = None
    for line in r.stdout.splitlines():
        if line.startswith('{"rccl_selftest"'):
            rec = json.loads(line)["rccl_selftest"]
    if rec is None:
        rec = {"ok": False, "error": f"exit code {r.returncode}: {r.stderr.strip()[-600:]}"}
    rec["exit_code"] = r.returncode
    rec["child_wall_s"] = round(time.perf_counter() - t0, 1)
    rec["what"] = ("child process `bench.py --gpus 1 --nccl-selftest`: init_process_group('nccl', world_size=1, device_id=cuda:0) before any "
                   "other GPU call, then sharding.collectives_selftest on device tensors (ms per step); not a scaling measurement")
    return rec


def stub_main(args) -> None:
    """`--stub`: the launch / barrier / max-over-ranks / one-JSON-line plumbing with a no-op step on gloo, no GPU.  Exists
    for the CPU test of the multi-rank path; its line says so and carries no measurement."""
    import torch
    import torch.distributed as dist

    dev, rank, world = init_ranks(args.gpus, backend="gloo")
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    seen = torch.ones(1, dtype=torch.int32)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"metric": "plumbing test (no measurement)", "value": None, "unit": None, "n_gpus": world,
                          "ranks_seen": int(seen.item()), "steps": args.steps, "warmup": args.warmup, "data": "stub",
                          "scaling": "weak", "higher_is_better": True}))
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(w: np.ndarray, budget_s: float = 20.0):
    """Time the oracle (checker, never the product) on the host cores: kind = "port"."""
    from bench_gptq import cpu_info

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oq_oracle as O
    t_best, runs = float("inf"), 0
    t_start = time.perf_counter()
    while runs < 5 and (runs < 2 or time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        q, s, z = O.rtn_quantize(w, "uint4", "group", GROUP)
        t_best = min(t_best, time.perf_counter() - t0)
        runs += 1
    return {
        "value": round(K_DIM * N_DIM / t_best / 1e6, 2), "unit": "M-param/s", "cores": 1,
        "kind": "port",
        "sample": f"full workload (one 4096x11008 matrix), best of {runs} runs, {t_best:.3f} s each; "
                  f"NumPy {np.__version__} elementwise path is single-threaded (as the reference runs it)",
        "seconds": round(t_best, 4),
        "digest_ok": None, **cpu_info(),
    }, (q, s, z)


# ------------------------------------------------------------------------------------------------ the seam, host to host
class _Tensor:
    def __init__(self, a):
        self._a = a

    def numpy(self):
        return self._a


class _Value:
    def __init__(self, name, a):
        self.name, self.const_value = name, _Tensor(a)


def seam_bench(w_host: np.ndarray, digest: dict, count: int = 12) -> dict:
    """Host -> host through the plugin seam on `count` config-2 weights (NumPy in, the three MatMulNBits arrays out)."""
    import torch

    from onnx_quantize_amd import QConfig, QuantType, QWeightArgs, seam
    from onnx_quantize_amd.algorithms.rtn import _rtn_quantize
    from onnx_quantize_amd.wire_format import _prepare_for_matmul_nbits

    qc = QConfig(weights=QWeightArgs(dtype=QuantType.QUInt4, group_size=GROUP))
    a = qc.weights
    params = K_DIM * N_DIM * count
    fresh = lambda: [w_host.copy() for _ in range(count)]   # noqa: E731 -- every route gets arrays the GPU has never touched, like a model's

    def check(blob, scale, zp) -> bool:
        full = np.empty((N_DIM, K_DIM // GROUP, GROUP), np.uint8)
        full[..., 0::2] = blob & 0x0F
        full[..., 1::2] = blob >> 4
        z = np.empty((N_DIM, K_DIM // GROUP), np.uint8)
        z[:, 0::2] = zp & 0x0F
        z[:, 1::2] = zp >> 4
        return bool(sha16(np.ascontiguousarray(full.reshape(N_DIM, K_DIM).T)) == digest["q_sha"] and
                    sha16(scale.reshape(-1, 1)) == digest["s_sha"] and sha16(z.reshape(-1, 1)) == digest["z_sha"])

    def round1_route(w):
        """What round 1 shipped behind the seam: pageable upload, [K,N] kernel, 45 MB pageable download, then the packer as
        a second round trip (upload q, pack, download)."""
        wd = torch.from_numpy(w).cuda()
        q, s, z = ops.rtn_quantize(wd, "uint4", "group", GROUP)
        q, s, z = q.cpu().numpy(), s.cpu().numpy(), z.cpu().numpy()
        blob = ops.pack_matmul_nbits(torch.from_numpy(q).cuda(), GROUP, 4).cpu().numpy()
        pz = ops.pack_zero_points_u4(torch.from_numpy(z.reshape(-1)).cuda(), N_DIM, K_DIM // GROUP).cpu().numpy()
        return blob, s.reshape(-1, K_DIM // GROUP), pz

    from onnx_quantize_amd.hip import ops

    # one untimed call per route: first launches of a kernel / first use of a stream carry 0.1 s of one-time set-up
    warm = w_host.copy()
    round1_route(warm)
    _prepare_for_matmul_nbits(*_rtn_quantize(warm, a.dtype, a.strategy, a.group_size, a.symmetric, a.reduce_range, a.clip_ratio, a.mse,
                                             a.scale_dtype, a.zp_dtype), qc)
    seam.weight_arrays(_Value("warm", warm), qc, None, True)
    mats = fresh()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w in mats:
        last = round1_route(w)
    t_before = time.perf_counter() - t0
    ok_before = check(*last)
    # the reference-compatible NumPy functions of this round, still two round trips
    mats = fresh()
    t0 = time.perf_counter()
    for w in mats:
        q, s, z = _rtn_quantize(w, a.dtype, a.strategy, a.group_size, a.symmetric, a.reduce_range, a.clip_ratio, a.mse, a.scale_dtype, a.zp_dtype)
        last = _prepare_for_matmul_nbits(q, s, z, qc)
    t_plugin = time.perf_counter() - t0
    # device-resident seam (upload per call).  Host-side effects (page faults of fresh arrays) make single passes scatter by
    # a factor of two between boxes, and the first pass of a process is 8-14 ms per weight everywhere (the runtime's own staging
    # warms up): five passes, the median is reported and all five are listed.
    trials = []
    for _ in range(5):
        mats = fresh()
        t0 = time.perf_counter()
        for i, w in enumerate(mats):
            last = seam.weight_arrays(_Value(f"w{i}", w), qc, None, True)
        trials.append(time.perf_counter() - t0)
    t_after = sorted(trials)[len(trials) // 2]
    ok_after = check(*last)
    rate = lambda t: round(params / t / 1e6, 1)  # noqa: E731
    return {"what": "host->host through quantize_weights' arrays (qrules/_common.py:133-137), uint4 g128 4096x11008, "
                    f"{count} weights in sequence, NumPy in / MatMulNBits arrays out, PCIe included",
            "unit": "M-param/s",
            "before": {"route": "round 1: pageable upload, [K,N] kernel, 45 MB pageable download, packer as a second round trip",
                       "value": rate(t_before), "ms_per_weight": round(t_before * 1e3 / count, 2), "digest_ok": ok_before},
            "plugin_functions": {"route": "_rtn_quantize + _prepare_for_matmul_nbits (NumPy in / NumPy out, two round trips)",
                                 "value": rate(t_plugin), "ms_per_weight": round(t_plugin * 1e3 / count, 2)},
            "after": {"route": "seam.weight_arrays: one blocking upload, fused blob kernel, one download (the only route since round 3: "
                               "the worker-thread prefetcher of round 2 was slower in the driver record and was removed)",
                      "value": rate(t_after), "ms_per_weight": round(t_after * 1e3 / count, 2),
                      "ms_per_weight_trials": [round(t * 1e3 / count, 2) for t in trials], "digest_ok": ok_after},
            "speedup": round(t_before / t_after, 2),
            "note": "every route is timed on fresh host arrays (memory the GPU has not mapped yet), as a model's weights are"}


# ------------------------------------------------------------------------------------------------ a model's weights in one call
def model_rtn_bench(dev, layout: str, headline_out, w_headline, layers: int = 32) -> dict:
    """RTN uint4 g128 of every MatMul weight of a Llama-2-7B-shaped model (q/k/v/o 4096x4096, gate/up 4096x11008, down
    11008x4096, x 32 layers = 224 matrices, 6.48 B parameters) resident in HBM, through `ops.rtn_quantize_many`
    (oq_rtn_quantize_ptrs_f32): what replaces the reference's per-node loop (qrules/_common.py:126-142) for a device-resident
    model.  Timed with HIP events on the launch stream around the second call (the first warms the allocator); the per-matrix
    loop over `ops.rtn_quantize` on the same weights is timed next to it; outputs of one matrix per shape are compared bit for
    bit with the single-matrix entry point, and the 4096 x 11008 KAT2 matrix is one of the weights (checked against the
    headline run's output)."""
    import torch

    from onnx_quantize_amd.hip import ops

    gen = torch.Generator(device=dev).manual_seed(5)
    shapes = [(4096, 4096)] * 4 + [(4096, 11008)] * 2 + [(11008, 4096)]
    base = {sh: torch.randn(sh, generator=gen, device=dev) * 0.02 for sh in set(shapes)}
    ws = []
    for layer in range(layers):
        for j, sh in enumerate(shapes):
            if layer == 0 and j == 4:
                ws.append(w_headline)                                # the KAT2 matrix rides along
            else:
                ws.append(base[sh].clone())                          # distinct buffers: 25.9 GB, nothing is served from a cache
    params = sum(w.numel() for w in ws)
    n_ws = len(ws)
    groups = params // GROUP
    alg = params * 4 + params // 2 + groups * 5
    ops.rtn_quantize_many(ws, "uint4", GROUP, layout=layout)         # warm-up (allocator, table upload path)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    res = ops.rtn_quantize_many(ws, "uint4", GROUP, layout=layout)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    ok = True
    for i in (0, 4, 6):                                              # one matrix per shape against the single-matrix entry point
        q, s, z = ops.rtn_quantize(ws[i], "uint4", "group", GROUP, layout=layout)
        ok = ok and bool(torch.equal(q.reshape(-1), res[i][0].reshape(-1)) and torch.equal(s.reshape(-1), res[i][1].reshape(-1)) and
                         torch.equal(z.reshape(-1), res[i][2].reshape(-1)))
    ok = ok and bool(torch.equal(res[4][0].reshape(-1), headline_out[0].reshape(-1)))
    del res
    torch.cuda.synchronize()
    l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    l0.record()
    keep = [ops.rtn_quantize(w, "uint4", "group", GROUP, layout=layout) for w in ws]   # outputs kept: every matrix writes its own 24 MB
    l1.record()
    torch.cuda.synchronize()
    loop_ms = l0.elapsed_time(l1)
    del keep
    gbs = alg / (dev_ms * 1e-3) / 1e9
    # the same 224 weights in the [K, N]-family layouts (what configs 4 / 5 serialise: int4 two columns per byte; and one value
    # per byte).  A single-matrix launch pays the 5 us parameter-transpose launch per matrix (`other_layout`, `packed_kn_layout`:
    # 0.55-0.56); in a model-sized call it is one launch per ~1.6e8 parameters.
    layouts = {}
    for qt, lay, qbytes in (("int4", "kn_packed4", 0.5), ("int8", "kn", 1.0)):
        ops.rtn_quantize_many(ws, qt, GROUP, layout=lay)
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        r2 = ops.rtn_quantize_many(ws, qt, GROUP, layout=lay)
        a1.record()
        torch.cuda.synchronize()
        ms2 = a0.elapsed_time(a1)
        q1, s1, z1 = ops.rtn_quantize(ws[5], qt, "group", GROUP, layout=lay)
        same = bool(torch.equal(q1.reshape(-1), r2[5][0].reshape(-1)) and torch.equal(s1.reshape(-1), r2[5][1].reshape(-1)) and
                    torch.equal(z1.reshape(-1), r2[5][2].reshape(-1)))
        alg2 = params * 4 + int(params * qbytes) + groups * 5
        layouts[f"{qt}_{lay}"] = {"device_ms": round(ms2, 3), "algorithmic_bytes": alg2, "frac": round(alg2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "equals_single_matrix_outputs": same}
        ok = ok and same
        del r2, q1, s1, z1
    del ws, base
    torch.cuda.empty_cache()
    # the small-matrix regime: gemma-3-270m's 126 MatMul weights (18 layers x {q 640x1024, k / v 640x256, o 1024x640, gate / up
    # 640x2048, down 2048x640}; 100 M parameters), int8 g128 (the examples' RTN configuration): one launch per shape
    gshapes = [(640, 1024), (640, 256), (640, 256), (1024, 640), (640, 2048), (640, 2048), (2048, 640)]
    gws = [torch.randn(sh, generator=gen, device=dev) * 0.05 for _ in range(18) for sh in gshapes]
    gparams = sum(w.numel() for w in gws)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        r = fn()
        a1.record()
        torch.cuda.synchronize()
        return a0.elapsed_time(a1), r
    g_many_ms, g_many = timed(lambda: ops.rtn_quantize_many(gws, "int8", GROUP, layout="kn"))
    g_loop_ms, g_loop = timed(lambda: [ops.rtn_quantize(w, "int8", "group", GROUP, layout="kn") for w in gws])
    g_ok = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) for a, b in zip(g_many, g_loop))
    small = {"what": "gemma-3-270m's 126 MatMul weights (100 M parameters), int8 g128, [K,N] layout: ops.rtn_quantize_many (one launch "
                     "per distinct shape: 5) against the per-matrix loop",
             "matrices": len(gws), "params": gparams, "one_call_device_ms": round(g_many_ms, 3), "per_matrix_loop_device_ms": round(g_loop_ms, 3),
             "speedup": round(g_loop_ms / g_many_ms, 2), "equals_per_matrix_outputs": bool(g_ok)}
    del gws, g_many, g_loop
    return {"small_matrices": small,
            "what": "Llama-2-7B's 224 MatMul weights (6.48 B parameters, 25.9 GB fp32) resident in HBM, uint4 g128, ONE call: "
                    "ops.rtn_quantize_many -> oq_rtn_quantize_ptrs_f32 (a device table of pointers, ~1.6e8 parameters per launch; the call's own "
                    "allocations and table upload are inside the timed region); at this footprint every matrix' 24 MB of output is really "
                    "written to HBM, while the single headline matrix' output stays in the 256 MiB Infinity Cache",
            "out_layout": layout, "matrices": n_ws, "params": params,
            "device_ms": round(dev_ms, 3), "host_wall_ms": round(wall * 1e3, 3),
            "value": round(params / (dev_ms * 1e-3) / 1e6, 1), "unit": "M-param/s",
            "algorithmic_bytes": alg, "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
            "per_matrix_loop_device_ms": round(loop_ms, 3), "per_matrix_loop_frac": round(alg / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "kn_family_layouts": layouts,
            "equals_single_matrix_outputs": ok}


# ------------------------------------------------------------------------------------------------ AWQ searches
def awq_bench(dev, k: int = 4096, n: int = 4096, t: int = 4096) -> dict:
    """pre_passes/awq.py:114-184 / :207-259 on one Llama-2-7B-sized layer (q_proj: K = N = 4096, 4096 calibration rows), uint4
    g128, device resident (oq_awq_scale_search_f32 / oq_awq_clip_search_f32: own kernels, the 20 / 10 products on the fp16-piece
    GEMM with the loss reduced in its epilogue).  Verified: the winning candidate's loss recomputed in float64 from this
    package's own RTN / dequantize kernels agrees within 2e-3."""
    import torch

    from onnx_quantize_amd.hip import ops

    gen = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn((t, k), generator=gen, device=dev) * (0.1 + 3.9 * torch.rand(k, generator=gen, device=dev))
    w = torch.randn((k, n), generator=gen, device=dev) * 0.02

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(reps):
            r = fn()
        a1.record()
        torch.cuda.synchronize()
        return a0.elapsed_time(a1) / reps, r
    ms_scale, (best_scale, losses) = timed(lambda: ops.awq_scale_search(x, w, "uint4", "group", GROUP))
    ms_clip, (best_ratio, closses) = timed(lambda: ops.awq_clip_search(x, w, "uint4", "group", GROUP))
    i = int(losses.argmin())
    # float64 check on 1024 rows (a float64 product of all 4096 would dominate the bench): the search on those rows, then
    # the loss of ITS winning scale recomputed from this package's RTN / dequantize kernels and a float64 product
    x64 = x[:1024].double()
    sub = ops.awq_scale_search(x[:1024], w, "uint4", "group", GROUP)
    q2, s2, z2 = ops.rtn_quantize(w * sub[0].reshape(-1, 1), "uint4", "group", GROUP)
    wh2 = ops.dequantize(q2, s2, z2, "uint4", mode="group", group=GROUP) / sub[0].reshape(-1, 1)
    ref2 = float(((x64 @ (w.double() - wh2.double())) ** 2).mean())
    ok = abs(float(sub[1].min()) - ref2) <= 2e-3 * ref2
    # a long calibration set (32768 rows = 16 sequences of 2048 tokens, T = 8 K): the Gram route of awq.hip -- X^T X once per
    # search on the Hessian kernels, per candidate the quadratic form <D, G D>.  Checked in float64 through the same identity
    # with a float64 Gram matrix and the winning scale's D from this package's RTN / dequantize kernels.
    tl = 32768
    xl = torch.randn((tl, k), generator=gen, device=dev) * (0.1 + 3.9 * torch.rand(k, generator=gen, device=dev))
    ms_scale_l, (best_l, losses_l) = timed(lambda: ops.awq_scale_search(xl, w, "uint4", "group", GROUP), reps=2)
    ms_clip_l, _ = timed(lambda: ops.awq_clip_search(xl, w, "uint4", "group", GROUP), reps=2)
    g64 = torch.zeros((k, k), dtype=torch.float64, device=dev)
    for r0 in range(0, tl, 4096):
        xb = xl[r0:r0 + 4096].double()
        g64 += xb.t() @ xb
    ql, sl, zl = ops.rtn_quantize(w * best_l.reshape(-1, 1), "uint4", "group", GROUP)
    dl = w.double() - (ops.dequantize(ql, sl, zl, "uint4", mode="group", group=GROUP) / best_l.reshape(-1, 1)).double()
    ref_l = float((dl * (g64 @ dl)).sum() / (tl * n))
    ok_l = abs(float(losses_l.min()) - ref_l) <= 2e-3 * ref_l
    return {"what": "AWQ scale search (20 candidates) and clip search (10 ratios) of one 4096 x 4096 layer with 4096 calibration rows, uint4 g128, "
                    "device resident; round 2 (torch elementwise + rocBLAS): 23.4 / 11.2 ms; round 3 (22-bit loss product): 7.7 / 3.6 ms; round 4 (first pieces only, fused residual kernel): 3.7 / 1.8 ms; round 5: the residual kernel writes the product's fp16 pieces itself",
            "scale_search_ms": round(ms_scale, 3), "clip_search_ms": round(ms_clip, 3), "best_grid_point": i, "best_clip_ratio": best_ratio,
            "loss_at_best": float(losses[i]), "loss_float64_check_rows": 1024, "loss_float64": ref2, "loss_kernel": float(sub[1].min()),
            "long_calibration_set": {"rows": tl, "route": "gram: X^T X once (Hessian kernels), <D, G D> per candidate", "scale_search_ms": round(ms_scale_l, 3),
                                     "clip_search_ms": round(ms_clip_l, 3), "loss_kernel": float(losses_l.min()), "loss_float64": ref_l,
                                     "verified": bool(ok_l)},
            "verified": bool(ok and ok_l)}


def search_bench(dev, w) -> dict:
    """The two iterative weight-only searches on the headline matrix (4096 x 11008, uint4 g128, device resident), from the same
    run: the MSE range search of utils.py:140-239 (`mse=True`: 20 shrink candidates per group, fake-quantize, |.|^2.4 error) and
    HQQ's zero-point optimisation (hqq.py:106-144: up to 20 rounds of quantize / shrink / row mean with ONE global early-stop
    decision per round, taken on the device).  Verified through what each search is for: the MSE parameters do not raise any
    group's |.|^2.4 error above plain RTN's, HQQ does not raise the mean |w - w_r| of its starting point."""
    import torch

    from onnx_quantize_amd.hip import ops

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(reps):
            r = fn()
        a1.record()
        torch.cuda.synchronize()
        return a0.elapsed_time(a1) / reps, r
    k, n = w.shape
    ms_plain, (q0, s0, z0) = timed(lambda: ops.rtn_quantize(w, "uint4", "group", GROUP))
    ms_mse, (q1, s1, z1) = timed(lambda: ops.rtn_quantize(w, "uint4", "group", GROUP, mse=True))
    ms_hqq, (q2, s2, z2, rounds) = timed(lambda: ops.hqq_quantize(w, GROUP))

    def group_err(q, s, z, p):      # per (column, k-group) sum |w - dequant|^p over a 512-column strip (checker, float64)
        c = 512
        d = ops.dequantize(q, s.reshape(-1), z.reshape(-1), "uint4", mode="group", group=GROUP)[:, :c].double() - w[:, :c].double()
        return d.abs().pow(p).reshape(k // GROUP, GROUP, c).sum(dim=1)
    e_rtn, e_mse = group_err(q0, s0, z0, 2.4), group_err(q1, s1, z1, 2.4)
    mse_ok = bool((e_mse <= e_rtn * (1 + 1e-6)).all()) and bool((e_mse < e_rtn).any())
    # HQQ keeps float zero points: w_r = (q - z) * s with them
    c = 512
    zf = z2.reshape(n, k // GROUP)[:c].t().repeat_interleave(GROUP, dim=0)
    sf = s2.reshape(n, k // GROUP)[:c].t().repeat_interleave(GROUP, dim=0)
    l1_hqq = float(((q2[:, :c].double() - zf.double()) * sf.double() - w[:, :c].double()).abs().mean())
    l1_rtn = float(group_err(q0, s0, z0, 1.0).sum() / (k * c))
    hqq_ok = l1_hqq <= l1_rtn * (1 + 1e-6) and int(rounds.item()) >= 1
    params = k * n
    return {"what": "MSE range search (QWeightArgs(mse=True)) and HQQ zero-point optimisation of the headline matrix, uint4 g128, device resident",
            "rtn_ms": round(ms_plain, 4), "mse_ms": round(ms_mse, 3), "mse_M_params_per_s": round(params / ms_mse / 1e3, 1),
            "hqq_ms": round(ms_hqq, 3), "hqq_rounds": int(rounds.item()), "hqq_M_params_per_s": round(params / ms_hqq / 1e3, 1),
            "mse_groups_improved_frac": round(float((e_mse < e_rtn).double().mean()), 4),
            "mean_abs_error_rtn": l1_rtn, "mean_abs_error_hqq": l1_hqq, "verified": bool(mse_ok and hqq_ok)}


def strategies_bench(dev, ws, rank: int) -> dict:
    """Per-channel and per-tensor int8 RTN of the headline matrix (rtn.py:54-109 with the reference's default QWeightArgs():
    core/_qconfig.py:232-268) -- `rtn_resident_stream` / `rtn_tensor_onepass` of rtn_resident.hip, W read once.  HIP events on
    the launch stream over rotating inputs; algorithmic bytes = W once + one byte per value + the parameters; outputs
    checked against the digests of what the reference itself returned (tests/golden/digests.json, rank 0's matrix)."""
    import torch

    from onnx_quantize_amd.hip import ops

    with open(os.path.join(ROOT, "tests", "golden", "digests.json")) as f:
        digests = json.load(f)
    out = {"what": "int8 RTN of the 4096x11008 matrix with one range per column / per tensor; W read once (rtn_resident.hip); "
                   "rounds 1-3 read it twice in three launches (89 / 94 us)"}
    for strategy, kernel in (("channel", "oq::rtn_resident_stream"), ("tensor", "oq::rtn_tensor_onepass")):
        outs = ops.rtn_quantize(ws[0], "int8", strategy, -1)
        for i in range(10):
            ops.rtn_quantize(ws[i % len(ws)], "int8", strategy, -1, out=outs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 100
        e0.record()
        for i in range(reps):
            ops.rtn_quantize(ws[i % len(ws)], "int8", strategy, -1, out=outs)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        nparam = N_DIM if strategy == "channel" else 1
        alg = K_DIM * N_DIM * 4 + K_DIM * N_DIM + nparam * 5
        rec = {"kernel": kernel, "launch_us": round(us, 2), "algorithmic_bytes": alg, "achieved_GBs": round(alg / us / 1e3, 1),
               "frac": round(alg / us / 1e3 / HBM_PEAK_GBS, 4)}
        d = digests.get(f"headline_int8_{strategy}")
        if rank == 0 and d is not None:
            q, sc, z = outs
            rec["verified_vs_reference_digest"] = bool(sha16(q.cpu().numpy()) == d["q_sha"] and sha16(sc.cpu().numpy()) == d["s_sha"] and
                                                       sha16(z.cpu().numpy()) == d["z_sha"])
        out[strategy] = rec
    # Llama's down_proj shape (11008 rows per column) on the same kernel; rank 0's last result is checked against the digest of
    # what the reference returned for that matrix (tests/golden/digests.json: tall_int8_channel, seed 6)
    dt = digests.get("tall_int8_channel")
    wt = [torch.from_numpy(np.random.default_rng(6 + 100 * j + 1000 * rank).standard_normal((N_DIM, K_DIM), dtype=np.float32)).to(dev) for j in range(3)]
    outs = ops.rtn_quantize(wt[0], "int8", "channel", -1)
    for i in range(6):
        ops.rtn_quantize(wt[i % 3], "int8", "channel", -1, out=outs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 61                                                   # ends on wt[0]: the digest's matrix on rank 0
    e0.record()
    for i in range(reps):
        ops.rtn_quantize(wt[i % 3], "int8", "channel", -1, out=outs)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    alg = K_DIM * N_DIM * 4 + K_DIM * N_DIM + K_DIM * 5
    rec = {"shape": f"{N_DIM}x{K_DIM}", "kernel": "oq::rtn_resident_stream", "launch_us": round(us, 2), "algorithmic_bytes": alg,
           "achieved_GBs": round(alg / us / 1e3, 1), "frac": round(alg / us / 1e3 / HBM_PEAK_GBS, 4)}
    if rank == 0 and dt is not None:
        q, sc, z = outs
        rec["verified_vs_reference_digest"] = bool(sha16(q.cpu().numpy()) == dt["q_sha"] and sha16(sc.cpu().numpy()) == dt["s_sha"] and
                                                   sha16(z.cpu().numpy()) == dt["z_sha"])
    out["channel_tall"] = rec
    return out


# ------------------------------------------------------------------------------------------------ main
def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rotate", type=int, default=4, help="distinct input/output buffer sets cycled through")
    ap.add_argument("--rotate-out", type=int, default=12,
                    help="output sets of the SECOND timed run (`roofline.frac_outputs_streamed`): 12 x 24.3 MB = 291 MB, more than "
                         "the 256 MiB Infinity Cache, so every output byte is written to HBM between the reads (0: skip)")
    ap.add_argument("--layout", choices=["kn", "nbits"], default="nbits",
                    help="kn: [K,N] one value per byte (what _rtn_quantize returns); "
                         "nbits: MatMulNBits blob [N,K/g,g/2] (what the emitted graph holds for this config)")
    ap.add_argument("--symmetric", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-extra", type=int, default=4,
                    help="also time oq_rtn_quantize_batched_f32 with this many matrices per launch (0/1: skip)")
    ap.add_argument("--no-extras", action="store_true", help="time only the headline configuration (used under rocprofv3)")
    ap.add_argument("--no-gptq", action="store_true", help="skip the `gptq` object (configs 4 / 5)")
    ap.add_argument("--no-seam", action="store_true", help="skip the `seam` object")
    ap.add_argument("--no-model-rtn", action="store_true", help="skip the `model_rtn` object (224 Llama-2-7B weights in one call)")
    ap.add_argument("--no-awq", action="store_true", help="skip the `awq` object (AWQ scale / clip searches of one layer)")
    ap.add_argument("--no-model-file", action="store_true", help="skip the `model_file` object (an ONNX file quantized file to file, bench_model.py)")
    ap.add_argument("--no-model-file-7b", action="store_true",
                    help="skip the `model_file_7b` object (north_star's sentence: a 7B-shaped ONNX file -> GPTQ int4 g128 -> file, ~1 min)")
    ap.add_argument("--no-calibration", action="store_true", help="skip the `calibration` object (config 3 stand-in)")
    ap.add_argument("--gptq-extra-passes", default="corrected,f32",
                    help="further whole-model GPTQ passes of the `gptq` object (bench_gptq.py --extra-passes)")
    ap.add_argument("--gptq-gather", choices=["streamed", "padded"], default="streamed", help="bench_gptq.py --gather (N > 1)")
    ap.add_argument("--gptq-layers", type=int, default=32)
    ap.add_argument("--gptq-tokens", type=int, default=128 * 2048)
    ap.add_argument("--qparams-only", action="store_true", help="diagnostic: scales/zero-points only (read path ceiling)")
    ap.add_argument("--stub", action="store_true", help="plumbing test on gloo without a GPU (tests/test_bench_launch.py)")
    ap.add_argument("--nccl-selftest", action="store_true",
                    help="bring up an RCCL communicator of `--gpus` ranks (1: a one-rank communicator on cuda:0) BEFORE any other GPU call, "
                         "run every kind of exchange the multi-rank path issues once (sharding.collectives_selftest), print its record as "
                         "one JSON line and exit 0 / 1.  With --backend gloo: the same steps on the CPU (tests)")
    ap.add_argument("--selftest-backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--no-rccl-selftest", action="store_true", help="skip the `rccl_selftest` object (a child process running --nccl-selftest)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # before anything touches a GPU
        raise SystemExit(self_launch(os.path.abspath(__file__), args.gpus, sys.argv[1:]))
    if args.stub:
        return stub_main(args)
    if args.nccl_selftest:
        return selftest_main(args)

    # ---- first contact with RCCL, in a child of its own (N = 1, before this process touches the GPU): a one-rank communicator
    # on cuda:0 and every kind of exchange the N-rank path issues.  Never the measured thing; a failure or a
    # hang of the child is reported inside the object and the headline still goes out.
    rccl_selftest = None
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_extras and not args.no_rccl_selftest and not args.qparams_only:
        rccl_selftest = run_selftest_child()

    import torch
    import torch.distributed as dist

    dev, rank, world = init_ranks(args.gpus)
    if world > 1:          # before any clock: a bounded wait that names missing ranks, then the count the driver asked for
        from onnx_quantize_amd.sharding import await_all_ranks

        await_all_ranks("oq/bench", timeout_s=120.0)
        early = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(early)
        if int(early.item()) != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but {int(early.item())} ranks joined")

    from onnx_quantize_amd.hip import _lib as L
    lib = L.load()

    # ---- synthetic input: the BASELINE.json config-2 matrix (seed = rank so ranks hold different weights;
    # rank 0 is the KAT2 matrix whose reference digests are committed in tests/golden/digests.json)
    w_host = np.random.default_rng(rank).standard_normal((K_DIM, N_DIM), dtype=np.float32)
    w_src = torch.from_numpy(w_host).to(dev)
    ws = [w_src] + [w_src.clone() for _ in range(max(1, args.rotate) - 1)]
    groups = K_DIM * N_DIM // GROUP
    q_elems = K_DIM * N_DIM if args.layout == "kn" else K_DIM * N_DIM // 2
    outs = [(torch.empty(q_elems, dtype=torch.uint8, device=dev),
             torch.empty(groups, dtype=torch.float32, device=dev),
             torch.empty(groups, dtype=torch.uint8, device=dev)) for _ in ws]
    ws_bytes = lib.oq_rtn_workspace_bytes(K_DIM, N_DIM, L.OQ_GROUP, GROUP, 0)
    wsbuf = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    layout = L.OQ_LAYOUT_KN if args.layout == "kn" else L.OQ_LAYOUT_NBITS
    sym = int(args.symmetric)

    calls = []
    for w, (q, s, z) in zip(ws, outs):
        calls.append((C.c_void_p(w.data_ptr()), C.c_void_p(q.data_ptr()), C.c_void_p(s.data_ptr()),
                      C.c_void_p(z.data_ptr())))
    wsp, wsn = C.c_void_p(wsbuf.data_ptr()), wsbuf.numel()
    fn = lib.oq_rtn_quantize_f32

    def step(i: int) -> None:
        wp, qp, sp, zp = calls[i % len(calls)]
        if args.qparams_only:
            st = lib.oq_rtn_qparams_f32(wp, K_DIM, N_DIM, N_DIM, L.OQ_UINT4, L.OQ_GROUP, GROUP, sym, 0, 1.0, 0, sp, zp,
                                        wsp, wsn, stream)
        else:
            st = fn(wp, K_DIM, N_DIM, N_DIM, L.OQ_UINT4, L.OQ_GROUP, GROUP, sym, 0, 1.0, 0, qp, sp, zp, layout,
                    wsp, wsn, stream)
        if st != 0:
            L.check(st)

    def fence() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(i)
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)            # HIP events on the launch stream: device time of the K launches

    # the same launches with outputs that do not fit the Infinity Cache: `--rotate` output sets are 96 MB and
    # stay on chip from one step to the next, as the output of ONE quantized matrix does in real use; with `--rotate-out`
    # sets every output byte of a step is really written to HBM before its buffer comes round again
    streamed_us = None
    if world == 1 and not args.no_extras and not args.qparams_only and args.rotate_out > len(outs):
        more = [(torch.empty(q_elems, dtype=torch.uint8, device=dev), torch.empty(groups, dtype=torch.float32, device=dev),
                 torch.empty(groups, dtype=torch.uint8, device=dev)) for _ in range(args.rotate_out - len(outs))]
        sets = [(C.c_void_p(q.data_ptr()), C.c_void_p(sc.data_ptr()), C.c_void_p(z.data_ptr())) for (q, sc, z) in list(outs) + more]

        def sstep(i: int) -> None:
            qp, sp, zp = sets[i % len(sets)]
            st = fn(calls[i % len(calls)][0], K_DIM, N_DIM, N_DIM, L.OQ_UINT4, L.OQ_GROUP, GROUP, sym, 0, 1.0, 0, qp, sp, zp, layout, wsp, wsn, stream)
            if st != 0:
                L.check(st)
        for i in range(max(args.warmup, len(sets))):
            sstep(i)
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for i in range(max(args.steps, 2 * len(sets))):
            sstep(i)
        s1.record()
        torch.cuda.synchronize()
        streamed_us = s0.elapsed_time(s1) * 1e3 / max(args.steps, 2 * len(sets))
        streamed_same = bool(torch.equal(more[-1][0], outs[0][0]) and torch.equal(more[-1][1], outs[0][1]))   # same matrix, other buffers
        del more, sets

    # per-launch distribution (median, p10 / p90): 100 more launches, one event pair each
    pct = None
    if world == 1 and not args.no_extras:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
        for i, (a, b) in enumerate(evs):
            a.record()
            step(i)
            b.record()
        torch.cuda.synchronize()
        d = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        pct = [round(d[10], 2), round(d[50], 2), round(d[90], 2)]

    # the same launches, many of them: the driver's 20 steps are 0.8 ms of device time, over before the GPU's clocks have
    # settled; 400 launches (16 ms) show what a caller that quantizes a model's worth of matrices one by one sees per launch
    sustained_us = None
    if world == 1 and not args.no_extras and not args.qparams_only:
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        u0, u1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        u0.record()
        for i in range(400):
            step(i)
        u1.record()
        torch.cuda.synchronize()
        sustained_us = u0.elapsed_time(u1) * 1e3 / 400

    # the other output layout, shorter run, same buffers (reported next to the headline, never as `value`)
    other = "kn" if args.layout == "nbits" else "nbits"
    other_us = None
    if world == 1 and not args.qparams_only and not args.no_extras:
        olayout = L.OQ_LAYOUT_KN if other == "kn" else L.OQ_LAYOUT_NBITS
        oq = torch.empty(K_DIM * N_DIM if other == "kn" else K_DIM * N_DIM // 2, dtype=torch.uint8, device=dev)
        oqp = C.c_void_p(oq.data_ptr())

        # [K,N] bytes: the stateful entry point (what ops.rtn_quantize and the seam call): parameters transposed inside the launch (round 6)
        ostate = torch.zeros(lib.oq_rtn_state_bytes(K_DIM, N_DIM, L.OQ_GROUP, GROUP) + 256, dtype=torch.uint8, device=dev)
        ostp, ostn = (C.c_void_p(ostate.data_ptr()), ostate.numel()) if other == "kn" else (C.c_void_p(0), 0)

        def ostep(i: int) -> None:
            wp, _, sp, zp = calls[i % len(calls)]
            st = lib.oq_rtn_quantize_stateful_f32(wp, K_DIM, N_DIM, N_DIM, L.OQ_UINT4, L.OQ_GROUP, GROUP, sym, 0, 1.0, 0, oqp, sp, zp, olayout, wsp, wsn,
                                                  ostp, ostn, stream)
            if st != 0:
                L.check(st)
        for i in range(20):
            ostep(i)
        torch.cuda.synchronize()
        o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        o0.record()
        for i in range(200):
            ostep(i)
        o1.record()
        torch.cuda.synchronize()
        other_us = o0.elapsed_time(o1) * 5.0
        o2, o3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      # once more: the first 200 launches of a process run slower
        o2.record()
        for i in range(200):
            ostep(i)
        o3.record()
        torch.cuda.synchronize()
        other_runs = [round(other_us, 2), round(o2.elapsed_time(o3) * 5.0, 2)]
        other_us = min(other_runs)
        for i in range(len(calls)):          # restore the headline layout's outputs for the digest check below
            step(i)
        torch.cuda.synchronize()
        del oq

    # the packed [K, N/2] layout of the int4 configurations (configs 4 / 5 are QInt4: not MatMulNBits-eligible, the reference
    # serialises their [K, N] integers two per byte, core/_pack.py:8-22): written by the fused kernel's epilogue
    packed_kn = None
    if world == 1 and not args.qparams_only and not args.no_extras and not args.symmetric:
        pq = torch.empty(K_DIM * N_DIM // 2, dtype=torch.uint8, device=dev)
        pz = torch.empty(groups, dtype=torch.int8, device=dev)
        pqp, pzp = C.c_void_p(pq.data_ptr()), C.c_void_p(pz.data_ptr())

        # the stateful entry point (what ops.rtn_quantize and the seam call): with a zeroed, self-cleaning state the parameters are
        # transposed inside the launch (round 6) instead of by a second launch
        pstate = torch.zeros(lib.oq_rtn_state_bytes(K_DIM, N_DIM, L.OQ_GROUP, GROUP) + 256, dtype=torch.uint8, device=dev)
        pstp, pstn = C.c_void_p(pstate.data_ptr()), pstate.numel()

        def pstep(i: int) -> None:
            wp, _, sp, _ = calls[i % len(calls)]
            st = lib.oq_rtn_quantize_stateful_f32(wp, K_DIM, N_DIM, N_DIM, L.OQ_INT4, L.OQ_GROUP, GROUP, 0, 0, 1.0, 0, pqp, sp, pzp, L.OQ_LAYOUT_KN_PACKED4,
                                                  wsp, wsn, pstp, pstn, stream)
            if st != 0:
                L.check(st)
        for i in range(20):
            pstep(i)
        torch.cuda.synchronize()
        p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p0.record()
        for i in range(200):
            pstep(i)
        p1.record()
        torch.cuda.synchronize()
        p_us = p0.elapsed_time(p1) * 5.0
        # the same call without the state (staged parameters + a transpose launch: what rounds 4-5 measured), same process, right behind
        def pstep2(i: int) -> None:
            wp, _, sp, _ = calls[i % len(calls)]
            st = fn(wp, K_DIM, N_DIM, N_DIM, L.OQ_INT4, L.OQ_GROUP, GROUP, 0, 0, 1.0, 0, pqp, sp, pzp, L.OQ_LAYOUT_KN_PACKED4, wsp, wsn, stream)
            if st != 0:
                L.check(st)
        for i in range(20):
            pstep2(i)
        torch.cuda.synchronize()
        p2, p3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p2.record()
        for i in range(200):
            pstep2(i)
        p3.record()
        torch.cuda.synchronize()
        p_us_two = p2.elapsed_time(p3) * 5.0
        for i in range(20):          # and the stateful call once more: the order of the two measurements must not decide
            pstep(i)
        torch.cuda.synchronize()
        p4, p5 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p4.record()
        for i in range(200):
            pstep(i)
        p5.record()
        torch.cuda.synchronize()
        p_us_again = p4.elapsed_time(p5) * 5.0
        palg = algorithmic_bytes("nbits")            # W once + half a byte per value + (scale, zero point) per group: the same count
        packed_kn = {"what": "int4 g128 RTN of the same matrix in OQ_LAYOUT_KN_PACKED4 ([K, N/2] nibble pairs, core/_pack.py order), "
                             "rtn_group_fused<16>, parameters transposed inside the launch (stateful entry point: zeroed, self-cleaning state); "
                             "22.5 MB of integers written instead of 45",
                     "qtype": "int4", "launch_us": round(min(p_us, p_us_again), 2), "launch_us_runs": [round(p_us, 2), round(p_us_again, 2)],
                     "launch_us_staged_plus_transpose_launch": round(p_us_two, 2), "achieved_GBs": round(palg / (min(p_us, p_us_again) * 1e-6) / 1e9, 1),
                     "frac": round(palg / (min(p_us, p_us_again) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        if rank == 0:
            with open(os.path.join(ROOT, "tests", "golden", "digests.json")) as f:
                dpk = json.load(f).get("headline_int4_g128_packed")
            if dpk is not None:
                packed_kn["verified_vs_reference_digest"] = bool(sha16(pq.cpu().numpy()) == dpk["packed_sha"] and
                                                                  sha16(pz.cpu().numpy()) == dpk["z_sha"])
        for i in range(len(calls)):          # the scales of the headline outputs were overwritten (shared buffers): restore
            step(i)
        torch.cuda.synchronize()
        del pq, pz

    # batched entry point: `--batch-extra` matrices per launch (stacked weights); reported separately
    batched = None
    if world == 1 and not args.qparams_only and not args.no_extras and args.batch_extra > 1:
        nb = args.batch_extra
        wb = torch.stack([w_src] * nb).contiguous()
        qb = torch.empty(nb * q_elems, dtype=torch.uint8, device=dev)
        sb = torch.empty(nb * groups, dtype=torch.float32, device=dev)
        zb = torch.empty(nb * groups, dtype=torch.uint8, device=dev)
        bws = torch.empty(lib.oq_rtn_batched_workspace_bytes(nb, K_DIM, N_DIM, GROUP) + 256, dtype=torch.uint8, device=dev)

        def bstep() -> None:
            st = lib.oq_rtn_quantize_batched_f32(C.c_void_p(wb.data_ptr()), nb, K_DIM * N_DIM, K_DIM, N_DIM, N_DIM, L.OQ_UINT4, GROUP, sym, 0,
                                                 1.0, C.c_void_p(qb.data_ptr()), C.c_void_p(sb.data_ptr()), C.c_void_p(zb.data_ptr()), layout,
                                                 C.c_void_p(bws.data_ptr()), bws.numel(), stream)
            if st != 0:
                L.check(st)
        for _ in range(5):
            bstep()
        torch.cuda.synchronize()
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        for _ in range(50):
            bstep()
        b1.record()
        torch.cuda.synchronize()
        per_matrix_us = b0.elapsed_time(b1) * 1e3 / 50 / nb
        same = bool(torch.equal(qb[:q_elems], outs[0][0]) and torch.equal(sb[:groups], outs[0][1]))
        batched = {"matrices_per_launch": nb, "us_per_matrix": round(per_matrix_us, 2),
                   "achieved_GBs": round(algorithmic_bytes(args.layout) / (per_matrix_us * 1e-6) / 1e9, 1),
                   "frac": round(algorithmic_bytes(args.layout) / (per_matrix_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                   "equals_single_launch_output": same}
        del wb, qb, sb, zb, bws

    # a whole model's weights in one call: Llama-2-7B's 224 MatMul weights resident in HBM (25.9 GB fp32), uint4 g128, through
    # oq_rtn_quantize_ptrs_f32 -- one launch per distinct shape (3), a device table of pointers instead of a per-node loop
    model_rtn = None
    if world == 1 and not args.qparams_only and not args.no_extras and not args.no_model_rtn and not args.symmetric:
        model_rtn = model_rtn_bench(dev, args.layout, outs[0], w_src)

    # the reference's DEFAULT strategies (QWeightArgs(): int8, group_size=None -> per tensor; and per channel) on the same matrix
    strategies = None
    if world == 1 and not args.qparams_only and not args.no_extras and not args.symmetric:
        strategies = strategies_bench(dev, ws, rank)

    t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev)
    ranks_seen = torch.ones(1, dtype=torch.int32, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks_seen)
    wall, dev_ms = float(t[0]), float(t[1])
    if int(ranks_seen.item()) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but {int(ranks_seen.item())} ranks joined")

    # ---- N > 1: every rank's last result to rank 0 through the RCCL gather north_star names
    gather = None
    if world > 1:
        from onnx_quantize_amd.sharding import LayerSpec, gather_device_results

        specs = [LayerSpec(f"rank{r}.w", K_DIM, N_DIM) for r in range(world)]
        plan = [[r] for r in range(world)]
        q, s, z = outs[(args.steps - 1) % len(outs)] if args.steps else outs[0]
        fence()
        tg = time.perf_counter()
        got, nbytes = gather_device_results(specs, plan, {rank: (q, s, z)})
        fence()
        tg = torch.tensor([time.perf_counter() - tg], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather = {"collective": "one padded gather to rank 0 (RCCL over xGMI, backend nccl) + a size all_reduce",
                  "backend": dist.get_backend(),
                  "gather_s": round(float(tg[0]), 5), "gather_bytes": int(nbytes), "ranks_seen": int(ranks_seen.item()),
                  "GBs_into_rank0": round(nbytes * (world - 1) / world / float(tg[0]) / 1e9, 1),
                  "rank0_result_intact": None if got is None else bool(torch.equal(got["rank0.w"][0].to(q.device), q))}

    # ---- verify rank 0's last outputs against the reference digests (asymmetric / symmetric KAT2)
    verified = None
    digest = None
    if rank == 0:
        with open(os.path.join(ROOT, "tests", "golden", "digests.json")) as f:
            digests = json.load(f)
        digest = digests["config2_sym" if args.symmetric else "config2_asym"]
        q, s, z = outs[(args.steps - 1) % len(outs)] if args.steps else outs[0]
        ok = sha16(s.cpu().numpy()) == digest["s_sha"] and sha16(z.cpu().numpy()) == digest["z_sha"]
        if args.qparams_only:
            pass
        elif args.layout == "kn":
            ok = ok and sha16(q.cpu().numpy()) == digest["q_sha"]
        else:   # unpack the blob back to [K, N] and compare with the same digest
            b = q.cpu().numpy().reshape(N_DIM, K_DIM // GROUP, GROUP // 2)
            full = np.empty((N_DIM, K_DIM // GROUP, GROUP), np.uint8)
            full[..., 0::2] = b & 0x0F
            full[..., 1::2] = b >> 4
            ok = ok and sha16(np.ascontiguousarray(full.reshape(N_DIM, K_DIM).T)) == digest["q_sha"]
        verified = bool(ok)

    # ---- the seam (host to host) on the same configuration, rank 0 at N = 1
    seam = None
    if world == 1 and not args.no_seam and not args.no_extras and not args.symmetric:
        seam = seam_bench(w_host, digests["config2_asym"])

    searches = None
    if world == 1 and not args.no_awq and not args.no_extras and not args.symmetric:
        searches = search_bench(dev, w_src)
    del ws, outs, calls, w_src
    torch.cuda.empty_cache()
    # ---- config 3 stand-in: min-max calibration of a gemma-3-270m-shaped activation population (rank 0 at N = 1)
    calibration = None
    if world == 1 and not args.no_calibration and not args.no_extras:
        import bench_calib

        calibration = bench_calib.run(dev, cpu=not args.no_cpu_baseline)
        torch.cuda.empty_cache()
    awq = None
    if world == 1 and not args.no_awq and not args.no_extras:
        awq = awq_bench(dev)
        torch.cuda.empty_cache()
    # ---- the file path: an ONNX file of two Llama-2-7B-width decoder layers -> quantize_file -> an ONNX file
    model_file = None
    if world == 1 and not args.no_model_file and not args.no_extras:
        import bench_model

        try:
            model_file = bench_model.run(bench_model.build_parser().parse_args(["--layers", "2", "--config", "uint4_g128"]))
        except Exception as e:   # noqa: BLE001 -- reported in the object (e.g. no room for the 1.6 GB source); the headline line still goes out
            import traceback
            sys.stderr.write(f"[bench] the model_file object failed:\n{traceback.format_exc()}\n")
            model_file = {"error": f"{type(e).__name__}: {e}", "verified": False}
        torch.cuda.empty_cache()
    # ---- north_star's sentence, file to file: ALL MatMul weights of a 7B-shaped ONNX file (32 decoder layers of Llama-2-7B's widths,
    # 224 weights, 6.48 G parameters, 25.9 GB of fp32 in a side file) -> quantize_file-equivalent GPTQ QInt4 g128 -> an ONNX file.
    # 65 536 calibration tokens run through the graph on this GPU, Hessians streamed, both modes (the reference's loop as written =
    # `parity`, the default; and `corrected`), phases apart, one weight of each file verified against the per-layer device path.
    # PCIe- and file-inclusive: never `value`.
    model_file_7b = None
    if world == 1 and not args.no_model_file and not args.no_model_file_7b and not args.no_extras:
        import bench_model

        try:
            model_file_7b = bench_model.run(bench_model.build_parser().parse_args(
                ["--layers", "32", "--config", "gptq_int4_g128_parity,gptq_int4_g128", "--samples", "32", "--seq", "2048", "--repeat", "1", "--phases"]))
            model_file_7b["what"] = ("7B-shaped ONNX file (32 layers x q/k/v/o/gate/up/down, random weights, written by this package) -> GPTQ QInt4 "
                                     "group_size=128 of all 224 MatMul weights -> ONNX file, one pass per mode on one GPU; calibration: 32 sequences "
                                     "of 2048 random tokens walked through the graph in 4 batches (graph_runner on torch-ROCm, products on "
                                     "oq_matmul_pieces_f32), Hessians streamed; PCIe, page cache and file writing included")
        except Exception as e:   # noqa: BLE001 -- reported in the object (e.g. no room for the 26 GB source); the headline line still goes out
            import traceback
            sys.stderr.write(f"[bench] the model_file_7b object failed:\n{traceback.format_exc()}\n")
            model_file_7b = {"error": f"{type(e).__name__}: {e}", "verified": False}
        torch.cuda.empty_cache()
    # ---- configs 4 / 5: GPTQ of a Llama-2-7B-shaped model from the same run (all ranks take part)
    gptq = None
    if not args.no_gptq and not args.no_extras:
        import bench_gptq

        gargs = bench_gptq.build_parser().parse_args(["--gpus", str(world), "--layers", str(args.gptq_layers),
                                                      "--tokens", str(args.gptq_tokens), "--extra-passes", args.gptq_extra_passes,
                                                      "--gather", args.gptq_gather] +
                                                     (["--no-cpu-baseline"] if args.no_cpu_baseline else []))
        if world == 1:
            gptq = bench_gptq.run(gargs, dev, rank, world)
        else:
            # N ranks: the headline above is weak scaling without any exchange; the GPTQ object is the only part of this line
            # that talks between ranks.  Should its first contact with a real N-GPU node fail, the headline still gets printed,
            # with the failure in the object instead of a measurement.
            try:
                gptq = bench_gptq.run(gargs, dev, rank, world)
            except Exception as e:   # noqa: BLE001 -- reported, not swallowed
                import traceback
                sys.stderr.write(f"[bench] rank {rank}: the gptq object failed at {world} ranks:\n{traceback.format_exc()}\n")
                gptq = {"error": f"{type(e).__name__}: {e}", "n_gpus": world, "verified": False}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    params_total = K_DIM * N_DIM * args.steps * world
    ms_per_step = wall * 1e3 / max(args.steps, 1)
    launch_us = dev_ms * 1e3 / max(args.steps, 1)
    alg = algorithmic_bytes(args.layout)
    achieved = alg / (launch_us * 1e-6) / 1e9 if launch_us > 0 else 0.0
    traffic, traffic_source = None, None
    pmc = os.path.join(ROOT, "profiles", "rtn_pmc_traffic.json")
    if os.path.exists(pmc):
        with open(pmc) as f:
            pj = json.load(f)
        traffic = pj.get(args.layout)
        traffic_source = ("stored profile, not measured in this run: " + pj.get("source", "profiles/rtn_pmc_traffic.json (separate rocprofv3 "
                          "--pmc passes of this command, FETCH_SIZE x2 correction)"))

    result = {
        "metric": "M-params quantized/sec, group-128 uint4 RTN on 4096x11008 fp32",
        "value": round(params_total / wall / 1e6, 1),
        "unit": "M-param/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"rtn_quint4_g128_{'sym' if args.symmetric else 'asym'}_4096x11008_f32",
                   "out_layout": args.layout, "rotating_buffers": max(1, args.rotate), "matrices_per_step_per_gpu": 1},
        "verified_vs_reference_digest": verified,
        "batched_launch": batched,
        "model_rtn": model_rtn,
        "other_layout": None if other_us is None else {
            "out_layout": other, "launch_us": round(other_us, 2), "launch_us_runs": other_runs,
            "entry_point": "oq_rtn_quantize_stateful_f32 (zeroed, self-cleaning state: [K,N] parameters transposed inside the launch)",
            "achieved_GBs": round(alg / (other_us * 1e-6) / 1e9, 1), "frac": round(alg / (other_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "oq::rtn_group_wave<8,true,5>" if args.layout == "nbits" else "oq::rtn_group_fused<16,true,true,true> + oq::transpose_qparams",
                     "launch_us": round(launch_us, 2), "launch_us_p10_p50_p90": pct,
                     "launch_us_400_launches": None if sustained_us is None else round(sustained_us, 2),
                     "frac_400_launches": None if sustained_us is None else round(alg / (sustained_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "frac_outputs_streamed": None if streamed_us is None else round(alg / (streamed_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "launch_us_outputs_streamed": None if streamed_us is None else round(streamed_us, 2),
                     "outputs_streamed": None if streamed_us is None else {
                         "output_sets": args.rotate_out, "output_bytes_in_rotation": args.rotate_out * (moved_bytes(args.layout) - K_DIM * N_DIM * 4),
                         "same_bytes_as_headline": streamed_same,
                         "what": "`frac` is BASELINE's single-matrix regime: 4 rotating output sets (96 MB) stay in the 256 MiB Infinity "
                                 "Cache between steps; here 12 sets (291 MB) force every output byte to HBM, as inside a model-sized call"},
                     "algorithmic_bytes_per_launch": alg, "moved_bytes_per_launch": moved_bytes(args.layout)},
        "packed_kn_layout": packed_kn,
        "strategies": strategies,
        "seam": seam,
        "rccl_selftest": rccl_selftest,
        "gather": gather,
        "calibration": calibration,
        "model_file": model_file,
        "model_file_7b": model_file_7b,
        "awq": awq,
        "searches": searches,
        "gptq": gptq,
    }
    if os.environ.get("OQ_BENCH_REHEARSAL", "0") == "1":
        result["rehearsal"] = "ranks share one GPU, collectives on gloo: the multi-rank code path executed, NOT a scaling measurement"
    if world == 1 and not args.no_cpu_baseline:
        base, (cq, cs, cz) = cpu_baseline(w_host)
        base["digest_ok"] = bool(sha16(cq) == digests["config2_asym"]["q_sha"] and sha16(cs) == digests["config2_asym"]["s_sha"])
        result["cpu_baseline"] = base
    else:
        result["cpu_baseline"] = None
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
