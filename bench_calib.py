#!/usr/bin/env python3
"""Calibration benchmark (BASELINE.json config 3 stand-in): static int8 activation calibration over 512 samples of
a gemma-3-270m-shaped model.  gemma-3-270m itself is not obtainable offline, so the activation population is
synthetic with the same shapes: per batch of 10 samples, 18 layers x { [10,512,640] x 2 (attention
and MLP inputs), [10,512,1024] (o_proj input), [10,512,2048] (down_proj input) } fp32; 51 batches (the reference
drops the 2 remaining samples, calibrate.py:161-170).  Every batch goes through MinMaxCalibrator.collect_many on the GPU
(ONE launch pair per batch of 72 tensors, running state on the device; the per-tensor `collect` loop of
calibrate.py:264-266 is timed next to it), then compute_range + _compute_qparams per tensor (calibrate.py:268-285).

`run(dev)` returns the object `bench.py` prints as `calibration`; `python bench_calib.py` prints it alone.
  value         activation GB/s reduced end to end (51 batches x 72 tensors through collect_many), HIP events on the
                launch stream
  roofline      oq::minmax_partial on one 1.3 GB tensor: algorithmic bytes = 4 B / element read once,
                HIP events over 20 launches, against the 8 TB/s HBM peak
  cpu_baseline  the oracle's MinMaxOracle.collect (minmax.py:40-64 restated: np.min + np.max per tensor) over ONE batch of
                the same 72 tensors on the host (kind "port", one core: NumPy reductions are single-threaded)
  hessians      the GPTQ side of a batch: the 72 Hessian updates in one grouped call against one call per tensor
  verified      every (min, max) of the timed run equals torch's own reduction of the same tensors and the oracle's on the
                sampled batch, bit for bit; the two Hessian routes agree to 2e-5 of max |H|
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def run(dev, cpu: bool = True, layers: int = 18, batches: int = 51) -> dict:
    import numpy as np
    import torch

    from onnx_quantize_amd import QuantType
    from onnx_quantize_amd.algorithms.functional import _compute_qparams
    from onnx_quantize_amd.calibration import MinMaxCalibrator
    from onnx_quantize_amd.hip import ops

    gen = torch.Generator(device=dev).manual_seed(3)
    shapes = [("attn_in", 640), ("mlp_in", 640), ("o_in", 1024), ("down_in", 2048)]
    # one batch worth of distinct tensors (1.6 GB, far beyond the 256 MiB Infinity Cache) is generated once and re-fed for
    # every batch, so the timed region measures the reductions, not the random-number generator
    acts = {f"l{l}.{nm}": torch.randn((10, 512, c), generator=gen, device=dev) * (0.1 + 9.9 * torch.rand(c, generator=gen, device=dev))
            for l in range(layers) for nm, c in shapes}
    nbytes = sum(t.numel() * 4 for t in acts.values())
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    cal = MinMaxCalibrator()
    for name, t in acts.items():            # warm-up
        cal.collect(name, t)
    torch.cuda.synchronize()
    cal = MinMaxCalibrator()
    t0 = time.perf_counter()
    for b in range(batches):
        for name, t in acts.items():
            cal.collect(name, t)
    torch.cuda.synchronize()
    t_per_tensor = time.perf_counter() - t0
    # the same statistics with one launch pair per batch (oq_minmax_collect_many_f32): what the on-device driver calls
    cal_many = MinMaxCalibrator()
    cal_many.collect_many(acts)
    torch.cuda.synchronize()
    cal_many = MinMaxCalibrator()
    e0, e1 = ev(), ev()
    t0 = time.perf_counter()
    e0.record()
    for b in range(batches):
        cal_many.collect_many(acts)
    e1.record()
    torch.cuda.synchronize()
    t_collect = time.perf_counter() - t0
    dev_collect = e0.elapsed_time(e1) * 1e-3
    ok = True
    for name, t in acts.items():
        d = cal_many.data[name]
        ok = ok and d.min_val == t.min().item() and d.max_val == t.max().item()
        ok = ok and d.min_val == cal.data[name].min_val and d.max_val == cal.data[name].max_val
    t1 = time.perf_counter()
    qparams_ref = {}
    for name in acts:                      # the reference's call pattern: one round trip per name (calibrate.py:268-285)
        lo, hi = cal.compute_range(name)
        qparams_ref[name] = _compute_qparams(lo, hi, QuantType.QInt8, False, False, "float32", QuantType.QInt8.np_dtype)
    t_params_per_name = time.perf_counter() - t1
    cal_many.compute_qparams_many(list(acts)[:2], QuantType.QInt8)
    t1 = time.perf_counter()
    qparams = cal_many.compute_qparams_many(list(acts), QuantType.QInt8)     # one kernel, one copy
    t_params = time.perf_counter() - t1
    for name in acts:
        ok = ok and qparams[name][0].tobytes() == qparams_ref[name][0].tobytes() and int(qparams[name][1]) == int(qparams_ref[name][1])

    # the weight side of the same configuration (static QInt8, per-tensor symmetric like BASELINE config 1) on the
    # gemma-3-270m MatMul shapes: 18 layers x {q 640x1024, k / v 640x256, o 1024x640, gate / up 640x2048, down 2048x640}
    wshapes = [(640, 1024), (640, 256), (640, 256), (1024, 640), (640, 2048), (640, 2048), (2048, 640)]
    weights = [torch.randn(kn, generator=gen, device=dev) * 0.05 for _ in range(layers) for kn in wshapes]
    for w in weights[:7]:
        ops.rtn_quantize(w, "int8", "tensor", -1, True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    wq = [ops.rtn_quantize(w, "int8", "tensor", -1, True) for w in weights]
    torch.cuda.synchronize()
    t_weights = time.perf_counter() - t2
    ops.rtn_quantize_tensor_many(weights, "int8", True)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    wq_many = ops.rtn_quantize_tensor_many(weights, "int8", True)
    torch.cuda.synchronize()
    t_weights_many = time.perf_counter() - t3
    ok = ok and all(torch.equal(a[0], b[0]) and float(a[1]) == float(b[1]) for a, b in zip(wq[:5], wq_many[:5]))
    wparams = sum(w.numel() for w in weights)

    # the GPTQ side of a calibration batch (calibrate.py:292-305 -> gptq.py:246-260): the Hessian updates of the 72 tapped
    # inputs, one grouped launch chain per batch (oq_hessian_accumulate_many_f32) against one call per tensor
    names = list(acts)
    xs = [acts[n] for n in names]
    def hessian_batches(many: bool, reps: int):
        hs = [torch.zeros((x.shape[-1], x.shape[-1]), device=dev) for x in xs]
        n = [0] * len(xs)
        def one():
            nonlocal n
            n = ops.hessian_accumulate_many(xs, hs, n) if many else [ops.hessian_accumulate(x, h, k) for x, h, k in zip(xs, hs, n)]
        one()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            one()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps, hs
    t_h_many, hs_many = hessian_batches(True, 10)
    t_h_each, hs_each = hessian_batches(False, 10)
    h_dev = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(hs_many, hs_each))
    h_flop = sum(2.0 * x.shape[0] * x.shape[1] * x.shape[2] ** 2 for x in xs)
    ok = ok and h_dev <= 2e-5 and all(bool(torch.equal(h, h.T)) for h in hs_many[:4])
    del hs_many, hs_each

    # the reduction kernel alone: one 1.3 GB tensor, HIP events over 20 launches on the launch stream
    big = torch.randn((64, 2048, 2560), generator=gen, device=dev)
    st = ops.minmax_state(dev)
    for _ in range(3):
        ops.minmax_collect(big, st)
    torch.cuda.synchronize()
    # three trials of 20 launches, the fastest counts: the events bracket the GPU timeline, so one host hiccup between two
    # launches (64 ms once, in a driver-style run of bench.py) would otherwise be booked as kernel time; all trials are reported
    big_trials = []
    for _ in range(3):
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(20):
            ops.minmax_collect(big, st)
        e1.record()
        torch.cuda.synchronize()
        big_trials.append(e0.elapsed_time(e1) * 1e3 / 20)
    big_us = min(big_trials)
    big_bytes = big.numel() * 4
    big_gbs = big_bytes / (big_us * 1e-6) / 1e9
    ok = ok and float(st[0]) == big.min().item() and float(st[1]) == big.max().item()
    del big

    traffic, traffic_source = None, None
    pmc_path = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            pj = json.load(f)
        if pj.get("calibration", {}).get("tensor_bytes") == big_bytes:
            traffic = pj["calibration"]["traffic_bytes_per_launch"]
            traffic_source = "stored profile, not measured in this run: " + pj["source"] + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_targets.py calibration: the same tensor, FETCH_SIZE x2 gfx950 correction)"
    out = {
        "metric": "activation GB/s reduced, min-max calibration, gemma-3-270m-shaped synthetic stand-in, 512 samples",
        "value": round(nbytes * batches / dev_collect / 1e9, 1), "unit": "GB/s", "n_gpus": 1, "higher_is_better": True,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "calibration_minmax_gemma3_270m_shapes_static_qint8", "layers": layers, "batches": batches,
                   "samples": batches * 10, "tensors_per_batch": len(acts), "bytes_per_batch": nbytes,
                   "entry_point": "MinMaxCalibrator.collect_many -> oq_minmax_collect_many_f32 (one launch pair per batch)"},
        "seconds": {"collect_device": round(dev_collect, 4), "collect_wall": round(t_collect, 4),
                    "ranges_and_qparams": round(t_params, 4),
                    "collect_per_tensor_calls_wall": round(t_per_tensor, 4), "ranges_and_qparams_per_name_calls": round(t_params_per_name, 4),
                    "weights_rtn_int8_per_tensor": round(t_weights, 4), "weights_rtn_int8_per_tensor_one_call": round(t_weights_many, 5)},
        "weights": {"matrices": len(weights), "params": wparams, "M_params_per_s": round(wparams / t_weights / 1e6, 1),
                    "M_params_per_s_one_call": round(wparams / t_weights_many / 1e6, 1)},
        "hessians": {"what": "gptq.py:246-260 for the 72 tapped inputs of one batch: ops.hessian_accumulate_many (one launch chain, fp16-piece "
                             "products) against ops.hessian_accumulate per tensor, wall clock over 10 batches",
                     "ms_per_batch_one_call": round(t_h_many * 1e3, 3), "ms_per_batch_per_tensor_calls": round(t_h_each * 1e3, 3),
                     "fp32_equivalent_TFLOPs_one_call": round(h_flop / t_h_many / 1e12, 1),
                     "max_abs_diff_between_routes_over_max_abs_h": h_dev},
        "tensors_per_s": round(len(acts) * batches / t_collect, 1),
        "frac_of_hbm_peak_end_to_end": round(nbytes * batches / dev_collect / 1e9 / HBM_PEAK_GBS, 4),
        "per_tensor_calls": {"GBs": round(nbytes * batches / t_per_tensor / 1e9, 1),
                             "tensors_per_s": round(len(acts) * batches / t_per_tensor, 1)},
        "roofline": {"bound": "hbm", "achieved": round(big_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(big_gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "oq::minmax_partial<float> (+ oq::minmax_update, one block)", "launch_us": round(big_us, 2), "launch_us_trials": [round(t, 2) for t in big_trials],
                     "algorithmic_bytes_per_launch": big_bytes, "bytes_per_element": 4},
        "verified": bool(ok),
    }
    if cpu:
        from bench_gptq import cpu_info

        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oq_oracle as O

        host = {name: t.cpu().numpy() for name, t in acts.items()}
        orc = O.MinMaxOracle(0.0)
        t4 = time.perf_counter()
        for name, a in host.items():
            orc.collect(name, a)
        dt = time.perf_counter() - t4
        same = all(tuple(float(v) for v in orc.compute_range(n)) == tuple(float(v) for v in cal_many.compute_range(n)) for n in host)
        out["cpu_baseline"] = {"value": round(nbytes / dt / 1e9, 2), "unit": "GB/s", "cores": 1, "kind": "port",
                               "sample": f"one batch of the same {len(host)} tensors ({nbytes / 1e9:.2f} GB of the {nbytes * batches / 1e9:.0f} GB "
                                         f"workload) through the oracle's MinMaxOracle.collect (np.min + np.max per tensor), {dt:.2f} s; "
                                         "NumPy reductions are single-threaded",
                               "seconds": round(dt, 3), "ranges_equal_gpu": bool(same), **cpu_info()}
        out["verified"] = bool(ok and same)
    return out


def main():
    import torch

    print(json.dumps(run(torch.device("cuda", 0), cpu="--no-cpu-baseline" not in sys.argv)))


if __name__ == "__main__":
    main()
