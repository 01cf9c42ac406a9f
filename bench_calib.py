#!/usr/bin/env python3
"""Calibration benchmark (BASELINE.json config 3 stand-in): static int8 activation calibration over 512 samples of
a gemma-3-270m-shaped model.  gemma-3-270m itself is not obtainable offline, so the activation population is
synthetic with the same shapes: per batch of 10 samples, 18 layers x { [10,512,640] x 2 (attention
and MLP inputs), [10,512,1024] (o_proj input), [10,512,2048] (down_proj input) } fp32; 51 batches (the reference
drops the 2 remaining samples, calibrate.py:161-170).  Every batch goes through MinMaxCalibrator.collect_many on the GPU (ONE launch pair per batch of 72 tensors, running
state on the device; the per-tensor `collect` loop of calibrate.py:264-266 is timed next to it), then compute_range +
_compute_qparams per tensor.
Prints one JSON line: tensors/s, GB/s of activation bytes reduced, and the large-tensor reduction rate.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    import torch

    from onnx_quantize_amd import QuantType
    from onnx_quantize_amd.algorithms.functional import _compute_qparams
    from onnx_quantize_amd.calibration import MinMaxCalibrator
    from onnx_quantize_amd.hip import ops

    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(3)
    layers, batches = 18, 51
    shapes = [("attn_in", 640), ("mlp_in", 640), ("o_in", 1024), ("down_in", 2048)]
    # one batch worth of distinct tensors (1.6 GB) is generated once and re-fed with a per-batch scale factor, so the
    # timed region measures the reductions, not the random-number generator; values differ per batch through the scale
    acts = {f"l{l}.{nm}": torch.randn((10, 512, c), generator=gen, device=dev) * (0.1 + 9.9 * torch.rand(c, generator=gen, device=dev))
            for l in range(layers) for nm, c in shapes}
    nbytes = sum(t.numel() * 4 for t in acts.values())
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
    # the same statistics with one launch pair per batch (oq_minmax_collect_many_f32): what an on-device driver calls
    cal_many = MinMaxCalibrator()
    cal_many.collect_many(acts)
    torch.cuda.synchronize()
    cal_many = MinMaxCalibrator()
    t0 = time.perf_counter()
    for b in range(batches):
        cal_many.collect_many(acts)
    torch.cuda.synchronize()
    t_collect = time.perf_counter() - t0
    for name in list(acts)[:8]:
        assert cal_many.data[name].min_val == cal.data[name].min_val and cal_many.data[name].max_val == cal.data[name].max_val
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
        assert qparams[name][0].tobytes() == qparams_ref[name][0].tobytes() and int(qparams[name][1]) == int(qparams_ref[name][1])
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
    # the same 126 matrices through the many-tensor entry point (three launches for the whole model)
    ops.rtn_quantize_tensor_many(weights, "int8", True)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    wq_many = ops.rtn_quantize_tensor_many(weights, "int8", True)
    torch.cuda.synchronize()
    t_weights_many = time.perf_counter() - t3
    assert all(torch.equal(a[0], b[0]) and float(a[1]) == float(b[1]) for a, b in zip(wq[:5], wq_many[:5]))
    wparams = sum(w.numel() for w in weights)
    # correctness spot check against torch
    for name in list(acts)[:6]:
        assert cal.data[name].min_val == acts[name].min().item() and cal.data[name].max_val == acts[name].max().item()
    # large-tensor reduction rate (kernel-level): one 1.3 GB tensor
    big = torch.randn((64, 2048, 2560), generator=gen, device=dev)
    st = ops.minmax_state(dev)
    for _ in range(3):
        ops.minmax_collect(big, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.minmax_collect(big, st)
    e1.record()
    torch.cuda.synchronize()
    big_gbs = big.numel() * 4 * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    print(json.dumps({
        "metric": "activation GB/s reduced, min-max calibration, gemma-3-270m-shaped synthetic, 512 samples",
        "value": round(nbytes * batches / t_collect / 1e9, 1), "unit": "GB/s", "n_gpus": 1,
        "config": {"workload": "calibration_minmax_gemma3_270m_shapes", "layers": layers, "batches": batches,
                   "tensors_per_batch": len(acts), "bytes_per_batch": nbytes},
        "seconds": {"collect": round(t_collect, 4), "ranges_and_qparams": round(t_params, 4),
                    "collect_per_tensor_calls": round(t_per_tensor, 4), "ranges_and_qparams_per_name_calls": round(t_params_per_name, 4),
                    "weights_rtn_int8_per_tensor": round(t_weights, 4), "weights_rtn_int8_per_tensor_one_call": round(t_weights_many, 5)},
        "weights": {"matrices": len(weights), "params": wparams, "M_params_per_s": round(wparams / t_weights / 1e6, 1),
                    "M_params_per_s_one_call": round(wparams / t_weights_many / 1e6, 1)},
        "tensors_per_s": round(len(acts) * batches / t_collect, 1),
        "per_tensor_calls": {"GBs": round(nbytes * batches / t_per_tensor / 1e9, 1),
                             "tensors_per_s": round(len(acts) * batches / t_per_tensor, 1)},
        "roofline": {"bound": "hbm", "achieved": round(big_gbs, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(big_gbs / 8000.0, 4), "kernel": "oq::minmax_partial<float> (1.3 GB tensor)"},
    }))


if __name__ == "__main__":
    main()
