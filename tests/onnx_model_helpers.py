"""Helpers shared by tests/test_onnx_model.py (CPU) and tests/test_onnx_model_gpu.py: the oracle as the numeric provider
of `model_quantize.quantize_model` (weights, bias, calibration)."""
import os

import numpy as np
import torch

import oq_oracle as O
from conftest import ROOT
from onnx_quantize_amd import onnx_proto as P
from onnx_quantize_amd.graph_runner import GraphRunner
from onnx_quantize_amd.model_quantize import quantize_model

FIXTURES = os.path.join(ROOT, "tests", "golden", "onnx")


def fixture(name):
    return P.load_model(os.path.join(FIXTURES, name + ".onnx"))


def oracle_weight_arrays(value, cfg, out, nbits):
    """What `seam.weight_arrays` returns, from the oracle."""
    a = cfg.weights
    w = value.const_value.numpy()
    if w.ndim == 1:                                          # QDQ Gemm bias: per-tensor RTN on a vector
        q, s, z = O.rtn_quantize(w.reshape(1, -1), a.dtype.key, "tensor", -1, a.symmetric, a.reduce_range, a.clip_ratio, a.mse)
        return q.reshape(w.shape), s, z
    x = None if out is None else out.producer().meta.get("input")
    tag = getattr(a.algorithm, "algorithm_type", "rtn")
    algo = {k: getattr(a.algorithm, k) for k in ("block_size", "percdamp", "actorder", "lp_norm", "beta", "kappa", "iters", "early_stop")
            if hasattr(a.algorithm, k)}
    return O.seam_arrays(w, tag, a.dtype.key, a.strategy.value, a.group_size, a.symmetric, a.reduce_range, a.clip_ratio, a.mse,
                         x=x, nbits=nbits, **algo)


def oracle_calibrate(runner_device="cpu"):
    """calibrate.py:310-380 restated on the oracle: the activations of each batch from a `GraphRunner` on `runner_device`
    (downloaded: the reference's list of NumPy dicts), `O.calibrate_flow` for the ranges, `O.gptq_inputs` for the GPTQ inputs."""

    def calibrate(model, G, targets, qconfig, device, keep_inputs=False):
        from onnx_quantize_amd.calibration_driver import generate_random_calibration_data
        from onnx_quantize_amd.model_quantize import _model_inputs

        cal_in = qconfig.input_activations is not None and qconfig.input_activations.is_static
        cal_out = qconfig.output_activations is not None and qconfig.output_activations.is_static
        algo = qconfig.weights.algorithm.requires_calibration or any(p.requires_calibration for p in qconfig.preprocessors)
        in_names, out_names = [n.input[0] for n in targets], [n.output[0] for n in targets]
        wanted = list(dict.fromkeys((in_names if (cal_in or algo) else []) + (out_names if cal_out else [])))
        params = qconfig.calibration_params
        inputs = _model_inputs(G)
        data = qconfig.calibration_data
        if data is None:
            data = generate_random_calibration_data(params.num_samples, inputs)
        named = data if isinstance(data, dict) else {inputs[0][0]: data}
        batched = {k: O.prepare_calibration_data(np.asarray(v), params.batch_size, params.num_samples) for k, v in named.items()}
        runner = GraphRunner(model, outputs=wanted, device=runner_device)
        acts = []
        for i in range(len(next(iter(batched.values())))):
            got = runner({k: torch.from_numpy(np.ascontiguousarray(v[i])) for k, v in batched.items()})
            acts.append({k: v.cpu().numpy() for k, v in got.items()})
        args = lambda a: (a.dtype.key, a.symmetric, a.reduce_range)      # noqa: E731
        flow = O.calibrate_flow(acts, in_names if cal_in else [], out_names if cal_out else [], params.momentum,
                                args(qconfig.input_activations) if cal_in else None, args(qconfig.output_activations) if cal_out else None)
        meta = {id(n): {} for n in targets}
        for n, i_name, o_name in zip(targets, in_names, out_names):
            for kind, name in (("input", i_name), ("output", o_name)):
                if (kind, name) in flow:
                    s, z = flow[(kind, name)]
                    meta[id(n)][f"{kind}_scale"], meta[id(n)][f"{kind}_zero_point"] = np.asarray(s), np.asarray(z)
        if algo:
            xs = O.gptq_inputs([{k: b[k] for k in dict.fromkeys(in_names)} for b in acts])
            for n, name in zip(targets, in_names):
                meta[id(n)]["input"] = xs[name]
        return meta

    return calibrate


class OracleSearches:
    """pre_passes/smooth_quant.py:62-113 and awq.py:114-259 from the oracle (NumPy inputs)."""

    @staticmethod
    def _np(x):
        return x if isinstance(x, np.ndarray) else x.cpu().numpy()

    def smooth_quant_scale(self, x, w, alpha):
        return O.smooth_quant_scale(self._np(x), w, alpha)

    def awq_scale_search(self, x, w, a):
        best, _losses = O.awq_scale_search(self._np(x), w, a.dtype.key, a.strategy.value, a.group_size, a.symmetric, a.reduce_range)
        return best

    def awq_clip_search(self, x, w, a):
        ratio, _losses = O.awq_clip_search(self._np(x), w, a.dtype.key, a.strategy.value, a.group_size, a.symmetric, a.reduce_range)
        return ratio


def q_oracle(model, qc, runner_device="cpu"):
    return quantize_model(model, qc, weight_arrays=oracle_weight_arrays, quantize_bias=O.quantize_bias,
                          calibrate=oracle_calibrate(runner_device), searches=OracleSearches())
