"""On-device calibration driver.

The reference runs the augmented model through onnxruntime, keeps EVERY batch of EVERY tapped activation as a NumPy
array in a list (`core/_calibration/calibrate.py:204-251`), and only then walks that list: once per activation kind
through the calibrator (`_set_qparams`, :254-285) and once to concatenate the GPTQ inputs (`_set_qparams_gptq`,
:288-307).  Here the activations never leave HBM and are consumed batch by batch as the model produces them:

* range statistics  -> `MinMaxCalibrator.collect_many` (one launch pair per batch, `oq_minmax_collect_many_f32`);
* GPTQ inputs       -> one `HessianAccumulator` per distinct value name, fed through `oq_hessian_accumulate_f32`
                       (the running form of gptq.py:246-260: the same H as the reference's single call on the
                       concatenation, see `HessianAccumulator`);
* SmoothQuant stats -> running per-channel absmax (`oq_absmax_f32`, smooth_quant.py:62-69);
* AWQ needs the activations themselves: `keep_names` holds those (and only those) in HBM.

What produces the activations is a *runner*: any callable ``runner(feed) -> {value name: tensor in HBM}``.
`TorchRunner` taps the inputs / outputs of sub-modules of a torch-ROCm model; an onnxruntime session with the ROCm or
MIGraphX provider and I/O binding fits the same protocol.  Nothing here falls back to a CPU path: the consumers are
the HIP library's kernels and raise if it is missing.
"""
from __future__ import annotations

from collections.abc import Callable, Iterable, Mapping

import numpy as np

from .calibration import MinMaxCalibrator

__all__ = ["prepare_calibration_data", "generate_random_calibration_data", "HessianAccumulator", "ActivationStream",
           "TorchRunner", "run_calibration", "quantize_weights_gptq"]


def prepare_calibration_data(calibration_data, batch_size: int, num_samples: int):
    """calibrate.py:150-172: ``[B, batch, ...]`` view of the first ``num_samples`` samples; a trailing partial batch is
    dropped; ``batch_size >= num_samples`` gives one batch.  NumPy arrays and torch tensors alike (no copy)."""
    total = calibration_data.shape[0]
    num_samples = min(num_samples, total)
    data = calibration_data[:num_samples]
    if batch_size >= num_samples:
        return data.reshape((1, num_samples, *data.shape[1:]))
    batches = num_samples // batch_size
    return data[: batches * batch_size].reshape((batches, batch_size, *data.shape[1:]))


def generate_random_calibration_data(num_samples: int, inputs):
    """calibrate.py:127-147.  ``inputs``: [(name, shape, numpy dtype)], symbolic dimensions given as str / None.  One
    generator seeded 0 serves all inputs in order; integer inputs draw from [0, 100); a single input gives a bare array."""
    rng = np.random.default_rng(0)

    def one(shape, dtype):
        dims = [num_samples] + [d if isinstance(d, int) else 1 for d in list(shape)[1:]]
        dtype = np.dtype(dtype)
        if np.issubdtype(dtype, np.integer):
            return rng.integers(0, 100, size=dims, dtype=dtype)
        return rng.standard_normal(size=dims).astype(dtype)

    inputs = list(inputs)
    if len(inputs) == 1:
        return one(inputs[0][1], inputs[0][2])
    return {name: one(shape, dtype) for name, shape, dtype in inputs}


class HessianAccumulator:
    """gptq.py:246-260 as a running sum over batches.  The reference calls `_accumulate_hessian` once on the
    concatenation of all batches (gptq.py:304-305 after calibrate.py:301-302), i.e. H = (2 / n) sum_b X_b^T X_b with n =
    the total leading dimension; feeding the same update rule batch by batch gives that H exactly in real arithmetic
    (H <- H n/(n+b) + (2/(n+b)) X_b^T X_b telescopes) and to fp32 rounding on the device, without ever holding more
    than one batch.  ``n`` counts leading-dimension entries (samples), not tokens (:247)."""

    def __init__(self, k: int, device):
        import torch
        self.h = torch.zeros((k, k), dtype=torch.float32, device=device)
        self.n = 0

    def add(self, x) -> None:
        from .hip import ops
        self.n = ops.hessian_accumulate(x, self.h, self.n)


class ActivationStream:
    """Consumer side of the driver: `_set_qparams` x kinds + `_set_qparams_gptq` (calibrate.py:254-307), streamed.

    input_names / output_names: value names whose range statistics feed static input / output activation
    quantization; hessian_names: inputs of the nodes a calibration-hungry weight algorithm (GPTQ) quantizes;
    absmax_names: per-channel absmax (SmoothQuant); keep_names: activations held whole (AWQ).

    Reference detail kept: with BOTH kinds calibrated the reference walks the activation list twice through the SAME
    calibrator (calibrate.py:355-373), every tapped name each time; input parameters are read after the first walk,
    output parameters after the second.  For running min / max that changes nothing; with ``momentum > 0`` the
    output ranges are an EMA over the batch sequence seen twice.  The stream reproduces that from the per-batch
    extrema (two floats per name and batch, kept on the device) instead of from the activations."""

    def __init__(self, *, calibrator=None, input_names: Iterable[str] = (), output_names: Iterable[str] = (),
                 hessian_names: Iterable[str] = (), absmax_names: Iterable[str] = (), keep_names: Iterable[str] = (),
                 hessian_streams: int = 0, statistics_names: Iterable[str] = (), statistics_after_bytes: int = 8 << 30):
        self.calibrator = calibrator if calibrator is not None else MinMaxCalibrator()
        self.input_names, self.output_names = list(dict.fromkeys(input_names)), list(dict.fromkeys(output_names))
        self.hessian_names, self.absmax_names = set(hessian_names), set(absmax_names)
        self.keep_names = set(keep_names)
        # statistics_names: what the AWQ / SmoothQuant searches need of a value, as running statistics (`ops.SearchStatistics`:
        # Gram matrix, |x| sums and maxima) instead of the activations themselves -- `keep_names` without the memory
        self.statistics_names = set(statistics_names)
        self.statistics: dict = {}
        # ... but only once it matters: up to `statistics_after_bytes` of them the batches are simply held (a small model's
        # walk costs nothing that way and the searches see the reference's arrays); past it, what is held is folded into the
        # statistics and every later batch goes straight there
        self.statistics_after_bytes = int(statistics_after_bytes)
        self._held_for_search: dict[str, list] = {}
        self._held_bytes = 0
        self._search_inputs: dict = {}
        self.hessians: dict[str, HessianAccumulator] = {}
        # 0 (default): the Hessian updates of a batch in one grouped launch chain (per-tensor calls on 4 side streams when a
        # Hessian method the grouped chain does not run is selected); n > 0: per-tensor calls on n side streams
        self.hessian_streams = max(0, int(hessian_streams))
        self._side = None
        self.absmax: dict = {}
        self._kept: dict[str, list] = {}
        self._ranged = bool(self.input_names or self.output_names)
        momentum = getattr(self.calibrator, "momentum", 0.0)
        self._replay = bool(self.input_names and self.output_names and momentum > 0)
        self._extrema: list[dict] = []          # per batch {name: state tensor [4]} when the second walk must be replayed
        self._after_first_walk = None
        self.batches = 0

    # ------------------------------------------------------------------ one batch
    def feed(self, activations: Mapping) -> None:
        import torch
        from .hip import ops

        for name, t in activations.items():
            if not (isinstance(t, torch.Tensor) and t.is_cuda):
                raise TypeError(f"ActivationStream.feed: '{name}' is not a tensor in GPU memory (the driver keeps "
                                "activations on the device; use MinMaxCalibrator.collect for host arrays)")
        if self._ranged:
            # calibrate.py:264-266 collects every tapped name, whichever kind is being set
            if self._replay:
                scratch = MinMaxCalibrator(0.0)
                scratch.collect_many(activations)                      # the batch's own extrema
                per_batch = {n: d._state for n, d in scratch.data.items()}
                self._extrema.append(per_batch)
                self.calibrator.collect_many({n: st[:2] for n, st in per_batch.items()})
            else:
                self.calibrator.collect_many(activations)
        wanted = sorted(self.hessian_names & activations.keys())
        streams = self.hessian_streams
        if wanted and streams == 0 and ops.hessian_method() not in ("auto", "f16x3"):
            streams = 4      # the grouped chain runs the fp16-piece kernels only: other methods would become serial per-tensor
                             # calls on one stream, so they take the side-stream route below
        if wanted and streams == 0:
            # one launch chain for all tapped inputs of the batch (ops.hessian_accumulate_many): a batch of a small model is
            # 72 tensors of 6 to 36 product tiles each, launch-bound and never filling the chip one at a time
            xs = []
            for name in wanted:
                x = activations[name]
                x32 = x if x.dtype == torch.float32 else x.to(torch.float32)
                if name not in self.hessians:
                    self.hessians[name] = HessianAccumulator(x32.shape[-1], x32.device)
                xs.append(x32)
            accs = [self.hessians[name] for name in wanted]
            for acc, n in zip(accs, ops.hessian_accumulate_many(xs, [a.h for a in accs], [a.n for a in accs])):
                acc.n = n
        elif wanted:
            # per-tensor calls spread over a few side streams (fork / join by events): the route for the Hessian methods
            # the grouped call does not run (hessian_streams > 0 selects it)
            cur = torch.cuda.current_stream()
            if self._side is None:
                self._side = [torch.cuda.Stream(device=activations[wanted[0]].device) for _ in range(streams)]
            fork = cur.record_event()
            for i, name in enumerate(wanted):
                x = activations[name]
                side = self._side[i % len(self._side)] if self._side else cur
                side.wait_event(fork)
                with torch.cuda.stream(side):
                    x32 = x if x.dtype == torch.float32 else x.to(torch.float32)
                    acc = self.hessians.get(name)
                    if acc is None:
                        acc = self.hessians[name] = HessianAccumulator(x32.shape[-1], x32.device)
                    acc.add(x32)
                x.record_stream(side)
            for side in self._side:
                cur.wait_stream(side)
        for name in self.absmax_names & activations.keys():
            x = activations[name]
            cur = ops.absmax(x if x.dtype == torch.float32 else x.to(torch.float32))
            self.absmax[name] = cur if name not in self.absmax else torch.maximum(self.absmax[name], cur)
        names = sorted(self.statistics_names & activations.keys())
        if names and not self.statistics and self._held_bytes <= self.statistics_after_bytes:
            for name in names:
                self._held_for_search.setdefault(name, []).append(activations[name])
                self._held_bytes += activations[name].numel() * activations[name].element_size()
            if self._held_bytes > self.statistics_after_bytes:         # too much to hold: fold what is there, stream from now on
                held, self._held_for_search = self._held_for_search, {}
                for name, batches in held.items():
                    self.statistics[name] = ops.SearchStatistics(batches[0].shape[-1], batches[0].device)
                for i in range(max(len(b) for b in held.values())):
                    part = [n for n in held if i < len(held[n])]
                    ops.SearchStatistics.add_many([self.statistics[n] for n in part], [held[n][i] for n in part])
        elif names:
            for name in names:
                if name not in self.statistics:
                    x = activations[name]
                    self.statistics[name] = ops.SearchStatistics(x.shape[-1], x.device)
            ops.SearchStatistics.add_many([self.statistics[n] for n in names], [activations[n] for n in names])
        for name in self.keep_names & activations.keys():
            self._kept.setdefault(name, []).append(activations[name])
        self.batches += 1

    # ------------------------------------------------------------------ results
    def _qparams(self, calibrator, names, qargs):
        have = [n for n in names if n in calibrator.data]                # calibrate.py:271 unseen names are skipped
        return calibrator.compute_qparams_many(have, qargs.dtype, qargs.symmetric, qargs.reduce_range)

    def input_qparams(self, qargs) -> dict:
        """{value name: (scale fp32 0-d, zero_point 0-d)} for the input kind (calibrate.py:355-363)."""
        if self._replay:
            self._finish_replay()
            return self._qparams(self._after_first_walk, self.input_names, qargs)
        return self._qparams(self.calibrator, self.input_names, qargs)

    def output_qparams(self, qargs) -> dict:
        """Same for the output kind (calibrate.py:365-373), after the second walk when both kinds are calibrated."""
        if self._replay:
            self._finish_replay()
        return self._qparams(self.calibrator, self.output_names, qargs)

    def _finish_replay(self) -> None:
        if self._after_first_walk is not None:
            return
        snap = MinMaxCalibrator(self.calibrator.momentum)
        for name, d in self.calibrator.data.items():
            snap.data[name] = type(d)(d._state.clone(), d._np_dtype)
        self._after_first_walk = snap
        for per_batch in self._extrema:                                  # the second walk, from the recorded extrema
            self.calibrator.collect_many({n: st[:2] for n, st in per_batch.items()})
        self._extrema = []

    def search_input(self, name: str):
        """What the AWQ / SmoothQuant searches get for the value `name` (`statistics_names`): its batches concatenated along axis
        0 while the walk could hold them (calibrate.py:301-302), its `ops.SearchStatistics` once it could not.  One object per
        name, whoever asks."""
        if name not in self._search_inputs:
            import torch
            if name in self.statistics:
                self._search_inputs[name] = self.statistics[name]
            elif name in self._held_for_search:
                self._search_inputs[name] = torch.cat(self._held_for_search.pop(name), dim=0)
            else:
                raise KeyError(name)
        return self._search_inputs[name]

    def kept(self, name: str):
        """calibrate.py:301-302: the batches of one kept activation concatenated along axis 0 (on the device)."""
        import torch
        return torch.cat(self._kept[name], dim=0)


class TorchRunner:
    """Runner over a torch-ROCm module: ``taps`` maps a value name to (sub-module path, "input" | "output"); the
    tensor entering (first positional argument) resp. leaving that sub-module is handed to the stream under that name.
    Hooks hold the tensors only until the batch has been consumed."""

    def __init__(self, module, taps: Mapping[str, tuple[str, str]]):
        self.module = module
        self._captured: dict = {}
        self._handles = []
        modules = dict(module.named_modules())
        for value_name, (path, kind) in taps.items():
            if path not in modules:
                raise KeyError(f"TorchRunner: no sub-module '{path}'")
            if kind == "input":
                hook = (lambda name: lambda _m, args: self._captured.__setitem__(name, args[0].detach()))(value_name)
                self._handles.append(modules[path].register_forward_pre_hook(hook))
            elif kind == "output":
                hook = (lambda name: lambda _m, _a, out: self._captured.__setitem__(name, out.detach()))(value_name)
                self._handles.append(modules[path].register_forward_hook(hook))
            else:
                raise ValueError("tap kind must be 'input' or 'output'")

    def __call__(self, feed) -> dict:
        import torch
        self._captured = {}
        with torch.no_grad():
            if isinstance(feed, Mapping):
                self.module(**feed)
            else:
                self.module(feed)
        out, self._captured = self._captured, {}
        return out

    def close(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []


def run_calibration(runner: Callable, calibration_data, stream: ActivationStream, *, num_samples: int = 100,
                    batch_size: int = 10, input_names=None) -> ActivationStream:
    """calibrate.py:225-251 without the list: batches the calibration data like the reference, runs ``runner`` on each
    batch and hands what it returns straight to ``stream``.  ``calibration_data``: one array / tensor (single-input
    model) or {input name: array}; ``input_names``: the model's input names (needed for the reference's
    multi-input check, :228-233, and to name a bare array)."""
    import torch

    if input_names is not None and len(input_names) > 1 and not isinstance(calibration_data, Mapping):
        raise ValueError("Calibration data must be a dict mapping input names to arrays for multi-input models.")
    bare = not isinstance(calibration_data, Mapping)
    named = {(input_names[0] if input_names else "input"): calibration_data} if bare else dict(calibration_data)
    batched = {}
    for name, data in named.items():
        if isinstance(data, np.ndarray):
            data = torch.from_numpy(np.ascontiguousarray(data))
        batched[name] = prepare_calibration_data(data, batch_size, num_samples)
    n_batches = len(next(iter(batched.values())))
    for i in range(n_batches):
        feed = {name: data[i].cuda(non_blocking=True) for name, data in batched.items()}
        arg = next(iter(feed.values())) if bare else feed
        if getattr(runner, "takes_sink", False):
            # a runner that can hand each tapped value over as it is produced (GraphRunner): consumed one by one, the tapped
            # activations of a batch never coexist in HBM
            left = runner(arg, sink=lambda name, t: stream.feed({name: t}))
            if left:
                stream.feed(left)
        else:
            stream.feed(runner(arg))
    return stream


def quantize_weights_gptq(layers: Mapping, hessians: Mapping[str, HessianAccumulator], quant_type: str, strategy: str,
                          group_size, *, symmetric=False, reduce_range=False, clip_ratio=1.0, block_size=128,
                          percdamp=0.01, actorder=False, mse=False, mode: str = "parity",
                          factor_batch_bytes: int = 24 << 30) -> dict:
    """`_gptq` (gptq.py:76-243) for every ``{layer name: (W [K, N] in HBM, input value name)}``, with everything that
    depends on H alone (dead channels, permutation, inverse factor) computed once per distinct input: the reference
    repeats it per node, nodes that share an input (q/k/v, gate/up) share it here.  Returns
    {layer name: (q, scale, zero_point, info)} on the device."""
    import torch

    from .hip import ops

    by_input: dict[str, list[str]] = {}
    for name, (_w, value) in layers.items():
        if value not in hessians:
            raise KeyError(f"quantize_weights_gptq: no Hessian accumulated for input '{value}' of '{name}'")
        by_input.setdefault(value, []).append(name)
    out = {}
    # Inputs of one width are factored in lock-step (oq_gptq_factor_batched_f32: the latency of one chain of diagonal
    # blocks for the whole batch), in batches bounded by `factor_batch_bytes` of workspace + stacked copies.
    by_width: dict[int, list[str]] = {}
    for value in by_input:
        by_width.setdefault(int(hessians[value].h.shape[0]), []).append(value)
    for k, values in by_width.items():
        per_matrix = 7 * k * k * 4                       # stacked H, U, and the factor's P, Lt, X, Y (+ Dinv)
        step = 1 if actorder else max(1, min(len(values), factor_batch_bytes // per_matrix))
        for b0 in range(0, len(values), step):
            chunk = values[b0:b0 + step]
            if len(chunk) == 1:
                shared_list = [ops.gptq_shared_factor(hessians[chunk[0]].h, percdamp, actorder)]
            else:
                shared_list = ops.gptq_shared_factors(torch.stack([hessians[v].h for v in chunk]).contiguous(), percdamp)
            for value, shared in zip(chunk, shared_list):
                h = hessians[value].h
                for name in by_input[value]:
                    out[name] = ops.gptq_quantize(layers[name][0], h, quant_type, strategy, group_size, symmetric, reduce_range,
                                                  clip_ratio, block_size, percdamp, actorder, mse, mode=mode, shared=shared)
    return {name: out[name] for name in layers}
