"""The reference's single call site of the numeric hot path, device resident (reference: qrules/_common.py:126-142).

``quantize_weights(op, w, qconfig, out, is_matmul_nbits_compatible)`` is what every rewrite rule of the reference calls
(qrules/_qdq/matmul_to_qmatmul.py:41,59,93,108; _qdq/gemm_to_qgemm.py; _qlinear/*.py).  As written it runs the algorithm
plugin NumPy -> NumPy and then, for MatMulNBits, repacks the result in NumPy (`_prepare_for_matmul_nbits`, :65-123).
Behind the same signature this module keeps the weight in HBM from upload to the final wire format:

    upload once (one blocking copy, staging.py)
      -> RTN | GPTQ | HQQ kernels on the device
      -> MatMulNBits blob written by the kernel epilogue (RTN, HQQ) or by oq_pack_matmul_nbits (GPTQ, RTN + mse)
      -> zero points nibble-packed by oq_pack_zero_points_u4
      -> ONE download of exactly the three arrays the reference would have handed to `op.initializer`.

`weight_arrays` returns those three arrays; `quantize_weights` wraps them into initializers like the reference.  It works
on this package's config classes and, after `integration.install_into_reference()`, on the reference's own (same field
names); an algorithm plugin this package does not know keeps the reference's NumPy route.
"""
from __future__ import annotations

import logging
from collections import OrderedDict  # noqa: F401  (annotation of _SHARED_INPUTS)

import numpy as np

__all__ = ["quantize_weights", "weight_arrays", "clear_shared_inputs", "shared_input_stats", "prefactor_streamed"]

logger = logging.getLogger(__name__)

_GPTQ_FALLBACK_WARNING = (
    "Failed to invert hessian due to numerical instability. Consider increasing percdamp, increasing the "
    "number of calibration samples, or shuffling the calibration dataset. Falling back to round-to-nearest "
    "for this module.")


# Nodes that read the same value (q / k / v, gate / up) carry the SAME calibration array in `node.meta["input"]`
# (calibrate.py:301-307 concatenates once per value name and hands that object to every consumer).  Everything `_gptq`
# derives from the input alone -- the Hessian, dead channels, permutation, inverse factor (gptq.py:118-150, :246-260) -- is
# computed once per CONTENT and kept for the next nodes; consumers of one value are neighbours in graph order, so two entries
# are enough.  An array input is uploaded (it has to be in HBM for the Hessian anyway) and fingerprinted there in one pass at
# the HBM rate (`oq_fingerprint64`: 64 bits over every byte, shape and dtype beside it in the key): an array rewritten in
# place between two nodes -- the reference's AWQ pass does exactly that to this object, awq.py:191 -- is a different content
# and is computed afresh, wherever the edit sits (rounds 3-5 compared a 512-element host sample).  A Hessian streamed by the
# calibration walk (`StreamedGptqInput`) is keyed by the object: nothing rewrites it.
_SHARED_INPUTS: "OrderedDict[tuple, tuple]" = None
_SHARED_KEEP = 2
shared_input_stats = {"hits": 0, "misses": 0}


def clear_shared_inputs() -> None:
    """Drop the cached Hessians / factors (end of a `quantize()` run)."""
    if _SHARED_INPUTS is not None:
        _SHARED_INPUTS.clear()


def _hessian_and_factor(x, k, device, percdamp, actorder):
    """(H [K, K], shared factor dictionary) of calibration input ``x`` on ``device``."""
    global _SHARED_INPUTS
    from collections import OrderedDict

    import torch

    from .hip import ops

    from .reference_passes import StreamedGptqInput

    if _SHARED_INPUTS is None:
        _SHARED_INPUTS = OrderedDict()
    streamed = isinstance(x, StreamedGptqInput)       # the calibration walk already accumulated H on the device
    tail = (int(k), str(device), float(percdamp), bool(actorder), ops.hessian_method())
    key, x_dev = None, None
    if streamed:
        key = ("streamed", id(x), *tail)
    elif isinstance(x, (np.ndarray, torch.Tensor)):
        if isinstance(x, np.ndarray):
            from .staging import upload

            with torch.cuda.device(device):
                x_dev = upload(x)
        else:
            x_dev = x.to(device, torch.float32).contiguous()
        if x_dev.numel():
            key = ("content", tuple(x.shape), str(x.dtype), ops.fingerprint64(x_dev), *tail)
    if key is not None:
        hit = _SHARED_INPUTS.get(key)
        if hit is not None and (not streamed or hit[0] is x):
            _SHARED_INPUTS.move_to_end(key)
            shared_input_stats["hits"] += 1
            return hit[1], hit[2]
    shared_input_stats["misses"] += 1
    if streamed:
        if tuple(x.h.shape) != (k, k):
            raise ValueError(f"streamed Hessian of '{x.name}' is {tuple(x.h.shape)}, the weight has {k} input channels")
        h = x.h if x.h.device == torch.device(device) else x.h.to(device)
        ready = getattr(x, "factors", {}).get((float(percdamp), bool(actorder), ops.hessian_method(), str(h.device)))
        if ready is not None:                              # factored ahead, in one batched chain with its peers (prefactor_streamed)
            shared_input_stats["hits"] += 1
            return h, ready
    else:
        h = torch.zeros((k, k), dtype=torch.float32, device=device)
        n = 0
        if x_dev is not None:
            n = ops.hessian_accumulate(x_dev, h, n)
        else:
            batches = x if isinstance(x, (list, tuple)) or hasattr(x, "__next__") else [x]
            for b in batches:
                xb = b if isinstance(b, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32))
                n = ops.hessian_accumulate(xb.to(device, torch.float32), h, n)
    shared = ops.gptq_shared_factor(h, percdamp, actorder)
    if key is not None:
        _SHARED_INPUTS[key] = (x if streamed else None, h, shared)     # a content entry does not keep the caller's array alive
        while len(_SHARED_INPUTS) > _SHARED_KEEP:
            _SHARED_INPUTS.popitem(last=False)
    return h, shared


def prefactor_streamed(inputs, percdamp: float, actorder: bool = False, batch_bytes: int = 24 << 30) -> int:
    """Everything `_gptq` derives from H alone (gptq.py:118-150) for ALL streamed calibration inputs of a model at once: inputs of
    one width are factored in lock-step (`ops.gptq_shared_factors`: one chain of diagonal blocks for the whole batch instead of
    one per input -- 128 factors of Llama-2-7B in 0.22 s instead of 1.3-1.5 s), in batches bounded by `batch_bytes`.  The results
    are left on the inputs (`StreamedGptqInput.factors`) where `_hessian_and_factor` finds them.  Returns the number factored."""
    import torch

    from .hip import ops

    if actorder:                                           # the permutation makes every factor its own problem: per input, on demand
        return 0
    by_width: dict = {}
    for x in {id(v): v for v in inputs}.values():
        by_width.setdefault(int(x.h.shape[0]), []).append(x)
    done = 0
    for k, group in by_width.items():
        per_matrix = 7 * k * k * 4
        step = max(1, min(len(group), batch_bytes // per_matrix))
        for b0 in range(0, len(group), step):
            chunk = group[b0:b0 + step]
            if len(chunk) == 1:
                shared = [ops.gptq_shared_factor(chunk[0].h, percdamp, False)]
            else:
                shared = ops.gptq_shared_factors(torch.stack([x.h for x in chunk]).contiguous(), percdamp)
            for x, sh in zip(chunk, shared):
                if not hasattr(x, "factors"):
                    continue
                x.factors[(float(percdamp), False, ops.hessian_method(), str(x.h.device))] = sh
                done += 1
    return done


def _key(qtype) -> str:
    """"QUInt4" -> "uint4" for this package's QuantType and the reference's alike."""
    return qtype.name[1:].lower()


def _strategy(s) -> str:
    return s.value if hasattr(s, "value") else str(s)


def _upload(name, array):
    from .staging import upload

    return upload(array)


def _host(t, dtype=None):
    from .staging import download

    return download(t, dtype)


def _device_algorithm(w_dev, name, qconfig, out, want_blob: bool, want_packed4: bool = False):
    """Run the configured algorithm on the device.  Returns (q, scale, zp, is_blob) as device tensors -- q is the
    MatMulNBits blob when the algorithm's kernel wrote it directly -- or None for an algorithm this package has no
    kernels for.  ``want_packed4``: 4-bit integers of the plain route as [K, N/2] nibble pairs straight from the kernel's
    epilogue where it has one (RTN groups, the GPTQ loop; `is_blob` is then the string "packed4")."""
    import torch

    from .hip import ops

    a = qconfig.weights
    algo = a.algorithm
    tag = getattr(algo, "algorithm_type", None)
    qt, st = _key(a.dtype), _strategy(a.strategy)
    g = -1 if a.group_size is None else a.group_size
    k = w_dev.shape[0]
    if tag == "rtn":
        blob = want_blob and not a.mse and k % ops.resolve_group(st, k, g) == 0
        packed = (want_packed4 and not blob and not a.mse and a.dtype.bitwidth == 4 and st == "group" and w_dev.shape[1] % 2 == 0
                  and k % ops.resolve_group(st, k, g) == 0)
        q, s, z = ops.rtn_quantize(w_dev, qt, st, g, bool(a.symmetric), bool(a.reduce_range), float(a.clip_ratio), bool(a.mse),
                                   layout="nbits" if blob else ("kn_packed4" if packed else "kn"))
        return q, s, z, ("packed4" if packed else blob)
    if tag == "hqq":
        assert a.zp_dtype == a.scale_dtype                                          # hqq.py:175
        if qt != "uint4":
            raise ValueError("the GPU HQQ path implements the reference's only legal configuration: uint4")
        blob = want_blob and k % ops.resolve_group("group", k, g) == 0
        q, s, z, _ = ops.hqq_quantize(w_dev, g, bool(a.reduce_range), float(a.clip_ratio), bool(a.mse), float(algo.lp_norm),
                                      float(algo.beta), float(algo.kappa), int(algo.iters), bool(algo.early_stop),
                                      layout="nbits" if blob else "kn")
        return q, s, z, blob
    if tag == "gptq":
        assert out is not None, "Output value is required for GPTQ quantization."      # gptq.py:54
        node = out.producer()
        assert "input" in node.meta, "GPTQ requires calibration data in node meta."    # gptq.py:56
        h, shared = _hessian_and_factor(node.meta["input"], k, w_dev.device, float(algo.percdamp), bool(algo.actorder))
        packed = want_packed4 and not want_blob and a.dtype.bitwidth == 4 and w_dev.shape[1] % 2 == 0 and not bool(algo.actorder)
        q, s, z, info = ops.gptq_quantize(w_dev, h, qt, st, a.group_size, bool(a.symmetric), bool(a.reduce_range),
                                          float(a.clip_ratio), int(algo.block_size), float(algo.percdamp), bool(algo.actorder),
                                          bool(a.mse), mode=getattr(algo, "mode", "parity"), shared=shared,
                                          layout="kn_packed4" if packed else "kn")
        if int(info.item()) != 0:                                                     # gptq.py:143-150
            logger.warning(_GPTQ_FALLBACK_WARNING)
        return q, s, z, ("packed4" if packed else False)
    return None


def weight_arrays(w, qconfig, out=None, is_matmul_nbits_compatible: bool = False, packed4: bool = False):
    """The three NumPy arrays `qrules/_common.py:133-137` produces for weight value ``w``: what
    ``algorithm.quantize_weights`` returns (rtn.py:106-109 shapes: q [K, N]; scale / zero point 0-d | [N] | [N*K/g, 1]) or,
    with ``is_matmul_nbits_compatible``, what `_prepare_for_matmul_nbits` makes of it (B uint8 [N, K/g, g*bits/8], scales
    [N, K/g], zero points uint8 [N, ceil(K/g / 2)] nibble-packed | [N, K/g]).  ``packed4`` (an extension for writers): 4-bit
    integers of the plain route come back nibble-packed, flat uint8 [ceil(K N / 2)], instead of one per byte."""
    from .hip import ops

    a = qconfig.weights
    w_np = w.const_value.numpy()
    res = None
    if w_np.ndim == 2 and getattr(a.algorithm, "algorithm_type", None) in ("rtn", "hqq", "gptq"):
        resident = getattr(w, "device_value", None)      # a caller that already holds this weight in HBM (the file path's
        if resident is not None and (tuple(resident.shape) != tuple(w_np.shape) or str(resident.dtype) != "torch.float32"   # calibration walk)
                                     or not resident.is_cuda or not resident.is_contiguous()):
            resident = None
        if resident is None and getattr(w, "placeholder", False):      # the host array of such a value is zeros that carry the shape
            raise RuntimeError(f"'{w.name}': the weight lives in HBM only and its device copy cannot be used ({'missing' if getattr(w, 'device_value', None) is None else 'not contiguous fp32 of the declared shape'})")
        res = _device_algorithm(resident if resident is not None else _upload(w.name, w_np), w.name, qconfig, out, is_matmul_nbits_compatible,
                                want_packed4=packed4 and not is_matmul_nbits_compatible)
    if res is None:        # an algorithm plugin without kernels here: its own NumPy route, then the wire format on the GPU
        if getattr(w, "placeholder", False):
            raise RuntimeError(f"'{w.name}': the weight lives in HBM only; the NumPy route of this algorithm plugin would read a placeholder")
        w_q, w_scale, w_zp = a.algorithm.quantize_weights(w, qconfig, out=out)
        if is_matmul_nbits_compatible:
            from .wire_format import _prepare_for_matmul_nbits

            w_q, w_scale, w_zp = _prepare_for_matmul_nbits(w_q, w_scale, w_zp, qconfig)
        return w_q, w_scale, w_zp

    q, s, z, is_blob = res
    qdt = a.dtype.np_dtype
    if not is_matmul_nbits_compatible:
        if packed4 and a.dtype.bitwidth == 4:
            # for a writer that serialises the integers as an ONNX INT4 / UINT4 tensor (core/_pack.py:8-22: two values per byte
            # in flat order; with an even N the [K, N/2] pairs of the kernels' epilogue ARE that order): packed on the device --
            # by the RTN / GPTQ kernel itself where it has the epilogue, else by oq_pack_nibbles --, half the download, no NumPy
            # pass over the values on the host
            q_packed = q.reshape(-1) if is_blob == "packed4" else ops.pack_nibbles(q)
            return _host(q_packed), _host(s, a.scale_dtype), _host(z, a.zp_dtype)
        return _host(q, qdt), _host(s, a.scale_dtype), _host(z, a.zp_dtype)

    # qrules/_common.py:65-123 on the device
    k, n = w_np.shape
    bits = a.dtype.bitwidth
    g = a.group_size
    assert k % g == 0                                                                 # _common.py:72
    blocks = k // g
    if not is_blob:
        q = ops.pack_matmul_nbits(q, g, bits)
    float_zp = np.dtype(a.zp_dtype) == np.dtype(a.scale_dtype)                        # HQQ: zero points stay floats (:96-99)
    if bits == 4 and blocks > 1 and not float_zp:
        z = ops.pack_zero_points_u4(z.reshape(-1), n, blocks)
    blob = _host(q)
    scale = _host(s, a.scale_dtype).reshape(-1, blocks)
    zp = _host(z).reshape(n, -1).astype(a.zp_dtype if float_zp else np.uint8, copy=False)
    if bits != 4:
        blob = blob.astype(qdt, copy=False)
    return blob, scale, zp


def quantize_weights(op, w, qconfig, out=None, is_matmul_nbits_compatible: bool = False):
    """qrules/_common.py:126-142, same signature and same three initializers (`w`, `w/scale`, `w/zero_point`)."""
    import onnx_ir as ir

    w_q, w_scale, w_zero_point = weight_arrays(w, qconfig, out, is_matmul_nbits_compatible)
    w_q = op.initializer(ir.tensor(w_q), name=w.name)
    w_scale = op.initializer(ir.tensor(w_scale), name=f"{w.name}/scale")
    w_zero_point = op.initializer(ir.tensor(w_zero_point), name=f"{w.name}/zero_point")
    return w_q, w_scale, w_zero_point
