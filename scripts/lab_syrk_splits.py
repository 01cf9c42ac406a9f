"""Lab (library built with --define OQ_SYRK_LAB): the fp16-piece SYRK product alone (pieces prepared before the clock) by
slice count, K and rows per call.  Prints ms per product call (GEMM + slab reduce) by HIP events."""
import ctypes as C, os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for K, T, splits in [(4096, 65536, (0, 1, 2, 3, 5, 7, 9, 15)), (4096, 262144, (0, 3, 7, 15)), (11008, 65536, (0, 1, 4)), (11008, 262144, (0, 1, 4))]:
    x = torch.randn((T, K), generator=g, device=dev) * (0.1 + 3.9 * torch.rand(K, generator=g, device=dev))
    pb = lib.oq_hessian_pieces_bytes(T, K)
    pieces = torch.empty(pb + 256, dtype=torch.uint8, device=dev)
    off = (-pieces.data_ptr()) % 256
    sb = lib.oq_hessian_slab_bytes(K)
    slabs = torch.empty(sb, dtype=torch.uint8, device=dev)
    h = torch.zeros((K, K), device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.oq_hessian_prepare_f32(C.c_void_p(x.data_ptr()), T, K, K, T, C.c_void_p(pieces.data_ptr() + off), pb, st))
    for c in splits:
        if c: os.environ["OQ_SYRK_SPLITS"] = str(c)
        else: os.environ.pop("OQ_SYRK_SPLITS", None)
        def prod():
            L.check(lib.oq_hessian_accumulate_prepared_f32(C.c_void_p(pieces.data_ptr() + off), T, K, 0, T, C.c_void_p(h.data_ptr()),
                                                           C.c_void_p(slabs.data_ptr()), sb, st))
        for _ in range(3): prod()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10 if T <= 65536 else 4
        e0.record()
        for _ in range(reps): prod()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tiles = ((K + 255) // 256) * ((K + 255) // 256 + 1) // 2
        print(json.dumps({"K": K, "T": T, "splits": c or "auto", "ms": round(ms, 3), "PFLOPs_executed": round(tiles * 65536 * T * 6 / ms / 1e12, 3)}), flush=True)
    del x, pieces, slabs, h
