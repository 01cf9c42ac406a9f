"""Phase stamps of gptq_rows16_kernel (lab build: python -m onnx_quantize_amd._build --define OQ_LOOP_STAMPS)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import ops, _lib as L
lib = C.CDLL(L.LIB_PATH)
for K, N in ((4096, 4096), (4096, 11008)):
    x = torch.randn((8, 1024, K), device="cuda")
    h = torch.zeros((K, K), device="cuda")
    ops.hessian_accumulate(x, h, 0)
    u, info = ops.gptq_factor(h, 0.01)
    w = torch.randn((K, N), device="cuda") * 0.02
    _, s0, z0 = ops.rtn_quantize(w, "int4", "channel", -1, emit_q=False)
    for rep in range(2):
        ops.gptq_loop(w.clone(), u, "int4", 128, False, False, 1.0, False, 128, "corrected", s0, z0)
        torch.cuda.synchronize()
    st = (C.c_ulonglong * 16)()
    lib.oq_lab_loop_stamps(st)
    t = list(st)
    names = ["start", "issued", "vmcnt0", "barrier"] + [f"slab{k}" for k in range(8)] + ["end"]
    print(f"K={K} N={N} (last launch of the loop), cycles since start:")
    print("  " + "  ".join(f"{n}={t[i] - t[0]}" for i, n in enumerate(names)))
    print(f"  dma_issued={t[13] - t[0]} tile_issued={t[14] - t[0]}")
