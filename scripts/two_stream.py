import os, sys, ctypes as C, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_quantize_amd.hip import _lib as L
lib = L.load()
K, N, G = 4096, 11008, 128
dev = torch.device("cuda")
ws = [torch.randn((K, N), device=dev) for _ in range(4)]
groups = K * N // G
for layout, qn in ((L.OQ_LAYOUT_NBITS, K * N // 2), (L.OQ_LAYOUT_KN, K * N)):
    outs = [(torch.empty(qn, dtype=torch.uint8, device=dev), torch.empty(groups, device=dev), torch.empty(groups, dtype=torch.uint8, device=dev)) for _ in ws]
    wsb = [torch.empty(lib.oq_rtn_workspace_bytes(K, N, 2, G, 0) + 256, dtype=torch.uint8, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def step(i, s):
        w = ws[i % 4]; q, sc, z = outs[i % 4]; wb = wsb[s]
        st = lib.oq_rtn_quantize_f32(C.c_void_p(w.data_ptr()), K, N, N, L.OQ_UINT4, L.OQ_GROUP, G, 0, 0, 1.0, 0, C.c_void_p(q.data_ptr()),
                                     C.c_void_p(sc.data_ptr()), C.c_void_p(z.data_ptr()), layout, C.c_void_p(wb.data_ptr()), wb.numel(),
                                     C.c_void_p(streams[s].cuda_stream))
        assert st == 0
    for nstreams in (1, 2):
        for i in range(40): step(i, i % nstreams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 400
        for i in range(iters): step(i, i % nstreams)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / iters * 1e6
        print(f"layout={layout} streams={nstreams}: {us:.2f} us/matrix  {204660736/us/1e6:.2f} TB/s", flush=True)
