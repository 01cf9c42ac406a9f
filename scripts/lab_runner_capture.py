import sys, os, time, logging, importlib.util
sys.path.insert(0, '/root/repo')
import torch
logging.basicConfig(level=logging.DEBUG)
from onnx_quantize_amd.graph_runner import GraphRunner
spec = importlib.util.spec_from_file_location("g", "/root/repo/examples/gemma3_shapes/gemma3_onnx_file.py"); ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
model = ex.build_model(layers=18, vocab=4096)
targets = [n for n in model.graph.node if n.op_type == "MatMul" and "lm_head" not in n.name]
wanted = list(dict.fromkeys(n.input[0] for n in targets))
data = ex.make_calibration_data(18, 4096, 24, 256)
for cap in (False, True):
    r = GraphRunner(model, outputs=wanted, device="cuda", capture=cap)
    ts = []
    for i in range(24):
        feed = {k: torch.from_numpy(v[i:i + 1]) for k, v in data.items()}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = r(feed)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("capture", cap, "ms per pass:", [round(t * 1e3, 1) for t in ts[:4]], "...", round(sum(ts[4:]) / len(ts[4:]) * 1e3, 2), "graphs", [v is not None for v in r._graphs.values()], flush=True)
