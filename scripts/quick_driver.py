"""On-device calibration driver on a gemma-3-270m-shaped stack: 18 blocks x 4 Linear layers, 51 batches of
[10, 512, 640]; static ranges for every Linear input and output + a streamed GPTQ Hessian for every Linear input.
python scripts/quick_driver.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onnx_quantize_amd.calibration import MinMaxCalibrator
from onnx_quantize_amd.calibration_driver import ActivationStream, TorchRunner, quantize_weights_gptq
from onnx_quantize_amd.config import QActivationArgs
from onnx_quantize_amd.dtypes import QuantType

dev = torch.device("cuda", 0)
torch.manual_seed(0)


class Block(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.qkv = torch.nn.Linear(640, 1024, bias=False)
        self.o = torch.nn.Linear(1024, 640, bias=False)
        self.up = torch.nn.Linear(640, 2048, bias=False)
        self.down = torch.nn.Linear(2048, 640, bias=False)

    def forward(self, x):
        x = x + self.o(torch.tanh(self.qkv(x)))
        return x + self.down(torch.nn.functional.gelu(self.up(x)))


model = torch.nn.Sequential(*[Block() for _ in range(18)]).to(dev)
taps, in_names, out_names = {}, [], []
for i in range(18):
    for name in ("qkv", "o", "up", "down"):
        taps[f"{i}.{name}/in"] = (f"{i}.{name}", "input")
        taps[f"{i}.{name}/out"] = (f"{i}.{name}", "output")
        in_names.append(f"{i}.{name}/in")
        out_names.append(f"{i}.{name}/out")
runner = TorchRunner(model, taps)
batches = [torch.randn((10, 512, 640), device=dev) for _ in range(4)]       # reused round robin: 51 batches
args = QActivationArgs(dtype=QuantType.QUInt8, is_static=True)


def run(hessians: bool, hessian_streams: int = 0):
    stream = ActivationStream(calibrator=MinMaxCalibrator(), input_names=in_names, output_names=out_names,
                              hessian_names=in_names if hessians else (), hessian_streams=hessian_streams)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    t_model = t_feed = 0.0
    t0 = time.perf_counter()
    for b in range(51):
        ta = time.perf_counter()
        acts = runner(batches[b % 4])
        torch.cuda.synchronize(); tb = time.perf_counter()
        stream.feed(acts)
        torch.cuda.synchronize(); tc = time.perf_counter()
        t_model += tb - ta; t_feed += tc - tb
    qi, qo = stream.input_qparams(args), stream.output_qparams(args)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    nbytes = sum(a.numel() * 4 for a in acts.values())
    return stream, total, t_model, t_feed, nbytes, torch.cuda.max_memory_allocated() / 2**30, len(qi) + len(qo)


run(False)
for hess, hs in ((False, 0), (True, 4), (True, 0)):
    stream, total, t_model, t_feed, nbytes, peak, nq = run(hess, hs)
    print(f"hessians={hess} ({'one grouped call per batch' if hs == 0 else f'per-tensor calls on {hs} side streams'}): 51 batches x {len(taps)} tapped tensors ({nbytes / 1e9:.2f} GB per batch, {51 * nbytes / 1e9:.0f} GB in all): "
          f"{total * 1e3:.1f} ms (model forward {t_model * 1e3:.1f}, stream.feed {t_feed * 1e3:.1f} = {51 * nbytes / t_feed / 1e12:.2f} TB/s of activations), "
          f"{nq} (scale, zp) pairs, peak device memory {peak:.2f} GiB", flush=True)
layers = {f"{i}.{n}": (getattr(model[i], n).weight.detach().t().contiguous(), f"{i}.{n}/in") for i in range(18) for n in ("qkv", "o", "up", "down")}
torch.cuda.synchronize(); t0 = time.perf_counter()
res = quantize_weights_gptq(layers, stream.hessians, "int4", "group", 128)
torch.cuda.synchronize()
print(f"GPTQ of the 72 weights from the streamed Hessians: {(time.perf_counter() - t0) * 1e3:.1f} ms")
