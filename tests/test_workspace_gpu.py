"""The per-stream scratch buffers of `hip.ops._workspace`: reuse on one stream, separation between streams, and results
that do not depend on how large the buffer has grown (some calls size their split-K slabs by the room they are given)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_large_workspaces_are_reused_per_stream_and_results_do_not_depend_on_their_size():
    from onnx_quantize_amd.hip import ops

    ops.release_workspaces()
    dev = torch.device("cuda", 0)
    a = ops._workspace(64 << 20, dev)
    b = ops._workspace(48 << 20, dev)
    assert a.data_ptr() == b.data_ptr() and b.numel() == 48 << 20          # same buffer, exactly the bytes asked for
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        c = ops._workspace(64 << 20, dev)
    assert c.data_ptr() != a.data_ptr()                                     # another stream never shares it
    small1, small2 = ops._workspace(4096, dev), ops._workspace(4096, dev)
    assert small1.data_ptr() != small2.data_ptr()                           # small requests are ordinary allocations

    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((4, 2048, 4096), generator=g, device=dev)
    ops.release_workspaces()
    h1 = torch.zeros((4096, 4096), device=dev)
    ops.hessian_accumulate(x, h1, 0)
    ops._workspace(3 << 30, dev)                                            # grow the buffer far beyond what the call needs
    h2 = torch.zeros((4096, 4096), device=dev)
    ops.hessian_accumulate(x, h2, 0)
    assert torch.equal(h1, h2)
    ops.release_workspaces()
    assert not ops._ARENA
    # a caller cycling through many streams does not pin one buffer per stream
    streams = [torch.cuda.Stream(device=dev) for _ in range(ops._ARENA_MAX_STREAMS + 3)]
    for s_ in streams:
        with torch.cuda.stream(s_):
            ops._workspace(40 << 20, dev)
    assert len(ops._ARENA) == ops._ARENA_MAX_STREAMS
    assert (0, streams[-1].cuda_stream) in ops._ARENA and (0, streams[0].cuda_stream) not in ops._ARENA
    ops.release_workspaces()
