"""HIP-graph capture of the matmul operators (torch.cuda.CUDAGraph): a training step that keeps `ptr` on the device
makes no host round trip -- the tile tables are planned by a kernel -- so the whole call, ticket counters included, can
be captured once and replayed on new data."""
import pytest
import torch

from pyg_lib_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('dtype,F,rows,variant', [
    (torch.bfloat16, 128, 400_000, 'mfma_bf16_k128_mc128_ticket'),
    (torch.bfloat16, 128, 40_000, 'mfma_bf16_k128_mc128_ring'),
    (torch.bfloat16, 256, 60_000, 'mfma_bf16_k256_regw'),
    (torch.float32, 128, 60_000, None),
])
def test_segment_matmul_replays_from_a_captured_graph(dtype, F, rows, variant):
    g = torch.Generator(device=DEV).manual_seed(3)
    B = 23
    cuts = torch.sort(torch.randint(0, rows, (B - 1,), device=DEV, generator=g)).values
    ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=DEV), cuts, torch.tensor([rows], device=DEV)])
    x = torch.randn(rows, F, device=DEV, generator=g).to(dtype)
    w = (torch.randn(B, F, F, device=DEV, generator=g) / F ** 0.5).to(dtype)
    bias = torch.randn(B, F, device=DEV, generator=g).to(dtype)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())  # the inputs are produced on the current stream
    with torch.cuda.stream(side):  # warm-up off the default stream: kernel attributes, allocator pools
        for _ in range(2):
            ops.segment_matmul(x, ptr, w, bias)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if variant is not None:
        assert ops.matmul_last_variant() == variant
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ops.segment_matmul(x, ptr, w, bias)
    for trial in range(3):
        x.copy_(torch.randn(rows, F, device=DEV, generator=g).to(dtype))
        w.copy_((torch.randn(B, F, F, device=DEV, generator=g) / F ** 0.5).to(dtype))
        graph.replay()
        torch.cuda.synchronize()
        ref = ops.segment_matmul(x, ptr, w, bias)
        view = torch.int32 if dtype == torch.float32 else torch.int16
        assert torch.equal(out.view(view), ref.view(view)), trial
