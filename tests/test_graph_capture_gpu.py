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


def test_atomic_free_fused_layer_replays_from_a_captured_graph():
    """The grouped R-GCN layer makes no host round trip either (records in the kernel argument, index validation
    deferred, the rows-longer-than-16 switch a device-side flag compared with an id baked into the launch): captured
    once, replayed on new features, weights AND new edge lists of the same sizes -- including a replay whose rows are
    longer than 16 edges although the captured call's were not."""
    from pyg_lib_amd import rgcn
    g = torch.Generator(device=DEV).manual_seed(11)
    n, F, E = 3000, 128, 20_000
    ets = [('a', 'r0', 'a'), ('a', 'r1', 'a')]
    x = torch.randn(n, F, device=DEV, generator=g).bfloat16()
    w = (torch.randn(2, F, F, device=DEV, generator=g) / F ** 0.5).bfloat16()
    rows = [torch.sort(torch.randint(0, n, (E,), device=DEV, generator=g)).values for _ in ets]    # ~7 edges per row
    cols = [torch.randint(0, n, (E,), device=DEV, generator=g) for _ in ets]
    out = torch.empty(n, F, device=DEV, dtype=torch.bfloat16)

    def call():
        return torch.ops.pyg.rgcn_fused(x, cols, rows, [0, 0], [0, 0], w, out, True)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            call()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    for trial in range(4):
        x.copy_(torch.randn(n, F, device=DEV, generator=g).bfloat16())
        w.copy_((torch.randn(2, F, F, device=DEV, generator=g) / F ** 0.5).bfloat16())
        hi = n if trial % 2 == 0 else 400          # odd trials: 50 edges per row -> the item-at-a-time walk
        for r, c in zip(rows, cols):
            r.copy_(torch.sort(torch.randint(0, hi, (E,), device=DEV, generator=g)).values)
            c.copy_(torch.randint(0, n, (E,), device=DEV, generator=g))
        graph.replay()
        torch.cuda.synchronize()
        got = out.clone()
        ref = torch.ops.pyg.rgcn_fused(x, cols, rows, [0, 0], [0, 0], w, torch.empty_like(out), True)
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), trial
    assert rgcn.pending_index_error() == 0
