"""segment_matmul over 4 Mi rows cut into B equal segments: how the kernels take relation changes (bf16 / fp32, F = 128 / 256)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyg_lib_amd import ops
from bench_legs import _kernel_ms
dev = 'cuda:0'
n = 1 << 22
which = sys.argv[1:] or ['bf16_128', 'bf16_256', 'f32_128']
cfg = {'bf16_128': (torch.bfloat16, 128, ('ring', 'ticket', 'contiguous')), 'bf16_256': (torch.bfloat16, 256, ('auto',)),
       'f32_128': (torch.float32, 128, ('ring', 'contiguous'))}
for name in which:
    dtype, F, scheds = cfg[name]
    x = torch.randn(n, F, device=dev).to(dtype)
    for B in (16, 256, 1024, 4096, 16384, 65536):
        w = (torch.randn(B, F, F, device=dev) / 11).to(dtype)
        ptr = torch.arange(0, n + 1, n // B)
        esz = x.element_size()
        ref = None
        for sched in scheds:
            ops.set_matmul_schedule(sched)
            ms = _kernel_ms(lambda: ops.segment_matmul(x, ptr, w), iters=6, warmup=2)
            out = ops.segment_matmul(x, ptr, w)
            same = ''
            if ref is None:
                ref = out
            elif dtype != torch.float32:
                same = 'bitwise-equal' if torch.equal(out.view(torch.int16), ref.view(torch.int16)) else 'DIFFERENT'
            print(name, B, 'segments of', n // B, 'rows:', ops.matmul_last_variant(), '%.3f ms' % ms,
                  '%.2f TB/s' % ((2 * n * F * esz + B * F * F * esz) / ms / 1e9), same)
        del w
    del x
ops.set_matmul_schedule('auto')
