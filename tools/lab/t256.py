import torch, sys
sys.path.insert(0, '/root/repo')
from pyg_lib_amd import ops
cases = [[70000, 90000], [70000, 1, 90000], [70000, 0, 90000], [69888, 90000], [1, 90000], [256, 90000], [70000, 256], [300, 300]]
which = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for ci, rows in enumerate(cases):
    if which >= 0 and ci != which:
        continue
    n = sum(rows)
    x = torch.randn(n, 256, device='cuda').bfloat16()
    w = (torch.randn(len(rows), 256, 256, device='cuda') / 16).bfloat16()
    ptr = torch.tensor([0] + torch.tensor(rows).cumsum(0).tolist())
    out = ops.segment_matmul(x, ptr, w)
    torch.cuda.synchronize()
    ref = torch.cat([x[ptr[i]:ptr[i + 1]].float() @ w[i].float() for i in range(len(rows))])
    err = (out.float() - ref).abs().max().item()
    bad = ((out.float() - ref).abs() > 0.1) | torch.isnan(out.float())
    br = bad.any(1).nonzero().flatten()
    print('bad rows', br.numel(), br[:8].tolist(), br[-4:].tolist())
    print(rows, ops.matmul_last_variant(), 'max err', err, flush=True)
