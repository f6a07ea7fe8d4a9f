import torch, sys
sys.path.insert(0, '/root/repo')
from pyg_lib_amd import ops
rows = [70000, 256]
n = sum(rows)
# x[r, c] = small integers identifying row and column: exactly representable
r = torch.arange(n, device='cuda')
x = ((r[:, None] % 251).float() + (torch.arange(256, device='cuda')[None, :] % 4).float() / 4).bfloat16()
w = torch.eye(256, device='cuda').bfloat16()[None].repeat(2, 1, 1)
w[1] = w[1] * 2
ptr = torch.tensor([0, 70000, n])
out = ops.segment_matmul(x, ptr, w)
torch.cuda.synchronize()
ref = torch.cat([x[:70000].float(), x[70000:].float() * 2])
bad = (out.float() != ref).any(1).nonzero().flatten()
print('bad rows', bad.numel(), bad[:6].tolist())
for b in bad[:6].tolist() + bad[-3:].tolist():
    print(b, 'got', out[b, :8].float().tolist(), 'want', ref[b, :8].tolist())
for b in bad[:3].tolist() + bad[-2:].tolist():
    d = (out[b].float() != ref[b]).nonzero().flatten()
    print(b, 'ncols', d.numel(), 'cols', d[:16].tolist(), '...', d[-4:].tolist())
    c = d[0].item()
    print('   got', out[b, c:c + 8].float().tolist(), 'want', ref[b, c:c + 8].tolist())
cols = (out[bad].float() != ref[bad]).any(0).nonzero().flatten()
print('bad cols overall', cols.numel(), cols[:40].tolist())
