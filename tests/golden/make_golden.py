"""Generates the committed golden fixtures (run in the build container; needs torch CPU only).

    python tests/golden/make_golden.py

* mt19937_words.npz  -- words torch's CPU generator yields for the exact calls the reference's
  RandintEngine makes (pyg_lib/csrc/random/cpu/rand_engine.h:79-91): at::randint(INT64_MIN,
  INT64_MAX, [128]) once, then in-place Tensor.random_(INT64_MIN, INT64_MAX) refills.
* matmul_golden.npz  -- segment_matmul / grouped_matmul expectations computed the way the
  reference's tests do (test/ops/test_matmul.py:14-45,48-93: per-segment `inputs[a:b] @ other[i]`
  with torch on CPU), for BASELINE config C1 (fp32), the docstring example, and ragged bf16
  segments including empty ones.  bf16 tensors are stored as uint16 bit patterns.
"""
import os.path as osp

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
I64_MIN, I64_MAX = -2**63, 2**63 - 1


def rng_words():
    out = {}
    for seed in (0, 12345, 123456):
        torch.manual_seed(seed)
        buf = torch.randint(I64_MIN, I64_MAX, (128,), dtype=torch.long)
        blocks = [buf.clone()]
        for _ in range(2):
            buf.random_(I64_MIN, I64_MAX)
            blocks.append(buf.clone())
        out[f'seed_{seed}'] = torch.stack(blocks).numpy()
    np.savez_compressed(osp.join(HERE, 'mt19937_words.npz'), **out)


def bits(t):
    return t.view(torch.int16).numpy().view(np.uint16)


def seg_ref(x, ptr, w, bias=None):
    out = torch.empty(x.size(0), w.size(-1), dtype=x.dtype)
    for i in range(w.size(0)):
        out[ptr[i]:ptr[i + 1]] = x[ptr[i]:ptr[i + 1]] @ w[i]
    if bias is not None:
        for i in range(w.size(0)):
            out[ptr[i]:ptr[i + 1]] += bias[i]
    return out


def matmul():
    d = {}
    # C1: BASELINE.json configs[0]: seed 0, x=randn(1000,64), ptr=arange(0,1001,100)
    torch.manual_seed(0)
    x = torch.randn(1000, 64)
    ptr = torch.arange(0, 1001, 100)
    w = torch.randn(10, 64, 64)
    d['c1_x'], d['c1_ptr'], d['c1_w'], d['c1_out'] = x.numpy(), ptr.numpy(), w.numpy(), seg_ref(x, ptr, w).numpy()
    # docstring example / test_matmul.py shape (fp32) + bias
    torch.manual_seed(1)
    x = torch.randn(8, 16)
    ptr = torch.tensor([0, 5, 8])
    w = torch.randn(2, 16, 32)
    b = torch.randn(2, 32)
    d['doc_x'], d['doc_ptr'], d['doc_w'], d['doc_bias'] = x.numpy(), ptr.numpy(), w.numpy(), b.numpy()
    d['doc_out'] = seg_ref(x, ptr, w).numpy()
    d['doc_out_bias'] = seg_ref(x, ptr, w, b).numpy()
    # ragged bf16, K=M=128, with empty segments and a 1-row segment (MFMA path shapes)
    torch.manual_seed(2)
    sizes = [0, 37, 128, 1, 0, 300, 129, 5]
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n = int(ptr[-1])
    x = torch.randn(n, 128).bfloat16()
    w = (torch.randn(len(sizes), 128, 128) / 128**0.5).bfloat16()
    b = torch.randn(len(sizes), 128).bfloat16()
    d['bf_x'], d['bf_ptr'], d['bf_w'], d['bf_bias'] = bits(x), ptr.numpy(), bits(w), bits(b)
    d['bf_out'] = bits(seg_ref(x, ptr, w))
    d['bf_out_bias'] = bits(seg_ref(x, ptr, w, b))
    # same in fp32 (the 1e-5 parity case on MFMA shapes)
    xf, wf = x.float(), w.float()
    d['f32r_out'] = seg_ref(xf, ptr, wf).numpy()
    # grouped: variable K/M incl. a transposed (non-contiguous) `other` as in test_matmul.py:48-93
    torch.manual_seed(3)
    ins = [torch.randn(5, 16), torch.randn(6, 9), torch.randn(3, 32)]
    oth = [torch.randn(16, 48), torch.randn(9, 42), torch.randn(32, 64)]
    for i, (a, o) in enumerate(zip(ins, oth)):
        d[f'g_in{i}'], d[f'g_ot{i}'], d[f'g_out{i}'] = a.numpy(), o.numpy(), (a @ o).numpy()
    np.savez_compressed(osp.join(HERE, 'matmul_golden.npz'), **d)


if __name__ == '__main__':
    rng_words()
    matmul()
    print('wrote fixtures to', HERE)
