"""The C-ABI driven with raw device pointers through ctypes, no torch operator in between.

PyTorch only allocates the device buffers and names the stream here; every call goes
`ctypes -> libpyg_hip.so` exactly as a cgo / JNI / N-API host would bind it.  Results against the oracle.
"""
import ctypes
import os.path as osp

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
DEV = 'cuda:0'
PYG_F32, PYG_BF16, PYG_I64 = 0, 3, 8


@pytest.fixture(scope='module')
def lib():
    L = ctypes.CDLL(osp.join(ROOT, 'pyg_lib_amd', 'libpyg_hip.so'))
    c = ctypes
    L.pyg_hip_last_error.restype = c.c_char_p
    L.pyg_hip_matmul_workspace_size.restype = c.c_size_t
    L.pyg_hip_matmul_workspace_size.argtypes = [c.c_int64]
    L.pyg_hip_matmul_last_variant.restype = c.c_char_p
    L.pyg_hip_segment_matmul.restype = c.c_int
    L.pyg_hip_segment_matmul.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p,
                                         c.c_int64, c.c_int64, c.c_int64, c.c_int64, c.c_void_p, c.c_size_t, c.c_int, c.c_void_p]
    L.pyg_hip_index_sort_workspace_size.restype = c.c_size_t
    L.pyg_hip_index_sort_workspace_size.argtypes = [c.c_int, c.c_int64]
    L.pyg_hip_index_sort.restype = c.c_int
    L.pyg_hip_index_sort.argtypes = [c.c_int, c.c_void_p, c.c_int64, c.c_int64, c.c_int, c.c_void_p, c.c_void_p,
                                     c.c_void_p, c.c_size_t, c.c_void_p]
    return L


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize('ptr_on_device', [0, 1])
@pytest.mark.parametrize('K,M,dtype', [(128, 128, torch.bfloat16), (100, 72, torch.bfloat16), (64, 64, torch.float32),
                                       (100, 47, torch.float32)])
def test_segment_matmul_raw_pointers(lib, ptr_on_device, K, M, dtype):
    torch.manual_seed(K + M)
    sizes = [300, 0, 129, 1000, 37]
    ptr_host = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N, B = int(ptr_host[-1]), len(sizes)
    x = torch.randn(N, K).to(dtype)
    w = (torch.randn(B, K, M) / K ** 0.5).to(dtype)
    xd, wd = x.to(DEV), w.to(DEV)
    out = torch.empty(N, M, dtype=dtype, device=DEV)
    ws_bytes = lib.pyg_hip_matmul_workspace_size(B)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    ptr_dev = torch.from_numpy(ptr_host).to(DEV)
    stream = torch.cuda.current_stream().cuda_stream
    p = ptr_dev.data_ptr() if ptr_on_device else ptr_host.ctypes.data
    rc = lib.pyg_hip_segment_matmul(PYG_BF16 if dtype == torch.bfloat16 else PYG_F32, xd.data_ptr(), p, ptr_on_device,
                                    wd.data_ptr(), None, out.data_ptr(), N, K, M, B, ws.data_ptr(), ws_bytes, 0, stream)
    assert rc == 0, lib.pyg_hip_last_error()
    torch.cuda.synchronize()
    assert lib.pyg_hip_matmul_last_variant().startswith(b'mfma_')
    if dtype == torch.bfloat16:
        ref = oracle.segment_matmul(bits(x), ptr_host, bits(w), dtype=oracle.BF16)
        np.testing.assert_allclose(oracle.bf16_bits_to_f32(bits(out)), oracle.bf16_bits_to_f32(ref), rtol=2 ** -7, atol=1e-3)
    else:
        ref = oracle.segment_matmul(x.numpy(), ptr_host, w.numpy())
        assert np.linalg.norm(out.cpu().numpy() - ref) <= 1e-5 * np.linalg.norm(ref)
    # error convention: int status + thread-local message, nothing thrown across the boundary
    rc = lib.pyg_hip_segment_matmul(PYG_F32, xd.data_ptr(), p, ptr_on_device, wd.data_ptr(), None, out.data_ptr(), N, K, M,
                                    B, ws.data_ptr(), 16, 0, stream)
    assert rc != 0 and b'workspace' in lib.pyg_hip_last_error()
    # unknown mode bits are refused, not ignored
    rc = lib.pyg_hip_segment_matmul(PYG_F32, xd.data_ptr(), p, ptr_on_device, wd.data_ptr(), None, out.data_ptr(), N, K, M,
                                    B, ws.data_ptr(), ws_bytes, 0x40, stream)
    assert rc != 0 and b'flags' in lib.pyg_hip_last_error()


@pytest.mark.parametrize('n,max_value', [(0, 10), (1, 10), (32769, 5), (1_000_003, 2_449_029), (300_000, 2 ** 40)])
def test_index_sort_raw_pointers(lib, n, max_value):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, max_value, n, dtype=np.int64)
    kd = torch.from_numpy(keys).to(DEV)
    keys_out = torch.empty(n, dtype=torch.int64, device=DEV)
    idx_out = torch.empty(n, dtype=torch.int64, device=DEV)
    ws_bytes = lib.pyg_hip_index_sort_workspace_size(PYG_I64, n)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=DEV)
    rc = lib.pyg_hip_index_sort(PYG_I64, kd.data_ptr(), n, max_value, 1, keys_out.data_ptr(), idx_out.data_ptr(),
                                ws.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.pyg_hip_last_error()
    torch.cuda.synchronize()
    ref_v, ref_i = oracle.index_sort(keys)
    assert np.array_equal(keys_out.cpu().numpy(), ref_v) and np.array_equal(idx_out.cpu().numpy(), ref_i)


@pytest.mark.parametrize('dtype,K', [(torch.float32, 128), (torch.bfloat16, 64), (torch.int64, 3), (torch.float32, 1032)])
def test_segment_and_gather_csr_hub_rows_with_and_without_scratch(lib, dtype, K):
    """pyg_hip_segment_csr_ws / pyg_hip_gather_csr_ws: rows of more than 512 positions per lane are hubs.  No scratch: one
    workgroup per hub row; the advertised scratch: chunks of 2048 positions dealt to all workgroups; a third of it: longer
    chunks.  Integer-valued data: every variant must give the oracle's bits, arg included."""
    c = ctypes
    L = lib
    L.pyg_hip_csr_hub_workspace_size.restype = c.c_size_t
    L.pyg_hip_csr_hub_workspace_size.argtypes = [c.c_int, c.c_int, c.c_int64, c.c_int64, c.c_int64]
    L.pyg_hip_segment_csr_ws.restype = c.c_int
    L.pyg_hip_segment_csr_ws.argtypes = [c.c_int, c.c_int, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p, c.c_int,
                                         c.c_int64, c.c_int64, c.c_int64, c.c_int64, c.c_void_p, c.c_size_t, c.c_void_p]
    L.pyg_hip_gather_csr_ws.restype = c.c_int
    L.pyg_hip_gather_csr_ws.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_int64, c.c_int64,
                                        c.c_int64, c.c_void_p, c.c_size_t, c.c_void_p]
    L.pyg_hip_fill_reduce_identity.restype = c.c_int
    L.pyg_hip_fill_reduce_identity.argtypes = [c.c_int, c.c_int, c.c_void_p, c.c_int64, c.c_void_p]
    code = {torch.float32: PYG_F32, torch.bfloat16: PYG_BF16, torch.int64: PYG_I64}[dtype]
    rng = np.random.default_rng(K)
    lens = rng.integers(0, 10, 4000)
    lens[[5, 1700, 3999]] = [30_000, 600, 9000] if K < 1000 else [5000, 600, 2100]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    E, N = int(indptr[-1]), len(lens)
    src = torch.from_numpy(rng.integers(-6, 7, (E, K)).astype(np.float32)).to(dtype)
    bf16 = dtype == torch.bfloat16
    src_np = bits(src) if bf16 else src.numpy()
    sd, ip = src.to(DEV), torch.from_numpy(indptr).to(DEV)
    stream = torch.cuda.current_stream().cuda_stream
    assert L.pyg_hip_csr_hub_workspace_size(0, code, 1, 512, K) == 0
    for op in (0, 1, 2, 3):   # sum, mean, min, max
        if op == 1 and not dtype.is_floating_point:
            continue
        want, warg = oracle.segment_csr(op, src_np, indptr, None, oracle.BF16 if bf16 else None)
        full = L.pyg_hip_csr_hub_workspace_size(op, code, 1, E, K)
        assert full > 0
        for ws_bytes in (0, full, full // 3):
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=DEV)
            # fresh = 1: neither `out` nor `arg_out` is read or needs a pre-fill (ABI 8): hand over rubbish
            out = torch.full((N, K), 123, dtype=dtype, device=DEV)
            arg = torch.full((N, K), -7, dtype=torch.int64, device=DEV)
            rc = L.pyg_hip_segment_csr_ws(op, code, sd.data_ptr(), ip.data_ptr(), 0, out.data_ptr(), arg.data_ptr() if op >= 2 else None,
                                          1, 1, N, E, K, ws.data_ptr() if ws_bytes else None, ws_bytes, stream)
            assert rc == 0, L.pyg_hip_last_error()
            torch.cuda.synchronize()
            got = bits(out) if bf16 else out.cpu().numpy()
            if op == 1:
                np.testing.assert_allclose(oracle.bf16_bits_to_f32(got) if bf16 else got,
                                           oracle.bf16_bits_to_f32(want) if bf16 else want, rtol=2 ** -7 if bf16 else 1e-6)
            else:
                assert np.array_equal(got, want), (op, ws_bytes)
            if op >= 2:
                assert np.array_equal(arg.cpu().numpy(), warg), (op, ws_bytes)
    rows = torch.from_numpy(rng.integers(-50, 50, (N, K)).astype(np.float32)).to(dtype)
    want = torch.repeat_interleave(rows, torch.from_numpy(lens), dim=0)
    full = L.pyg_hip_csr_hub_workspace_size(4, code, 1, E, K)
    rd = rows.to(DEV)
    for ws_bytes in (0, full):
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=DEV)
        out = torch.zeros(E, K, dtype=dtype, device=DEV)
        rc = L.pyg_hip_gather_csr_ws(code, rd.data_ptr(), ip.data_ptr(), 0, out.data_ptr(), 1, N, E, K,
                                     ws.data_ptr() if ws_bytes else None, ws_bytes, stream)
        assert rc == 0, L.pyg_hip_last_error()
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), want), ws_bytes
