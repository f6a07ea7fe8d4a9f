"""torch.classes.pyg.CUDAHashMap on the HIP device, modelled on the reference's
test/classes/test_hash_map.py (same expectations) plus larger random key sets."""
import numpy as np
import pytest
import torch

import pyg_lib_amd  # noqa: F401  (loads libpyg.so)

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)
INT_TO_DTYPE = {2: torch.short, 3: torch.int, 4: torch.long}


@pytest.mark.parametrize('dtype', [torch.short, torch.int, torch.long])
def test_hash_map(dtype):
    key = torch.tensor([0, 10, 30, 20], device=DEV, dtype=dtype)
    query = torch.tensor([30, 10, 20, 40], device=DEV, dtype=dtype)
    hash_map = torch.classes.pyg.CUDAHashMap(key, 0.5)
    assert hash_map.size() == 4
    assert INT_TO_DTYPE[hash_map.dtype()] == dtype
    assert hash_map.device() == DEV
    assert hash_map.keys().equal(key)
    assert hash_map.keys().dtype == dtype
    expected = torch.tensor([2, 1, 3, -1], device=DEV)
    assert hash_map.get(query).equal(expected)
    assert hash_map.get(query).dtype == torch.long


def test_large_random_keys_and_duplicates():
    rng = np.random.default_rng(0)
    key = torch.from_numpy(rng.permutation(5_000_000)[:1_000_000] * 7 - 3).to(DEV)
    m = torch.classes.pyg.CUDAHashMap(key, 0.5)
    assert m.size() == 1_000_000
    pos = torch.from_numpy(rng.integers(0, 1_000_000, 200_000)).to(DEV)
    assert m.get(key[pos]).equal(pos)
    absent = torch.from_numpy(rng.integers(0, 5_000_000, 200_000) * 7 - 2).to(DEV)  # never congruent to a key
    assert (m.get(absent) == -1).all()
    # duplicates map to their first position; keys() lists distinct keys in insertion order
    dup = torch.tensor([5, 9, 5, 7, 9, 5], device=DEV)
    d = torch.classes.pyg.CUDAHashMap(dup, 0.5)
    assert d.size() == 3
    assert d.get(torch.tensor([5, 7, 9, 1], device=DEV)).tolist() == [0, 3, 1, -1]
    assert d.keys().tolist() == [5, 9, 7]
    # negative keys, empty map
    neg = torch.tensor([-1, -(2 ** 40), 3], device=DEV)
    assert torch.classes.pyg.CUDAHashMap(neg, 0.5).get(neg).tolist() == [0, 1, 2]
    e = torch.classes.pyg.CUDAHashMap(torch.zeros(0, dtype=torch.long, device=DEV), 0.5)
    assert e.size() == 0 and e.get(torch.tensor([1], device=DEV)).tolist() == [-1]


class Foo(torch.nn.Module):
    def __init__(self, key):
        super().__init__()
        self.map = torch.classes.pyg.CUDAHashMap(key, 0.5)


def test_serialization(tmp_path):
    key = torch.tensor([0, 10, 30, 20], device=DEV)
    scripted = torch.jit.script(Foo(key))
    path = str(tmp_path / 'foo.pt')
    scripted.save(path)
    loaded = torch.jit.load(path)
    assert loaded.map.keys().equal(key)


def test_cpu_keys_are_rejected():
    with pytest.raises(RuntimeError):
        torch.classes.pyg.CUDAHashMap(torch.tensor([1, 2]), 0.5)
