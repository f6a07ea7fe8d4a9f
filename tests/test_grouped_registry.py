"""The samplers remember their `row` outputs by identity (pyg_lib_amd.sampler.rows_are_grouped): host logic, no GPU."""
import gc

import torch


def test_registry_remembers_by_identity_and_forgets_with_the_tensor():
    from pyg_lib_amd import sampler
    t = torch.arange(7)
    assert not sampler.rows_are_grouped(t)
    sampler._mark_grouped(t)
    assert sampler.rows_are_grouped(t)
    assert not sampler.rows_are_grouped(t.clone()) and not sampler.rows_are_grouped(t[:3]) and not sampler.rows_are_grouped(t + 0)
    key = id(t)
    del t
    gc.collect()
    assert key not in sampler._grouped_rows
    sampler._mark_grouped(None)   # (an absent output)


def test_cpu_sampler_marks_rows_unless_csc():
    from pyg_lib_amd import sampler
    rowptr = torch.tensor([0, 2, 4, 6, 8])
    col = torch.tensor([1, 2, 2, 3, 3, 0, 0, 1])
    out = sampler.neighbor_sample(rowptr, col, torch.tensor([0, 1]), [2, 2])
    assert sampler.rows_are_grouped(out[0]) and not sampler.rows_are_grouped(out[1])
    assert bool((out[0][1:] >= out[0][:-1]).all())
    out = sampler.neighbor_sample(rowptr, col, torch.tensor([0, 1]), [2, 2], csc=True)
    assert not sampler.rows_are_grouped(out[0])
