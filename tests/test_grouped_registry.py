"""The samplers remember their expanded-node outputs (`row` for csc=False, `col` for csc=True) by identity and version
counter (pyg_lib_amd.sampler.rows_are_grouped): host logic, no GPU."""
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


def test_cpu_sampler_marks_the_expanded_node_vector():
    from pyg_lib_amd import sampler
    rowptr = torch.tensor([0, 2, 4, 6, 8])
    col = torch.tensor([1, 2, 2, 3, 3, 0, 0, 1])
    out = sampler.neighbor_sample(rowptr, col, torch.tensor([0, 1]), [2, 2])
    assert sampler.rows_are_grouped(out[0]) and not sampler.rows_are_grouped(out[1])
    assert bool((out[0][1:] >= out[0][:-1]).all())
    # csc=True: get_sampled_edges swaps the two vectors (neighbor_kernel.cpp:155-159) -- `col` holds the expanded nodes
    out = sampler.neighbor_sample(rowptr, col, torch.tensor([0, 1]), [2, 2], csc=True)
    assert not sampler.rows_are_grouped(out[0]) and sampler.rows_are_grouped(out[1])
    assert bool((out[1][1:] >= out[1][:-1]).all())


def test_hetero_cpu_sampler_marks_col_for_csc():
    from pyg_lib_amd import sampler
    et = ('a', 'r', 'b')
    # csc=True: a CSC over the 3 `b` nodes holding `a` ids
    colptr = torch.tensor([0, 2, 4, 5])
    row = torch.tensor([0, 4, 1, 2, 3])
    out = sampler.hetero_neighbor_sample({et: colptr}, {et: row}, {'b': torch.tensor([2, 0])}, {et: [2]}, csc=True)
    assert sampler.rows_are_grouped(out[1][et]) and not sampler.rows_are_grouped(out[0][et])
    assert out[1][et].tolist() == [0, 1, 1] and sorted(out[2]['a'][out[0][et]].tolist()) == [0, 3, 4]


def test_a_tensor_written_to_in_place_no_longer_counts():
    from pyg_lib_amd import sampler
    t = torch.arange(9)
    sampler._mark_grouped(t)
    assert sampler.rows_are_grouped(t)
    t.sort(descending=True)              # out of place: untouched
    assert sampler.rows_are_grouped(t)
    t[2] = 0                             # in place: the version counter moved
    assert not sampler.rows_are_grouped(t)
    for mutate in (lambda u: u.copy_(torch.zeros(9, dtype=torch.long)), lambda u: u.add_(1),
                   lambda u: torch.sort(u, descending=True, out=(u, torch.empty(9, dtype=torch.long))),
                   lambda u: u.__setitem__(u > 3, 0)):
        u = torch.arange(9)
        sampler._mark_grouped(u)
        mutate(u)
        assert not sampler.rows_are_grouped(u)
