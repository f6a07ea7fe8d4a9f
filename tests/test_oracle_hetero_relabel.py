"""oracle.hetero_relabel_neighborhood against the reference's golden vectors
(test/csrc/sampler/test_dist_relabel.cpp:81-275) and the reference's own consistency property: relabelling the
globally-numbered output of hetero_neighbor_sample gives back its (row, col)."""
import numpy as np
import pytest

import oracle
from tests.golden import sampler_reference_vectors as G

ET = ('paper', 'to', 'paper')


@pytest.mark.parametrize('case', G.HETERO_RELABEL_CASES, ids=[c['name'] for c in G.HETERO_RELABEL_CASES])
def test_golden(case):
    batch = {'paper': np.array(case['batch'])} if 'batch' in case else None
    row, col = oracle.hetero_relabel_neighborhood(['paper'], [ET], {'paper': np.array(case['seed'])},
                                                  {'paper': np.array(case['sampled'])}, {ET: case['counts']},
                                                  batch_dict=batch, **case['kwargs'])
    assert row[ET].tolist() == case['row'] and col[ET].tolist() == case['col']


def test_golden_equals_hetero_sample_on_cycle_graph():
    # test_dist_relabel.cpp:123-137: same (row, col) as hetero_neighbor_sample with fan-out [2] from seeds {2, 3}
    rowptr, col = G.cycle_graph(6)
    out = oracle.hetero_neighbor_sample(['paper'], [ET], {ET: rowptr}, {ET: col}, {'paper': np.array([2, 3])}, {ET: [2]})
    assert out[0][ET].tolist() == [0, 0, 1, 1] and out[1][ET].tolist() == [2, 1, 0, 3]
