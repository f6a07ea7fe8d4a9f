"""Golden vectors held by the reference's own sampler tests (data only).

Source of the numbers: pyg-lib v0.9.0 ``test/csrc/sampler/test_neighbor.cpp`` (test name and
line range given per case) on the 6-node cycle graph of ``test/csrc/graph.h:5-13``.  Cases that
need ``edge_weight`` (BiasedNeighborTest :300-328, HeteroBiasedNeighborTest :330-377) consume
at::multinomial / Tensor.uniform_ and are outside the restated path.
"""
import numpy as np


def cycle_graph(n=6):
    """test/csrc/graph.h:5-13: rowptr = [0,2,..,2n], col = [(i-1)%n, (i+1)%n]."""
    rowptr = np.arange(0, 2 * n + 1, 2, dtype=np.int64)
    col = np.stack([(np.arange(-1, n - 1) % n), (np.arange(1, n + 1) % n)], axis=1).reshape(-1).astype(np.int64)
    return rowptr, col


ROWPTR, COL = cycle_graph(6)
# NodeLevelTemporalNeighborTest sorts each neighbourhood by node id (:156):
COL_SORTED = np.sort(COL.reshape(-1, 2), axis=1).reshape(-1)

CASES = [
    dict(  # BasicNeighborTest :8-31
        name='basic', rowptr=ROWPTR, col=COL, seed=[2, 3], num_neighbors=[-1, -1], kwargs={},
        row=[0, 0, 1, 1, 2, 2, 3, 3], col_out=[2, 1, 0, 3, 4, 0, 1, 5], node=[2, 3, 1, 4, 0, 5],
        edge=[4, 5, 6, 7, 2, 3, 8, 9], nodes_per_hop=[2, 2, 2], edges_per_hop=[4, 4]),
    dict(  # ZeroNeighborTest :33-57
        name='zero_degree', rowptr=np.zeros(6, dtype=np.int64), col=np.zeros(0, dtype=np.int64),
        seed=[0, 1, 2, 3, 4], num_neighbors=[-1, -1], kwargs={},
        row=[], col_out=[], node=[0, 1, 2, 3, 4], edge=[], nodes_per_hop=[5, 0, 0], edges_per_hop=[0, 0]),
    dict(  # WithoutReplacementNeighborTest :59-85 (at::manual_seed(123456))
        name='without_replacement_seeded', rowptr=ROWPTR, col=COL, seed=[2, 3], num_neighbors=[1, 1],
        kwargs=dict(replace=False), manual_seed=123456,
        row=[0, 1, 2, 3], col_out=[2, 3, 0, 4], node=[2, 3, 1, 4, 5], edge=[4, 7, 3, 9]),
    dict(  # WithReplacementNeighborTest :87-113 (at::manual_seed(123456))
        name='with_replacement_seeded', rowptr=ROWPTR, col=COL, seed=[2, 3], num_neighbors=[1, 1],
        kwargs=dict(replace=True), manual_seed=123456,
        row=[0, 1, 2, 3], col_out=[2, 3, 0, 4], node=[2, 3, 1, 4, 5], edge=[4, 7, 3, 9]),
    dict(  # DisjointNeighborTest :115-144
        name='disjoint', rowptr=ROWPTR, col=COL, seed=[2, 3], num_neighbors=[2, 2],
        kwargs=dict(replace=False, directed=True, disjoint=True),
        row=[0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5], col_out=[2, 3, 4, 5, 6, 0, 0, 7, 8, 1, 1, 9],
        node=[[0, 2], [1, 3], [0, 1], [0, 3], [1, 2], [1, 4], [0, 0], [0, 4], [1, 1], [1, 5]],
        edge=[4, 5, 6, 7, 2, 3, 6, 7, 4, 5, 8, 9]),
    dict(  # NodeLevelTemporalNeighborTest out1 :158-180
        name='node_temporal', rowptr=ROWPTR, col=COL_SORTED, seed=[2, 3], num_neighbors=[2, 2],
        kwargs=dict(node_time=np.arange(6, dtype=np.int64), replace=False, directed=True, disjoint=True),
        row=[0, 1, 2, 2, 3, 3], col_out=[2, 3, 4, 0, 5, 1],
        node=[[0, 2], [1, 3], [0, 1], [1, 2], [0, 0], [1, 1]], edge=[4, 6, 2, 3, 4, 5]),
    dict(  # NodeLevelTemporalNeighborTest out2 :182-201 ("last" strategy, same expectation)
        name='node_temporal_last', rowptr=ROWPTR, col=COL_SORTED, seed=[2, 3], num_neighbors=[1, 2],
        kwargs=dict(node_time=np.arange(6, dtype=np.int64), replace=False, directed=True, disjoint=True,
                    temporal_strategy='last'),
        row=[0, 1, 2, 2, 3, 3], col_out=[2, 3, 4, 0, 5, 1],
        node=[[0, 2], [1, 3], [0, 1], [1, 2], [0, 0], [1, 1]], edge=[4, 6, 2, 3, 4, 5]),
    dict(  # EdgeLevelTemporalNeighborTest out :214-237
        name='edge_temporal', rowptr=ROWPTR, col=COL, seed=[2, 3], num_neighbors=[2, 2],
        kwargs=dict(edge_time=np.arange(12, dtype=np.int64), seed_time=np.array([5, 6], dtype=np.int64),
                    replace=False, directed=True, disjoint=True),
        row=[0, 0, 1, 2, 2, 4, 4], col_out=[2, 3, 4, 5, 0, 6, 1],
        node=[[0, 2], [1, 3], [0, 1], [0, 3], [1, 2], [0, 0], [1, 1]], edge=[4, 5, 6, 2, 3, 4, 5]),
    dict(  # EdgeLevelTemporalNeighborTest out2 :239-256
        name='edge_temporal_none', rowptr=ROWPTR, col=COL, seed=[2, 3], num_neighbors=[1, 1],
        kwargs=dict(edge_time=np.arange(12, dtype=np.int64), seed_time=np.array([-1, -1], dtype=np.int64),
                    replace=True, directed=True, disjoint=True),
        row=[], col_out=[], node=[[0, 2], [1, 3]], edge=[]),
]

# HeteroNeighborTest :259-298 (one node type "paper", one relation, fanout [2, 2]): identical
# expectation to BasicNeighborTest because 2 >= degree.
HETERO_CASE = dict(
    node_types=['paper'], edge_types=[('paper', 'to', 'paper')], seed=[2, 3], num_neighbors=[2, 2],
    row=[0, 0, 1, 1, 2, 2, 3, 3], col_out=[2, 1, 0, 3, 4, 0, 1, 5], node=[2, 3, 1, 4, 0, 5],
    edge=[4, 5, 6, 7, 2, 3, 8, 9], nodes_per_hop=[2, 2, 2], edges_per_hop=[4, 4])


# test/csrc/sampler/test_dist_neighbor.cpp: (node_id, edge_id, cumsum_neighbors_per_node) of ONE hop
DIST_CASES = [
    dict(name='dist_basic', col=COL, seed=[2, 3], num_neighbors=-1, kwargs={},  # :8-27
         node=[2, 3, 1, 3, 2, 4], edge=[4, 5, 6, 7], cumsum=[2, 4, 6]),
    dict(name='dist_without_replacement_seeded', col=COL, seed=[2, 3], num_neighbors=1, kwargs={},  # :29-49
         manual_seed=123456, node=[2, 3, 1, 4], edge=[4, 7], cumsum=[2, 3, 4]),
    dict(name='dist_with_replacement_seeded', col=COL, seed=[2, 3], num_neighbors=2,  # :51-77
         kwargs=dict(replace=True), manual_seed=123456, node=[2, 3, 1, 3, 4, 4], edge=[4, 5, 7, 7],
         cumsum=[2, 4, 6]),
    dict(name='dist_disjoint', col=COL, seed=[2, 3], num_neighbors=2,  # :79-107
         kwargs=dict(replace=False, directed=True, disjoint=True),
         node=[[0, 2], [1, 3], [0, 1], [0, 3], [1, 2], [1, 4]], edge=[4, 5, 6, 7], cumsum=[2, 4, 6]),
    dict(name='dist_temporal', col=COL_SORTED, seed=[2, 3], num_neighbors=2,  # :109-144
         kwargs=dict(node_time=np.arange(6, dtype=np.int64), replace=False, directed=True, disjoint=True,
                     temporal_strategy='uniform'),
         node=[[0, 2], [1, 3], [0, 1], [1, 2]], edge=[4, 6], cumsum=[2, 3, 4]),
]


# test/csrc/sampler/test_dist_merge_outputs.cpp (:7-49, :51-93, :95-132): merge_sampler_outputs
MERGE_CASES = [
    dict(name='merge_basic', node_ids=[[2, 7, 8], [0, 1, 4, 5, 6], [3, 9, 10]], edge_ids=[[17, 18], [14, 15, 16], [19, 20]],
         cumsum=[[1, 3], [2, 4, 5], [1, 3]], partition_ids=[1, 1, 0, 2], partition_orders=[0, 1, 0, 0], num_partitions=3,
         num_neighbors=2, batch=None, disjoint=False,
         nodes=[4, 5, 6, 7, 8, 9, 10], edges=[14, 15, 16, 17, 18, 19, 20], out_batch=None, counts=[2, 1, 2, 2]),
    dict(name='merge_all_neighbors', node_ids=[[2, 7, 8], [0, 1, 4, 5, 6], [3, 9, 10, 11]],
         edge_ids=[[17, 18], [14, 15, 16], [19, 20, 21]], cumsum=[[1, 3], [2, 4, 5], [1, 4]],
         partition_ids=[1, 1, 0, 2], partition_orders=[0, 1, 0, 0], num_partitions=3, num_neighbors=-1, batch=None,
         disjoint=False, nodes=[4, 5, 6, 7, 8, 9, 10, 11], edges=[14, 15, 16, 17, 18, 19, 20, 21], out_batch=None,
         counts=[2, 1, 2, 3]),
    dict(name='merge_disjoint', node_ids=[[2, 7, 8], [0, 1, 4, 5, 6], [3, 9, 10]], edge_ids=[[17, 18], [14, 15, 16], [19, 20]],
         cumsum=[[1, 3], [2, 4, 5], [1, 3]], partition_ids=[1, 1, 0, 2], partition_orders=[0, 1, 0, 0], num_partitions=3,
         num_neighbors=2, batch=[0, 1, 2, 3], disjoint=True,
         nodes=[4, 5, 6, 7, 8, 9, 10], edges=[14, 15, 16, 17, 18, 19, 20], out_batch=[0, 0, 1, 2, 2, 3, 3],
         counts=[2, 1, 2, 2]),
]

# test/csrc/sampler/test_dist_relabel.cpp (:9-37, :39-79): relabel_neighborhood
RELABEL_CASES = [
    dict(name='relabel_basic', seed=[2, 3], sampled=[1, 3, 2, 4], counts=[2, 2], num_nodes=6, batch=None, disjoint=False,
         row=[0, 0, 1, 1], col=[2, 1, 0, 3]),
    dict(name='relabel_disjoint', seed=[2, 3], sampled=[1, 3, 2, 4], counts=[2, 2], num_nodes=6, batch=[0, 0, 1, 1],
         disjoint=True, row=[0, 0, 1, 1], col=[2, 3, 4, 5]),
]


# test/csrc/sampler/test_dist_relabel.cpp:81-275 -- hetero_relabel_neighborhood on the cycle graph, one node
# type "paper", one edge type; num_sampled_neighbors_per_node = 2 layers x [2]
HETERO_RELABEL_CASES = [
    dict(name='hetero_relabel', kwargs={}, seed=[2, 3], sampled=[1, 3, 2, 4], counts=[[2], [2]],
         row=[0, 0, 1, 1], col=[2, 1, 0, 3]),                                    # :81-138
    dict(name='hetero_relabel_csc', kwargs=dict(csc=True), seed=[2, 3], sampled=[1, 3, 2, 4], counts=[[2], [2]],
         row=[2, 1, 0, 3], col=[0, 0, 1, 1]),                                    # :140-204
    dict(name='hetero_relabel_disjoint', kwargs=dict(disjoint=True), seed=[2, 3], sampled=[1, 3, 2, 4],
         counts=[[2], [2]], batch=[0, 0, 1, 1], row=[0, 0, 1, 1], col=[2, 3, 4, 5]),  # :206-275
]
