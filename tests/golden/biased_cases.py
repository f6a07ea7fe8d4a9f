"""Loader of tests/golden/biased_golden.npz (written by make_biased_golden.py with the real torch ops)."""
import os.path as osp

import numpy as np

HERE = osp.dirname(osp.abspath(__file__))


def load():
    z = np.load(osp.join(HERE, 'biased_golden.npz'))
    cases = []
    for c in range(int(z['num_cases'][0])):
        pre = f'c{c}_'
        meta = [int(x) for x in z[pre + 'meta']]
        hetero, disjoint, f64, manual_seed, n_rel = meta[:5]
        replace = bool(meta[5]) if len(meta) > 5 else False
        if hetero:
            node_types = ['a', 'b']
            edge_types = [('a', 'x', 'b'), ('b', 'y', 'a'), ('a', 'z', 'a')]
        else:
            node_types = ['n']
            edge_types = [('n', 'to', 'n')]
        assert n_rel == len(edge_types)
        case = dict(id=c, hetero=bool(hetero), disjoint=bool(disjoint), f64=bool(f64), manual_seed=manual_seed, replace=replace,
                    node_types=node_types, edge_types=edge_types,
                    rowptr={e: z[pre + f'rowptr{j}'] for j, e in enumerate(edge_types)},
                    col={e: z[pre + f'col{j}'] for j, e in enumerate(edge_types)},
                    weight={e: z[pre + f'weight{j}'] for j, e in enumerate(edge_types)},
                    fan={e: z[pre + f'fan{j}'].tolist() for j, e in enumerate(edge_types)},
                    seed={t: z[pre + f'seed_{t}'] for t in node_types},
                    row_out={e: z[pre + f'row_out{j}'] for j, e in enumerate(edge_types)},
                    col_out={e: z[pre + f'col_out{j}'] for j, e in enumerate(edge_types)},
                    edge_out={e: z[pre + f'edge_out{j}'] for j, e in enumerate(edge_types)},
                    ehops={e: z[pre + f'ehops{j}'].tolist() for j, e in enumerate(edge_types)},
                    node={t: z[pre + f'node_{t}'] for t in node_types},
                    nhops={t: z[pre + f'nhops_{t}'].tolist() for t in node_types})
        cases.append(case)
    return cases
