"""Reader for tests/golden/reduce_golden.npz (outputs of the REAL reference CPU kernels, recorded
by tests/golden/make_ref_golden.py through oracle/_ref)."""
import os.path as osp

import numpy as np

_D = None


def data():
    global _D
    if _D is None:
        _D = np.load(osp.join(osp.dirname(osp.abspath(__file__)), 'reduce_golden.npz'))
    return _D


def names(prefix):
    return [str(n) for n in data()['__cases__'] if str(n).startswith(prefix)]


def get(key):
    """(array, is_bf16) or (None, False)"""
    d = data()
    if key not in d:
        return None, False
    return d[key], (key + '__bf16') in d


def case(name):
    """name: 'scatter3_min' | 'coo2_sum' | 'gather0' | 'sort1_0'.  Returns a dict of numpy arrays."""
    d = data()
    kind = name.rstrip('0123456789_sumulmeaninx')  # not used for parsing below
    base, _, op = name.partition('_')
    out = {'name': name, 'op': op}
    if name.startswith('sort'):
        out['keys'] = d[base + '_keys']
        out['idx'] = d[base + '_idx']
        out['max'] = int(d[name + '_max']) if (name + '_max') in d else None
        return out
    if name.startswith('gather'):
        for k in ('src', 'index', 'res'):
            out[k], bf = get(f'{name}_{k}')
            if k == 'src':
                out['bf16'] = bf
        return out
    out['src'], out['bf16'] = get(base + '_src')
    out['index'], _ = get(base + '_index')
    out['out0'], _ = get(base + '_out0')
    ds, _ = get(base + '_dim_size')
    out['dim_size'] = None if ds is None else int(ds)
    dm, _ = get(base + '_dim')
    out['dim'] = None if dm is None else int(dm)
    out['res'], _ = get(name + '_res')
    out['arg'], _ = get(name + '_arg')
    return out
