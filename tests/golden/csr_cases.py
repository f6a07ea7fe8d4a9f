"""Reader for tests/golden/csr_golden.npz (outputs of the REAL reference CPU kernels for
segment_*_csr / gather_csr / softmax_csr, recorded by tests/golden/make_csr_golden.py through
oracle/_ref)."""
import os.path as osp

import numpy as np

_D = None


def data():
    global _D
    if _D is None:
        _D = np.load(osp.join(osp.dirname(osp.abspath(__file__)), 'csr_golden.npz'))
    return _D


def names(kind):
    """kind: 'reduce' (csr<i>_{sum,mean,min,max}), 'gather' (csr<i>_gather), 'softmax'."""
    out = []
    for n in data()['__cases__']:
        n = str(n)
        if kind == 'softmax' and n.startswith('softmax'):
            out.append(n)
        elif kind == 'gather' and n.endswith('_gather'):
            out.append(n)
        elif kind == 'reduce' and n.startswith('csr') and not n.endswith('_gather'):
            out.append(n)
    return out


def _get(key):
    d = data()
    if key not in d:
        return None, False
    return d[key], (key + '__bf16') in d


def case(name):
    d = data()
    out = {'name': name}
    if name.startswith('softmax'):
        for k in ('src', 'ptr', 'res', 'out_grad', 'in_grad'):
            out[k] = d[f'{name}_{k}']
        out['dim'] = int(d[name + '_dim'])
        return out
    base, _, op = name.partition('_')
    out['op'] = op
    out['indptr'] = d[base + '_indptr']
    if op == 'gather':
        out['src'], out['bf16'] = _get(name + '_src')
        out['out0'], _ = _get(name + '_out0')
        out['res'], _ = _get(name + '_res')
        return out
    out['src'], out['bf16'] = _get(base + '_src')
    out['out0'], _ = _get(base + '_out0')
    out['res'], _ = _get(name + '_res')
    out['arg'], _ = _get(name + '_arg')
    return out
