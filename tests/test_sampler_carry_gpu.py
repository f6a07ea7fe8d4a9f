"""The random-word stream kept between sampler calls (include/pyg_hip.h, pyg_hip_sampler_rng_carry_stats; csrc/hip/
sampler_rng.hip, RngCarry).  A loader samples batch after batch on ONE generator; the library continues the words it left on
the device when the presented engine is the one it handed back -- and must produce, call after call, exactly what the
reference's engine produces when it keeps drawing 128-word blocks from that generator (random/cpu/rand_engine.h:79-91).

The oracle is run on the SAME torch CPU generator through its `fill` hook (every prefetch = the next 128 int64 of the
stream), so outputs and the generator state after every call are compared with the reference's protocol itself."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def torch_fill(arr):
    arr[:] = torch.randint(-2 ** 63, 2 ** 63 - 1, (128,)).numpy()


def random_csr(rng, n, mean_deg):
    deg = rng.poisson(mean_deg, n).astype(np.int64)
    return np.concatenate([[0], np.cumsum(deg)]).astype(np.int64), rng.integers(0, n, int(deg.sum()), dtype=np.int64)


def same(got, ref):
    for g, r in zip(got[:4], ref[:4]):
        assert np.array_equal(g.cpu().numpy().reshape(-1), np.asarray(r).reshape(-1))
    assert list(got[4]) == list(ref[4]) and list(got[5]) == list(ref[5])


def run_pair(calls, seed):
    """calls: list of (device call, oracle call) thunks; returns after checking every output and the generator."""
    torch.manual_seed(seed)
    got = [g() for g, _ in calls]
    torch.cuda.synchronize()
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    ref = [o() for _, o in calls]
    assert torch.equal(state, torch.get_rng_state())     # the generator ends where the reference's engine leaves it
    return got, ref


def test_consecutive_calls_on_one_generator_continue_the_stream():
    import oracle
    from pyg_lib_amd import sampler
    rng = np.random.default_rng(5)
    n = 60_000
    rp, cl = random_csr(rng, n, 20)
    rpd, cld = torch.from_numpy(rp).cuda(), torch.from_numpy(cl).cuda()
    seeds = [rng.permutation(n)[:512].astype(np.int64) for _ in range(30)]
    fan = [10, 6, 3]
    calls = [(lambda s=s: sampler.neighbor_sample(rpd, cld, torch.from_numpy(s).cuda(), fan),
              lambda s=s: oracle.neighbor_sample(rp, cl, s, fan, fill=torch_fill)) for s in seeds]
    a0, k0 = sampler.rng_carry_stats()
    got, ref = run_pair(calls, 1234)
    for g, r in zip(got, ref):
        same(g, r)
    a1, k1 = sampler.rng_carry_stats()
    assert sampler.last_mode() == 'fused'
    # the first call after manual_seed starts cold, the others adopt what their predecessor left -- until the kept buffer
    # (8 x two calls' worth) is used up and the stream restarts cold in a fresh one
    assert a1 - a0 >= 24 and 2 <= k1 - k0 <= 6


def test_generator_used_in_between_misses_and_stays_exact():
    import oracle
    from pyg_lib_amd import sampler
    rng = np.random.default_rng(6)
    n = 20_000
    rp, cl = random_csr(rng, n, 12)
    rpd, cld = torch.from_numpy(rp).cuda(), torch.from_numpy(cl).cuda()
    seeds = [rng.permutation(n)[:256].astype(np.int64) for _ in range(5)]
    fan = [8, 4]

    def use_generator():
        torch.rand(37)      # anything that draws from the CPU generator: the kept engine no longer matches
        return None

    calls = []
    for i, s in enumerate(seeds):
        calls.append((lambda s=s: sampler.neighbor_sample(rpd, cld, torch.from_numpy(s).cuda(), fan),
                      lambda s=s: oracle.neighbor_sample(rp, cl, s, fan, fill=torch_fill)))
        if i in (1, 2):
            calls.append((use_generator, use_generator))
    _, k0 = sampler.rng_carry_stats()
    got, ref = run_pair(calls, 99)
    for g, r in zip(got, ref):
        if g is not None:
            same(g, r)
    _, k1 = sampler.rng_carry_stats()
    assert k1 - k0 >= 3          # after manual_seed and after each foreign draw


@pytest.mark.parametrize('disjoint,replace', [(False, False), (True, False), (False, True)])
def test_many_small_calls_of_varying_size(disjoint, replace):
    """40 small calls of 1 ... 200 seeds (a round of the generator covers many of them), every one against the oracle on
    the same generator."""
    import oracle
    from pyg_lib_amd import sampler
    rng = np.random.default_rng(7)
    n = 3000
    rp, cl = random_csr(rng, n, 9)
    rpd, cld = torch.from_numpy(rp).cuda(), torch.from_numpy(cl).cuda()
    fan = [5, 3]
    calls = []
    for b in range(40):
        s = rng.permutation(n)[:int(rng.integers(1, 200))].astype(np.int64)
        calls.append((lambda s=s: sampler.neighbor_sample(rpd, cld, torch.from_numpy(s).cuda(), fan, disjoint=disjoint, replace=replace),
                      lambda s=s: oracle.neighbor_sample(rp, cl, s, fan, disjoint=disjoint, replace=replace, fill=torch_fill)))
    a0, k0 = sampler.rng_carry_stats()
    got, ref = run_pair(calls, 4321)
    for g, r in zip(got, ref):
        same(g, r)
    a1, k1 = sampler.rng_carry_stats()
    assert a1 - a0 >= 30 and k1 - k0 >= 1      # adopted but for the cold start (and wherever the buffer ended)


def test_growing_batches_outgrow_the_kept_stream():
    """Small batches, then much larger ones on the same generator: a call that adopted the kept words and needs more than
    the kept buffer holds moves them to a larger one (the adopted mark has no event to wait for: that wait was issued on a
    null handle -- `invalid resource handle` -- before tools/sampler_sweep.py found it).  Every call against the oracle on the
    same generator, incl. the generator state at the end."""
    import oracle
    from pyg_lib_amd import sampler
    rng = np.random.default_rng(9)
    n = 200_000
    rp, cl = random_csr(rng, n, 30)
    rpd, cld = torch.from_numpy(rp).cuda(), torch.from_numpy(cl).cuda()
    fan = [15, 10, 5]
    sizes = [64, 64, 64, 64, 1024, 1024, 8192, 8192, 8192, 128, 8192, 20000, 20000]
    calls = []
    for b in sizes:
        s = rng.permutation(n)[:b].astype(np.int64)
        calls.append((lambda s=s: sampler.neighbor_sample(rpd, cld, torch.from_numpy(s).cuda(), fan),
                      lambda s=s: oracle.neighbor_sample(rp, cl, s, fan, fill=torch_fill)))
    a0, _ = sampler.rng_carry_stats()
    got, ref = run_pair(calls, 777)
    for g, r in zip(got, ref):
        same(g, r)
    a1, _ = sampler.rng_carry_stats()
    assert a1 - a0 >= 4                       # the stream was adopted along the way (the largest batches run 'queued')
    assert sampler.last_mode() in ('fused', 'queued')


def test_hetero_and_homogeneous_calls_share_the_stream_and_release_drops_it():
    import oracle
    from pyg_lib_amd import sampler
    from tests.test_rgcn_gpu import MAG_TYPES, MAG_ETS, build_graph
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rng = np.random.default_rng(8)
    sizes = {'paper': 4000, 'author': 6000, 'institution': 90, 'field_of_study': 400}
    hrp, hcl = build_graph(rng, sizes, MAG_ETS, 10)
    hrpd, hcld = {e: dev(v) for e, v in hrp.items()}, {e: dev(v) for e, v in hcl.items()}
    hfan = {e: [5, 4] for e in MAG_ETS}
    n = 9000
    rp, cl = random_csr(rng, n, 10)
    rpd, cld = dev(rp), dev(cl)

    def hetero_dev(s):
        return sampler.hetero_neighbor_sample(hrpd, hcld, {'paper': dev(s)}, hfan)

    def hetero_ora(s):
        return oracle.hetero_neighbor_sample(MAG_TYPES, MAG_ETS, hrp, hcl, {'paper': s}, hfan, fill=torch_fill)

    def release():
        sampler.release_table_cache()
        return None

    calls = []
    for b in range(6):
        s = rng.permutation(4000)[:128].astype(np.int64)
        calls.append((lambda s=s: hetero_dev(s), lambda s=s: hetero_ora(s)))
        s2 = rng.permutation(n)[:300].astype(np.int64)
        calls.append((lambda s=s2: sampler.neighbor_sample(rpd, cld, dev(s), [6, 3]),
                      lambda s=s2: oracle.neighbor_sample(rp, cl, s, [6, 3], fill=torch_fill)))
        if b == 3:
            calls.append((release, lambda: None))
    got, ref = run_pair(calls, 2468)
    for g, r in zip(got, ref):
        if g is None:
            continue
        if isinstance(g[0], dict):
            for e in MAG_ETS:
                assert np.array_equal(g[0][e].cpu().numpy(), r[0][e]) and np.array_equal(g[1][e].cpu().numpy(), r[1][e])
                assert np.array_equal(g[3][e].cpu().numpy(), r[3][e])
            for t in MAG_TYPES:
                assert np.array_equal(g[2][t].cpu().numpy(), r[2][t])
        else:
            same(g, r)


def test_batched_calls_neither_use_nor_disturb_the_kept_stream():
    import oracle
    from pyg_lib_amd import sampler
    rng = np.random.default_rng(9)
    n = 20_000
    rp, cl = random_csr(rng, n, 12)
    rpd, cld = torch.from_numpy(rp).cuda(), torch.from_numpy(cl).cuda()
    seeds = [rng.permutation(n)[:256].astype(np.int64) for _ in range(4)]
    fan = [8, 4]
    torch.manual_seed(31)
    first = sampler.neighbor_sample(rpd, cld, torch.from_numpy(seeds[0]).cuda(), fan)
    a0, k0 = sampler.rng_carry_stats()
    outs = sampler.neighbor_sample_batched(rpd, cld, [torch.from_numpy(s).cuda() for s in seeds], fan, [5, 6, 7, 8])
    assert sampler.rng_carry_stats() == (a0, k0)      # per-batch engines: no lookup at all
    for s, sd, o in zip(seeds, [5, 6, 7, 8], outs):
        same(o, oracle.neighbor_sample(rp, cl, s, fan, rng_seed=sd))
    second = sampler.neighbor_sample(rpd, cld, torch.from_numpy(seeds[1]).cuda(), fan)   # continues `first`'s stream
    a1, _ = sampler.rng_carry_stats()
    assert a1 == a0 + 1
    torch.manual_seed(31)
    same(first, oracle.neighbor_sample(rp, cl, seeds[0], fan, fill=torch_fill))
    same(second, oracle.neighbor_sample(rp, cl, seeds[1], fan, fill=torch_fill))
