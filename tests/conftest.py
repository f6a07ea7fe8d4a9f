import os
import os.path as osp
import subprocess
import sys

import pytest

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_HEALTH = {}
_DIAGNOSED = []


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver with -m gpu)')


def _gpu_health():
    """Rounds 3 / 4: on one lease per round a full pass failed exactly the tests whose kernels accumulate through hardware
    floating-point atomics (missing, extra and NaN values), same binaries green everywhere else; cause unknown
    (profiles/NOTES_r4.md section 1).  Since round 5 the weight gradients have no atomics; what still has them (small
    scatter sums, the fused R-GCN layer) can run on compare-and-swap loops.  At session start this records, for the
    report header: the GPU's uuid, the stand-alone probe (tools/probe/atomic_probe: hipMalloc memory, own process) and
    the in-process self-test (pyg_hip_atomic_selftest: the caching allocator's memory and torch's stream -- 5 add flavours
    x 3 ways of clearing x 2 readbacks).  Information only: a failing test stays a failing test."""
    if _HEALTH:
        return _HEALTH
    _HEALTH['probe'] = 'not run'
    try:
        import torch
        if not torch.cuda.is_available():
            return _HEALTH
        try:
            uuid = subprocess.run(['rocminfo'], capture_output=True, text=True, timeout=60).stdout
            _HEALTH['uuid'] = ', '.join(sorted({ln.split()[-1] for ln in uuid.splitlines() if 'Uuid' in ln and 'GPU-' in ln}))
        except Exception:  # noqa: BLE001
            _HEALTH['uuid'] = '?'
        try:   # partition modes of the box (C4's 0.60 vs 0.695 is a property of the box: recorded next to every pass)
            smi = subprocess.run(['rocm-smi', '--showcomputepartition', '--showmemorypartition'], capture_output=True, text=True,
                                 timeout=60).stdout
            _HEALTH['partition'] = '; '.join(ln.split(':', 1)[1].strip() for ln in smi.splitlines() if 'Partition:' in ln)
        except Exception:  # noqa: BLE001
            _HEALTH['partition'] = '?'
        exe = osp.join(ROOT, 'tools', 'probe', 'atomic_probe')
        lines = []
        if osp.exists(exe):
            out = subprocess.run([exe, '30'], capture_output=True, text=True, timeout=300)
            lines = [ln for ln in out.stdout.splitlines()
                     if 'repetitions wrong' in ln and not ln.split(':')[1].strip().startswith('0 of')]
            _HEALTH['probe'] = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else f'no output (rc {out.returncode})'
        else:
            _HEALTH['probe'] = 'tools/probe/atomic_probe not built'
        _HEALTH['lost'] = lines
        try:
            from pyg_lib_amd import diagnostics
            bad, text = diagnostics.atomic_selftest(rounds=4)
            _HEALTH['selftest'] = text.strip().splitlines()
        except Exception as e:  # noqa: BLE001
            _HEALTH['selftest'] = [f'in-process self-test failed to run: {e!r}']
        os.makedirs(osp.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(osp.join(ROOT, 'gpurun_out', 'gpu_health.txt'), 'a') as f:
            f.write(f"gpu {_HEALTH.get('uuid')} ({_HEALTH.get('partition')}): {_HEALTH['probe']}\n" + ''.join(ln + '\n' for ln in lines) +
                    ''.join(ln + '\n' for ln in _HEALTH['selftest']))
    except Exception as e:  # noqa: BLE001 - diagnostics only
        _HEALTH['probe'] = f'probe failed to run: {e!r}'
    return _HEALTH


def _wants_gpu(config):
    return 'not gpu' not in (config.getoption('-m') or '')


def pytest_sessionstart(session):
    if _wants_gpu(session.config):
        _gpu_health()


def _health_lines(h):
    return ([f"GPU {h.get('uuid', '?')} ({h.get('partition', '?')})", f"stand-alone float-atomic probe: {h['probe']}"] +
            [f'    {ln}' for ln in h.get('lost', [])] + [f'in-process {ln}' if i == 0 else f'  {ln}'
                                                         for i, ln in enumerate(h.get('selftest', []))])


def pytest_report_header(config):
    if not _wants_gpu(config):
        return None
    h = _gpu_health()
    if h.get('probe') == 'not run':
        return None
    return _health_lines(h)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    """A failing GPU test is diagnosed where it failed, while the GPU is in whatever state made it fail: the in-process
    self-test again (8 rounds), what the last atomically accumulating launch was (accumulator address and
    hipPointerGetAttributes, who cleared it and how, stream, add flavour), and the SAME test body again with the
    remaining float atomics as hardware adds and as compare-and-swap loops -- the report says which of the two
    repeats failed.  (The weight gradients have no atomics: a failure there repeats in both modes.)"""
    outcome = yield
    rep = outcome.get_result()
    if rep.when != 'call' or not rep.failed or item.get_closest_marker('gpu') is None and 'gpu' not in item.nodeid:
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
        from pyg_lib_amd import diagnostics
        lines = [f'last accumulating launch: {diagnostics.last_accumulate_info()}']
        bad, text = diagnostics.atomic_selftest(rounds=8)
        lines += text.strip().splitlines()
        if len(_DIAGNOSED) < 8:   # (re-running bodies costs time: the first failures of a session only)
            for mode in ('hw', 'cas'):
                before = diagnostics.set_float_atomic_mode(mode)
                try:
                    item.runtest()
                    torch.cuda.synchronize()
                    lines.append(f'same test body again, float atomics = {mode}: PASSED')
                except (Exception, pytest.fail.Exception, pytest.skip.Exception) as e:  # noqa: BLE001
                    lines.append(f'same test body again, float atomics = {mode}: FAILED ({type(e).__name__}: {str(e)[:200]})')
                finally:
                    diagnostics.set_float_atomic_mode(before)
        _DIAGNOSED.append((item.nodeid, lines))
        rep.sections.append(('float-atomic diagnosis', '\n'.join(lines)))
    except Exception as e:  # noqa: BLE001 - diagnostics must never mask the failure
        rep.sections.append(('float-atomic diagnosis', f'could not run: {e!r}'))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    h = _HEALTH
    if not h or h.get('probe') == 'not run':
        return
    terminalreporter.section('GPU health')
    for ln in _health_lines(h):
        terminalreporter.write_line(ln)
    for nodeid, lines in _DIAGNOSED:
        terminalreporter.write_line(f'diagnosis of {nodeid}:')
        for ln in lines:
            terminalreporter.write_line('    ' + ln)
