"""Differential fuzzing of the HIP sampler against the oracle (tools/fuzz_sampler.py): random homogeneous /
heterogeneous graphs, fan-outs incl. 0, -1 and > 64, duplicate seeds, replace / disjoint / temporal modes --
all outputs and the generator state bit for bit, in both launch modes of the sampler."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_fully_queued_mode():
    from tools import fuzz_sampler
    assert fuzz_sampler.run(400, 101)


def test_fuzz_synchronising_mode(monkeypatch):
    from tools import fuzz_sampler
    monkeypatch.setenv('PYG_HIP_SAMPLER_SYNC_MODE', '1')
    assert fuzz_sampler.run(250, 102)
