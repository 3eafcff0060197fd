import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the developer switches and test hooks of libnagp.so (NAGP_NO_PIPELINE, NAGP_TEST_FAKE_DEVICES, ...) are only read with this set: the tests
# use them to pin one schedule / kernel against another; without it (bench.py, a MATLAB host) the library ignores every one of them
os.environ.setdefault('NAGP_DEVELOPER', '1')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def nagp_lib():
    """Build (if needed) and load libnagp.so; the GPU tests call the product ONLY through it."""
    import nagp
    nagp.build()
    return nagp.lib()


@pytest.fixture(scope='session')
def full_length_refs():
    """The sequential CPU algorithm (compiled oracle, structured form) on the seven full-length bench workloads -- the two audio files
    BASELINE names among them -- one thread each, started on first use and running BESIDE the rest of the GPU suite
    (tools/full_length_parity.py: about 35 / 90 / 200 / 115 s of one core); the tests at the end of tests/test_gpu_parity.py join them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('full_length_parity', os.path.join(ROOT, 'tools', 'full_length_parity.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    legs = mod.CpuLegs(['cfg4', 'cfg4audio', 'cfg5seg', 'cfg2', 'cfg2audio', 'cfg3', 'cfg3sqrt'])          # longest first
    legs.mod = mod
    return legs
