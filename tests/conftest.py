import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "obj-gan_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _bounded_cpu_threads():
    """The CPU oracle runs on the host cores of whatever box executes the tests (8 here, 256 on the GPU
    box).  Hundreds of threads make the many small fp64 GEMMs of an Inception pass crawl (fork-join cost per
    call), so every test starts from a moderate pool; the full-batch oracle step raises it for itself."""
    import torch
    n = min(32, os.cpu_count() or 1)
    torch.set_num_threads(n)
    yield
    torch.set_num_threads(n)


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel_l2(a, b):
    import torch
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = torch.linalg.vector_norm(b)
    if den == 0:
        return float(torch.linalg.vector_norm(a - b))
    return float(torch.linalg.vector_norm(a - b) / den)


def note(key, value):
    """Append an observed parity number to gpurun_out/parity_numbers.txt (copied to profiles/ per round):
    the bounds asserted by the tests say what must hold, this file says what was measured."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_numbers.txt"), "a") as f:
            f.write("%-70s %s\n" % (key, value))
    except OSError:
        pass
