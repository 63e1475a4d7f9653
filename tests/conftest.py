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


def poison_gpu_allocator(device, total_bytes=6 << 30):
    """Fill the caching allocator's free lists with NaN-filled blocks of many sizes and release them: a kernel that
    leaves part of a `torch.empty` output (or of a workspace) unwritten then produces NaNs instead of silently
    inheriting plausible stale values.  The GPU tests run this way (outputs are no longer zero-filled)."""
    import torch
    blocks, used = [], 0
    size = 256 << 20
    while size >= 512 and used < total_bytes:
        n = max(1, min(64, (total_bytes // 12) // size))
        for _ in range(n):
            blocks.append(torch.full((size // 4,), float("nan"), dtype=torch.float32, device=device))
            used += size
        size //= 4
    torch.cuda.synchronize()
    del blocks


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _poisoned_gpu_memory(request):
    """Before every GPU test: NaN-fill the allocator's free blocks (see poison_gpu_allocator)."""
    if request.node.get_closest_marker("gpu") is None or os.environ.get("OG_NO_POISON") == "1":
        yield
        return
    import torch
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
        poison_gpu_allocator(torch.device("cuda:0"), total_bytes=3 << 30)
    yield


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
