"""pytest configuration: registers the ``gpu`` marker and puts the product package
(``nir-gan_amd/``) and the oracle (``oracle/``, test infrastructure) on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nir-gan_amd")
ORACLE = os.path.join(ROOT, "oracle")
for p in (PKG, ORACLE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def host_cores() -> int:
    """CPU share of this container (affinity mask / cgroup quota), not the machine's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def pytest_configure(config):
    import torch
    torch.set_num_threads(host_cores())
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # NIRGAN_TEST_ORDER=reverse | shuffle:<seed>: the collected tests in another order (tests that hand raw device addresses to the library
    # must keep their tensors alive themselves; an order-dependent pass is a bug -- round 4 found one this way)
    order = os.environ.get("NIRGAN_TEST_ORDER", "")
    if order == "reverse":
        items.reverse()
    elif order.startswith("shuffle:"):
        import random
        random.Random(int(order.split(":", 1)[1])).shuffle(items)
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
