import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the first HIP call of the process (kanzi_amd/__init__.py)
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    # PyTorch ships its own copy of the HIP runtime: in a process that uses both, torch has to be loaded before libkanzi_hip.so
    # (build() loads the library), or torch finds "No HIP GPUs" later.  Some GPU tests use torch for device buffers.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def ctx(built):
    # PyTorch ships its own copy of the HIP runtime: in a process that uses both, torch has to come first (a torch imported
    # after libkanzi_hip.so has initialised /opt/rocm's runtime finds "No HIP GPUs").  Some tests use torch for device buffers.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    import kanzi_amd as kz
    c = kz.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _switches_follow_the_environment(monkeypatch):
    """The library reads its KZ_* environment switches once, when a context is created.  Tests flip them with
    monkeypatch on live contexts (the session's `ctx`): every setenv / delenv of a test is followed by a re-read in the live
    contexts, and so is the restoration at the end of the test."""
    import kanzi_amd as kz
    set0, del0 = monkeypatch.setenv, monkeypatch.delenv

    def setenv(*a, **k):
        set0(*a, **k)
        kz.reload_switches()

    def delenv(*a, **k):
        del0(*a, **k)
        kz.reload_switches()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.undo()
    kz.reload_switches()
