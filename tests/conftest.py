import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest

# Pins that take minutes each on the GPU and add little beyond a faster sibling in the default suite (which already runs config C4's whole
# million-sequence NJ phase, C5's 20 000-protein pipeline and the 100 000-sequence threaded pipeline): VFT_TEST_HEAVY=1 runs them too
# (profiles/r06_pytest_heavy.txt is this round's run of them).
heavy = pytest.mark.skipif(not os.environ.get("VFT_TEST_HEAVY"), reason="minutes-long extra pin: set VFT_TEST_HEAVY=1")


def free_port():
    """a TCP port nobody is listening on right now, for a torch.distributed.run rendezvous on 127.0.0.1 (a fixed port shared by the cases of
    a parametrised test can still be held by the previous case's store for a moment: one spurious failure in a dozen runs)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
