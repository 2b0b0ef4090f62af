import os
import sys

# The CPU oracle's LAPACK calls (scipy / OpenBLAS) start one thread per core the machine reports -- 256 on a GPU box whose
# process group is limited to a fraction of them.  Such a burst exhausts the group's CPU quota and the kernel then throttles
# every thread of the process, the one that enqueues GPU work included, for tens of milliseconds: bounded hand-off waits on the
# GPU (cocons_fit_engine_state counts them) have run out that way right behind an oracle call.  Sixteen threads are what a
# one-GPU box is given; set before numpy loads its BLAS.
os.environ.setdefault("OPENBLAS_NUM_THREADS", str(min(16, os.cpu_count() or 16)))
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 16)))

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "shared_gpu: several processes use the GPU at once (engine time-outs allowed)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _engine_never_times_out(request):
    """Every fit handle a GPU test creates is checked at the end of the test: the resident diagonal-block engine must not
    have timed out (cocons_fit_engine_state: retries == 0) -- a silent fall-back to the plain schedule would otherwise
    pass every parity test.  Tests that SHARE the GPU between processes are exempt (marker `shared_gpu`)."""
    if request.node.get_closest_marker("gpu") is None or request.node.get_closest_marker("shared_gpu") is not None:
        yield
        return
    from cocons_amd import host
    made = []
    orig_init, orig_taper_init, orig_close = host.CoconsFit.__init__, host.CoconsTaperFit.__init__, host.CoconsFit.close

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        made.append(self)

    def taper_init(self, *a, **k):
        orig_taper_init(self, *a, **k)
        made.append(self)

    retries = []

    def close(self):
        if getattr(self, "_h", None):
            st = self.engine_state()
            if st["retries"]:
                # who gave up (0x600 = the start-up gate: engine not resident in time; 0x1tt / 0x2tt the engine waiting for
                # tile tt; 0x3tt / 0x5.. a panel kernel waiting for the engine; 0x900 the reductions) and on what problem
                retries.append("n=%d retries=%d last_abort=0x%x active=%d"
                               % (getattr(self, "n", -1), st["retries"], st["last_abort"], int(st["active"])))
        orig_close(self)

    host.CoconsFit.__init__, host.CoconsTaperFit.__init__, host.CoconsFit.close = init, taper_init, close
    try:
        yield
        for f in made:
            f.close()
    finally:
        host.CoconsFit.__init__, host.CoconsTaperFit.__init__, host.CoconsFit.close = orig_init, orig_taper_init, orig_close
    request.node.user_properties.append(("engine_timeouts", list(retries)))
    assert not retries, "engine hand-off time-outs in this test: %s" % "; ".join(retries)
