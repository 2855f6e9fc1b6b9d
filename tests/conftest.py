import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout(seconds): per-test limit (pytest-timeout, when it is installed)")
    # the tests bind the built libraries: build them when a fresh checkout has none (hipcc cross-compiles
    # without a GPU; on the GPU box the .so files arrive with the snapshot)
    lib = os.path.join(ROOT, "trueconsense_amd", "lib", "libtcmi.so")
    orc = os.path.join(ROOT, "oracle", "_build", "libtcoracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()
