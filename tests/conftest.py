import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG_DIR = os.path.join(ROOT, "datafusion-bio-formats_amd")


def load_pkg():
    """The package directory name has a hyphen (it mirrors the upstream repo name), so it is
    loaded by path under the importable alias `datafusion_bio_formats_amd`."""
    name = "datafusion_bio_formats_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle():
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import bam_oracle
    return bam_oracle


def scratch_dir(need_bytes):
    """A directory with room for a generated input of `need_bytes` (tmpfs first: the files are read back at once);
    the test is skipped when neither /dev/shm nor /tmp can hold it."""
    for cand in ("/dev/shm", "/tmp"):
        try:
            if os.path.isdir(cand) and os.access(cand, os.W_OK):
                v = os.statvfs(cand)
                if v.f_bavail * v.f_frsize > need_bytes * 1.3:
                    return cand
        except OSError:
            pass
    pytest.skip("no scratch space for %d MB" % (need_bytes >> 20))


def full_size_blocks(full: int, fallback: int, bytes_per_block: int = 27000) -> int:
    """Member count of a large-input property test: BIOSCAN_TEST_LARGE_BLOCKS when set, else the BASELINE.json size
    (`full`) when /dev/shm or /tmp can hold the file with room to spare and the box has the cores to generate it in well
    under a minute, else `fallback`."""
    env = os.environ.get("BIOSCAN_TEST_LARGE_BLOCKS")
    if env:
        return int(env)
    if (os.cpu_count() or 1) < 16:
        return fallback
    for cand in ("/dev/shm", "/tmp"):
        try:
            if os.path.isdir(cand) and os.access(cand, os.W_OK):
                v = os.statvfs(cand)
                if v.f_bavail * v.f_frsize > full * bytes_per_block * 2:
                    return full
        except OSError:
            pass
    return fallback


def report_size(test, **sizes):
    """Large-input tests pick their size from the box (scratch space, cores); a UserWarning survives `pytest -q` and lands
    in the driver's record, so the record says which size a run actually covered."""
    import warnings
    warnings.warn("%s ran with %s" % (test, ", ".join("%s=%s" % kv for kv in sizes.items())), UserWarning, stacklevel=2)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    return load_oracle()


@pytest.fixture(scope="session")
def golden():
    return GOLDEN
