import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    """The checker (oracle) is built on demand; the product library must already exist
    (python -c 'import __graft_entry__ as g; g.build()') -- tests never paper over a
    missing libcsadp.so."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "libcsa_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(os.path.join(ROOT, "csa_amd", "libcsadp.so")):
        import __graft_entry__
        __graft_entry__.build()
    yield


@pytest.fixture(autouse=True)
def csadp_switches(monkeypatch):
    """libcsadp.so reads its environment switches once per process (csadp_config.h).  Tests flip them with monkeypatch inside
    one process: the library re-reads them before every test (the previous test's variables are restored by then) and after
    every variable a test sets or deletes."""
    import csa_amd
    csa_amd.reload_config()
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def set_and_reload(name, value, *args, **kwargs):
        setenv(name, value, *args, **kwargs)
        csa_amd.reload_config()

    def del_and_reload(name, *args, **kwargs):
        delenv(name, *args, **kwargs)
        csa_amd.reload_config()

    monkeypatch.setenv = set_and_reload
    monkeypatch.delenv = del_and_reload
    yield
