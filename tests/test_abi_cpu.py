"""CPU: the C-ABI library builds, loads and exports every symbol include/oai_hip.h declares."""
import ctypes
import os
import re

from oai_analysis_2_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "oai_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oai_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_everything():
    path = build.build_library(verbose=False)
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/oai_hip.h but not exported"
    # the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.load().oai_version() >= 100


def test_bad_arguments_are_reported_not_crashed():
    lib = _lib.load()
    rc = lib.oai_grid_sample3d(None, 1, 4, 4, 4, None, 4, 4, 4, None, None)
    assert rc != 0 and b"null" in lib.oai_last_error()
    rc = lib.oai_avgpool2_3d(None, 0, 0, 0, 0, None, None)
    assert rc != 0
