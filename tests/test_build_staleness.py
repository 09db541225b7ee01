"""mucon_amd/build.py decides what to recompile from the dependency files hipcc wrote (-MMD), not from a hand-kept header list:
touching a header that only one translation unit includes recompiles that unit alone, and a header nobody listed by hand
(csrc/decoder_mw.hpp, included by shead.hip) cannot leave a stale library behind."""
import os
import shutil
import time

import pytest

from mucon_amd import build as hb


def _built():
    return os.path.exists(hb.LIB) and all(os.path.exists(hb._obj(s) + ".d") for s, _ in hb.SOURCES)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_touching_one_header_recompiles_the_units_that_include_it_only():
    hb.build()                      # up to date first (a no-op when the tree was built already)
    assert hb.stale_units() == [] and not hb._stale()
    hdr = os.path.join(hb.CSRC, "decoder_mw.hpp")
    deps = {s: hb._deps_of(s) for s, _ in hb.SOURCES}
    users = [s for s, d in deps.items() if any(os.path.samefile(f, hdr) for f in d)]
    assert users == ["shead.hip"], users
    old = os.stat(hdr)
    try:
        future = time.time() + 5
        os.utime(hdr, (future, future))
        assert hb.stale_units() == ["shead.hip"] and hb._stale()
        before = {s: os.path.getmtime(hb._obj(s)) for s, _ in hb.SOURCES}
        os.utime(hdr, (old.st_atime, time.time()))      # "edited now"
        hb.build()
        assert hb.last_compiled == ["shead.hip"]
        after = {s: os.path.getmtime(hb._obj(s)) for s, _ in hb.SOURCES}
        assert [s for s in before if after[s] != before[s]] == ["shead.hip"]
        assert hb.stale_units() == [] and not hb._stale()
    finally:
        os.utime(hdr, (old.st_atime, old.st_mtime))


def test_every_included_header_is_a_dependency_of_its_unit():
    if not _built():
        pytest.skip("library not built here")
    import re

    for src, _ in hb.SOURCES:
        deps = {os.path.basename(f) for f in hb._deps_of(src)}
        text = open(os.path.join(hb.CSRC, src)).read()
        for inc in re.findall(r'#include "([^"]+)"', text):
            assert os.path.basename(inc) in deps, (src, inc)


def test_a_missing_dependency_file_means_rebuild(tmp_path, monkeypatch):
    if not _built():
        pytest.skip("library not built here")
    d = hb._obj("metrics.hip") + ".d"
    moved = str(tmp_path / "metrics.o.d")
    shutil.move(d, moved)
    try:
        assert "metrics.hip" in hb.stale_units() and hb._stale()
    finally:
        shutil.move(moved, d)
