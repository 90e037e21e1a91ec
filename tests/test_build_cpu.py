"""CPU: the incremental build never accepts a stale object (the command-line stamp used to be written
BEFORE hipcc ran, so an interrupted compile after a flag change left the old object next to the new stamp)."""
import os

from neo360_amd import build


def test_failed_compile_leaves_no_stamp_and_no_object(tmp_path, monkeypatch):
    csrc, out = tmp_path / "csrc", tmp_path / "lib"
    csrc.mkdir()
    (csrc / "good.hip").write_text('extern "C" int neo_good(void) { return 1; }\n')
    (csrc / "bad.hip").write_text('extern "C" int neo_bad(void) { return 1; }\n')
    (csrc / "x.h").write_text("// header\n")
    monkeypatch.setattr(build, "CSRC", str(csrc))
    monkeypatch.setattr(build, "OUT_DIR", str(out))
    monkeypatch.setattr(build, "LIB", str(out / "libtest.so"))
    monkeypatch.setattr(build, "EXTRA_FLAGS", {})
    build.build()
    assert os.path.exists(out / "good.o") and os.path.exists(out / "good.o.cmd") and os.path.exists(out / "libtest.so")
    # a flag change AND a source that no longer compiles: the old object and its stamp must both be gone afterwards
    monkeypatch.setattr(build, "EXTRA_FLAGS", {"bad.hip": ["-DNEO_TEST_FLAG=1"]})
    (csrc / "bad.hip").write_text("this does not compile\n")
    try:
        build.build()
        raise AssertionError("the broken translation unit compiled?")
    except RuntimeError as e:
        assert "bad.hip" in str(e)
    assert not os.path.exists(out / "bad.o") and not os.path.exists(out / "bad.o.cmd")
    assert os.path.exists(out / "good.o.cmd")                       # untouched
    # fixed source: rebuilt with the new flags, stamped after success, library relinked
    (csrc / "bad.hip").write_text('extern "C" int neo_bad(void) { return NEO_TEST_FLAG; }\n')
    lib_before = os.path.getmtime(out / "libtest.so")
    build.build()
    assert "-DNEO_TEST_FLAG=1" in (out / "bad.o.cmd").read_text()
    assert os.path.getmtime(out / "libtest.so") >= lib_before
    # an object newer than the library forces a relink even when nothing was recompiled
    os.utime(out / "libtest.so", (1, 1))
    build.build()
    assert os.path.getmtime(out / "libtest.so") > 1
