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


def _disassemble_library(tmp_path):
    """ISA text of every gfx950 code object inside the built library (llvm-objdump --offloading writes the bundles next
    to its input, so it works on a copy)."""
    import shutil
    import subprocess
    lib = build.build()
    llvm = os.path.join(os.path.dirname(os.path.realpath(build._hipcc())), "..", "lib", "llvm", "bin")
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        llvm = "/opt/rocm/lib/llvm/bin"
    cp = tmp_path / "lib.so"
    shutil.copyfile(lib, cp)
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", str(cp)], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    cos = sorted(f for f in os.listdir(tmp_path) if f.endswith("amdgcn-amd-amdhsa--" + build.ARCH))
    assert len(cos) >= 10                      # one per translation unit with device code
    text = []
    for co in cos:
        text.append(subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--no-show-raw-insn", str(tmp_path / co)],
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout)
    return "\n".join(text)


def test_library_isa_has_no_inconsistent_split_conversions(tmp_path):
    """hi/lo splits must round hi ONCE.  hipcc folds (float)(_Float16)(a * b) into v_fma_mixlo_f16 (a single rounding from
    the exact product) in one use of the value and converts the rounded product in another: hi + lo then misses x by an ulp
    of fp16 (measured 1e-5 outliers in rgb).  split_tile.h:split makes its argument opaque; this keeps the instruction
    from coming back through some other route.  Also: the kernels that run fp16 MFMAs carry no packed fp32 op with an
    op_sel modifier (the forms that glitch beside v_mfma_f32_32x32x16_f16: profiles/r02_pk_f32_repro.log)."""
    isa = _disassemble_library(tmp_path)
    assert "v_mfma_f32_32x32x16_f16" in isa                                     # the disassembly is the real thing
    # Round 5: split_tile.h:split2 USES the instruction pair explicitly, operand by operand - lo = fp16(x - float(hi)) with hi
    # read out of the packed pair: `v_fma_mix{lo,hi}_f16 vD, vHI, -1.0, vX op_sel_hi:[1,0,0]`.  That form is the only one allowed;
    # the compiler's own folding of a product (two fp32 register sources, no -1.0) must still never appear.
    mixed = [l.strip() for l in isa.splitlines() if "v_fma_mixlo_f16" in l or "v_fma_mixhi_f16" in l]
    rogue = [l for l in mixed if ", -1.0, " not in l or "op_sel_hi:[1,0,0]" not in l]
    assert not rogue, rogue[:5]
    # walk kernel by kernel: a kernel with split-fp16 MFMAs must not contain op_sel'd packed fp32 arithmetic
    kernel, has_mfma, bad = None, {}, {}
    for line in isa.splitlines():
        if line.endswith(">:") and "<" in line:
            kernel = line.split("<", 1)[1][:-2]
        elif kernel:
            if "v_mfma_f32_32x32x16_f16" in line:
                has_mfma[kernel] = True
            elif "v_pk_" in line and "_f32" in line and "op_sel" in line:
                bad.setdefault(kernel, line.strip())
    offenders = {k: v for k, v in bad.items() if has_mfma.get(k)}
    assert not offenders, offenders
