"""Builds libneo360_hip.so in-tree with hipcc for gfx950 (no GPU needed: hipcc
cross-compiles).  Incremental: a translation unit is recompiled only when it or
a header is newer than its object."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libneo360_hip.so")
ARCH = "gfx950"
# -ffp-contract=off: the sampling / encoding kernels mirror the reference's
# separate fp32 multiply and add (e.g. o + t*d) so encodings see identical
# arguments; dot products inside the MLP go through MFMA and are unaffected.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]

# Per-file additions: no packed-fp32 VALU ops (v_pk_mul/add/fma_f32) in any file that runs fp32 VALU work next to a
# v_mfma_f32_32x32x16_f16 stream.  Measured on MI355X (profiles/r01_tp_h_race_bisect.log, stand-alone reproducer
# tools/pk_f32_repro.hip + profiles/r02_pk_f32_repro.log): a packed fp32 op that uses op_sel to broadcast one half of
# a source pair, executed while another wave of the same SIMD runs that MFMA, occasionally returns wrong values in
# lanes 48-63; no wait state fixes it.  The compiler picks those forms on its own (21 of them in mlp_vanilla_h.hip),
# so the feature is switched off per file; the cost is nil (profiles/r02_power_envelope.log, "nopk").
# tests/test_gpu_repeatable.py checks bitwise repeatability of every evaluator.
_NO_PK_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
EXTRA_FLAGS = {
    "mlp_vanilla_h.hip": _NO_PK_F32,
    "mlp_tp_h.hip": _NO_PK_F32,
    "mlp_mip_h.hip": _NO_PK_F32,
    "mlp_pix_h.hip": _NO_PK_F32,
    "mlp_tp_hp.hip": _NO_PK_F32,
    "mlp_tp_hpp.hip": _NO_PK_F32,
    "pillar.hip": _NO_PK_F32,        # built with packed ops it returned ~20 wrong rows of 786,432, differently on every run (r02)
}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "neo360_hip.h"))
    newest_header = max(os.path.getmtime(h) for h in headers)
    objs, rebuilt = [], False
    procs = []
    for src in sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(OUT_DIR, src[:-4] + ".o")
        objs.append(op)
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", sp, "-o", op]
        # the object is stale when the source, a header OR its command line changed (a per-file flag added to
        # EXTRA_FLAGS must rebuild that file: round 2 chased a "race" that was an object built with the old flags)
        stamp = op + ".cmd"
        same_cmd = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if force or not same_cmd or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), newest_header):
            # the stale object and its stamp go first; the stamp is written only after a successful compile, so an
            # interrupted or failed build can never leave an old object next to a new command line
            for stale in (op, stamp):
                if os.path.exists(stale):
                    os.remove(stale)
            if verbose:
                print(" ".join(cmd))
            procs.append((src, stamp, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            rebuilt = True
    failed = []
    for src, stamp, cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append("hipcc failed on %s:\n%s" % (src, out))
            continue
        with open(stamp, "w") as f:
            f.write(" ".join(cmd))
        if verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("\n".join(failed))
    if not rebuilt and os.path.exists(LIB):
        rebuilt = any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)     # an object newer than the library
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
