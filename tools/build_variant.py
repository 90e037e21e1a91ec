"""Link a variant of the library with one translation unit (or several: a,b,c) recompiled with extra flags.
usage: build_variant.py <name> <source.hip>[,<source2.hip>...] [extra hipcc flags...]  ->  tools/build/libneo_<name>.so
(select it with NEO360_HIP_LIB=<path>; kernel experiments only; NEO_VARIANT_NO_EXTRA=1 drops build.py's per-file flags)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neo-360_amd"))
import build as B
B.build()
name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
out = os.path.join(ROOT, "tools", "build")
os.makedirs(out, exist_ok=True)
srcs = src.split(",")
vobjs = []
for one in srcs:
    obj = os.path.join(out, "%s_%s.o" % (one[:-4], name))
    subprocess.check_call([B._hipcc()] + B.FLAGS + ([] if os.environ.get("NEO_VARIANT_NO_EXTRA") else B.EXTRA_FLAGS.get(one, [])) + extra + ["-c", os.path.join(B.CSRC, one), "-o", obj],
                          stderr=subprocess.DEVNULL)
    vobjs.append(obj)
objs = [os.path.join(B.OUT_DIR, s[:-4] + ".o") for s in B.sources() if s not in srcs] + vobjs
lib = os.path.join(out, "libneo_%s.so" % name)
subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", lib] + objs)
print(lib)
