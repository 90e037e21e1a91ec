"""Hand-perturbed ISA variants of the failing kernel of tools/pk_f32_repro.hip (k_mix<4,true>), to find which
ingredient of the packed-fp32 glitch matters.  Usage (no GPU needed to build):
    python tools/pk_f32_variants.py            # writes tools/build/pkv/<variant>.co
then on the GPU box: for f in tools/build/pkv/*.co; do tools/build/pk_f32_co_run $f; done
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(ROOT, "build", "pkv")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_Z5k_mixILi4ELb1EEvPKDv4_fjiPS0_Pf"

PK_MOV = re.compile(r"^\tv_pk_mov_b32 v\[(\d+):(\d+)\], v\[102:103\], v\[102:103\]$")
PK_FMA = re.compile(r"^\tv_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] (op_sel.*)$")
PK_MUL = re.compile(r"^\tv_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel_hi:\[1,0\]$")


def scalar_fma(m):
    d0, d1, a0, a1, w0, w1, c0, c1, sel = m.groups()
    w = w1 if sel.startswith("op_sel:[0,1,0]") else w0          # both halves use the high / the low weight
    return [f"\tv_fma_f32 v{d0}, v{a0}, v{w}, v{c0}", f"\tv_fma_f32 v{d1}, v{a1}, v{w}, v{c1}"]


def variant(lines, name):
    out, inside, in_loop = [], False, False
    for ln in lines:
        if ln.startswith(KERNEL + ":"):
            inside = True
        if inside and re.match(r"^\.LBB\d+_6:", ln):
            in_loop = True
        if not (inside and in_loop):
            out.append(ln)
            continue
        mov, fma = PK_MOV.match(ln), PK_FMA.match(ln)
        if name == "no_war" and mov:
            out.append("\ts_nop 0")
        elif name == "nop7_between" and mov:
            out += ["\ts_nop 7", ln]
        elif name == "nop1_between" and mov:
            out += ["\ts_nop 1", ln]
        elif name == "war_by_v_mov" and mov:
            out += [f"\tv_mov_b32 v{mov.group(1)}, v102", f"\tv_mov_b32 v{mov.group(2)}, v103"]
        elif name == "scalar_fma_keep_war" and fma:
            out += scalar_fma(fma)
        elif name == "no_mfma" and "v_mfma_" in ln:
            out.append("\ts_nop 0")
        elif name == "junk_const" and ("v_mul_f32_e32 v102," in ln or "v_add_f32_e32 v103," in ln):
            out.append("\tv_mov_b32 v102, 1.0" if "v102" in ln else "\tv_mov_b32 v103, 2.0")
        elif name == "war_other_regs" and mov:
            # the same write, but to a pair nobody reads (v[106:107] is unused in this kernel): no WAR
            out.append("\tv_pk_mov_b32 v[106:107], v[102:103], v[102:103]")
        elif name == "no_op_sel" and "s_waitcnt vmcnt(3) lgkmcnt(0)" in ln:
            # splat the four weights into pairs v[106:113]; the packed ops below then need no op_sel
            out.append(ln)
            for k in range(4):
                out += [f"\tv_mov_b32 v{106 + 2 * k}, v{98 + k}", f"\tv_mov_b32 v{107 + 2 * k}, v{98 + k}"]
        elif name == "no_op_sel" and (fma or PK_MUL.match(ln)):
            if fma:
                d0, d1, a0, a1, w0, w1, c0, c1, sel = fma.groups()
                k = (int(w0) - 98) + (1 if sel.startswith("op_sel:[0,1,0]") else 0)
                out.append(f"\tv_pk_fma_f32 v[{d0}:{d1}], v[{a0}:{a1}], v[{106 + 2 * k}:{107 + 2 * k}], v[{c0}:{c1}]")
            else:
                d0, d1, a0, a1, w0, w1 = PK_MUL.match(ln).groups()
                k = int(w0) - 98
                out.append(f"\tv_pk_mul_f32 v[{d0}:{d1}], v[{a0}:{a1}], v[{106 + 2 * k}:{107 + 2 * k}]")
        elif name == "copy_weights" and "s_waitcnt vmcnt(3) lgkmcnt(0)" in ln:
            # op_sel kept, but the packed ops read VALU-written copies of the weights instead of the ds_read destination
            out += [ln, "\tv_mov_b64 v[106:107], v[98:99]", "\tv_mov_b64 v[108:109], v[100:101]"]
        elif name == "copy_weights" and (fma or PK_MUL.match(ln)):
            out.append(ln.replace("v[98:99]", "v[106:107]").replace("v[100:101]", "v[108:109]"))
        elif name == "wait_after_lds" and "s_waitcnt vmcnt(3) lgkmcnt(0)" in ln:
            out += [ln, "\ts_nop 7", "\ts_nop 7", "\ts_nop 7", "\ts_nop 7"]
        elif name == "scalar_mul_keep_pk_fma" and PK_MUL.match(ln):
            d0, d1, a0, a1, w0, w1 = PK_MUL.match(ln).groups()
            out += [f"\tv_mul_f32 v{d0}, v{a0}, v{w0}", f"\tv_mul_f32 v{d1}, v{a1}, v{w0}"]
        elif name == "mfma_32x32x8" and "v_mfma_f32_32x32x16_f16" in ln:
            m = re.match(r"^\tv_mfma_f32_32x32x16_f16 (v\[\d+:\d+\]), v\[(\d+):\d+\], v\[(\d+):\d+\], (v\[\d+:\d+\])$", ln)
            a, b = int(m.group(2)), int(m.group(3))
            out.append(f"\tv_mfma_f32_32x32x8_f16 {m.group(1)}, v[{a}:{a + 1}], v[{b}:{b + 1}], {m.group(4)}")
        elif name == "nop3_after_mfma" and "v_mfma_" in ln:
            out += [ln, "\ts_nop 3"]
        elif name == "nop15_after_mfma" and "v_mfma_" in ln:
            out += [ln, "\ts_nop 7", "\ts_nop 7"]
        else:
            out.append(ln)
        if "s_cbranch_scc1 .LBB" in ln and "_6" in ln:
            in_loop = inside = False
    return out


def bump_vgprs(text, count=120):
    out, seen, meta = [], False, False
    for ln in text:
        if ln.startswith("\t.amdhsa_kernel " + KERNEL):
            seen = True
        if seen and ".amdhsa_next_free_vgpr" in ln:
            ln = f"\t\t.amdhsa_next_free_vgpr {count}"
        if seen and ".amdhsa_accum_offset" in ln:
            ln, seen = f"\t\t.amdhsa_accum_offset {count}", False
        if ".name:" in ln and KERNEL in ln and ".kd" not in ln:
            meta = True
        if meta and ".vgpr_count:" in ln:
            ln, meta = f"    .vgpr_count:     {count}", False
        out.append(ln)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    base = os.path.join(OUT, "base.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                           "-o", base, os.path.join(ROOT, "pk_f32_repro.hip")], stderr=subprocess.DEVNULL)
    lines = open(base).read().split("\n")
    # v[106:107] must be free for war_other_regs: bump the kernel's VGPR count if needed
    for name in ("base", "no_war", "nop1_between", "nop7_between", "war_by_v_mov", "scalar_fma_keep_war", "no_mfma",
                 "junk_const", "war_other_regs", "no_op_sel", "scalar_mul_keep_pk_fma", "mfma_32x32x8", "nop3_after_mfma",
                 "nop15_after_mfma", "copy_weights", "wait_after_lds"):
        text = lines if name == "base" else variant(lines, name)
        if name != "base" and text == lines:
            sys.exit(f"variant {name}: nothing changed")
        if name in ("no_op_sel", "copy_weights"):                      # v[106:113] are new: raise this kernel's VGPR allocation
            text = bump_vgprs(text)
        s = os.path.join(OUT, name + ".s")
        open(s, "w").write("\n".join(text))
        subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s,
                               "-o", s[:-2] + ".o"])
        subprocess.check_call([f"{LLVM}/ld.lld", "-shared", s[:-2] + ".o", "-o", s[:-2] + ".co"])
        print("built", name)


if __name__ == "__main__":
    main()
