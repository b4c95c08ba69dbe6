#!/usr/bin/env python3
"""Opcode histograms of the gfx950 code objects behind the extraction kernels, priced with the measured per-opcode issue rates
(profiles/valu_rate_r02.txt, tools/c/valu_rate.hip): what a vector instruction of THIS kernel costs on average, instead of the flat
4 cycles bench.py used through round 5.

    python tools/opcode_histogram.py [--out profiles/opcode_mix_r06.json]            (no GPU needed: it reads csrc/*.o)

For every kernel: the device code object is cut out of the object file (llvm-objcopy .hip_fatbin, clang-offload-bundler), disassembled
(llvm-objdump -d), and its instructions are classified (VALU / SALU / LDS / VMEM / SMEM / other).  Loops are found from backward
branches; an instruction INSIDE a loop counts with weight 1, the straight-line prologue / epilogue with weight 0 (`all` keeps the
unweighted mix beside it).  A static mix is not a dynamic one -- loops iterate different numbers of times -- so the result is a
weighted MEAN COST PER VECTOR INSTRUCTION (cycles), which bench.py multiplies with the dynamic count the hardware reports
(SQ_INSTS_VALU): issue_frac_weighted = SQ_INSTS_VALU x mean_cycles / (1024 SIMDs x 2.4 GHz x duration).

The rates: a microbenchmark of 2^20 dependent-free instructions per wave, 8 waves per SIMD (tools/c/valu_rate.hip).  Plain 32-bit
logic / add / shift-right / move and fp32 add / mul issue in ~2.4 cycles; everything packed, every min / max / compare / v_perm /
three-operand integer op, dot products and cross-lane ops take ~4.2.  An opcode the microbenchmark did not cover is priced at 4.2
(the conservative end) and listed under `unpriced`.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(ROOT, "gf-orb-slam2_amd", "csrc")

# kernel (substring of the mangled name) -> object file; the first symbol that contains the substring is taken
KERNELS = {
    "fast": ("k_fast.o", "k_fastILi48ELi44ELb1E"),
    "orient_desc": ("k_orient_desc.o", "k_orient_desc"),
    "blur": ("k_blur.o", "11k_blur_mfma"),
    "blur_streaming": ("k_blur.o", "6k_blurPK"),
    "resize": ("k_pyramid.o", "8k_resize"),
    "resize_tail": ("k_pyramid.o", "k_resize_tail"),
    "pyramid_bands": ("k_pyramid.o", "k_pyramid_bands"),
    "quadtree": ("k_quadtree.o", "10k_quadtree"),
    "stereo_match": ("k_stereo.o", "k_stereo_match_rows"),
}

FULL = 2.4     # measured 2.28 .. 2.63
HALF = 4.2     # measured 4.09 .. 4.44
# mnemonic (without _e32 / _e64 / _sdwa / _dpp suffix) -> cycles per wave-instruction, from profiles/valu_rate_r02.txt
RATES = {
    "v_and_b32": FULL, "v_or_b32": FULL, "v_xor_b32": FULL, "v_add_u32": FULL, "v_sub_u32": FULL, "v_subrev_u32": FULL,
    "v_lshrrev_b32": FULL, "v_ashrrev_i32": FULL, "v_mov_b32": FULL, "v_mul_f32": FULL, "v_add_f32": FULL, "v_sub_f32": FULL,
    "v_add_u16": FULL, "v_not_b32": FULL,
    "v_min_u32": HALF, "v_max_u32": HALF, "v_min_i32": HALF, "v_max_i32": HALF, "v_lshl_or_b32": HALF, "v_add3_u32": HALF, "v_bfe_u32": HALF,
    "v_bfe_i32": HALF, "v_pk_min_i16": HALF, "v_pk_max_i16": HALF, "v_pk_sub_i16": HALF, "v_pk_add_u16": HALF, "v_pk_add_i16": HALF,
    "v_pk_sub_u16": HALF, "v_pk_lshlrev_b16": HALF, "v_pk_mad_i16": HALF, "v_pk_mad_u16": HALF, "v_pk_mul_lo_u16": HALF, "v_min_f32": HALF,
    "v_max_f32": HALF, "v_fma_f32": HALF, "v_fmac_f32": HALF, "v_mul_i32_i24": HALF, "v_mul_u32_u24": HALF, "v_mul_lo_u32": HALF,
    "v_perm_b32": HALF, "v_alignbit_b32": HALF, "v_alignbyte_b32": HALF, "v_cndmask_b32": 4.3, "v_dot4_u32_u8": 4.4, "v_dot2_u32_u16": 4.4,
    "v_sad_u32": HALF, "v_bcnt_u32_b32": HALF, "v_mbcnt_lo_u32_b32": HALF, "v_mbcnt_hi_u32_b32": HALF, "v_lshlrev_b32": HALF,
    "v_and_or_b32": HALF, "v_or3_b32": HALF, "v_lshl_add_u32": HALF, "v_add_lshl_u32": HALF, "v_xad_u32": HALF, "v_min3_u32": HALF,
    "v_med3_u32": HALF, "v_min3_i32": HALF, "v_max3_i32": HALF, "v_max3_u32": HALF, "v_med3_i32": HALF, "v_cvt_f32_i32": HALF, "v_cvt_f32_u32": HALF,
    "v_cvt_i32_f32": HALF, "v_cvt_u32_f32": HALF, "v_cvt_f32_ubyte0": HALF, "v_cvt_f64_f32": HALF, "v_cvt_f32_f64": HALF,
    # matrix instructions hold the SIMD's vector issue for 8 of their 16 cycles (MI355X_MICROARCH.md); tools/c/mfma_i8_layout.hip: one per 8.6 ns
    "v_mfma_i32_16x16x64_i8": 8.0, "v_mfma_i32_16x16x32_i8": 8.0, "v_permlane32_swap_b32": HALF, "v_permlane16_swap_b32": HALF,
}
CMP = re.compile(r"^v_cmpx?_")   # every v_cmp measured 4.24


def base(mn):
    for suf in ("_e32", "_e64", "_sdwa", "_dpp", "_e64_dpp"):
        if mn.endswith(suf):
            return mn[:-len(suf)]
    return mn


def klass(mn):
    if mn.startswith(("ds_",)):
        return "lds"
    if mn.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if mn.startswith("s_load") or mn.startswith("s_buffer_load") or mn.startswith("s_store"):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("v_"):
        return "valu"
    return "other"


def rate(mn):
    b = base(mn)
    if b in RATES:
        return RATES[b], True
    if CMP.match(b):
        return 4.24, True
    if b.startswith("v_readlane") or b.startswith("v_readfirstlane") or b.startswith("v_writelane"):
        return HALF, True
    if "f64" in b:                 # double precision: priced at the half rate (the microbenchmark has no f64 op; gfx950's f64 FMA rate
        return 8.0, False          # is half the fp32 one) -- only gfo_sincosf uses them, ~60 per keypoint
    return HALF, False


def disassemble(obj, tmp):
    fat = os.path.join(tmp, os.path.basename(obj) + ".fat")
    co = os.path.join(tmp, os.path.basename(obj) + ".co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(tmp, "discard.o")])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--output={co}"])
    return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)


FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:$")
INS = re.compile(r"^\s+(\S+)(.*?)//\s*([0-9A-F]+):")
TARGET = re.compile(r"<[^>]+\+0x([0-9a-f]+)>|<([^>+]+)>$")


def functions(asm):
    out, cur = {}, None
    for line in asm.splitlines():
        m = FUNC.match(line)
        if m:
            cur = m.group(2)
            out[cur] = {"addr": int(m.group(1), 16), "ins": []}
            continue
        if cur is None:
            continue
        m = INS.match(line)
        if m:
            mn, addr = m.group(1), int(m.group(3), 16)
            tgt = None
            if mn.startswith(("s_cbranch", "s_branch")):
                t = TARGET.search(line)
                if t:
                    tgt = out[cur]["addr"] + (int(t.group(1), 16) if t.group(1) else 0)
            out[cur]["ins"].append((addr, mn, tgt))
    return out


def analyse(fn):
    ins = fn["ins"]
    loops = [(t, a) for a, mn, t in ins if t is not None and t <= a]      # backward branch: the region [target, branch] is a loop
    def in_loop(a):
        return any(lo <= a <= hi for lo, hi in loops)
    res = {}
    for scope in ("loops", "all"):
        cnt = collections.Counter()
        kl = collections.Counter()
        for a, mn, _ in ins:
            if scope == "loops" and not in_loop(a):
                continue
            kl[klass(mn)] += 1
            if klass(mn) == "valu":
                cnt[base(mn)] += 1
        nv = sum(cnt.values())
        cyc = sum(rate(m)[0] * c for m, c in cnt.items())
        unpriced = {m: c for m, c in cnt.items() if not rate(m)[1]}
        full = sum(c for m, c in cnt.items() if rate(m)[0] <= FULL)
        res[scope] = {"instructions": dict(kl), "valu": nv, "mean_cycles_per_valu": round(cyc / nv, 3) if nv else None,
                      "valu_at_full_rate": full, "valu_share_at_full_rate": round(full / nv, 3) if nv else None,
                      "unpriced_valu": unpriced, "top_valu": dict(cnt.most_common(24))}
    res["loop_regions"] = len(loops)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "opcode_mix_latest.json"))
    ap.add_argument("--csrc", default=CSRC)
    args = ap.parse_args()
    out = {"what": "static opcode mix of the gfx950 code objects (tools/opcode_histogram.py), priced with profiles/valu_rate_r02.txt; "
                   "`loops` = instructions inside a loop (weight 1), `all` = every instruction of the kernel",
           "rates": {"full_rate_cycles": FULL, "half_rate_cycles": HALF, "source": "profiles/valu_rate_r02.txt (tools/c/valu_rate.hip, one MI355X)"},
           "kernels": {}}
    with tempfile.TemporaryDirectory() as tmp:
        cache = {}
        for name, (obj, sym) in KERNELS.items():
            path = os.path.join(args.csrc, obj)
            if not os.path.exists(path):
                print(f"{path} missing: build the library first", file=sys.stderr)
                return 1
            if obj not in cache:
                cache[obj] = functions(disassemble(path, tmp))
            match = [k for k in cache[obj] if sym in k]
            if not match:
                print(f"no symbol containing {sym} in {obj}", file=sys.stderr)
                continue
            r = analyse(cache[obj][match[0]])
            r["symbol"] = match[0]
            out["kernels"][name] = r
            # a kernel whose work is straight-line code (fully unrolled: k_orient_desc) has next to nothing inside loops: its mix is `all`
            use = "loops" if r["loops"]["valu"] >= 0.5 * r["all"]["valu"] else "all"
            r["mean_cycles_per_valu"] = r[use]["mean_cycles_per_valu"]
            r["priced_scope"] = use
            u = r[use]
            print(f"{name:14s} [{use:5s}] {u['valu']:5d} VALU, mean {u['mean_cycles_per_valu']} cycles ({u['valu_share_at_full_rate']:.0%} at full rate), "
                  f"LDS {u['instructions'].get('lds', 0)}, VMEM {u['instructions'].get('vmem', 0)}, SALU {u['instructions'].get('salu', 0)}; "
                  f"whole kernel: {r['all']['valu']} VALU, mean {r['all']['mean_cycles_per_valu']}")
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
