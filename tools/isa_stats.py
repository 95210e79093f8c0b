#!/usr/bin/env python3
"""Developer tool: static instruction mix and register use per kernel, from the device assembly of pbrs_gpu.hip.

    python tools/isa_stats.py [--json PATH] [extra hipcc flags ...]     (CPU only; hipcc cross-compiles gfx950)

--json PATH also writes, per kernel, the static instruction mix bench.py prices the vector issue share with (pbrs_amd/roofline.py):
vector instructions in all, those in a 32-bit encoding (`_e32`: VOP1 / VOP2 / VOPC forms without a literal — 2.7 issue cycles per
wave on gfx950, tools/microbench/issue_rates.hip; the 64-bit encodings, DPP / SDWA forms and instructions with a 32-bit literal
take about 4) and scalar instructions, stamped with the hash of the sources (profiles/latest_isa_mix.json).

Prints, per kernel: VGPRs, SGPRs, scratch bytes, waves per SIMD, and the static counts of the instruction families that
matter for the VALU-bound stages (correctly rounded f32 division = v_div_scale x2 + v_rcp + v_div_fmas + v_div_fixup,
square roots, f64 products of the division-free box test, memory and LDS operations).  Static counts say what a kernel
is made of, not what it executes: the dynamic mix comes from rocprofv3 (profiles/*_pmc_*).
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pbrs_amd", "csrc", "pbrs_gpu.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize",
         "-Wno-unused-function", "--cuda-device-only", "-S"]

FAMILIES = [
    ("div", r"v_div_fixup_f32"), ("rcp", r"v_rcp_f32"), ("sqrt", r"v_sqrt_f32"), ("rsq", r"v_rsq_f32"), ("f64", r"v_\w+_f64"),
    ("cvt64", r"v_cvt_f(64_f32|32_f64)"), ("fma", r"v_fma_f32|v_fmac_f32"), ("cndmask", r"v_cndmask"), ("valu", r"^v_"), ("valu32", r"^v_\w+_e32$"), ("salu", r"^s_(?!waitcnt|nop|endpgm|branch|cbranch|barrier)"),
    ("branch", r"^s_c?branch"), ("waitcnt", r"^s_waitcnt"), ("vmem_ld", r"^(global|buffer|flat)_load"), ("vmem_st", r"^(global|buffer|flat)_store"),
    ("atomic", r"^(global|buffer|flat)_atomic"), ("scratch", r"^scratch_"), ("lds", r"^ds_"), ("bpermute", r"ds_b?permute"),
]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    argv = sys.argv[1:]
    json_path = None
    if "--json" in argv:
        k = argv.index("--json")
        json_path = argv[k + 1]
        del argv[k:k + 2]
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "dev.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + argv + ["-o", asm, SRC], cwd=os.path.dirname(SRC))
        text = open(asm).read()
    kernels = {}
    cur = None
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            pass
        s = line.strip()
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            m = re.match(r";\s*(NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|TotalNumSgprs):\s*(\d+)", s)
            if m:
                kernels[cur]["@" + m.group(1)] = int(m.group(2))
            continue
        op = s.split()[0]
        kernels[cur]["insts"] += 1
        for fam, pat in FAMILIES:
            if re.search(pat, op):
                if fam == "valu32" and re.search(r"\b0x[0-9a-f]+\b", s):
                    continue  # a 32-bit literal follows the instruction word: 64 bits in all
                kernels[cur][fam] += 1
    names = demangle(list(kernels))
    cols = ["insts"] + [f for f, _ in FAMILIES]
    print("%-44s %5s %5s %7s %4s | " % ("kernel", "vgpr", "sgpr", "scratch", "occ") + " ".join("%7s" % c for c in cols))
    for k, c in kernels.items():
        if "@NumVgprs" not in c:
            continue
        n = names[k].split("(")[0].replace("void ", "")
        print("%-44s %5d %5d %7d %4d | " % (n[:44], c["@NumVgprs"], c.get("@TotalNumSgprs", c["@NumSgprs"]), c["@ScratchSize"], c["@Occupancy"]) +
              " ".join("%7d" % c[f] for f in cols))
    if json_path:
        import json
        sys.path.insert(0, ROOT)
        from pbrs_amd import roofline
        doc = {"source_hash": roofline.source_hash(), "flags": " ".join(FLAGS + argv),
               "note": "static counts per kernel: valu = vector instructions, valu32 = those in a 32-bit encoding (no literal), salu = scalar instructions",
               "kernels": {names[k].split("(")[0].replace("void ", ""): {"valu": c["valu"], "valu32": c["valu32"], "salu": c["salu"], "vgpr": c["@NumVgprs"],
                                                                          "scratch": c["@ScratchSize"], "waves_per_simd": c["@Occupancy"]}
                           for k, c in kernels.items() if "@NumVgprs" in c}}
        with open(json_path, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
