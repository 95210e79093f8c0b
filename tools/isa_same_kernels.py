"""Developer tool (CPU): which kernels of two device assemblies of pbrs_gpu.hip are the same instruction for instruction.

usage: python tools/isa_same_kernels.py OLD.s NEW.s        (hipcc <the Makefile's flags> --cuda-device-only -S -o X.s pbrs_gpu.hip)

A counter file (profiles/*traffic*.json) is stamped with a hash of ALL kernel sources; a change to one header moves the hash although
most kernels compile to the same code.  This tool compares every kernel's body (label .. .Lfunc_end, local labels renumbered in order of
appearance, comments and debug directives dropped) and its resource directives (registers, scratch, LDS) and prints the kernels that
differ: a counter file may then name the new hash under `same_isa_as_measured` for the kernels this tool finds unchanged."""
import hashlib
import re
import subprocess
import sys


def kernels(path):
    out, cur, name = {}, None, None
    for line in open(path, errors="replace"):
        s = line.split(";")[0].rstrip()
        if not s.strip():
            continue
        m = re.match(r"^(_Z\w+):", s)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if s.startswith(".Lfunc_end"):
                out[name] = cur
                cur = None
                continue
            t = s.strip()
            if t.startswith((".loc", ".file", ".cfi", ".p2align")):
                continue
            cur.append(t)
    # resource directives: the .amdhsa_kernel blocks
    res, k = {}, None
    for line in open(path, errors="replace"):
        t = line.strip()
        if t.startswith(".amdhsa_kernel "):
            k = t.split()[1]
            res[k] = []
        elif t == ".end_amdhsa_kernel":
            k = None
        elif k:
            res[k].append(t)
    return out, res


def canon(body):
    ids = {}

    def ren(m):
        return ids.setdefault(m.group(0), ".L%d" % len(ids))
    return hashlib.sha256("\n".join(re.sub(r"\.L[A-Za-z_]*\d+(_\d+)?", ren, l) for l in body).encode()).hexdigest()[:16]


def demangle(names):
    try:
        return dict(zip(names, subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.splitlines()))
    except OSError:
        return {n: n for n in names}


a, ra = kernels(sys.argv[1])
b, rb = kernels(sys.argv[2])
names = sorted(set(a) | set(b))
dm = demangle(names)
same, diff = [], []
for n in names:
    if n in a and n in b and canon(a[n]) == canon(b[n]) and ra.get(n) == rb.get(n):
        same.append(n)
    else:
        diff.append(n)
print(f"kernels {len(names)}: same {len(same)}, different {len(diff)}")
for n in diff:
    why = "only in one" if not (n in a and n in b) else f"{len(a[n])} -> {len(b[n])} lines"
    print("  differs:", dm[n].split("(")[0], f"({why})")
