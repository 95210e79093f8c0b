"""tools/isa_same_kernels.py on two small hand-made device assemblies: a kernel whose local labels are merely renumbered counts as the
same, one with another instruction or another register budget does not (the comparison behind `same_isa_as_measured`)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KERNEL = """
\t.globl\t{name}
{name}:
; %bb.0:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\t.loc 1 {line} 0
\tv_mov_b32_e32 v1, {imm}
\ts_cbranch_scc1 .LBB{n}_2
.LBB{n}_2:
\ts_endpgm
.Lfunc_end{n}:
\t.amdhsa_kernel {name}
\t\t.amdhsa_next_free_vgpr {vgpr}
\t.end_amdhsa_kernel
"""


def asm(path, specs):
    with open(path, "w") as f:
        for n, (name, imm, vgpr) in enumerate(specs):
            f.write(KERNEL.format(name=name, n=n + 7 * (path.endswith("b.s")), imm=imm, vgpr=vgpr, line=10 + n))


def test_kernels_that_differ_are_named(tmp_path):
    a, b = str(tmp_path / "a.s"), str(tmp_path / "b.s")
    asm(a, [("_Z5k_onev", 1, 8), ("_Z5k_twov", 2, 8), ("_Z7k_threev", 3, 8)])
    asm(b, [("_Z5k_onev", 1, 8), ("_Z5k_twov", 5, 8), ("_Z7k_threev", 3, 16)])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_same_kernels.py"), a, b], capture_output=True, text=True, check=True).stdout
    assert "kernels 3: same 1, different 2" in out
    assert "k_two" in out and "k_three" in out and "k_one" not in out.split("\n", 1)[1]
