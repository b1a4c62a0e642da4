"""Build gate for the DP-ALU DPP instructions that sit in inline assembly (sac_amd/csrc/simt.h: row_bcast_fma / row_bcast,
`v_fmac_f64_dpp` / `v_mov_b64_dpp ... row_newbcast`).  LLVM's hazard recogniser does not look inside inline assembly, so nothing
inserts the wait states the ISA asks for when a DPP instruction READS (as its DPP source, src0) a VGPR that a VALU instruction wrote
one or two instructions earlier (2 wait states), or when EXEC was written by the five instructions before it.  The kernels are
arranged so that the DPP source registers come out of LDS loads (ordered by s_waitcnt), but a copy the register allocator places in
front of the asm would break that silently -- the CPU emulation cannot see it.  This script disassembles the gfx950 code objects of
libsac_amd.so and fails if any `row_newbcast` instruction has such a producer in its hazard window.

    python tools/check_dpp_hazard.py [path/to/libsac_amd.so]        (exit 0 = clean; used by tests/test_cpu_product.py)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def wait_states(ins):
    """wait states an instruction between producer and DPP consumer provides: itself (1) or s_nop N (N + 1)"""
    m = re.match(r"s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def check(lib):
    tmp = tempfile.mkdtemp(prefix="dpphaz_")
    try:
        so = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, capture_output=True, cwd=tmp)
        objs = [os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f]
        assert objs, "no gfx950 code object in " + lib
        n_dpp, bad = 0, []
        for obj in objs:
            dis = subprocess.run([OBJDUMP, "-d", obj], check=True, capture_output=True, text=True).stdout.split("\n")
            ins = [l.split("//")[0].strip() for l in dis]
            ins = [i for i in ins if i and not i.endswith(":") and not i.startswith("Disassembly") and not i.startswith(os.path.basename(obj))]
            for k, i in enumerate(ins):
                if "row_newbcast" not in i:
                    continue
                n_dpp += 1
                ops = i.split(None, 1)[1].split("row_newbcast")[0].split(",")
                # v_mov_b64_dpp dst, src0 ; v_fmac_f64_dpp dst, src0, src1 : the DPP source is the first source operand
                src0 = vregs(ops[1])
                ws = 0
                for back in range(1, 6):
                    if k - back < 0:
                        break
                    p = ins[k - back]
                    mn = p.split()[0]
                    if mn.startswith("v_") and ws < 2:
                        dst = vregs(p.split(None, 1)[1].split(",")[0]) if " " in p else set()
                        if mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                            dst = set()
                        if dst & src0:
                            bad.append((obj, k, p, i, "VALU write of the DPP source %d wait state(s) before" % ws))
                    if ws < 5 and re.match(r"(s_\w+\s+exec\b|s_\w+_saveexec|v_cmpx)", p):
                        bad.append((obj, k, p, i, "EXEC write %d wait state(s) before" % ws))
                    ws += wait_states(p)
        return n_dpp, bad
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sac_amd", "libsac_amd.so")
    n, bad = check(lib)
    for b in bad[:20]:
        print("HAZARD:", b[4], "|", b[2], "->", b[3])
    print(f"{n} row_newbcast instructions checked, {len(bad)} hazards")
    sys.exit(1 if bad else 0)
