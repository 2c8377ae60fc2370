"""Per-basic-block issue pattern of a gfx950 kernel: for every block that contains MFMAs, the number of other instructions
issued between consecutive MFMAs (the "gap fill" -- MI355X_MICROARCH.md: <= 5 single-issue fillers hide behind one 32x32x16 MFMA
when a wave has its SIMD to itself), split by class, plus accvgpr moves and s_nop states.

  python tools/isa_gaps.py /tmp/attn7.s flash_attn_v7_kernelIDF16_Li2ELi1E [--dump LABEL]
"""
import collections
import re
import sys


def cls(op):
    if op.startswith("v_mfma"):
        return "M"
    if "accvgpr" in op:
        return "a"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")):
        return "t"
    if op.startswith("v_"):
        return "v"
    if op.startswith("ds_"):
        return "d"
    if op.startswith("s_nop"):
        return "n"
    if op.startswith("s_waitcnt"):
        return "w"
    if op.startswith("s_barrier"):
        return "B"
    if op.startswith("s_"):
        return "s"
    return "g"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l)][0]
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        s = l.strip()
        if not s or s.startswith((".", ";")):
            continue
        blocks[cur].append(s.split(";")[0].strip())
    for name, ins in blocks.items():
        seq = "".join("|" if i.startswith("s_cbranch") else cls(i.split()[0]) for i in ins)
        if dump == name:
            print("\n".join(ins))
        if "M" not in seq:
            continue
        c = collections.Counter(seq)
        gaps = seq.split("M")
        print(f"{name}: {len(ins)} instr  " + " ".join(f"{k}{c[k]}" for k in "Mvtadgswn" if c[k]))
        print("   lead " + (gaps[0] or "-") + " : " + " ".join(g or "-" for g in gaps[1:]))


if __name__ == "__main__":
    main()
