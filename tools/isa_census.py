"""Instruction census of the loops of a gfx950 kernel (the method behind DESIGN.md section 3.5): compiles one source of
star_amd/csrc to assembly with the product flags and prints, for every backward branch of the chosen kernel, the number of MFMA /
VALU / SALU / LDS / vector-memory instructions between the loop label and the branch, plus the most frequent opcodes.

  python tools/isa_census.py attn.cpp flash_attn_v5_kernelIDF16_Li1ELi1E          (substring of the mangled kernel name)
  python tools/isa_census.py gemm_f16.cpp 'gemm_kernelIDF16_Li256ELi320ELi4ELi2ELi0ELi2ELb0ELb0ELi0ELi0ELi0E' [--bench]

(no GPU needed: hipcc cross-compiles)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, pat = sys.argv[1], sys.argv[2]
    bench = "--bench" in sys.argv
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-Wno-unused-value", "-Wno-inline-asm", "-ffp-contract=fast",
             "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S"]
    if src.startswith("attn"):
        flags += ["-fno-honor-nans", "-fno-slp-vectorize"]
    if bench:
        flags += ["-DSTAR_BENCH_VARIANTS=1"]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["hipcc"] + flags + [os.path.join(ROOT, "star_amd", "csrc", src), "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l)]
    if not starts:
        sys.exit(f"no kernel matching {pat}")
    start = starts[0]
    name = lines[start].split(":")[0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    meta = "\n".join(lines)
    m = re.search(r"\.name:\s+" + re.escape(name) + r"\n(.*?)\.vgpr_spill_count:\s+(\d+)", meta, re.S)
    vg = re.search(r"\.vgpr_count:\s+(\d+)", m.group(1)).group(1) if m else "?"
    print(f"{name}\n  {len([l for l in body if l.strip() and not l.strip().startswith((';', '.'))])} instructions, {vg} VGPRs, "
          f"{m.group(2) if m else '?'} spilled")
    labels = {mm.group(1): i for i, l in enumerate(body) if (mm := re.match(r"^(\.LBB\d+_\d+):", l))}

    def census(seg):
        c = collections.Counter()
        for s in seg:
            s = s.strip()
            if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
                continue
            c[s.split()[0]] += 1
        cls = collections.Counter()
        for op, n in c.items():
            k = "mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else \
                "salu" if op.startswith("s_") else "vmem"
            cls[k] += n
        return c, cls

    for i, l in enumerate(body):
        mm = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(2) in labels and labels[mm.group(2)] < i:
            c, cls = census(body[labels[mm.group(2)]:i + 1])
            print(f"  loop {mm.group(2)} [{labels[mm.group(2)]}..{i}]: " + ", ".join(f"{k} {cls[k]}" for k in ("mfma", "valu", "salu", "lds", "vmem")))
            print("     " + ", ".join(f"{op} {n}" for op, n in c.most_common(14)))


if __name__ == "__main__":
    main()
