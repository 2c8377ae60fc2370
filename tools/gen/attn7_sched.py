"""Generates the per-MFMA vector-instruction placement tables of flash_attn_v7_kernel (star_amd/csrc/attn7_sched.inc).

A pipeline step issues 18 MFMAs; the softmax of the current 32 x 64 score block is ~70 vector instructions with a dependency
DAG (exp -> pack -> sum tree -> probe).  This list-schedules the DAG into 18 chunks of at most CAP issues each (one chunk = the
shadow of one MFMA, MI355X_MICROARCH.md: <= 5 single-issue fillers per 32-cycle gap), a producer at least one chunk ahead of its
consumer (transcendental-use hazard, no s_nop).  Op encoding: kind << 8 | index;  kinds: 1 exp(n), 2 pack(i) = cvt_pk(e[2i],
e[2i+1]), 3 packed-sum add (node id), 4 final (0: lo half -> f32, 1: hi half -> f32, 2: lo + hi), 5 fp32 chain add of e[n] into chain n & 3,
6 fp32 final (0: c0 + c1, 1: c2 + c3, 2: sum), 7 overflow probe (compare + ballot).  Everything is done one chunk before the
end: the branch on the probe sits behind the LAST MFMA and must not wait for a compare issued just ahead of it.

  python tools/gen/attn7_sched.py > star_amd/csrc/attn7_sched.inc
"""
import sys

NCH = 18


def build(pksum):
    ops = {}   # id -> (code, deps)
    for n in range(32):
        ops[("e", n)] = ((1 << 8) | n, [])
    for i in range(16):
        ops[("c", i)] = ((2 << 8) | i, [("e", 2 * i), ("e", 2 * i + 1)])
    if pksum:
        # four running packed chains (chain j takes packs j, j + 4, j + 8, j + 12), then a 2-level combine: node ids
        #   0..11: chain adds (id = 4 * (step - 1) + j, step 1..3);  12: ch0 + ch1;  13: ch2 + ch3;  14: total
        for j in range(4):
            ops[("a", j)] = ((3 << 8) | j, [("c", j), ("c", j + 4)])
            ops[("a", 4 + j)] = ((3 << 8) | (4 + j), [("a", j), ("c", j + 8)])
            ops[("a", 8 + j)] = ((3 << 8) | (8 + j), [("a", 4 + j), ("c", j + 12)])
        ops[("a", 12)] = ((3 << 8) | 12, [("a", 8), ("a", 9)])
        ops[("a", 13)] = ((3 << 8) | 13, [("a", 10), ("a", 11)])
        ops[("a", 14)] = ((3 << 8) | 14, [("a", 12), ("a", 13)])
        ops[("f", 0)] = ((4 << 8) | 0, [("a", 14)])
        ops[("f", 1)] = ((4 << 8) | 1, [("a", 14)])
        ops[("f", 2)] = ((4 << 8) | 2, [("f", 0), ("f", 1)])
        ops[("p", 0)] = ((7 << 8) | 0, [("f", 2)])
    else:
        for n in range(32):
            deps = [("e", n)] + ([("s", n - 4)] if n >= 4 else [])
            ops[("s", n)] = ((5 << 8) | n, deps)
        ops[("g", 0)] = ((6 << 8) | 0, [("s", 28), ("s", 29)])
        ops[("g", 1)] = ((6 << 8) | 1, [("s", 30), ("s", 31)])
        ops[("g", 2)] = ((6 << 8) | 2, [("g", 0), ("g", 1)])
        ops[("p", 0)] = ((7 << 8) | 0, [("g", 2)])
    return ops


def schedule(ops, caps, tcaps):
    # priority = longest path to the sink (critical path first); at most 2 transcendentals per chunk (profiles/r01_overlap_probe.txt: MFMA + 2 exp hides, MFMA + 4 exp does not).
    # A consumer sits at least one chunk behind its producer (a back-to-back dependent pair costs an s_nop: transcendental-use and
    # packed-operand hazards) except in the last two chunks, where the tail of the sum chain has nowhere else to go.
    height = {}

    def h(k):
        if k not in height:
            users = [u for u, (_, d) in ops.items() if k in d]
            height[k] = 1 + max((h(u) for u in users), default=0)
        return height[k]

    for k in ops:
        h(k)
    done = {}
    chunks = []
    for c in range(NCH - 1):
        take, ntr = [], 0
        while len(take) < caps[c]:
            ready = [k for k in ops if k not in done and
                     all(d in done and (done[d] < c or (d[0] != "e" and c >= NCH - 3)) for d in ops[k][1]) and
                     not (k[0] == "e" and ntr >= tcaps[c])]
            if not ready:
                break
            ready.sort(key=lambda k: (-height[k], k))
            k = ready[0]
            take.append(k)
            done[k] = c
            ntr += k[0] == "e"
        chunks.append(take)
    chunks.append([])
    left = [k for k in ops if k not in done]
    return chunks, left


def emit(name, pksum):
    ops = build(pksum)
    # the two half-depth MFMAs of the augmented k-step offer half a shadow; raise the caps from the back until the DAG fits
    caps = [2, 2] + [4] * (NCH - 2)
    tcaps = [2, 2] + [3, 2] * 4 + [2] * (NCH - 10)     # MFMA + 2 exp hides, + 4 does not (profiles/r01_overlap_probe.txt); a third in every other early chunk
    bump = NCH - 2
    for _ in range(200):
        chunks, left = schedule(ops, caps, tcaps)
        if not left:
            break
        caps[bump] += 1
        bump = bump - 1 if bump > 2 else NCH - 2
    assert not left, left
    width = max(len(c) for c in chunks)
    print(f"// {name}: {sum(len(c) for c in chunks)} vector instructions in {NCH} chunks, at most {width} per chunk")
    print(f"constexpr int {name}_W = {width};")
    print(f"constexpr unsigned short {name}[{NCH}][{width}] = {{")
    for c in chunks:
        row = [ops[k][0] for k in c] + [0] * (width - len(c))
        print("  {" + ", ".join(f"0x{v:03x}" for v in row) + "},")
    print("};")


if __name__ == "__main__":
    print("// generated by tools/gen/attn7_sched.py -- do not edit")
    emit("V7_SCHED_PK", True)
    emit("V7_SCHED_F32", False)
