"""Per-shape traffic / matrix-pipe table from single-shape rocprofv3 --pmc passes on tools/cbench (tools/r06_calls.sh 1).

    python tools/pmc_traffic_table.py <spec file> <dir with one sub-directory per shape index and counter group> [cbench timing log]

For shape i of the spec file the passes live in <dir>/s<i>_FETCH, <dir>/s<i>_WRITE, <dir>/s<i>_SQ (rocpd .db files).  Per launch of the
op (a tail-split GEMM is two kernels per launch: their per-kernel means are summed) the table gives
  * FETCH = 2 x FETCH_SIZE KB (gfx950: 128-byte requests tallied at 64, MI355X_MICROARCH.md "HBM"), WRITE = WRITE_SIZE KB -- requests the
    L2 sends to the fabric (Infinity-Cache hits included), against the ALGORITHMIC bytes (every operand read once, the output written once);
  * matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs);
  * wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls), parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier).
Measurement tooling."""
import glob
import os
import re
import sqlite3
import sys


def counters(d):
    out = {}
    for db in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):
        c = sqlite3.connect(db)
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
        for k, cn, n, v in rows:
            if k.startswith("__amd_rocclr") or "fill" in k.lower() and "star" not in k:
                continue
            out.setdefault(cn, {})[k] = (n, v)
    return out


def per_launch(cs, name):
    ks = cs.get(name, {})
    if not ks:
        return None
    nmax = max(n for n, _ in ks.values())
    # kernels launched (about) once per op launch; helper kernels launched once per process (layer_norm_rowab for tq) are dropped
    return sum(v for n, v in ks.values() if n * 4 >= nmax)


def algorithmic_bytes(spec):
    t = spec.split()
    kind, a = t[0], [int(x.split(",")[0]) for x in t[1:]]
    if kind == "gemm":
        M, N, K, epi = a[:4]
        n_out = N // 2 if epi & 4 else N
        return 2 * (M * K + N * K + M * n_out * (2 if epi & 2 else 1)), 2.0 * M * N * K
    if kind == "conv":
        NB, H, W, Cin, Cout = a[:5]
        M = NB * H * W
        return 2 * (M * Cin + Cout * 9 * Cin + M * Cout), 2.0 * M * Cout * 9 * Cin
    if kind == "tconv":
        F, HW, C = a[:3]
        M = F * HW
        return 2 * (M * C + 3 * C * C + 2 * M * C), 2.0 * M * C * 3 * C
    if kind == "attn":
        B, h, Nq, Nk = a[:4]
        return 2 * B * h * 64 * (2 * Nq + 2 * Nk), 4.0 * B * h * Nq * Nk * 64
    if kind == "tq":
        F, HW = a[:2]
        M = F * HW
        return 2 * (M * 320 + 960 * 320 + M * 320), 2.0 * M * 960 * 320 + 4.0 * M * F * 320
    raise ValueError(spec)


def main():
    spec_file, root = sys.argv[1], sys.argv[2]
    timing = {}
    if len(sys.argv) > 3:
        lines = [l for l in open(sys.argv[3]) if " min " in l and " ms " in l]
        for i, l in enumerate(lines):
            m = re.search(r"mean\s+([0-9.]+) ms", l)
            timing[i] = float(m.group(1))
    specs = [l.split("#")[0].strip() for l in open(spec_file)]
    specs = [s for s in specs if s]
    print(f"{'shape':36s} {'ms':>7s} {'TF/s':>7s} {'alg GB':>7s} {'FETCH GB':>9s} {'WRITE GB':>9s} {'(F+W)/alg':>9s} {'GB/s moved':>10s} {'MFMA busy':>9s} {'wait':>6s} {'parked':>6s} {'VALU/MFMA':>9s}")
    for i, s in enumerate(specs):
        alg, flops = algorithmic_bytes(s)
        f = per_launch(counters(os.path.join(root, f"s{i}_FETCH")), "FETCH_SIZE")
        w = per_launch(counters(os.path.join(root, f"s{i}_WRITE")), "WRITE_SIZE")
        sq = counters(os.path.join(root, f"s{i}_SQ"))
        g = lambda n: per_launch(sq, n)
        ms = timing.get(i)
        fgb = None if f is None else 2 * f * 1024 / 1e9
        wgb = None if w is None else w * 1024 / 1e9
        busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024) if g("GRBM_GUI_ACTIVE") else None
        wait = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None
        park = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None else None
        vm = g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA") if g("SQ_INSTS_MFMA") else None
        fmt = lambda v, p="{:.3f}": "-" if v is None else p.format(v)
        tot = None if fgb is None or wgb is None else fgb + wgb
        print(f"{s:36s} {fmt(ms):>7s} {fmt(None if ms is None else flops / ms / 1e9, '{:.0f}'):>7s} {alg / 1e9:7.3f} {fmt(fgb):>9s} {fmt(wgb):>9s} "
              f"{fmt(None if tot is None else tot / (alg / 1e9), '{:.2f}'):>9s} {fmt(None if tot is None or ms is None else tot / ms * 1e3, '{:.0f}'):>10s} "
              f"{fmt(busy, '{:.1%}'):>9s} {fmt(wait, '{:.2f}'):>6s} {fmt(park, '{:.2f}'):>6s} {fmt(vm, '{:.2f}'):>9s}")


if __name__ == "__main__":
    main()
