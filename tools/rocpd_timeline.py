#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 rocpd database (the default output format of ROCm 7.2's rocprofv3 --kernel-trace):
splits the trace at the wav_prep launches (one per encoder forward), prints for ONE chosen forward+rest-of-step the kernels after the
last encoder layer (the "tail": weighted sum, head, loss, optimiser) with start offsets, durations and gaps, and a per-name summary.

    python tools/rocpd_timeline.py gpurun_out/prof/run_results.db [--step 3] [--csv profiles/x_kernel_stats.csv]
"""
import argparse
import collections
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--step", type=int, default=3)
    ap.add_argument("--csv", default=None, help="also write a kernel_stats.csv (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs)")
    ap.add_argument("--all", action="store_true", help="print every kernel of the step, not only the tail")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    if args.csv:
        agg = collections.OrderedDict()
        for n, s, e in rows:
            a = agg.setdefault(n, [0, 0, 1 << 62, 0])
            a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
        tot = sum(a[1] for a in agg.values())
        with open(args.csv, "w") as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
            for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f'"{n}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100.0 * a[1] / tot:.2f},{a[2]},{a[3]}\n')
    starts = [i for i, r in enumerate(rows) if "wav_prep" in r[0]]
    i0 = starts[args.step]
    i1 = starts[args.step + 1] if args.step + 1 < len(starts) else len(rows)
    step = rows[i0:i1]
    t0 = step[0][1]
    print(f"step {args.step}: {len(step)} launches, {(step[-1][2] - t0) / 1e6:.3f} ms from first kernel start to last kernel end")
    # tail = after the last gemm256 / attention / layernorm16 launch of the encoder (first wsum kernel)
    first_tail = next((i for i, r in enumerate(step) if "wsum" in r[0]), 0)
    sel = step if args.all else step[first_tail:]
    prev_end = step[first_tail - 1][2] if first_tail > 0 and not args.all else sel[0][1]
    busy = 0
    for n, s, e in sel:
        print(f"{(s - t0) / 1e3:10.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {n[:100]}")
        prev_end = max(prev_end, e)
        busy += e - s
    print(f"tail: {len(sel)} launches, busy {busy / 1e6:.3f} ms, span {(sel[-1][2] - sel[0][1]) / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
