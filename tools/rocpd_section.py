#!/usr/bin/env python3
"""Per-kernel totals of ONE train step from a rocprofv3 rocpd database, split into encoder / tail (first wsum launch onward), with
launch counts: the accounting DESIGN.md's per-recipe tables are made from.

    python tools/rocpd_section.py run_results.db [--step 3] [--top 25]
"""
import argparse
import collections
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--step", type=int, default=3)
    ap.add_argument("--top", type=int, default=30)
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if "wav_prep" in r[0]]
    i0 = starts[args.step]
    i1 = starts[args.step + 1] if args.step + 1 < len(starts) else len(rows)
    step = rows[i0:i1]
    first_tail = next((i for i, r in enumerate(step) if "wsum" in r[0]), len(step))
    print(f"step {args.step}: {len(step)} launches, {(step[-1][2] - step[0][1]) / 1e6:.3f} ms wall; encoder {first_tail} launches "
          f"{(step[first_tail - 1][2] - step[0][1]) / 1e6:.3f} ms, tail {len(step) - first_tail} launches "
          f"{(step[-1][2] - step[first_tail][1]) / 1e6:.3f} ms")
    for title, part in (("encoder", step[:first_tail]), ("tail", step[first_tail:])):
        agg = collections.OrderedDict()
        for n, s, e in part:
            short = n.replace("(anonymous namespace)::", "").replace("void ", "")
            short = short.split("(")[0] if not short.startswith("at::") else short[:80]
            a = agg.setdefault(short, [0, 0])
            a[0] += 1
            a[1] += e - s
        print(f"--- {title}")
        for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: args.top]:
            print(f"{a[0]:5d} x {a[1] / a[0] / 1e3:9.1f} us = {a[1] / 1e6:8.3f} ms  {n}")


if __name__ == "__main__":
    main()
