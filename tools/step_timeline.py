"""Timeline of ONE training step out of a rocprofv3 kernel trace (rocpd .db): every kernel's start / end relative to the
step's first kernel, its queue, and the union of busy time — to see what the step's wall time is made of when its kernels
run on several streams (sum of durations > step time) and how much of it no kernel covers at all (launch gaps).
usage: python tools/step_timeline.py <trace dir> <out.txt> [anchor kernel substring = field_fwd_taped] [which occurrence = -3 |
       a kernel substring: the last whole step that contains it, e.g. density_bwd_train = a step on which the proposal networks update]"""
import glob
import os
import sqlite3
import sys


def main():
    d, dst = sys.argv[1], sys.argv[2]
    anchor = sys.argv[3] if len(sys.argv) > 3 else "field_fwd_taped"
    which = sys.argv[4] if len(sys.argv) > 4 else "-3"
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    sel = "name, start, end" + (", " + qcol if qcol else ", 0")
    rows = sorted(cur.execute(f"select {sel} from kernels"), key=lambda r: r[1])
    starts = [i for i, r in enumerate(rows) if anchor in r[0]]
    if which.lstrip("-").isdigit():
        which = int(which)
    else:  # the last whole step holding a kernel of that name
        hit = [k for k in range(len(starts) - 1) if any(which in r[0] for r in rows[starts[k]:starts[k + 1]])]
        if not hit:
            raise SystemExit(f"no step contains {which!r}")
        which = hit[-1] - len(starts)
    if len(starts) < abs(which) + 1:
        raise SystemExit(f"anchor {anchor!r} found {len(starts)} times")
    a, b = starts[which], starts[which + 1] if which + 1 < 0 or which + 1 < len(starts) else len(rows)
    step = rows[a:b]
    # a step starts a few launches before the field forward (camera optimizer, proposal pass): rotate so that the window holds
    # one whole period anchor -> anchor
    t0 = step[0][1]
    with open(dst, "w") as out:
        out.write(f"# one training step (kernels between two consecutive {anchor} launches), times in us relative to the first\n")
        out.write(f"# source: {os.path.basename(db)}; columns: start, end, duration, {qcol or 'queue'}, kernel\n")
        busy, last_end, covered = [], None, 0.0
        for name, s, e, q in step:
            out.write(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {name[:110]}\n")
            busy.append((s, e))
        busy.sort()
        cs, ce = busy[0]
        gaps = []
        for s, e in busy[1:]:
            if s > ce:
                covered += ce - cs
                gaps.append((s - ce, ce - t0))
                cs, ce = s, e
            else:
                ce = max(ce, e)
        covered += ce - cs
        period = rows[b][1] - t0 if b < len(rows) else ce - t0
        out.write(f"# period (anchor to anchor) {period / 1e3:.1f} us; covered by at least one kernel {covered / 1e3:.1f} us; "
                  f"sum of durations {sum(e - s for _, s, e, _ in step) / 1e3:.1f} us; {len(step)} launches; "
                  f"{len(gaps)} gaps totalling {sum(g for g, _ in gaps) / 1e3:.1f} us "
                  f"(largest: {', '.join('%.1f@%.0f' % (g / 1e3, at / 1e3) for g, at in sorted(gaps, reverse=True)[:8])})\n")
    print(open(dst).read())


if __name__ == "__main__":
    main()
