"""Turn a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) into a small text summary that
can be committed under profiles/.   python tools/rocprof_summary.py <results.db> [--forward]"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"([A-Za-z0-9_]+_kernel(?:<[^>]*>)?)", name)
    if m:
        return m.group(1)
    return re.sub(r"\s+", " ", name)[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats   (durations in microseconds; source %s)" % sys.argv[1].split("/")[-1])
    print("%-44s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows:
        if pct < 0.05:
            continue
        print("%-44s %8d %12.1f %10.2f %6.2f%%" % (short(name), calls, tot, avg, pct))
    if "--forward" in sys.argv:
        ks = list(c.execute("select name, start, duration, grid_x, grid_y, lds_size, vgpr_count, accum_vgpr_count "
                            "from kernels order by start"))
        idx = [i for i, r in enumerate(ks) if "stem_" in r[0]]
        if len(idx) >= 5:
            s, e = idx[-5], idx[-4]
            print("\n# one forward (launch order), a step inside the timed region")
            tot = 0.0
            t0 = ks[s][1]
            for r in ks[s:e]:
                print("%-44s @%7.1f %7.1f us  grid=(%d,%d) lds=%d vgpr=%d agpr=%d" % (
                    short(r[0]), (r[1] - t0) / 1000.0, r[2] / 1000.0, r[3] // 256, r[4], r[5], r[6], r[7]))
                tot += r[2]
            print("# sum of kernel durations %.1f us, first-start to next-forward-start %.1f us" % (
                tot / 1000.0, (ks[e][1] - ks[s][1]) / 1000.0))
            # the two trunks run as two concurrent launch chains: the conv family's chip time is the union of its intervals
            conv = sorted((r[1], r[1] + r[2]) for r in ks[s:e] if re.search(r"conv3x3|conv_igemm|splitk_finish", r[0]))
            if conv:
                busy, lo, hi = 0, conv[0][0], conv[0][1]
                for a, b in conv[1:]:
                    if a > hi:
                        busy += hi - lo
                        lo, hi = a, b
                    else:
                        hi = max(hi, b)
                busy += hi - lo
                print("# conv family: %d launches, sum of durations %.1f us, union of intervals (chip time) %.1f us" % (
                    len(conv), sum(b - a for a, b in conv) / 1000.0, busy / 1000.0))


if __name__ == "__main__":
    main()
