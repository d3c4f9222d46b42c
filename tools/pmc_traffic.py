"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately as
MI355X_MICROARCH.md prescribes):  python tools/pmc_traffic.py <dir FETCH_SIZE> <dir WRITE_SIZE>
Counter unit KiB.  gfx950 correction (same guide): FETCH_SIZE reports half of a wide coalesced read stream
(16 B/lane global loads / LDS-DMA) -> fetch_x2; WRITE_SIZE is uncorrected."""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(conv_igemm_kernel<[^>]*>|conv3x3_patch_kernel<[^>]*>|conv3x3_c\d+_regw_kernel<[^>]*>|stem_pool_kernel<[^>]*>|"
                  r"[A-Za-z0-9_]+_kernel\b)", name)
    return m.group(1) if m else re.sub(r"\s+", " ", name)[:60]


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[(short(row["Kernel_Name"]), int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
    return acc


def main():
    fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    print("%-52s %9s %4s %10s %12s %10s" % ("kernel", "grid_thr", "n", "fetch_MiB", "fetch_x2_MiB", "write_MiB"))
    for key in sorted(set(fe) | set(wr)):
        if not (key[0].startswith("conv") or key[0].startswith("stem") or key[0].endswith("_kernel")) or "at::" in key[0]:
            continue
        f = fe.get(key, [0.0])
        w = wr.get(key, [0.0])
        fm, wm = sum(f) / len(f) / 1024.0, sum(w) / len(w) / 1024.0
        if fm + wm < 0.05:
            continue
        print("%-52s %9d %4d %10.1f %12.1f %10.1f" % (key[0][:52], key[1], len(f), fm, 2 * fm, wm))


if __name__ == "__main__":
    main()
