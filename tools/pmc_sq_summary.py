"""Per-kernel SQ counter digest from two rocprofv3 --pmc passes (see profiles/r01_j_sq_counters.txt for the commands).
python tools/pmc_sq_summary.py <dir pass A> <dir pass B> <kernel duration csv dir (any pass)>"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"([A-Za-z0-9_]+_kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


def load(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k is None or "at::" in row["Kernel_Name"]:
                continue
            acc[(k, int(row["Grid_Size"]))][row["Counter_Name"]].append(float(row["Counter_Value"]))
            acc[(k, int(row["Grid_Size"]))]["_dur"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    return acc


def mean(v):
    return sum(v) / len(v) if v else 0.0


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    simds, cus = 1024, 256
    print("%-50s %8s %7s | %6s %6s %6s %6s | %6s %6s" % ("kernel (grid threads)", "us", "calls", "mfma%", "issue%", "parked%", "stall%",
                                                        "lds%", "confl%"))
    for key in sorted(a, key=lambda k: -mean(a[k]["_dur"]) * len(a[k]["_dur"])):
        ca, cb = a[key], b.get(key, {})
        dur_ns = mean(ca["_dur"])
        if dur_ns < 3000:
            continue
        wave = mean(ca["SQ_WAVE_CYCLES"]) or 1.0
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; kernel cycles ~ dur * clk, estimate clk from BUSY is unreliable -> use 2.0 GHz
        cyc = dur_ns * 2.0
        mfma = mean(ca["SQ_VALU_MFMA_BUSY_CYCLES"]) / simds / cyc * 100
        lds = mean(cb.get("SQ_LDS_IDX_ACTIVE", [0])) / cus / cyc * 100
        confl = 100 * mean(cb.get("SQ_LDS_BANK_CONFLICT", [0])) / max(mean(cb.get("SQ_LDS_IDX_ACTIVE", [1])), 1)
        print("%-50s %8.1f %7d | %6.1f %6.1f %6.1f %6.1f | %6.1f %6.1f" % (
            (key[0] + " (%d)" % key[1])[:50], dur_ns / 1000, len(ca["_dur"]), mfma, 100 * mean(ca["SQ_ACTIVE_INST_ANY"]) / wave,
            100 * mean(ca["SQ_WAIT_ANY"]) / wave, 100 * mean(ca["SQ_WAIT_INST_ANY"]) / wave, lds, confl))


if __name__ == "__main__":
    main()
