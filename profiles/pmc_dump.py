"""Per-kernel averages of every counter in a rocprofv3 --pmc result database (rocpd SQLite).
Usage: python profiles/pmc_dump.py gpurun_out/pmc_x/x_results.db [kernel-substring]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1]).cursor()
    filt = sys.argv[2] if len(sys.argv) > 2 else "k_"
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
    ker = {}
    for name, ctr, n, v, dur in rows:
        if filt not in name:
            continue
        ker.setdefault(name, {"_calls": n, "_avg_us": dur / 1000.0})[ctr] = v
    for name, d in sorted(ker.items(), key=lambda kv: -kv[1]["_avg_us"]):
        print("%s  calls=%d avg_us=%.1f" % (name[:110], d["_calls"], d["_avg_us"]))
        for k in sorted(d):
            if not k.startswith("_"):
                print("    %-28s %18.1f" % (k, d[k]))


if __name__ == "__main__":
    main()
