"""One step of a traced bench.py as a timeline: reads rocprofv3's kernel trace (csv) and prints, for the LAST complete step, every kernel
between the end of the previous gpfq_blk_kernel and the end of this one -- start, end, duration in microseconds, and the stream (queue) it ran on.
usage: step_timeline.py <t_kernel_trace.csv> [main kernel substring = gpfq_blk_kernel<] [index of the launch that ends the step = -2]"""
import csv
import sys


def main():
    path = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else "gpfq_blk_kernel<"
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    mains = [i for i, r in enumerate(rows) if key in r[2]]
    if len(mains) < 3:
        print("fewer than three launches of", key)
        return
    k = int(sys.argv[3]) if len(sys.argv) > 3 else -2           # (bench.py: warm-up steps, the timed steps, then the same steps with the medians prefetched)
    a, b = mains[k - 1], mains[k]
    t0 = rows[a][1]
    print(f"microseconds from the end of the previous step's kernel; queue = HIP stream\n{'start':>9} {'end':>9} {'dur':>8}  queue  kernel")
    for s, e, name, q in rows[a + 1:b + 1]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {q:>5}  {name[:110]}")


if __name__ == "__main__":
    main()
