"""Per-call table of a rocprofv3 kernel_stats csv: python tools/stats_table.py <kernel_stats.csv> <calls of the workload> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
reps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 26
print("kernel total per call: %.1f us" % (sum(int(r["TotalDurationNs"]) for r in rows) / reps / 1000))
for r in rows[:top]:
    print("%-58s launches/call %7.1f  avg %8.1f us  per call %8.1f us" % (r["Name"].split("(")[0].replace("void ", "")[:58], int(r["Calls"]) / reps,
                                                                           float(r["AverageNs"]) / 1000, int(r["TotalDurationNs"]) / reps / 1000))
