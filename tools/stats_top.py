"""Print the top rows of a rocprofv3 kernel_stats.csv, per batch.  usage: stats_top.py <csv> <batches> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nb = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per batch %.3f, launches per batch %.0f" % (tot / 1e6 / nb, sum(int(r["Calls"]) for r in rows) / nb))
for r in rows[:top]:
    print("%8.3f ms/batch  x%6.1f  %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 1e6 / nb, int(r["Calls"]) / nb, float(r["Percentage"]), r["Name"][:120]))
