"""Prints the kernel timeline of the last LM step found in a rocprofv3 kernel trace
(python3 tools/lm_timeline.py <kernel_trace.csv>): name, start offset and duration in us."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at its (last) prep_cameras/landmark kernel
starts = [i for i, r in enumerate(rows) if "landmark_" in r["Kernel_Name"]]
first = starts[-1]
while first > 0 and ("prep_cameras" in rows[first - 1]["Kernel_Name"]):
    first -= 1
t0 = int(rows[first]["Start_Timestamp"])
prev_end = t0
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  +%6.1f gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:60]))
    prev_end = e
print("total %.1f us" % ((prev_end - t0) / 1e3))
