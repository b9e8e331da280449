"""Per-step view of a profiles/*_kernel_trace_train_*.txt summary.  usage: python tools/trace_table.py <file> [steps=42] [rows=40]"""
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 42
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = [r for r in csv.reader(l for l in open(path) if not l.startswith("#") and l.strip()) if len(r) >= 5 and r[0] != "kernel"]
tot = sum(float(r[2]) for r in rows)
print(f"{path}: {tot / steps:.1f} us of kernels per step, {sum(int(r[1]) for r in rows) / steps:.1f} launches per step")
for r in rows[:nrows]:
    print(f"{float(r[2]) / steps:8.1f} us/step  x{int(r[1]) / steps:5.2f}  avg {float(r[3]):8.1f}  {r[0][:100]}")
