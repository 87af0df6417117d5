"""Prints the kernels of the last timed step of a `rocprofv3 --kernel-trace` run of bench.py in launch order.
Usage: python3 tools/step_trace.py <dir with *_kernel_trace.csv> [kernels per step]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last ckks_tensor_kernel starts the last step
idx = [i for i, n in enumerate(names) if "ckks_tensor_kernel" in n]
a = idx[-1]
end = len(rows)
for i in range(a + 1, len(rows)):
    if "ntt" not in names[i] and "ks_" not in names[i] and "gather" not in names[i] and "tail_" not in names[i] and "copy" not in names[i].lower():
        end = i
        break
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in rows[a:end]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(st - t0) / 1e3:9.1f} us  +gap {(st - prev_end) / 1e3:6.1f}  dur {(en - st) / 1e3:8.1f}  {r['Kernel_Name'].split('(anonymous namespace)::')[-1][:70]}  grid {r.get('Grid_Size', '')}")
    prev_end = en
    tot += en - st
print(f"step span {(prev_end - t0) / 1e3:.1f} us, kernel time {tot / 1e3:.1f} us")
