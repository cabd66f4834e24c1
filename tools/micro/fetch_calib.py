"""FETCH_SIZE / WRITE_SIZE counted by rocprofv3 against the known byte counts of tools/micro/fetch_calib.hip: python fetch_calib.py <fetch_dir> <write_dir>"""
import csv, glob, sys
def load(d, name):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].split("(")[0]
                out[k] = out.get(k, 0.0) + float(r["Counter_Value"])
    return out
F, W = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
G = 4 << 30
known = {"k_read16": ("read", G, "16 B per lane, contiguous"), "k_read4": ("read", G, "4 B per lane, contiguous"), "k_read1": ("read", G // 4, "1 B per lane, contiguous"),
         "k_gather4": ("read", (G // 128) * 4, "4 B per lane, every access in its own 128-B line"), "k_write16": ("write", G, "16 B per lane, contiguous"),
         "k_write4": ("write", G, "4 B per lane, contiguous"), "k_scatter4": ("write", (G // 128) * 4, "4 B per lane, every access in its own 128-B line")}
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB counters) over a 4 GiB buffer touched once, MI355X")
print("kernel, pattern, known bytes, counted bytes, counted / known, counted per access")
for k, (kind, b, what) in known.items():
    c = (F if kind == "read" else W).get(k, 0.0) * 1024
    per = c / (b / (4 if "4 B" in what else 16 if "16 B" in what else 1))
    print(f"{k}, {what}, {b}, {c:.0f}, {c / b:.3f}, {per:.1f} B")
