"""gpurun_out/prof_<tag>/ -> profiles/<prefix>_*: copies the rocprofv3 summaries worth keeping and writes the
per-variant HBM byte counts (FETCH_SIZE x 2 on gfx950, WRITE_SIZE; MI355X_MICROARCH.md §HBM) as json.
usage: python tools/summarize_profiles.py <tag> <prefix>      e.g.  r01b r01b"""
import collections
import csv
import json
import os
import shutil
import sys

tag, prefix = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles", exist_ok=True)


def find(suffix, start):
    for root, _, files in os.walk(src):
        for f in files:
            if f.startswith(start) and f.endswith(suffix):
                return os.path.join(root, f)
    raise FileNotFoundError((start, suffix))


shutil.copy(find("kernel_stats.csv", tag), f"profiles/{prefix}_bench_kernel_stats.csv")
shutil.copy(os.path.join(src, "bench_line.json"), f"profiles/{prefix}_bench_line.json")
shutil.copy(os.path.join(src, "bench_line_under_rocprof.json"), f"profiles/{prefix}_bench_line_under_rocprof.json")
VARIANTS = 300000
PASSES = None   # passes over the cohort in that command: counted from the launches below (22 shards per pass)


def per_kernel(path, counter):
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[k] += float(r["Counter_Value"])
        n[k] += 1
    return acc, n


f, nf = per_kernel(find("counter_collection.csv", "f_"), "FETCH_SIZE")
w, _ = per_kernel(find("counter_collection.csv", "w_"), "WRITE_SIZE")
# one k_parse_fixed launch per shard and pass: bench.py runs a warm serial pass (since round 3c), the serial profiling pass
# and the timed step
PASSES = max(nf.get("k_parse_fixed", 0) // 22, 1)
out = {
    "_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_profiles.sh) on `bench.py --variants "
               f"{VARIANTS} --steps 1 --warmup 0 --no-overlap` = {PASSES} passes over {VARIANTS} variants x 2504 samples",
    "_correction": "FETCH_SIZE is in KB and counts 64 B per 128-B request for wide coalesced reads on gfx950 "
                   "(MI355X_MICROARCH.md §HBM): fetch bytes = FETCH_SIZE*1024*2; WRITE_SIZE*1024 is exact",
    "variants_per_pass": VARIANTS,
}
# the "lz4" stage is two launches per shard: the bit-plane coder, then the byte-wise coder over the streams it marked
for name, kerns in (("lz4", ("k_lz4_bitplanes", "k_lz4_bitplanes_uni", "k_lz4_blocks")), ("encode", ("k_encode_tiles", "k_encode_planes")), ("index", ("k_index_newlines", "k_index_hop")),
                    ("frame", ("k_frame_write",)), ("fixed", ("k_parse_fixed",))):
    kerns = [k for k in kerns if k in f]
    if not kerns:
        continue
    fb = sum(f[k] for k in kerns) * 1024 * 2 / (VARIANTS * PASSES)
    wb = sum(w[k] for k in kerns) * 1024 / (VARIANTS * PASSES)
    out[name] = {"kernel": "+".join(kerns), "fetch_bytes_per_variant": fb, "write_bytes_per_variant": wb,
                 "hbm_bytes_per_variant": fb + wb, "launches": {k: nf[k] for k in kerns}}
json.dump(out, open(f"profiles/{prefix}_pmc_summary.json", "w"), indent=1)
shutil.copy(find("counter_collection.csv", "sq_"), f"profiles/{prefix}_pmc_lz4_sq.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
grid = collections.Counter()
for r in csv.DictReader(open(f"profiles/{prefix}_pmc_lz4_sq.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")   # with the template arguments: <depth, planes input, exception-aware>
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        grid[k] += int(r["Grid_Size"])
# one 4 KiB plane stream per wave in k_lz4_bitplanes (128 lanes = one 8 KiB block = two planes)
lines = []
for k in acc:
    waves = acc[k].get("SQ_WAVES", 0) or 1
    lines.append(f"{k}: per wave " + json.dumps({c: round(v / waves, 1) for c, v in sorted(acc[k].items())}))
open(f"profiles/{prefix}_pmc_lz4_sq.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
for f_ in os.listdir(src):
    if f_.startswith("e2e_") and f_.endswith(".json") and os.path.getsize(os.path.join(src, f_)):
        shutil.copy(os.path.join(src, f_), f"profiles/{prefix}_{f_}")
# round 3: SQ counters of the bit-plane encode kernel, the config-4 leg and its kernel stats
try:
    shutil.copy(find("counter_collection.csv", "sqe_"), f"profiles/{prefix}_pmc_encode_planes_sq.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f"profiles/{prefix}_pmc_encode_planes_sq.csv")):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    with open(f"profiles/{prefix}_pmc_encode_planes_sq.txt", "w") as fo:
        for k in acc:
            waves = acc[k].get("SQ_WAVES", 0) or 1
            fo.write(f"{k}: per wave " + json.dumps({c: round(v / waves, 1) for c, v in sorted(acc[k].items())}) + "\n")
    shutil.copy(find("kernel_stats.csv", "c4"), f"profiles/{prefix}_c4_kernel_stats.csv")
    shutil.copy(os.path.join(src, "c4_line.json"), f"profiles/{prefix}_c4_line.json")
except FileNotFoundError as e:
    print("round-3 extras missing:", e)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
