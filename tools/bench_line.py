"""print the essentials of bench.py's JSON line(s): python tools/bench_line.py tag file [tag file ...]"""
import json, sys
for tag, f in zip(sys.argv[1::2], sys.argv[2::2]):
    l = [x for x in open(f) if x.startswith("{")]
    if not l:
        print(tag, "no json line"); continue
    d = json.loads(l[-1])
    print(tag, "%.2f Mvar/s" % (d["value"] / 1e6), "%.1f ms" % d["ms_per_step"],
          {k: round(v, 1) for k, v in d["stages_ms_per_step"].items()}, "ratio %.3f" % d["config"]["compression_ratio"])
