"""bench.py on a 300 k-variant workload, one single-stream step, no legs — the command the --pmc passes profile
(tools/pmc_kernel.sh <tag> <kernel regex> bench_small.py [more bench.py flags])"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(ROOT, "bench.py"), "--variants", "300000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-overlap",
            "--no-legs", "--no-check"] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
