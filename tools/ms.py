"""Print ms_per_step (and kernel ms) of a bench.py JSON line read from stdin; argv[1:] = a label."""
import json
import sys
d = json.loads(sys.stdin.readline())
print(" ".join(sys.argv[1:]), "ms/step", round(d["ms_per_step"], 3), "kernel_ms", round(d["roofline"]["kernel_ms_avg"], 3),
      "value", "%.4g" % d["value"])
