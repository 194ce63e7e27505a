"""one line of key numbers from a bench.py JSON file: python tools/abline.py <tag> <file>"""
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    d = [json.loads(l) for l in open(path) if l.startswith("{")][-1]
    r = d.get("roofline", {})
    print("%-10s crops/s %8.1f  ms/step %7.3f | hot convs %7.1f TF frac %.4f avg_launch_ms %.4f pass_ms %s" % (
        tag, d["value"], d["ms_per_step"], r.get("achieved", 0), r.get("frac", 0), r.get("avg_launch_ms", 0), r.get("pass_ms")))
except Exception as e:
    print(tag, "FAILED", e)
