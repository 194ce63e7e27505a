import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for ln in sys.stdin:
    if ln.startswith("{"):
        d = json.loads(ln)
        r = d.get("roofline", {})
        print(tag, "crops/s", d["value"], "ms/step", d["ms_per_step"], "net TF", d["whole_net_tflops"],
              "| hot convs TF", r.get("achieved"), "frac", r.get("frac"), "pass_ms", r.get("pass_ms"))
