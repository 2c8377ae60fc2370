import sys, json
for l in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    try: r = json.loads(l)
    except Exception: continue
    print("%-52s %8.3f ms %8.1f TF/s %8.1f GB/s" % (r["kernel"], r["ms"], r.get("TFLOP/s",0), r.get("GB/s",0)))
