import json,sys
for f in sys.argv[1:]:
    try:
        l=[x for x in open(f) if x.startswith("{")]
        d=json.loads(l[0]); r=d["roofline"]
        print(f.split("/")[-1], "pts/s %.3e ms/scan %.4f R %d lat %.3f | knn launch %.1f us frac %.3f | %s" % (d["value"], d["ms_per_step"], d["repeats"], d["config"]["single_stream_latency_ms_per_scan"], r["avg_launch_us"], r["frac"], json.dumps(r["other_kernels_us"])))
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-600:])
