import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def g(*ks):
    x=d
    for k in ks:
        x=x.get(k) if isinstance(x,dict) else None
        if x is None: return None
    return x
print(sys.argv[1], "value %.3fM ms_per_step %.4f kernel %.4f frac %.4f | build_each %s | stop_rule %s ms | scene_b %s | single %s | e2e %s" % (d["value"]/1e6, d["ms_per_step"], g("roofline","kernel_ms") or -1, g("roofline","frac") or -1, g("with_build_each_step","ms_per_step"), g("reference_stop_rule","ms_per_step"), g("scene_b","ms_per_step"), g("single_scan","kernel_ms"), g("e2e","update_batch_ms")))
