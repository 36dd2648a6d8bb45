import importlib, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
PKG="lins---lidar-inertial-slam_amd"
pkg=importlib.import_module(PKG); host=importlib.import_module(PKG+".host"); ieskf=importlib.import_module(PKG+".ieskf")
import test_gpu_edge_cases as T
raws=[T._wide_room_raw_scan(5), T._wide_room_raw_scan(6,60.0,2.0)]
want=[host.frontend_segment(r) for r in raws]
with ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024) as c:
    got=c.segment_batch(raws); feats=c.extract_features_batch(got)
for f,w in zip(feats,want):
    ref=host.frontend_extract_segmented(w)
    for k in ("corner_sharp","corner_less_sharp","surf_flat","surf_less_flat"):
        a,b=f[k],ref[k]
        if a.shape!=b.shape: print(k,"shape",a.shape,b.shape); continue
        d=np.where((a!=b).any(axis=1))[0]
        print(k,len(a),"rows differ:",len(d), d[:10])
        if len(d):
            print(a[d[:4]]); print(b[d[:4]])
