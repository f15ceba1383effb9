import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gpu_utils as G
import sys
shapes = [[int(v) for v in a.split('x')] for a in sys.argv[1:]] or [[3000,1000],[1500,1000],[6000,500],[2500,1200],[3000,768]]
for dims in shapes:
    p = G.make_descriptor(dims, "f32", batch=4).commit()
    i = p.info()
    print(dims, [ (i.dims[k].tier, list(i.dims[k].factors)[:i.dims[k].n_factors]) for k in range(2)], flush=True)
