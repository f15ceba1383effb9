import os, sys
sys.path.insert(0, "/root/repo")
import portfft_amd as pf
for prec, dims in (("f32", [1000, 1000]), ("f32", [768, 768]), ("f32", [1200, 1200]), ("f32", [1536, 1536]), ("f32", [4096, 4096]), ("f64", [1000, 1000])):
    d = pf.descriptor(dims, prec); d.number_of_transforms = 4
    info = d.commit().info()
    print(prec, dims, [[info.dims[i].factors[k] for k in range(info.dims[i].n_factors)] for i in range(2)], [info.dims[i].ffts_per_workgroup for i in range(2)], [info.dims[i].workgroup_size for i in range(2)], flush=True)
