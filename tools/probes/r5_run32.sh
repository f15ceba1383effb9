# round 5: which plan do the lengths of the dropped tuned entries end up with (pair, or handed back)?
mkdir -p gpurun_out/r5_run32
python3 - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r5_run32/plans.txt
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import gpu_utils as G
for prec, ns in (("f32", [10935, 11000, 11200, 11664, 12000, 12150, 12800, 12960, 13500, 13824, 14000, 14336, 14400, 15000, 15120, 15360, 10500, 13125, 16200]),
                 ("f64", [5400, 5760, 5832, 6000, 6400, 6480, 6750, 7200, 7680, 7000, 7168, 6656])):
    for n in ns:
        d = G.make_descriptor([n], prec, batch=4).commit().info().dims[0]
        print(prec, n, "factors", [int(d.factors[i]) for i in range(d.n_factors)], "lanes", d.workgroup_size, "lds", d.lds_bytes, "PAIR" if d.lds_bytes <= 80 * 1024 else "lds-resident", flush=True)
PY
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "register_resident" 2>&1 | tail -4 ) | tee gpurun_out/r5_run32/pytest_sel.txt
