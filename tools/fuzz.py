"""Random descriptors against NumPy: rank, lengths (61-smooth), batch, layout (packed / batch-interleaved / unpacked rows /
strided), storage, placement, offsets, scales, precision, direction.  usage: fuzz.py [seed] [iterations] [big2d|global|regres|pairs|stages|wide]
With `big2d` the shapes are 2-D / 3-D with a long last dimension (256...2048): the two-pass 2-D plan
(stockham_rows2d.hpp) and its fall-backs.  With `global` the lengths are four-step (GLOBAL tier) sizes: powers of two
2^15 ... 2^21 (the registered stage pairs), 3 / 5 / 6 / 10 times powers of two, powers of ten, lengths with a prime factor
37 ... 61 -- packed, both storages, both placements.  With `regres` the lengths are 31-smooth and lie just beyond one
work-group's LDS (fp32 20481 ... 40000, fp64 10241 ... 20000): the register-resident kernel (stockham_wg_hx.hpp) with
whatever radices and lanes its planner picks, or the four-step plan where it declines.  With `pairs` they are multiples
of 16 between 80 and 152 KiB (fp32 10241 ... 19000, fp64 5121 ... 9500): the same kernel planned as two work-groups per CU, or
the LDS-resident kernel where there is no such plan -- both storages, both placements, offsets, scales.  With `stages` the
lengths are 31-smooth lengths of 600 ... 1100 points (fp64: to 1100 too) in the batch-interleaved and mixed layouts with 16 ... 200
transforms, or four-step lengths one of whose factors lies in that band: the register-resident strided stage kernel
(stockham_strided_hx.hpp: groups that would sit alone on their CU) wherever its planner takes it, ragged passes and partial
groups included, and the XCD-contiguous walk for the unaligned row pitches these lengths have.  With `wide` the lengths are 13-smooth
lengths of 590 ... 2304 points, batch-interleaved (or the long column dimension of a 2-D array): groups of 74 ... 80 KiB on two
register-resident work-groups per CU and, from 1025 points, the WIDE groups -- one register-resident work-group per CU, kernels that
may spill a few registers -- or the two-stage plan where the planner finds none."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H
import gpu_utils as G
import portfft_amd as pf

PRIMES = [2, 2, 2, 2, 2, 2, 3, 3, 3, 5, 5, 7, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61]

def smooth(rng, lo, hi):
    lo = min(lo, hi)
    while True:
        n = 1
        while n < lo:
            n *= rng.choice(PRIMES)
        if n <= hi:
            return n

def expected_supported(prec, dims, layout):
    """Every length the fuzzer draws is 61-smooth.  Documented limits: a dimension
    fits one work-group up to 20480 (fp32) / 10240 (fp64) points; longer lengths run on the GLOBAL tier, which -- like
    the reference's (committed_descriptor_impl.hpp:757-764) -- takes 1-D packed data only, up to (LDS/2 elements)^2 in
    two stages.  Inside these limits `unsupported_configuration` is a FAILURE, not a skip: a length that silently
    became unsupported must not pass."""
    one_wg = 20480 if prec == "f32" else 10240
    if all(d <= one_wg for d in dims):
        return True
    return len(dims) == 1 and layout == "P" and dims[0] <= one_wg * one_wg


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    big2d = len(sys.argv) > 3 and sys.argv[3] == "big2d"
    glob = len(sys.argv) > 3 and sys.argv[3] == "global"
    regres = len(sys.argv) > 3 and sys.argv[3] in ("regres", "pairs")
    pairs = len(sys.argv) > 3 and sys.argv[3] == "pairs"
    wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
    stages = len(sys.argv) > 3 and sys.argv[3] in ("stages", "wide")
    stage_lo, stage_hi = (590, 2304) if wide else (600, 1100)
    rng = random.Random(seed)
    fails = 0
    for it in range(iters):
        prec = rng.choice(["f32", "f32", "f64"])
        dtype = np.complex64 if prec == "f32" else np.complex128
        rank = rng.choice([1, 1, 1, 2, 2, 3])
        if big2d:
            rank = rng.choice([2, 2, 3])
            last = rng.choice([256, 512, 1024, 2048, 1000, 768])
            mid = rng.choice([8, 16, 24, 40, 64, 96, 128, 250, 256, 512, 1024, 1500, 3000, 2560])
            dims = ([rng.choice([2, 3, 5, 8])] if rank == 3 else []) + [mid, last]
        elif stages:
            rank = 1
            while True:
                n = 1
                while n < stage_lo:
                    n *= rng.choice([2, 2, 2, 3, 3, 5, 5, 7, 11, 13])
                if n <= stage_hi:
                    break
            four_step = rng.random() < 0.3 and not wide
            dims = [n * rng.choice([64, 100, 104, 125, 128, 240])] if four_step else [n]
            if wide and rng.random() < 0.25:
                rank, dims = 2, [n, rng.choice([16, 24, 33, 40])]
        elif regres:
            rank = 1
            lo, hi = (20481, 40000) if prec == "f32" else (10241, 20000)
            if pairs:
                lo, hi = (10241, 19000) if prec == "f32" else (5121, 9500)
            while True:
                n = 16 if pairs else 1
                while n < lo:
                    n *= rng.choice([2, 2, 2, 2, 3, 3, 5, 5, 7, 11, 13, 17, 19, 23, 29, 31])
                if n <= hi:
                    break
            dims = [n]
        elif glob:
            rank = 1
            dims = [rng.choice([1 << 15, 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 3 << 15, 5 << 14, 3 << 17,
                                6 << 15, 10 << 13, 100000, 1000000, 60000, 37 << 12, 61 << 10, 9 * 5 * 7 * 11 * 13 * 16, 49152,
                                # k * 2^m: a registered stage B behind a runtime-specialised stage A (plan_global.cpp, half pairs)
                                3 << 13, 5 << 13, 7 << 14, 3 << 16, 5 << 15, 9 << 16, 15 << 15, 11 << 13, 3 << 18,
                                5 << 17, 3 << 20, 12288, 20480, 13 << 12, 25 << 12, 3 << 19, 5 << 18,
                                1 << 23, 3 << 22])]  # (three-stage plan)
        elif rank == 1:
            dims = [smooth(rng, rng.choice([2, 20, 300, 3000]), rng.choice([64, 2000, 20000, 200000]))]
        else:
            dims = [smooth(rng, 2, rng.choice([12, 40, 200])) for _ in range(rank)]
        n = int(np.prod(dims))
        batch = rng.choice([1, 2, 3, 7, 16, 33, 100])
        if stages and n <= stage_hi:
            batch = rng.choice([16, 17, 33, 48, 100, 133, 200])
        if n * batch > 4_000_000:
            batch = max(1, (8_000_000 if (big2d or glob or regres or stages) else 4_000_000) // n)
        storage = rng.choice([0, 0, 1])
        kw = {}
        place = rng.choice([0, 1])
        layout = "P"
        if rank == 1 and not glob and not regres and not (stages and n > stage_hi):
            layout = rng.choice(["BI", "BI", "BI", "PBI", "BIP"] if stages else ["P", "P", "BI", "ROWS", "STR", "PBI", "BIP"])
            if layout == "BI":
                kw = dict(fwd_strides=[batch], fwd_distance=1, bwd_strides=[batch], bwd_distance=1)
            elif layout == "PBI" and place == 1:
                kw = dict(bwd_strides=[batch], bwd_distance=1)
            elif layout == "BIP" and place == 1:
                kw = dict(fwd_strides=[batch], fwd_distance=1)
            elif layout == "ROWS":
                ld = n + rng.choice([1, 3, 16, 64])
                kw = dict(fwd_distance=ld, bwd_distance=ld)
            elif layout == "STR":
                s = rng.choice([2, 3])
                d = n * s + rng.choice([0, 1, 5])
                kw = dict(fwd_strides=[s], fwd_distance=d, bwd_strides=[s], bwd_distance=d)
        fo = rng.choice([0, 0, 3, 17])
        bo = fo if place == 0 else rng.choice([0, 5])
        fs, bs = rng.choice([1.0, 0.5]), rng.choice([1.0, 2.0])
        direction = rng.choice([pf.direction.FORWARD, pf.direction.BACKWARD])
        desc = "%s dims=%s batch=%d storage=%d place=%d layout=%s off=(%d,%d) dir=%s" % (prec, dims, batch, storage, place, layout, fo, bo, direction.name)
        try:
            d = G.make_descriptor(dims, prec, batch=batch, storage=storage, placement=place, fwd_offset=fo, bwd_offset=bo,
                                  fwd_scale=fs, bwd_scale=bs, **kw)
            x, y = H.gen_fourier_data(batch, dims, dtype, seed=it)
            if direction == pf.direction.FORWARD:
                got, _ = G.transform_packed(d, direction, x)
                ref = fs * y.astype(np.complex128)
            else:
                got, _ = G.transform_packed(d, direction, y)
                ref = bs * n * x.astype(np.complex128)
            err = H.rel_l2(np.asarray(got).reshape(batch, -1), ref.reshape(batch, -1))
            tol = (4e-6 if prec == "f32" else 1e-14)
            if not (err < tol):
                fails += 1
                print("FAIL err=%.2e  %s" % (err, desc), flush=True)
            if it % 10 == 9:
                print("... %d done, %d failures" % (it + 1, fails), flush=True)
        except pf.unsupported_configuration as e:
            if expected_supported(prec, dims, layout):
                fails += 1
                print("FAIL unsupported inside the documented limits (%s): %s" % (e, desc), flush=True)
            else:
                print("skip (beyond the documented limits: %s): %s" % (type(e).__name__, desc), flush=True)
        except pf.invalid_configuration as e:
            print("skip (%s): %s" % (type(e).__name__, desc), flush=True)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print("EXC %r  %s" % (e, desc), flush=True)
    print("fuzz seed %d: %d iterations, %d failures" % (seed, iters, fails))
    sys.exit(1 if fails else 0)

main()
