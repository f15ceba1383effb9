"""Worker of tests/test_distributed.py (launched with torch.distributed.run, gloo, CPU).

Every rank transforms its shard of a seeded global batch (with the oracle standing in for the GPU kernel: there is
no GPU here), then the ranks exchange only scalars: timing max and per-shard checksums."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from portfft_amd.sharding import process_group, shard_range  # noqa: E402
import oracle_binding as ob  # noqa: E402


def main():
    n, total = 256, 37  # ragged on purpose: 37 transforms over 2 ranks
    pg = process_group("gloo")
    lo, hi = shard_range(total, pg.world, pg.rank)
    rng = np.random.Generator(np.random.SFC64(0))
    x = (rng.uniform(-1, 1, (total, n)) + 1j * rng.uniform(-1, 1, (total, n))).astype(np.complex64)
    mine = np.ascontiguousarray(x[lo:hi])
    pg.barrier()
    t0 = time.perf_counter()
    y = ob.compute(ob.make_desc([n], "f32", batch=hi - lo), ob.FORWARD, mine.ravel()).reshape(hi - lo, n)
    elapsed = time.perf_counter() - t0
    pg.barrier()
    worst = pg.max(elapsed)
    table = pg.gather([lo, hi, float(np.abs(y).astype(np.float64).sum()), elapsed])
    if pg.rank == 0:
        ref = np.fft.fft(x.astype(np.complex128), axis=1)
        expected = [float(np.abs(ref[int(r[0]):int(r[1])]).sum()) for r in table]
        print(json.dumps({"world": pg.world, "table": table, "expected": expected, "max_elapsed": worst}))
    pg.close()


if __name__ == "__main__":
    main()
