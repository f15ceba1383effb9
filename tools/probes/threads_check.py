"""several host threads committing (runtime compilation included) and executing plans concurrently on their own streams"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import portfft_amd as pf
errors = []
def work(tid):
    try:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for n in ([4096], [1200 + 40 * tid], [60, 60 + tid], [1 << 16], [3000]):
                d = pf.descriptor(n, "f32"); d.number_of_transforms = 64
                plan = d.commit()
                tot = int(np.prod(n)) * 64
                x = torch.randn(tot, dtype=torch.complex64, device="cuda"); y = torch.empty_like(x)
                for _ in range(20):
                    plan.compute_forward(x, y)
                s.synchronize()
                ref = torch.fft.fftn(x.view(64, *n), dim=tuple(range(1, len(n) + 1))).reshape(-1)
                err = float((y - ref).norm() / ref.norm())
                if not err < 2e-6:
                    errors.append((tid, n, err))
    except Exception as e:  # noqa: BLE001
        errors.append((tid, repr(e)))
threads = [threading.Thread(target=work, args=(i,)) for i in range(6)]
[t.start() for t in threads]; [t.join() for t in threads]
print("errors:", errors); print("OK" if not errors else "FAIL"); sys.exit(1 if errors else 0)
