"""Cost of a cross-stream dependency on this platform: chains of small kernels on one stream against the same chains with
every other kernel on a second stream behind event waits (fork / join per pair), and two independent kernels side by side.
The question behind it: would weight gradients on a side stream pay at small batches?"""
import time

import torch

dev = torch.device("cuda:0")
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
N = 200


def run(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / N * 1e6


for n_el, label in ((1 << 12, "16 KB"), (1 << 22, "16 MB"), (1 << 24, "64 MB")):
    a = torch.zeros(n_el, device=dev)
    b = torch.zeros(n_el, device=dev)

    def serial():
        with torch.cuda.stream(s0):
            for _ in range(N):
                a.add_(1.0)
                b.add_(1.0)

    def forked():   # b's kernel on the side stream after a's predecessor; joined before the next pair
        for _ in range(N):
            e = torch.cuda.Event()
            e.record(s0)
            s1.wait_event(e)
            with torch.cuda.stream(s1):
                b.add_(1.0)
            with torch.cuda.stream(s0):
                a.add_(1.0)
            e2 = torch.cuda.Event()
            e2.record(s1)
            s0.wait_event(e2)

    def independent():
        for _ in range(N):
            with torch.cuda.stream(s1):
                b.add_(1.0)
            with torch.cuda.stream(s0):
                a.add_(1.0)

    print(f"{label}: per pair of kernels: one stream {run(serial):.2f} us, fork + join per pair {run(forked):.2f} us, "
          f"two free-running streams {run(independent):.2f} us", flush=True)
