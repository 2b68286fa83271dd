#!/usr/bin/env python3
"""Does plain streaming bandwidth depend on WHICH memory an allocation got? K pairs of 2 GiB buffers alive at once, each pair's
device copy timed alone (interleaved repetitions). Companion of tools/alloc_probe.py."""
import torch

K, N = 12, 1 << 31
pairs = [(torch.empty(N, dtype=torch.uint8, device="cuda"), torch.empty(N, dtype=torch.uint8, device="cuda")) for _ in range(K)]
for s, d in pairs:
    s.zero_()
    d.zero_()
res = [[] for _ in pairs]
for rep in range(3):
    for i, (s, d) in enumerate(pairs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d.copy_(s)
        e0.record()
        for _ in range(5):
            d.copy_(s)
        e1.record()
        torch.cuda.synchronize()
        res[i].append(2.0 * N * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
for (s, d), r in zip(pairs, res):
    print(f"{s.data_ptr():#x} -> {d.data_ptr():#x}: " + " ".join(f"{v:7.1f}" for v in r) + " GB/s")
