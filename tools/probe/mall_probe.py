#!/usr/bin/env python3
"""Does the 256 MB memory-side cache (MALL / Infinity Cache) serve data that a previous kernel has just written or read?
Times a streaming read (torch sum) of an S-byte buffer (a) right after a kernel wrote it, (b) right after a kernel read
it, (c) after 2 GiB of other traffic.  One line per size.   python tools/probe/mall_probe.py"""
import torch

def t(fn, n=5):
    best = 1e9
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        fn(a, b)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best

def main():
    dev = "cuda"
    big = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev)   # 2 GiB
    big.fill_(1.0)
    for mb in (32, 64, 128, 192, 256, 384, 768):
        n = mb * 1024 * 1024 // 4
        x = torch.empty(n, dtype=torch.float32, device=dev)
        def after_write(a, b):
            big.add_(1.0); x.fill_(2.0); a.record(); x.sum(); b.record()
        def after_read(a, b):
            big.add_(1.0); x.sum(); a.record(); x.sum(); b.record()
        def cold(a, b):
            x.fill_(2.0); big.add_(1.0); a.record(); x.sum(); b.record()
        r = [t(f) for f in (after_write, after_read, cold)]
        print("%4d MB: read after write %.4f ms (%.2f TB/s)  after read %.4f ms (%.2f TB/s)  cold %.4f ms (%.2f TB/s)" % (
            mb, r[0], mb * 1.048576e-3 / r[0], r[1], mb * 1.048576e-3 / r[1], r[2], mb * 1.048576e-3 / r[2]), flush=True)

if __name__ == "__main__":
    main()
