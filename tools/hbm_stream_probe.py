"""What a pure WRITE stream and a copy reach on this device (round 6: the edge_expand kernel writes 128 B per pixel and reads 4 -- is its
2.9 - 3.7 TB/s the kernel or the memory?).  torch's own fill / copy kernels over buffers far larger than the 256 MB Infinity Cache."""
import torch
for gb in (0.25, 1.0, 2.0):
    n = int(gb * (1 << 30) // 4)
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    for name, fn, bytes_ in (("fill (write only)", lambda: a.fill_(1.5), 4 * n), ("copy (read + write)", lambda: b.copy_(a), 8 * n),
                             ("sum (read only)", lambda: a.sum(), 4 * n), ("mul_ in place (read + write)", lambda: a.mul_(1.0001), 8 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%.2f GiB %-30s %8.1f us  %6.2f TB/s" % (gb, name, 1e3 * ms, bytes_ / ms / 1e9))
    del a, b
