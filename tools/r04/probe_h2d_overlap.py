"""Does a pinned host -> device copy on its own stream overlap with kernels on another stream on this box?"""
import os, time, torch
dev = torch.device("cuda")
a = torch.randn(8192, 8192, device=dev)
h = torch.empty(64, 3, 224, 224).pin_memory()            # 38.5 MB
copy = torch.cuda.Stream()
def work(n):
    x = a
    for _ in range(n):
        x = x @ a * 1e-4
    return x
def run(with_copy, n=20, copies=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if with_copy != "copy_only": work(n)
    if with_copy in ("both", "copy_only"):
        with torch.cuda.stream(copy):
            for _ in range(copies): h.to(dev, non_blocking=True)
    torch.cuda.synchronize(); return time.perf_counter() - t0
for _ in range(2): run("both")
print("env HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
for mode in ("kernels_only", "copy_only", "both", "kernels_only", "both"):
    print("%-13s %.1f ms" % (mode, 1e3 * run(mode)))
gb = 40 * h.numel() * 4 / 1e9
print("copied %.2f GB per run -> copy_only rate %.1f GB/s" % (gb, gb / run("copy_only")))
