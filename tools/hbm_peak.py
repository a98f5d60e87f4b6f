import torch, time
n = 8 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.random_(0, 255)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
c = t(lambda: b.copy_(a))
print(f"device copy 8 GiB: {c*1e3:.2f} ms = {2*n/c/1e12:.2f} TB/s (read+write)")
ai = a.view(torch.int64)
s = t(lambda: ai.sum())
print(f"read-only reduction (int64 sum) 8 GiB: {s*1e3:.2f} ms = {n/s/1e12:.2f} TB/s")
z = t(lambda: b.zero_())
print(f"fill 8 GiB: {z*1e3:.2f} ms = {n/z/1e12:.2f} TB/s (write)")
