import torch, time
n = 185*1024*1024
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
def run(k):
    ss = [torch.cuda.Stream() for _ in range(k)]
    ch = n // k
    torch.cuda.synchronize()
    t = time.perf_counter()
    for it in range(10):
        for i, s in enumerate(ss):
            with torch.cuda.stream(s):
                d[i*ch:(i+1)*ch].copy_(h[i*ch:(i+1)*ch], non_blocking=True)
    torch.cuda.synchronize()
    return n * 10 / (time.perf_counter() - t) / 1e9
for k in (1, 2, 4): print(k, "streams:", round(run(k), 1), "GB/s")
