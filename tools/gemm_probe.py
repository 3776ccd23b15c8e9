import torch, time
dev = torch.device("cuda:0")
def bench(bs, N, K=4096, nl=16):
    ws = [(torch.randn(N, K, device=dev) * 0.1).half() for _ in range(nl)]
    x = (torch.randn(bs, K, device=dev) * 0.1).half()
    outs = [torch.empty(bs, N, device=dev, dtype=torch.half) for _ in range(nl)]
    for w, o in zip(ws, outs): torch.matmul(x, w.t(), out=o)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for w, o in zip(ws, outs): torch.matmul(x, w.t(), out=o)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (20 * nl) * 1e6
for bs in (16, 32, 33, 48, 64, 128):
    print(bs, "qkv %.1f us" % bench(bs, 12288), "oproj %.1f us" % bench(bs, 4096, nl=32))
