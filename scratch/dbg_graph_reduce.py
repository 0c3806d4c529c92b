import torch
dev = "cuda"
n = 131072
x = torch.randn(64, 2048, device=dev)
big = torch.randn(n, 1024, device=dev, dtype=torch.bfloat16)
def body():
    a = x.mean(); b = x.std(); c = torch.nn.functional.mse_loss(x, torch.zeros_like(x))
    d = big.sum(0, dtype=torch.float32)
    e = big.float().pow(2).mean()
    return torch.stack([a, b, c, e]), d
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    st, d = body()
for trial in range(4):
    x.normal_(); big.normal_()
    g.replay(); torch.cuda.synchronize()
    st2, d2 = body(); torch.cuda.synchronize()
    print(trial, "graph", st.tolist(), "eager", st2.tolist(), "d err", float((d - d2).abs().max()))
