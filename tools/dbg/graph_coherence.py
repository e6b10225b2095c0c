"""does a hipGraph replay see (a) data written by the previous node on another XCD, (b) data written by an eager kernel before it?"""
import torch
dev = torch.device('cuda:0')
N = 1 << 22
x = torch.zeros(N, device=dev); y = torch.zeros(N, device=dev); z = torch.zeros(N, device=dev)
s = torch.cuda.Stream()
def body():
    x.add_(1.0)            # producer
    y.copy_(x.flip(0))     # consumer reads elements produced by other blocks (other XCDs)
    z.copy_(y * 2.0)
for _ in range(3): body()
torch.cuda.synchronize()
x.zero_(); y.zero_(); z.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
bad = 0
for i in range(1, 3001):
    g.replay()
    if i % 100 == 0:
        torch.cuda.synchronize()
        b = int((y != float(i)).sum()) + int((z != 2.0 * i).sum())
        bad += b
print('intra-graph producer/consumer mismatches over 3000 replays:', bad)
# eager writer -> graph reader
w = torch.zeros(N, device=dev); out = torch.zeros(N, device=dev)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    out.copy_(w.flip(0) + 1.0)
bad = 0
for i in range(1, 3001):
    w.fill_(float(i))      # eager
    g2.replay()
    if i % 100 == 0:
        torch.cuda.synchronize()
        bad += int((out != float(i) + 1.0).sum())
print('eager-writer / graph-reader mismatches over 3000 replays:', bad)
# graph on a side stream with wait_stream
bad = 0
for i in range(1, 3001):
    w.fill_(float(i))
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g2.replay()
    torch.cuda.current_stream().wait_stream(s)
    if i % 100 == 0:
        torch.cuda.synchronize()
        bad += int((out != float(i) + 1.0).sum())
print('same, graph replayed on a side stream:', bad)
