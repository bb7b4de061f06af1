import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import bench
dev = torch.device("cuda", 0)
wl, enc, dt, size = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
step, model, trainer = bench.build_leg(wl, enc, dt, 8, size, 23, dev, 0, 1, False)
for _ in range(5): step()
torch.cuda.synchronize()
def timeit(f, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / n
print("eager ms/step", timeit(step), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
print("capturing", flush=True)
with torch.cuda.graph(g, stream=s):
    loss = step()
print("captured", flush=True)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("replayed once, loss", float(loss), flush=True)
print("graph ms/step", timeit(g.replay), flush=True)
