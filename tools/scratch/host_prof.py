import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch
import bench
dev = torch.device("cuda", 0)
wl, enc, dt, size = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
step, model, trainer = bench.build_leg(wl, enc, dt, 8, size, 23, dev, 0, 1, False)
for _ in range(5): step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
    torch.cuda.synchronize()     # so that queue back-pressure does not show up as host time
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats(60)
