import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import bench
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
scan = bench.load_scan()
x_init, xs, tv = bench.make_inputs(pipe, scan, 10, 1000, dev)
with torch.no_grad():
    for rep in range(3):
        ts = []
        for j in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            bench.run_steps(pipe, x_init, xs, tv, j, 1)
            torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        print("pass", rep, " ".join(f"{t:.1f}" for t in ts), "sum", round(sum(ts), 1), flush=True)
