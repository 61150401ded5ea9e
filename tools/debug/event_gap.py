import torch, time
dev = torch.device("cuda:0")
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(dev)
x = torch.randn(8192, 8192, device=dev)
small = torch.zeros(64, device=dev)

def gap(label, prep):
    res = []
    for _ in range(20):
        torch.cuda.synchronize()
        prep()
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(main)
        t = torch.full((1,), 7, dtype=torch.int64, device=dev)
        b.record(main)
        u = torch.zeros(1, dtype=torch.int32, device=dev)
        c.record(main)
        torch.cuda.synchronize()
        res.append((a.elapsed_time(b), b.elapsed_time(c)))
    r = torch.tensor(res)
    print(f"{label:<50} event->fill->event {r[:,0].median()*1e3:8.1f} us   second {r[:,1].median()*1e3:8.1f} us")

gap("idle GPU", lambda: None)
def big():
    with torch.cuda.stream(side):
        for _ in range(3): y = x @ x
gap("big GEMMs running on a side stream", big)
def many():
    with torch.cuda.stream(side):
        for _ in range(300): small.add_(1)
gap("300 tiny kernels queued on a side stream", many)
def main_busy():
    for _ in range(2): y = x @ x
gap("behind 2 big GEMMs on main itself", main_busy)
def main_busy_side_wait():
    e = torch.cuda.Event()
    for _ in range(2): y = x @ x
    e.record(main)
    side.wait_event(e)
    with torch.cuda.stream(side):
        for _ in range(100): small.add_(1)
gap("behind GEMMs on main, side waits main then 100 tiny", main_busy_side_wait)
