import torch, time
dev = torch.device('cuda', 0)
x = torch.zeros(1 << 16, device=dev); y = torch.zeros(1 << 16, device=dev)
s0 = torch.cuda.Stream(); aux = [torch.cuda.Stream() for _ in range(3)]
def run(mode, reps=200):
    torch.cuda.synchronize()
    with torch.cuda.stream(s0):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s0)
        for _ in range(reps):
            x.add_(1.0)
            if mode:
                ev = torch.cuda.Event(); ev.record(s0)
                for s in aux[:mode]:
                    s.wait_event(ev)
                    with torch.cuda.stream(s):
                        y.add_(1.0)
                    e2 = torch.cuda.Event(); e2.record(s); s0.wait_event(e2)
            x.add_(1.0)
        b.record(s0)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000
for mode in (0, 1, 3, 0, 1, 3):
    print("aux streams", mode, "%.1f us per iteration" % run(mode), flush=True)
