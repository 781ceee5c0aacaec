"""Dev: how late does the host wake up from a blocking HIP wait on this box?  A fixed-length GPU busy kernel, then four ways of
waiting for it; the spread of (wall time - kernel time) is the wake-up delay.  (where do the 50-190 ms
steps of a loop that synchronises every step come from?)"""
import sys, time, ctypes
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000_000
x = torch.zeros(1, device="cuda")
torch.cuda._sleep(cycles); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(cycles); e1.record(); torch.cuda.synchronize()
kern_ms = e0.elapsed_time(e1)
print(f"busy kernel: {kern_ms:.2f} ms")

def run(name, wait):
    d = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.cuda._sleep(cycles)
        ev = torch.cuda.Event(); ev.record()
        wait(ev)
        d.append(1e3 * (time.perf_counter() - t0) - kern_ms)
    d.sort()
    print(f"{name:28s}: wake-up delay ms  median {d[len(d)//2]:7.3f}  p90 {d[int(.9*len(d))]:7.3f}  max {d[-1]:7.3f}   >5ms: {sum(v > 5 for v in d)}/{n}")

def poll(ev):
    while not ev.query():
        pass
def poll_sleep(ev):
    while not ev.query():
        time.sleep(2e-4)
run("event.synchronize()", lambda ev: ev.synchronize())
run("torch.cuda.synchronize()", lambda ev: torch.cuda.synchronize())
run(".item() of a device scalar", lambda ev: x.item())
run("event.query() spin", poll)
run("event.query() + sleep(0.2ms)", poll_sleep)
run("event.synchronize()", lambda ev: ev.synchronize())
