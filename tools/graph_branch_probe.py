"""How does the HIP graph executor run two independent branches of a captured graph?  (HISTORY.md section 6)

Chains of dummy kernels with known durations are captured with different creation orders / stream roles and the replay
time is compared with the serial sum and the ideal overlap.  python tools/graph_branch_probe.py"""
import time

import torch

dev = torch.device("cuda:0")
# latency-bound kernels (a few workgroups each, long dependent loops), like the step's own: two of them CAN share the GPU
big = (torch.randn(64, 16384, device=dev), torch.randn(16384, 64, device=dev), torch.zeros(64, 64, device=dev))
small = (torch.randn(32, 2048, device=dev), torch.randn(2048, 32, device=dev), torch.zeros(32, 32, device=dev))


def chain(t, n):
    for _ in range(n):
        torch.mm(t[0], t[1], out=t[2])


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def capture(body):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()                                  # warm-up
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        body()
    return g


side1, side2 = torch.cuda.Stream(), torch.cuda.Stream()
NA, NB = 16, 24


def pattern(order, a_stream, b_stream, pre=3):
    """pre small kernels; fork; chain A (big) on a_stream and chain B (small) on b_stream, created in `order`; join; 2 small"""
    def body():
        main = torch.cuda.current_stream()
        chain(small, pre)
        ev = torch.cuda.Event()
        ev.record(main)

        def run(which):
            st = {"main": main, "s1": side1, "s2": side2}[a_stream if which == "A" else b_stream]
            if st is not main:
                st.wait_event(ev)
            with torch.cuda.stream(st):
                chain(big if which == "A" else small2, NA if which == "A" else NB)

        for w in order:
            run(w)
        for st in (side1, side2):
            main.wait_stream(st)
        chain(small, 2)
    return body


small2 = (torch.randn(32, 2048, device=dev), torch.randn(2048, 32, device=dev), torch.zeros(32, 32, device=dev))
ta = timed(lambda: chain(big, NA))
tb = timed(lambda: chain(small2, NB))
print(f"eager: chain A {ta:.0f} us ({ta / NA:.1f}/kernel), chain B {tb:.0f} us ({tb / NB:.1f}/kernel)")
ga, gb = capture(lambda: chain(big, NA)), capture(lambda: chain(small2, NB))
print(f"graph: chain A {timed(ga.replay):.0f} us, chain B {timed(gb.replay):.0f} us   (ideal overlap = max, serial = sum)")
for order in ("AB", "BA"):
    for a_st, b_st in (("main", "s1"), ("s1", "main"), ("s1", "s2")):
        g = capture(pattern(order, a_st, b_st))
        print(f"created {order}: A on {a_st:4s} B on {b_st:4s} -> replay {timed(g.replay):7.0f} us")
