import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "stream_probe.so"))
lib.stream_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(8 + 4096 * 64, dtype=torch.int32, device=dev)
import sys
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for mb, wgs in ((1, 256), (9, 1024), (33, 1024)):
    nb = mb * 1024 * 1024
    nbuf = 64 if mb < 20 else 32
    bufs = [torch.randint(0, 255, (nb,), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    def run(st):
        for b in bufs: lib.stream_launch(MODE, 4, b.data_ptr(), nb, sink.data_ptr(), wgs, st)
    run(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    te = e0.elapsed_time(e1) * 1e3 / nbuf
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    tg = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        tg = min(tg, e0.elapsed_time(e1) * 1e3 / nbuf)
    print(f"{mb} MiB x{nbuf}: eager {te:.2f} us/kernel, graph {tg:.2f} us/kernel", flush=True)
