import ctypes as C, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "stream_probe.so")
lib = C.CDLL(so)
lib.stream_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(4, dtype=torch.int32, device=dev)
for mb in (33, 295):
    nb = mb * 1000 * 1000 // 1024 * 1024
    nbuf = max(2, (1 << 30) // nb + 1)
    bufs = [torch.randint(0, 255, (nb,), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    for mode in (0, 1, 2):
        for u in (2, 4, 8):
            for wgs in (512, 1024, 2048):
                st = torch.cuda.current_stream().cuda_stream
                for b in bufs: lib.stream_launch(mode, u, b.data_ptr(), nb, sink.data_ptr(), wgs, st)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for b in bufs: lib.stream_launch(mode, u, b.data_ptr(), nb, sink.data_ptr(), wgs, st)
                e1.record(); torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / 1e3 / nbuf
                print(f"MB={mb} mode={mode} U={u} wgs={wgs}: {t*1e6:.1f} us  {nb/t/1e12:.2f} TB/s", flush=True)
    del bufs
