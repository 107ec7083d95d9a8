"""Do parallel branches of a captured HIP graph run concurrently?  Two independent long streaming kernels on forked streams."""
import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "stream_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "stream_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.stream_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(4, dtype=torch.int32, device=dev)
nb = 256 * 1024 * 1024
a = torch.randint(0, 255, (nb,), dtype=torch.uint8, device=dev)
b = torch.randint(0, 255, (nb,), dtype=torch.uint8, device=dev)
def t_graph(parallel, wgs):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    s1.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        with torch.cuda.graph(g, stream=s1):
            if parallel:
                s2.wait_stream(s1)
                lib.stream_launch(2, 4, a.data_ptr(), nb, sink.data_ptr(), wgs, s1.cuda_stream)
                with torch.cuda.stream(s2):
                    lib.stream_launch(2, 4, b.data_ptr(), nb, sink.data_ptr(), wgs, s2.cuda_stream)
                s1.wait_stream(s2)
            else:
                lib.stream_launch(2, 4, a.data_ptr(), nb, sink.data_ptr(), wgs, s1.cuda_stream)
                lib.stream_launch(2, 4, b.data_ptr(), nb, sink.data_ptr(), wgs, s1.cuda_stream)
    torch.cuda.current_stream().wait_stream(s1)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for wgs in (64, 256, 1024):
    print(f"wgs={wgs}: serial {t_graph(False, wgs)*1e3:.1f} us, parallel-branch {t_graph(True, wgs)*1e3:.1f} us", flush=True)
