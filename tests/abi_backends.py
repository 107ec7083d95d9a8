"""Two ways to call the C-ABI launchers of libmistralrsquant.so from a test body:
  GpuBackend  -- the product library on cuda:0 (torch only owns the device buffers);
  HostBackend -- the SAME kernel sources compiled for the host on wave64 fibers (oracle/build_hip_host.sh -> oracle/_hiphost, test
                 infrastructure): buffers are numpy arrays, `stream` is NULL.  Lets a launcher be exercised before GPU time is spent on it.
A test body takes a backend `be` and uses be.sym / be.buf / be.stream only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _HostBuf:
    def __init__(self, a):
        self.a = np.ascontiguousarray(a).copy()
        self.ptr = self.a.ctypes.data

    def numpy(self):
        return self.a

    def fill(self, v):
        self.a[...] = v

    def write(self, a):
        """in-place upload: same address, new contents"""
        self.a[...] = np.asarray(a).reshape(self.a.shape)


class HostBackend:
    name = "host-emulation"
    stream = None
    _lib = None
    global_symbols = False  # `pytest --host-emulation` sets this: the C++ runner resolves launchers with dlsym(RTLD_DEFAULT)

    @classmethod
    def lib(cls):
        if cls._lib is None:
            so = os.environ.get("MRS_HIPHOST_SO")  # a build of oracle/build_hip_host.sh elsewhere (HIPHOST_OUT=...): used as it is, never rebuilt
            if so:
                cls._lib = C.CDLL(so, mode=C.RTLD_GLOBAL if cls.global_symbols else C.RTLD_LOCAL)
                return cls._lib
            so = os.path.join(ROOT, "oracle", "_hiphost", "libhiphost.so")
            srcs = [os.path.join(ROOT, "mistral.rs_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "mistral.rs_amd", "csrc")) if f.endswith((".cuh", ".hip"))]
            srcs += [os.path.join(ROOT, "oracle", "hip_host", "hip", "hip_runtime.h"), os.path.join(ROOT, "oracle", "build_hip_host.sh")]
            if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
                subprocess.check_call(["sh", os.path.join(ROOT, "oracle", "build_hip_host.sh")], stdout=subprocess.DEVNULL)
            cls._lib = C.CDLL(so, mode=C.RTLD_GLOBAL if cls.global_symbols else C.RTLD_LOCAL)
        return cls._lib

    def sym(self, name, argtypes, restype=None):
        f = getattr(self.lib(), name)
        f.argtypes, f.restype = argtypes, restype
        return f

    def buf(self, a, dtype=None):
        """dtype: 'f16' / 'bf16' store an f32 array in that 16-bit format (bf16 as raw uint16 bits)."""
        if dtype == "f16":
            return _HostBuf(np.asarray(a, dtype=np.float32).astype(np.float16))
        if dtype == "bf16":
            from oracle import oracle as O
            b = _HostBuf(O.to_bf16_bits(np.asarray(a, dtype=np.float32)))
            b.numpy = lambda b=b: O.from_bf16_bits(b.a)
            return b
        return _HostBuf(a)


class _GpuBuf:
    def __init__(self, t):
        self.t = t
        self.ptr = t.data_ptr()

    def numpy(self):
        return self.t.float().cpu().numpy() if self.t.is_floating_point() else self.t.cpu().numpy()

    def fill(self, v):
        self.t.fill_(v)

    def write(self, a):
        """in-place upload: same address, new contents"""
        import torch
        self.t.copy_(torch.from_numpy(np.ascontiguousarray(a)).reshape(self.t.shape).to(self.t.dtype))


class GpuBackend:
    name = "gpu"

    def __init__(self, dev):
        self.dev = dev

    @property
    def stream(self):
        import torch
        return torch.cuda.current_stream().cuda_stream

    def sym(self, name, argtypes, restype=None):
        from mistralrs_amd import _lib
        # the launchers live in four libraries (quant / paged_attn / core / ext): resolve the symbol in whichever exports it
        for key in ("quant", "ext", "paged_attn", "core"):
            if hasattr(_lib.load(key), name):
                return _lib.sym(key, name, argtypes, restype)
        raise AttributeError(f"{name}: exported by none of {list(_lib.NAMES.values())}")

    def buf(self, a, dtype=None):
        import torch
        from tests.util import torch_dtype
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint32:
            a = a.view(np.int32)
        t = torch.from_numpy(a.copy()).to(self.dev)
        if dtype in ("f16", "bf16"):
            t = t.to(torch_dtype(dtype))
        return _GpuBuf(t.contiguous())
