import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--host-emulation", action="store_true", default=False,
                     help="TEST INFRASTRUCTURE: run the `-m gpu` tests against oracle/_hiphost/libhiphost.so (the product kernel sources compiled for "
                          "the host on wave64 fibers, oracle/build_hip_host.sh) with CPU tensors instead of an MI355X.  Slow; pick tests with -k. "
                          "Kernels that need MFMA (prefill GEMM / prefill attention), RCCL or the C++ runner are not in that library.")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")
    if config.getoption("--host-emulation"):
        _enter_host_emulation()


def _enter_host_emulation():
    """Point the ctypes loader of the package at the host-emulation library and give torch a null 'current stream'.  Lives here, not in the
    package: the product loader (mistralrs_amd/_lib.py) has no CPU path and keeps failing loudly when its HIP libraries are missing."""
    import ctypes as C
    import torch
    from tests.abi_backends import HostBackend
    import mistralrs_amd  # noqa: F401
    from mistralrs_amd import _lib
    HostBackend.global_symbols = True
    lib = HostBackend.lib()
    for key in ("quant", "paged_attn", "core", "ext"):
        _lib._cache[key] = lib

    class _NullStream:
        cuda_stream = 0

        def synchronize(self):
            pass

    _from_numpy = torch.from_numpy
    torch.from_numpy = lambda a: _from_numpy(a).clone()  # `.to(cpu)` does not copy: keep the tests' numpy inputs out of reach of in-place kernels
    torch.Tensor.is_cuda = property(lambda self: True)  # the wrappers' "must live on the GPU" guards: here the emulated device memory IS host memory
    torch.cuda.current_stream = lambda *a, **k: _NullStream()
    torch.cuda.synchronize = lambda *a, **k: None


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure) -- built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def dev(request):
    import torch
    if request.config.getoption("--host-emulation"):
        return torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
