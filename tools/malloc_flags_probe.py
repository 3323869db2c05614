"""bench.py with the weight arenas / the KV region in device memory from hipExtMallocWithFlags (not part of the product): replaces
emmax.engine._device_bytes.  EMMAX_LAB_MALLOC_<ARENA|AUX|KV>=<flag> (1 = fine-grained, 3 = uncached); remaining arguments go to bench.py, e.g.
    EMMAX_LAB_MALLOC_ARENA=3 python tools/malloc_flags_probe.py --steps 2 --warmup 1 --no-cpu-baseline
Result (round 5, profiles/r05_malloc_flags_ab.txt): no difference at batch 1 / 8 -- the streams already bypass the caches as non-temporal loads."""
import ctypes as C
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "emma-x_amd")]
import torch  # noqa: E402

from emmax import engine  # noqa: E402

_hip = C.CDLL("libamdhip64.so")


class LabBuffer:
    """Device bytes with the two members the engine uses of a torch tensor."""

    def __init__(self, nbytes, flag):
        self._n = int(nbytes)
        self._p = C.c_void_p()
        rc = _hip.hipExtMallocWithFlags(C.byref(self._p), C.c_size_t(self._n), C.c_uint(flag))
        if rc != 0 or not self._p.value:
            raise RuntimeError(f"hipExtMallocWithFlags({nbytes}, {flag}) failed: {rc}")

    def data_ptr(self):
        return int(self._p.value)

    def numel(self):
        return self._n

    def __del__(self):
        if self._p.value:
            _hip.hipFree(self._p)


def device_bytes(nbytes, device, what):
    flag = os.environ.get("EMMAX_LAB_MALLOC_" + what)
    if flag:
        return LabBuffer(nbytes, int(flag))
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


engine._device_bytes = device_bytes
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
