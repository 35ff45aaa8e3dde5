"""Buffers for the device-resident entry points (vpt_predict_batch_device, vpt_fill_tags_batch_device): torch CUDA
tensors on the GPU box; plain numpy arrays when the kernels run on the CPU emulator (tests/test_kernel_emu.py), whose
"device pointers" are host pointers."""
import numpy as np

EMULATED = False   # set by tests/test_kernel_emu.py for the duration of that module


class Buf:
    def __init__(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        if EMULATED:
            self._a = arr.copy()
            self.ptr = self._a.ctypes.data
        else:
            import torch
            signed = {np.dtype(np.uint64): np.int64, np.dtype(np.uint32): np.int32}.get(arr.dtype)
            self._t = torch.from_numpy(arr.view(signed) if signed else arr).cuda()
            self._dtype = arr.dtype
            self.ptr = self._t.data_ptr()

    def set(self, arr: np.ndarray) -> None:
        """Overwrites the buffer in place (same address: what a caller reusing its device buffers does)."""
        arr = np.ascontiguousarray(arr)
        if EMULATED:
            self._a[:len(arr)] = arr
        else:
            import torch
            signed = {np.dtype(np.uint64): np.int64, np.dtype(np.uint32): np.int32}.get(arr.dtype)
            self._t[:len(arr)].copy_(torch.from_numpy(arr.view(signed) if signed else arr))
            torch.cuda.synchronize()

    def get(self, n=None) -> np.ndarray:
        if EMULATED:
            return self._a[:n].copy()
        return self._t[:n].cpu().numpy().view(self._dtype)


def put(arr) -> Buf:
    return Buf(np.asarray(arr))


def zeros(n: int, dtype) -> Buf:
    return Buf(np.zeros(n, dtype))


def stream() -> int:
    if EMULATED:
        return 0
    import torch
    return torch.cuda.current_stream().cuda_stream
