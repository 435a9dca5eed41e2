"""Stream priority range of the device, and an ExternalStream of the lowest priority (probe for model._side_stream)."""
import ctypes
import torch

hip = ctypes.CDLL("libamdhip64.so")
torch.cuda.init()
lo, hi = ctypes.c_int(), ctypes.c_int()
print("rc", hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), "least", lo.value, "greatest", hi.value)
s = ctypes.c_void_p()
print("create rc", hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, lo.value), hex(s.value or 0))   # 1 = non-blocking
ext = torch.cuda.ExternalStream(s.value)
with torch.cuda.stream(ext):
    x = torch.ones(1024, device="cuda") * 2
ext.synchronize()
print(x.sum().item(), torch.cuda.Stream(priority=-1).priority, torch.cuda.current_stream().priority)
