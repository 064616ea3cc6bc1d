import sys, ctypes
sys.path.insert(0, '.')
import torch
print("torch", torch.__version__)
def maps(tag):
    libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l})
    print(tag, libs)
maps("after import torch")
order = sys.argv[1]
from target_estimation_amd import capi
if order == "torch_first":
    print("is_available", torch.cuda.is_available())
    x = torch.zeros(4, device="cuda"); print(x)
    maps("after torch cuda")
l = capi.lib()
maps("after my lib")
hip = ctypes.CDLL("libamdhip64.so.7")
cnt = ctypes.c_int(-1)
rc = hip.hipGetDeviceCount(ctypes.byref(cnt)); print("hipGetDeviceCount rc", rc, "count", cnt.value)
m = l.target_manager_new(b"models/model_uniform_velocity_params.yaml"); print("handle", m)
