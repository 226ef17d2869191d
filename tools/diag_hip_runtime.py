import ctypes, os, re, sys
mode = sys.argv[1]
def maps():
    m = open('/proc/self/maps').read()
    return sorted(set(re.findall(r'\S*(?:libamdhip64|libhsa-runtime64)\S*', m)))
if mode == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.version.hip)
    print(maps())
lib = ctypes.CDLL("/root/repo/mathematical-modeling-of-infectious-diseases-v1_amd/libsepaihrd_hip.so")
print(maps())
hip = ctypes.CDLL("libamdhip64.so.7") if mode != "torch_first" else ctypes.CDLL(os.path.join(os.path.dirname(__import__('torch').__file__), "lib", "libamdhip64.so"))
n = ctypes.c_int(-1)
rc = hip.hipGetDeviceCount(ctypes.byref(n))
print("hipGetDeviceCount rc", rc, "n", n.value)
v = ctypes.c_int(0); hip.hipRuntimeGetVersion(ctypes.byref(v)); print("runtime version", v.value)
if mode == "mine_first":
    import torch
    print(maps())
    try:
        print("torch avail", torch.cuda.is_available())
        x = torch.zeros(4, device="cuda"); print(x)
    except Exception as e:
        print("torch failed:", e)
