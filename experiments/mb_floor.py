"""Fixed cost of a kernel inside a captured graph: run under rocprofv3 --kernel-trace and read the durations."""
import sys, os, ctypes
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "mb_floor.so"))
dev = "cuda"
nxt = torch.randperm(1024, device=dev, dtype=torch.int32)
out = torch.zeros(4, dtype=torch.int32, device=dev)
big = torch.empty(12 << 20, dtype=torch.uint8, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def chain():
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        lib.run_floor(0, 1, 64, 0, P(out), 0, None, st)                   # empty, 1 wave
        lib.run_floor(0, 256, 512, 0, P(out), 0, None, st)                # empty, 256 x 512
        lib.run_floor(0, 2048, 256, 0, P(out), 0, None, st)               # empty, 2048 x 256
        lib.run_floor(1, 256, 512, 117 * 1024, P(out), 0, None, st)       # 117 KB LDS blocks
        lib.run_floor(2, 256, 256, 0, P(nxt), 1, P(out), st)              # 1 round trip
        lib.run_floor(2, 255, 256, 0, P(nxt), 4, P(out), st)              # 4 round trips
        lib.run_floor(2, 254, 256, 0, P(nxt), 8, P(out), st)              # 8 round trips
        lib.run_floor(3, 3072, 256, 0, P(big), (12 << 20) // 16, None, st)  # 12 MB written
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): chain()
    for _ in range(20): g.replay()
torch.cuda.synchronize()
if os.environ.get("FLOOR_STREAM"):
    s2 = torch.cuda.Stream()
    lib.run_chain_stream(40, P(out), P(nxt), P(big), ctypes.c_void_p(s2.cuda_stream))
    torch.cuda.synchronize()
print("done")
