"""Does a 16-byte hipMemsetAsync captured into a hipGraph (torch.cuda.graph, ROCm 7.2) do on replay what it does eagerly?
sage_prepare_weights = hipMemsetAsync(trailer, 0, 16) + a kernel that only ORs into the trailer when W holds a huge value."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "graphsage-simple_amd")]
import torch
from sage355 import native
L = native.lib()
dev = "cuda"
w = torch.randn(128, 256, device=dev)
need = L.sage_prepared_weight_bytes(256, 128, 0)
for trial in range(3):
    junk = [torch.full((1 << 20,), 0x5A, dtype=torch.uint8, device=dev) for _ in range(8)]     # dirty the allocator's free list
    del junk
    buf = torch.full((need,), 7, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        native.check(L.sage_prepare_weights(w.data_ptr(), w.stride(0), 256, 128, 0, buf.data_ptr(), need, st.cuda_stream), "prep")
    torch.cuda.synchronize()
    print(f"trial {trial}: eager      trailer", buf[-16:].tolist())
    buf[-16:] = 9
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        native.check(L.sage_prepare_weights(w.data_ptr(), w.stride(0), 256, 128, 0, buf.data_ptr(), need, torch.cuda.current_stream().cuda_stream), "prep")
    torch.cuda.synchronize()
    print(f"trial {trial}: captured   trailer", buf[-16:].tolist(), "(capture does not execute)")
    for r in range(3):
        g.replay()
        torch.cuda.synchronize()
        print(f"trial {trial}: replay {r}   trailer", buf[-16:].tolist())
