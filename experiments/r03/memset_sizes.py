"""ROCm 7.2: hipMemsetAsync captured into a hipGraph -- for which sizes / alignments do replays 2, 3, ... write something else than
the value?  Raw hipMemsetAsync through ctypes on the capturing stream of torch.cuda.graph; the buffer is dirtied before every replay."""
import ctypes, sys
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = "cuda"
torch.zeros(1, device=dev)
for fn, name, unit in ((hip.hipMemsetAsync, "hipMemsetAsync", 1), (hip.hipMemsetD32Async, "hipMemsetD32Async", 4)):
    for size in (4, 8, 16, 32, 64, 128, 256, 1024, 4096, 65536, 1 << 20):
        for off in (0, 16):
            buf = torch.full((size + 64 + off,), 7, dtype=torch.uint8, device=dev)
            view = buf[off:off + size]
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                rc = fn(view.data_ptr(), 0, size // unit, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            bad = []
            for r in range(4):
                buf.fill_(9)
                torch.cuda.synchronize()
                g.replay()
                torch.cuda.synchronize()
                ok = bool((view == 0).all()) and bool((buf[off + size:] == 9).all()) and bool((buf[:off] == 9).all())
                if not ok:
                    bad.append((r, int((view != 0).sum()), view[:16].tolist()))
            print(f"{name:18s} {size:8d} B at +{off:2d}: " + ("every replay zeroes it" if not bad else f"WRONG on replays {[b[0] for b in bad]}: {bad[0][1]} bytes non-zero, first 16 = {bad[0][2]}"))
