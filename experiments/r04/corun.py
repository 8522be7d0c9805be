"""What does a kernel that SITS on the CUs like the layer-1 contraction (512 threads, 168 VGPRs, 117 KB of LDS, one block per CU on 224 CUs) take from a
kernel that READS like the layer-1 gather?  The gather stand-in alone, then beside a holder that does ONE kind of work (experiments/r04/corun.hip).
    python experiments/r04/corun.py"""
import ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
dev = torch.device("cuda", 0)
torch.cuda.init()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "corun.so"))
lib.launch_gather.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.launch_holder.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
table = torch.randn(1 << 28, device=dev)                    # 1 GiB = 2^23 granules of 128 B
granules = (1 << 28) // 32
units = 94_000                                              # x 16 granules x 128 B = 192 MB per launch (the real gather moves 195 MB past L2)
out = torch.empty(units * 32, device=dev)
src = torch.randn(1 << 26, device=dev)                      # 256 MiB for the streaming holder
sink = torch.zeros(1024, device=dev)
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
NAMES = {0: "nothing (s_sleep): footprint only", 1: "ds_read_b128 at full rate", 2: "v_mfma_f32_32x32x16_bf16 back to back", 3: "streaming global loads",
         4: "plain VALU (v_fma)", 5: "LDS reads + MFMA interleaved"}


def gather_ms(n=20, blocks=256 * 6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    with torch.cuda.stream(sA):
        ev[0].record()
        for i in range(n):
            assert lib.launch_gather(table.data_ptr(), granules, out.data_ptr(), units, blocks, sA.cuda_stream) == 0
            ev[i + 1].record()
    sA.synchronize()
    d = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))
    return d[len(d) // 2], d[0]


def holder_iters(mode, target_ms=4.0, blocks=224):
    """Calibrate the holder to run ~target_ms alone."""
    it = 2000
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sB):
            a.record(); lib.launch_holder(mode, src.data_ptr(), src.numel(), sink.data_ptr(), it, blocks, sB.cuda_stream); b.record()
        sB.synchronize()
        ms = a.elapsed_time(b)
        if 0.7 * target_ms < ms < 1.4 * target_ms:
            break
        it = max(50, int(it * target_ms / max(ms, 1e-3)))
    return it, ms


torch.cuda.synchronize()
gather_ms(5)
med, mn = gather_ms()
print(f"gather stand-in alone (6 blocks per CU): median {med:.1f} us, min {mn:.1f} us  = {units * 16 * 128 / med / 1e6:.2f} TB/s", flush=True)
med2, _ = gather_ms(blocks=256 * 2)
print(f"gather stand-in alone (2 blocks per CU): median {med2:.1f} us", flush=True)
for blocks in (224, 128):
    for mode in range(6):
        it, ms = holder_iters(mode, blocks=blocks)
        with torch.cuda.stream(sB):
            lib.launch_holder(mode, src.data_ptr(), src.numel(), sink.data_ptr(), it, blocks, sB.cuda_stream)
        time.sleep(0.0005)                                   # the holder's blocks are resident before the gathers arrive
        m, mn = gather_ms(n=20)
        torch.cuda.synchronize()
        print(f"beside a holder on {blocks} CUs doing {NAMES[mode]:42s}: gather median {m:6.1f} us (min {mn:5.1f}); the holder alone ran {ms:.2f} ms for {it} rounds", flush=True)
