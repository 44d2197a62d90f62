"""Feasibility probe (round 5): CU-masked HIP streams.  Can a stream be confined to k CUs (hipExtStreamCreateWithCUMask), what
does an HBM-streaming kernel reach on 8 / 16 / 32 CUs, and does it run CONCURRENTLY with work on the complementary mask?"""
import ctypes as C, sys, time
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value)


dev = torch.device("cuda:0")
torch.cuda.init()
print("CUs:", torch.cuda.get_device_properties(0).multi_processor_count)
x = torch.empty(1 << 30, device=dev, dtype=torch.float32).normal_()      # 4 GB
y = torch.empty_like(x)
A = torch.randn(8192, 8192, device=dev)
Bm = torch.randn(8192, 8192, device=dev)


def time_copy(stream, reps=3):
    with torch.cuda.stream(stream):
        y.copy_(x)
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(reps):
            y.copy_(x)
        e1.record()
    e1.synchronize()
    return 8 * x.numel() * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


def time_mm(stream, reps=5):
    with torch.cuda.stream(stream):
        torch.mm(A, Bm)
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(reps):
            torch.mm(A, Bm)
        e1.record()
    e1.synchronize()
    return 2 * 8192 ** 3 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


full = torch.cuda.Stream()
print(f"all CUs: copy {time_copy(full):.2f} TB/s, fp32 mm {time_mm(full):.1f} TFLOP/s")
for k in (8, 16, 32, 64):
    s = masked_stream(range(256 - k, 256))
    print(f"{k:3d} CUs (bits {256 - k}..255): copy {time_copy(s):.2f} TB/s, fp32 mm {time_mm(s):.1f} TFLOP/s")
for k in (16, 32):
    big, small = masked_stream(range(0, 256 - k)), masked_stream(range(256 - k, 256))
    mm_alone = time_mm(big)
    # concurrent: mm on the big mask, copies on the small one
    torch.cuda.synchronize()
    e = [torch.cuda.Event(True) for _ in range(4)]
    with torch.cuda.stream(big):
        e[0].record()
        for _ in range(20):
            torch.mm(A, Bm)
        e[1].record()
    with torch.cuda.stream(small):
        e[2].record()
        for _ in range(3):
            y.copy_(x)
        e[3].record()
    torch.cuda.synchronize()
    t_mm, t_cp = e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])
    print(f"{256 - k} + {k} CUs: mm alone {mm_alone:.1f} TFLOP/s; together: mm {2 * 8192 ** 3 * 20 / t_mm / 1e9:.1f} TFLOP/s over {t_mm:.1f} ms, "
          f"copy {8 * x.numel() * 3 / t_cp / 1e9:.2f} TB/s over {t_cp:.1f} ms (overlap if copy window lies inside mm window: "
          f"start skew {e[0].elapsed_time(e[2]):.1f} ms)")
