import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unpaired_image_captioning_amd import _lib as L
lib = L.load()
def timeit(fn, iters=60):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(d[iters // 8: iters - iters // 8]) / (iters - 2 * (iters // 8)) * 1e3
g = torch.Generator(device="cuda").manual_seed(3)
print("%-28s %9s %9s %9s %9s %9s" % ("shape M x N x K", "default", "128x128", "pp 256", "pp 192", "pp 128"))
for M, N, K in [(1920, 2048, 512), (2048, 2048, 512), (2048, 2048, 1024), (1920, 512, 2048), (2048, 512, 2048), (1920, 512, 512), (2048, 512, 512), (2048, 1024, 2048),
                (640, 512, 2048), (640, 2048, 512), (1920, 1024, 512), (1280, 2048, 512), (960, 2048, 512)]:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    B = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ts = []
    for force in (0, 0x100, 0x200, 0x400, 0x800):
        try:
            ts.append(timeit(lambda: L.check(lib.uic_linear(1, M, N, K, L.ptr(A), K, L.ptr(B), K, L.ptr(C), N, None, 4 | force, L.stream()))))
        except Exception as e:
            ts.append(float("nan"))
    print("%-28s %9.1f %9.1f %9.1f %9.1f %9.1f" % ("%d x %d x %d" % (M, N, K), *ts))
