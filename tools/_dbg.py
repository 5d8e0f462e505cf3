import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unpaired_image_captioning_amd import _lib as L
lib = L.load()
torch.manual_seed(0)
for (Mfull, N, K) in ((36, 40, 80), (36, 64, 128), (36, 40, 96)):
    A = torch.randn(Mfull, K, device="cuda") * 0.01
    B = torch.randn(N, K, device="cuda") * 0.1
    def run(r0, M):
        Cc = torch.zeros(M, N, device="cuda")
        a = A[r0:r0 + M].contiguous() if False else A[r0:r0 + M]
        L.check(lib.uic_linear(0, M, N, K, a.data_ptr(), K, L.ptr(B), K, L.ptr(Cc), N, None, 4, L.stream()))
        torch.cuda.synchronize()
        return Cc
    full = run(0, Mfull)
    for (r0, M) in ((0, 24), (6, 18), (6, 6), (12, 12), (1, 35), (8, 18), (4, 18), (6, 30)):
        c = run(r0, M)
        d = (c - full[r0:r0 + M]).abs().amax(1)
        print((Mfull, N, K), "r0", r0, "M", M, "rows differing:", [i for i in range(M) if d[i] > 0])
