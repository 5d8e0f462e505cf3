import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "none"]
import kbench as kb
for M in (640, 320, 128, 64):
    for Ks in ([64], [256], [512], [512, 512], [512, 512, 512]):
        kb.bench_lstm(M, 512, Ks)
kb.bench_linear(640, 512, 512, label="h2att fwd")
kb.bench_linear(640, 512, 64, label="tiny K")
kb.bench_linear(640, 1536, 2048, label="dX2")
kb.bench_linear(640, 1536, 256, label="dX2 K256")
