"""Throughput of step-wise decoding at the BASELINE model shape (V 30k, biLSTM 512, z 256): arg-max decoding and beam search
(beam 5) for a batch of sentences, sentences/s and tokens/s.  GPU box only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd.engine import Dims, Engine
from variational_mmt_amd.decode import beam_decode, greedy_decode
d = Dims(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=0.0)
e = Engine(d, dtype="bf16", device="cuda", seed=0)
g = torch.Generator().manual_seed(0)
S, L = 20, 24
for B, K in ((1, 1), (64, 1), (256, 1), (1, 5), (16, 5), (64, 5)):
    src = torch.randint(2, d.vs, (S, B), generator=g)
    sl = torch.full((B,), S, dtype=torch.int64)
    fn = (lambda: greedy_decode(e, src, sl, max_len=L)) if K == 1 else (lambda: beam_decode(e, src, sl, K, max_len=L))
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("batch %4d beam %d: %7.2f ms per batch of %d positions = %8.0f sentences/s, %9.0f target positions/s" % (B, K, dt * 1e3, L, B / dt, B * L / dt))
