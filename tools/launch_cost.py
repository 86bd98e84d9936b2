"""host cost of N back-to-back step launches on one stream (is the enqueue blocked by the GPU?)"""
import sys, os, time, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_mmt_amd.engine import Dims, Engine
from variational_mmt_amd import _lib as L
d = Dims(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True, dropout=0.0, conditional=True)
eng = Engine(d, dtype="bf16", device="cuda", seed=0)
ws = eng.workspace(256, 20, 20)
plan = ws.plan_fwd_train
chain = [(fn, args) for fn, args, name, keep, sid in plan if name == "vmmt_lstm_chain_fwd"][0]
fn, args = chain
st = torch.cuda.Stream()
torch.cuda.synchronize()
for n in (16, 64, 256):
    a = list(args); a[2] = n
    t0 = time.perf_counter(); fn(*a, st.cuda_stream); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("n=%d host %.1f us (%.2f us/launch), until done %.1f us" % (n, (t1 - t0) * 1e6, (t1 - t0) * 1e6 / n, (t2 - t0) * 1e6))
ws.backward_plan(1.0 / 256, 256.0, 1.0, False, 0.0, False)
fnb, argsb = [(fn, args) for fn, args, name, keep, sid in ws.plan_bwd if name == "vmmt_lstm_chain_bwd"][0]
for n in (64, 256):
    a = list(argsb); a[2] = n
    t0 = time.perf_counter(); fnb(*a, st.cuda_stream); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("bwd chain n=%d host %.1f us (%.2f us/launch), until done %.1f us" % (n, (t1 - t0) * 1e6, (t1 - t0) * 1e6 / n, (t2 - t0) * 1e6))
# same, behind ~3 ms of queued GPU work: does the enqueue wait for the GPU (kernarg pool / queue back-pressure)?
A = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
for n in (64, 128, 256, 512):
    with torch.cuda.stream(st):
        for _ in range(6): A @ A
    a = list(args); a[2] = min(n, 256)
    t0 = time.perf_counter()
    for _ in range(max(1, n // 256)): fn(*a, st.cuda_stream)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("busy stream: n=%d host %.1f us (%.2f us/launch), until done %.1f us" % (n, (t1 - t0) * 1e6, (t1 - t0) * 1e6 / n, (t2 - t0) * 1e6))
