"""Does replaying the train step as a HIP graph shorten it?  Captures Trainer.step (seed and Adam step frozen: a timing probe,
not a training mode) and times eager steps against graph replays on the same box.  usage: python tools/graph_probe.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import silent_speech_amd as ss
from silent_speech_amd import _lib as L

cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
L.load()
B, T, K, C = 256, 30, 40, 50
roi = 96 if cfgn == 5 else 64
if cfgn == 5:
    C = bench.C5["classes"]
sp = bench.spec_for(cfgn, B, T, K, roi, C)
X, lengths, R, y = bench.synth_inputs(L, dev, 0, B, T, K, sp["roi_hw"], sp["C"])
torch.manual_seed(0)
model = ss.BiGRUClassifier(sp["D"], sp["C"], use_roi=True, **sp["model_kw"]).to(dev).train()
tr = ss.Trainer(model)

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

eager = lambda: tr.step(X, lengths, R, y)
for _ in range(10): eager()
print("eager   ms/step", [round(timed(eager, 50), 4) for _ in range(3)], flush=True)
cs = torch.cuda.Stream()
cs.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cs):
    for _ in range(3): eager()
torch.cuda.current_stream().wait_stream(cs)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cs):
    eager()
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(10): g.replay()
print("graph   ms/step", [round(timed(g.replay, 50), 4) for _ in range(3)], flush=True)
print("eager   ms/step", [round(timed(eager, 50), 4) for _ in range(3)], flush=True)
print("loss", float(tr.scal[0]))
