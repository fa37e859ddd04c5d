import sys, time, torch
sys.path.insert(0, '/root/repo')
import silent_speech_amd as ss
dev = torch.device('cuda')
B, T = 256, 30
m = ss.BiGRUClassifier(84, 5, use_roi=True).to(dev).train()
X = torch.randn(B, T, 84, device=dev); R = torch.randint(0, 256, (B, T, 64, 64), device=dev, dtype=torch.uint8)
L = torch.full((B,), T, device=dev); y = torch.randint(0, 5, (B,), device=dev)
for mb in (1, 2):
    tr = ss.Trainer(m, micro_batches=mb)
    for _ in range(5): tr.step(X, L, R, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): tr.step(X, L, R, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"mb={mb}: enqueue {1000*(t1-t0)/20:.3f} ms/step, total {1000*(t2-t0)/20:.3f} ms/step")
