timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
timeout 300 python - <<'PY'
import torch, time, sys
sys.path.insert(0, '.')
from knn_cuda import KNN
dev = torch.device('cuda:0')
ref = torch.randn(1, 6890, 3, device=dev); q = torch.randn(1, 100000, 3, device=dev)
k1 = KNN(1, True); k2 = KNN(2, True)
for _ in range(3): k1(ref, q)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): k1(ref, q)
torch.cuda.synchronize(); print("k=1 100k x 6890: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
k2(q, q); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): k2(q, q)
torch.cuda.synchronize(); print("k=2 100k x 100k: %.3f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
PY
