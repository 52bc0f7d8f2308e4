"""List-scheduling simulation of the forward blend's heavy items (measured s_memtime lengths) under different orders/granularities."""
import ctypes, os, sys, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moss_amd import scenes, _lib
from tests import helpers as hp
dev = torch.device("cuda:0")
d = hp.inputs_of(scenes.config3(), "scale_rot")
L = _lib.lib()
L.moss_raster_debug_set_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(16 * 1024 * 8, dtype=torch.int64, device=dev)
for _ in range(3): hp.hip_forward(d, dev)
torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(buf.data_ptr())
hp.hip_forward(d, dev); torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(None)
s = buf.cpu().numpy().reshape(-1, 8)          # row = rank*16 + sub
n_rows = s.shape[0]
length = s[:, 0].astype(np.float64)
rank = np.arange(n_rows) // 16
valid = length > 0
print("heavy items", int(valid.sum()), "sum", int(length.sum()), "max", int(length.max()))

def simulate(queues, waves_per_queue=128):
    ends = []
    for q in queues:
        h = [0.0] * waves_per_queue
        heapq.heapify(h)
        for t in q:
            heapq.heappush(h, heapq.heappop(h) + t)
        ends.append(max(h))
    return max(ends), np.mean(ends)

items = [(int(r), float(l)) for r, l in zip(rank[valid], length[valid])]
# (a) current: rank r -> queue r % 8, in rank order
qa = [[l for r, l in items if r % 8 == q] for q in range(8)]
print("current order, 8 queues      : makespan %d (mean queue end %d)" % simulate(qa))
# (b) per queue sorted by true item length (ideal LPT)
qb = [sorted(q, reverse=True) for q in qa]
print("per-queue exact LPT          : makespan %d (mean %d)" % simulate(qb))
# (c) single global queue, exact LPT
print("global queue exact LPT       : makespan %d" % simulate([sorted([l for _, l in items], reverse=True)], 1024)[0])
print("global queue, current order  : makespan %d" % simulate([[l for _, l in items]], 1024)[0])
# (d) items split in two halves of 0.6x length each (8-slot estimate: trips halve, scan stays)
qd = [[x for l in q for x in (0.6 * l, 0.6 * l)] for q in qa]
print("items split in 2 (0.6x each) : makespan %d (mean %d)" % simulate(qd))
qe = [sorted(q, reverse=True) for q in qd]
print("split + exact LPT            : makespan %d (mean %d)" % simulate(qe))
print("lower bounds: max item %d, total/1024 %d" % (length.max(), length.sum() / 1024))
# (f) proxy orders that the scan kernel could produce: tiles sorted by their list length n (exact), or by quarter-octave classes
n_of_rank = {}
for row in np.nonzero(valid)[0]:
    n_of_rank[int(rank[row])] = int(s[row, 1])
ranks = sorted(n_of_rank)
def order_sim(tile_seq, label):
    queues = [[] for _ in range(8)]
    for k, r in enumerate(tile_seq):
        queues[k % 8].extend(float(length[r * 16 + b]) for b in range(16) if length[r * 16 + b] > 0)
    print("%-29s: makespan %d (mean %d)" % ((label,) + simulate(queues)))
order_sim(ranks, "as measured (clz classes)")
order_sim(sorted(ranks, key=lambda r: -n_of_rank[r]), "tiles sorted by n")
def qclass(n):
    b = n.bit_length() - 1
    return (b << 2) | ((n >> max(b - 2, 0)) & 3)
order_sim(sorted(ranks, key=lambda r: (-qclass(n_of_rank[r]), r)), "quarter-octave classes")
# (g) tiles ordered by their MEASURED cost (what the backward could know from the forward of the same frame)
tile_sum = {r: sum(length[r * 16 + b] for b in range(16)) for r in ranks}
tile_max = {r: max(length[r * 16 + b] for b in range(16)) for r in ranks}
def order_sim_per_queue(key, label):
    queues = [[] for _ in range(8)]
    for q in range(8):
        mine = sorted([r for k, r in enumerate(ranks) if k % 8 == q], key=key)       # same tile -> queue assignment, new order inside
        for r in mine:
            queues[q].extend(float(length[r * 16 + b]) for b in range(16) if length[r * 16 + b] > 0)
    print("%-29s: makespan %d (mean %d)" % ((label,) + simulate(queues)))
order_sim_per_queue(lambda r: -tile_sum[r], "queue tiles by measured sum")
order_sim_per_queue(lambda r: -tile_max[r], "queue tiles by measured max")
