"""MOSS's whole densification schedule on the headline's form: 17 events (train_ZJU.py:171-186: every 100 iterations from 400 to 2000) with
100 graph replays between them, starting from configs[1] (6 890 Gaussians).  Reports, per event, its cost, the driver allocations inside it
and the memory torch holds -- does the graph pool or the allocator grow without bound over a schedule?
usage: python scripts/densify_17_events.py [events=17] [reserve=1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                # noqa: E402
from moss_amd import scenes                                 # noqa: E402
from moss_amd.host import limit_cpu_threads                 # noqa: E402
from moss_amd.surgery import reserve_workspace              # noqa: E402
from tests.test_gpu_surgery import FormA, _target           # noqa: E402

limit_cpu_threads()
n_events = int(sys.argv[1]) if len(sys.argv) > 1 else 17
reserve = (int(sys.argv[2]) if len(sys.argv) > 2 else 1) != 0
gpu = torch.device("cuda", 0)
scene = scenes.config2()
gt, mask = _target(scenes.config2, gpu)
gT = torch.Generator().manual_seed(1234)
T = torch.eye(3) + 0.05 * torch.randn(scene.P, 3, 3, generator=gT)
a = FormA(scene, gpu, gt, mask, T, degree=3, graph=True)
if reserve:
    reserve_workspace(3072 * 8 * scene.P, gpu)             # the set grows ~5x over 17 events
    a.graphed.reserve_pool(max(2 * 512 * 8 * int(a.ctx.capacity), 64 << 20))
torch.cuda.synchronize(gpu)
mb = lambda x: round(x / 2 ** 20, 1)
print(f"start: P {scene.P}  reserved {mb(torch.cuda.memory_reserved(gpu))} MB  allocated {mb(torch.cuda.memory_allocated(gpu))} MB", flush=True)
for e in range(1, n_events + 1):
    torch.cuda.synchronize(gpu); t0 = time.perf_counter()
    for _ in range(100):
        a.step()
    torch.cuda.synchronize(gpu); dt = time.perf_counter() - t0
    ev = scenes.scripted_densification(a.tensors(), e, gpu, reset_opacity=(e % 6 == 0))
    rep = a.event(ev)
    a.ctx.check_status()
    print(f"event {e:2d}: P {rep['rows_before']:6d} -> {rep['rows_after']:6d}  steps {1e3 * dt / 100:.4f} ms  event {rep['event_ms']:6.2f} ms "
          f"(surgery {rep['surgery_ms']:.2f}, probe {rep['probe_ms']:.2f}, capture {rep['capture_ms']:.2f})  mallocs {rep['device_mallocs_by_phase']} frees {rep['device_frees']}  "
          f"reserved {mb(torch.cuda.memory_reserved(gpu))} MB  allocated {mb(torch.cuda.memory_allocated(gpu))} MB  capacity {a.ctx.capacity}  dropped {a.graphed.dropped_frames}", flush=True)
