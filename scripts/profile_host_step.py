"""Where the HOST time of an eager step goes: cProfile around the timed steps of one bench caller variant (the harness is bench.py's own).
usage: python scripts/profile_host_step.py [variant=patched_moss_pattern_one_call_loss]"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "patched_moss_pattern_one_call_loss"
orig = bench.Harness.time_steps
seen = []


def spy(self, n, barrier=None):
    if self.use_graph or n < 50:                            # the headline's harness replays a graph: not what is profiled here
        return orig(self, n, barrier)
    pr = cProfile.Profile()
    pr.enable()
    out = orig(self, n, barrier)
    pr.disable()
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(28)
    pstats.Stats(pr, stream=buf).sort_stats("cumtime").print_stats(45)
    print(f"==== {variant}: {n} eager steps in {1e3 * out[0]:.1f} ms ({1e3 * out[0] / n:.4f} ms per step; under the profiler)", file=sys.stderr)
    print(buf.getvalue(), file=sys.stderr)
    return out


bench.Harness.time_steps = spy
bench.main(["--callers-only", variant, "--no-cpu-baseline"])
