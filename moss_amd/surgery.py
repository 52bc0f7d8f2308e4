"""One densification EVENT on the step that produces the headline number.

MOSS adds and removes Gaussians every 100 iterations between iterations 400 and 2000 and resets the opacities
(``train_ZJU.py:171-186``): ``densify_and_prune`` -> ``densification_postfix`` (``cat_tensors_to_optimizer``) / ``prune_points``
(``_prune_optimizer``), ``reset_opacity`` (``replace_tensor_to_optimizer``) -- ``scene/gaussian_model.py:314-317, 362-454``.  Which
Gaussians it clones, splits or drops is MOSS's decision logic (``:456-620``) and stays there; THIS module carries out the decision on
the objects of the fast step, in the order MOSS does, with one call per event and outside graph capture:

    parameters + both AdamW moments   ``FlatAdamW.append_rows`` / ``prune_rows`` / ``reset_rows`` (new rows: zero moments; reset: zero moments)
    gradient bucket                   ``GradBucket.relayout`` (new offsets, the loss block moves with the tail)
    fused optimizer step              re-armed on the new moment addresses (``FlatAdamW.fuse_into_backward`` again, same names)
    densification statistics          ``DensifyStats.reset(P)`` after an append (densification_postfix :451-454), ``.prune`` after a prune
    binning capacity                  ``RasterContext.relearn_capacity()``: the next forward is synchronous and sizes it for the new set
    captured step                     ``GraphedStep.recapture(probe)``: parameter, moment, bucket and scratch addresses are baked into a graph

A ``reset_opacity`` alone changes no shape and no address: it is applied in place and the captured graph stays valid.
"""
from __future__ import annotations

import gc
import time

import torch

__all__ = ["densification_event"]


def reserve_workspace(nbytes, device):
    """Make torch's caching allocator hold ONE free segment of ``nbytes`` (allocate it, release it -- call after the first graph capture,
    which empties the cache).  A densification event builds new flat parameter / gradient / moment buffers and their gathered sources
    (seven tensors of 236 B per Gaussian) a little larger than the ones it frees, so none of them fits a cached block and every one
    is a hipMalloc (a few hundred microseconds each).  Blocks split off a large cached segment cost nothing and merge back when they
    are freed.  3 KB per Gaussian of the LARGEST set expected is ample; an MI355X has 288 GB."""
    t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    del t


def densification_event(pc, optimizer, *, append=None, prune=None, reset_opacity=False, stats=None, context=None, graphed=None,
                        probe=None, per_gaussian=None, after_surgery=None):
    """Carry out one event.  ``append``: dict with the six tensors of ``densification_postfix`` (``new_xyz, new_features_dc,
    new_features_rest, new_opacities, new_scaling, new_rotation``) or a list of such dicts (MOSS appends twice per event: clones, then
    splits) -- applied first, in order; ``prune``: bool mask over the Gaussians AFTER the appends, True = remove (``prune_points``);
    ``reset_opacity``: last.  ``per_gaussian``: optional dict name -> (P, ...) tensor the CALLER keeps per Gaussian (an LBS transform
    table, cached neighbours): appended rows are taken from ``append[i]["source"]`` (index of the Gaussian each new row derives from)
    and pruned with the mask; the re-indexed dict is returned in the report and -- BEFORE the probe and the re-capture, whose step
    function reads those tables -- handed to ``after_surgery(per_gaussian)``.

    Returns a report: rows before / after, what was re-captured, and the host-side cost of the event in milliseconds (it
    synchronises the device: the event is outside the step's asynchronous flow by nature)."""
    dev = pc._xyz.device
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()

    def counters():
        # the usual suspects when ONE event costs ten times the others: segments allocated from / returned to the driver (hipMalloc /
        # hipFree by torch's caching allocator), a full pass of Python's cyclic collector (45-55 ms in a process that has torch loaded).
        # (Zero in every event since round 6 -- the 40-95 ms outliers were the container's CPU quota: profiles/r06_notes.md section 10)
        st = torch.cuda.memory_stats(dev) if dev.type == "cuda" else {}
        return (int(st.get("segment.all.allocated", 0)), int(st.get("segment.all.freed", 0)), int(gc.get_stats()[2]["collections"]))
    c0 = counters()
    rows_before = int(pc._xyz.shape[0])
    per_gaussian = dict(per_gaussian or {})
    appends = [] if append is None else ([append] if isinstance(append, dict) else list(append))
    shape_changed = False
    for a in appends:
        n_new = int(a["new_xyz"].shape[0])
        if n_new == 0:
            continue
        pc.densification_postfix(a["new_xyz"], a["new_features_dc"], a["new_features_rest"], a["new_opacities"], a["new_scaling"],
                                 a["new_rotation"], optimizer, stats=stats)
        for k, t in list(per_gaussian.items()):
            if "source" not in a:
                raise ValueError("per_gaussian tensors need append['source']: the Gaussian each new row derives from")
            per_gaussian[k] = torch.cat((t, t[a["source"].to(t.device)]), dim=0).contiguous()
        shape_changed = True
    if prune is not None and bool(prune.any()):
        if int(prune.numel()) != int(pc._xyz.shape[0]):
            raise ValueError(f"prune mask of {int(prune.numel())} entries for {int(pc._xyz.shape[0])} Gaussians (it indexes the set AFTER the appends)")
        pc.prune_points(prune, optimizer, stats=stats)
        keep = ~prune.bool()
        for k, t in list(per_gaussian.items()):
            per_gaussian[k] = t[keep.to(t.device)].contiguous()
        shape_changed = True
    if reset_opacity:
        pc.reset_opacity(optimizer)
    recaptured = False
    if after_surgery is not None:
        after_surgery(per_gaussian)

    def lap():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        return time.perf_counter()
    t1 = t2 = t3 = lap()
    m1 = m2 = counters()[0]
    if shape_changed:
        if getattr(pc, "spatially_ordered", False):
            pc.spatially_ordered = False                     # (appended rows sit at the end: index neighbours are no longer spatial neighbours)
        if context is not None:
            context.relearn_capacity()
        if probe is not None:
            probe()
        t2 = t3 = lap()
        m2 = counters()[0]
        if graphed is not None:
            graphed.recapture()
            recaptured = True
            t3 = lap()
    c1 = counters()
    return {"rows_before": rows_before, "rows_after": int(pc._xyz.shape[0]), "recaptured": recaptured,
            "event_ms": round(1e3 * (t3 - t0), 3), "surgery_ms": round(1e3 * (t1 - t0), 3), "probe_ms": round(1e3 * (t2 - t1), 3),
            "capture_ms": round(1e3 * (t3 - t2), 3), "device_mallocs": c1[0] - c0[0], "device_mallocs_by_phase": [m1 - c0[0], m2 - m1, c1[0] - m2], "device_frees": c1[1] - c0[1],
            "gc_full_collections": c1[2] - c0[2], "per_gaussian": per_gaussian}
