"""Aggregate rocprofv3 --pmc counter_collection CSVs (one per pass) into per-kernel and per-stage JSON.

usage: summarize_pmc.py <dir with *counter_collection.csv> <out.json> [<stage_out.json>]
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (section on FETCH_SIZE / WRITE_SIZE): both counters are in KiB; on gfx950
FETCH_SIZE reports HALF of the bytes of WIDE COALESCED STREAMING reads (16 B per lane, global_load and LDS-DMA alike), so for the
kernels whose reads are of that class (STREAMING below: parameter / SH / record / image streams) it is doubled:
    hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE.
Other access widths are uncalibrated in the guide.  For the GATHER kernels (reads = one 16- or 48/64-byte record per lane from
unrelated lines, or 8-byte keys) the doubled figure is only an UPPER bound: they are reported as the interval
    hbm_bytes_interval = [FETCH_SIZE + WRITE_SIZE, 2 * FETCH_SIZE + WRITE_SIZE]
with hbm_bytes_per_launch = its upper end and "traffic_bound": "upper" (VERDICT r2, weak 9 / next 8).  `hbm_bytes_raw` keeps the
undoubled sum for every kernel.
"""
import collections, csv, glob, json, re, sys
src, out = sys.argv[1], sys.argv[2]
stage_out = sys.argv[3] if len(sys.argv) > 3 else None
KERNELS = ["preprocess_forward_kernel", "preprocess_backward_kernel", "scan_kernel", "scatter_kernel", "chunk_sort_kernel",
           "merge_gather_kernel", "blend_forward_wave_kernel", "blend_backward_wave_kernel", "blend_forward_kernel",
           "blend_backward_kernel", "clear_words_kernel", "ssim_pass1_kernel", "ssim_pass2_kernel", "loss_finish_kernel",
           "adamw_kernel", "adamw_tick_kernel", "activate_forward_kernel", "activate_backward_kernel", "mark_visible_kernel"]
STAGES = {"preprocess_fwd": ["preprocess_forward_kernel"], "scan": ["scan_kernel"], "scatter": ["scatter_kernel"],
          "chunk_sort": ["chunk_sort_kernel"], "merge_gather": ["merge_gather_kernel"], "blend_fwd": ["blend_forward_wave_kernel", "blend_forward_kernel"],
          "blend_bwd": ["blend_backward_wave_kernel", "blend_backward_kernel"], "preprocess_bwd": ["preprocess_backward_kernel"]}


STREAMING = {"preprocess_forward_kernel", "blend_forward_wave_kernel", "blend_backward_wave_kernel", "ssim_pass1_kernel", "ssim_pass2_kernel",
             "ssim_fused_kernel", "adamw_kernel", "activate_forward_kernel", "activate_backward_kernel", "clear_words_kernel"}
# everything else that reads by Gaussian index or sorts 8-byte keys: scatter, chunk_sort, tile_sort, merge_gather, preprocess_backward
KERNELS += ["ssim_fused_kernel", "tile_sort_kernel"]


def short(name):
    for k in KERNELS:
        if re.search(r"\b" + k + r"\b", name):
            return k
    return None


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(src + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
for k, c in res.items():
    c["launches_sampled"] = max(len(v) for v in agg[k].values())
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        c["hbm_bytes_raw"] = int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        c["traffic_bound"] = "calibrated (streaming reads: 2 x FETCH_SIZE)" if k in STREAMING else "upper"
        if k not in STREAMING:
            c["hbm_bytes_interval"] = [c["hbm_bytes_raw"], c["hbm_bytes_per_launch"]]
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
if stage_out:
    st = {}
    for s, ks in STAGES.items():
        present = [res[k] for k in ks if k in res and "hbm_bytes_per_launch" in res[k]]
        if present:
            st[s] = {"hbm_bytes_per_launch": sum(p["hbm_bytes_per_launch"] for p in present),
                     "hbm_bytes_raw": sum(p["hbm_bytes_raw"] for p in present), "kernels": [k for k in ks if k in res],
                     "traffic_bound": "upper" if any(p["traffic_bound"] == "upper" for p in present) else present[0]["traffic_bound"],
                     "hbm_bytes_interval": [sum(p["hbm_bytes_raw"] if p["traffic_bound"] == "upper" else p["hbm_bytes_per_launch"] for p in present),
                                            sum(p["hbm_bytes_per_launch"] for p in present)],
                     # reads and writes apart (a kernel can be well over its algorithmic bytes by DESIGN in one direction only: the
                     # backward blend writes a 48-byte gradient record per (entry, block) pair where the reference adds atomically)
                     "hbm_write_bytes": int(sum(p["WRITE_SIZE"] for p in present) * 1024),
                     "hbm_read_bytes": int(sum(2 * p["FETCH_SIZE"] for p in present) * 1024),
                     "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), eager-launch bench.py --graph 0 --steps 30 --warmup 10"}
            # vector-instruction issue (round 6, what bench.py's per-stage `bound` is read from): a SIMD issues at most one VALU instruction
            # of a wave64 per 4 cycles, so SQ_INSTS_VALU x 4 / (1024 SIMDs x kernel cycles) is the share of the kernel's issue slots that
            # held one, AVERAGED over the SIMDs (a kernel whose busiest SIMDs are full reads well below 1: the blend kernels).  Kernel
            # cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs), of the same counter pass.
            k0 = res[[k for k in ks if k in res][0]]
            if "SQ_INSTS_VALU" in k0 and k0.get("GRBM_GUI_ACTIVE", 0) > 0:
                cyc = k0["GRBM_GUI_ACTIVE"] / 8.0
                st[s].update(insts_valu_per_launch=int(k0["SQ_INSTS_VALU"]), insts_salu_per_launch=int(k0.get("SQ_INSTS_SALU", 0)),
                             kernel_cycles=int(cyc), valu_issue_frac=round(k0["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc), 4))
                if k0.get("SQ_WAVE_CYCLES", 0) > 0 and "SQ_WAIT_ANY" in k0:
                    st[s]["wave_cycles_waiting_frac"] = round(k0["SQ_WAIT_ANY"] / k0["SQ_WAVE_CYCLES"], 4)
    # stamp: the kernel sources these counters were measured on (bench.py emits `traffic` only while this matches the checkout)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    st["csrc_sha256"] = bench.csrc_sha256()
    json.dump(st, open(stage_out, "w"), indent=1, sort_keys=True)
for k, c in sorted(res.items()):
    print(k.ljust(30), {n: (round(v) if isinstance(v, float) else v) for n, v in c.items()})
