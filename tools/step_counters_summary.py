"""Turn the rocpd databases of the tools/step_counters.py counter passes into one JSON (per DDIM step and per VAE decode).

    python tools/step_counters_summary.py <out.json> <name>=<p_results.db> ...

Every database holds one rocprofv3 --pmc pass (counters cannot all share a pass on gfx950: FETCH_SIZE takes 3 of the 4 TCC
slots, WRITE_SIZE 2).  Dispatches are split into segments at the marker kernel (silu_kernel); one DDIM step =
(segment[2] - segment[1]) / 4 (6 steps minus 2 steps).  FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide
streaming reads at 64 bytes, MI355X_MICROARCH.md section HBM); WRITE_SIZE is taken as is.  Values are in the counters' units."""
import json
import os
import re
import sqlite3
import sys

CLOCK_GHZ = os.environ.get("SDEO_CLOCK_GHZ", "2.1")      # shader clock held under this load (in-kernel s_memtime probe)
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def segments(db):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection order by dispatch_id").fetchall()
    segs, cur = [], defaultdict(lambda: defaultdict(float))
    seen = set()
    for did, kn, cn, v, dur in rows:
        if "silu_kernel" in kn:
            if (did, "m") not in seen:
                seen.add((did, "m"))
                segs.append(cur)
                cur = defaultdict(lambda: defaultdict(float))
            continue
        k = short(kn)
        cur[cn][k] += v
        cur[cn]["__total__"] += v
        if (did, "d") not in seen:
            seen.add((did, "d"))
            cur["duration_ns"][k] += dur
            cur["duration_ns"]["__total__"] += dur
            cur["dispatches"][k] += 1
            cur["dispatches"]["__total__"] += 1
    return segs


def main():
    out_path = sys.argv[1]
    out = {"_how": "rocprofv3 --pmc <counter> --kernel-include-regex sdeo -- python3 tools/step_counters.py 512 vae (eager, one stream); "
                   "per DDIM step = (6-step segment - 2-step segment) / 4; FETCH_SIZE raw KB (double it for bytes: gfx950 counts "
                   "128-B requests at 64 B), WRITE_SIZE KB exact"}
    for spec in sys.argv[2:]:
        name, db = spec.split("=", 1)
        segs = segments(db)
        assert len(segs) >= 3, f"{db}: expected >= 3 marker-separated segments, found {len(segs)}"
        per = {}
        for cn in segs[2]:
            keys = set(segs[2][cn]) | set(segs[1][cn])
            d = {k: (segs[2][cn].get(k, 0.0) - segs[1][cn].get(k, 0.0)) / 4.0 for k in keys}
            tot = d.pop("__total__")
            top = dict(sorted(d.items(), key=lambda kv: -abs(kv[1]))[:12])
            per[cn] = {"per_ddim_step": tot, "top_kernels": top}
        entry = {"per_ddim_step": per}
        if len(segs) >= 4:
            entry["vae_decode"] = {cn: {"total": segs[3][cn]["__total__"],
                                        "top_kernels": dict(sorted(((k, v) for k, v in segs[3][cn].items() if k != "__total__"), key=lambda kv: -abs(kv[1]))[:8])}
                                   for cn in segs[3]}
        out[name] = entry
    # one-screen summary (what bench.py copies into its roofline object)
    try:
        f = out["fetch"]["per_ddim_step"]; w = out["write"]["per_ddim_step"]; m = out["mfma"]["per_ddim_step"]
        ms = m["duration_ns"]["per_ddim_step"] / 1e6
        rd, wr = f["FETCH_SIZE"]["per_ddim_step"] * 2 * 1024, w["WRITE_SIZE"]["per_ddim_step"] * 1024
        busy = m["SQ_VALU_MFMA_BUSY_CYCLES"]["per_ddim_step"]
        mops = m["SQ_INSTS_VALU_MFMA_MOPS_F16"]["per_ddim_step"] * 512
        clk = float(CLOCK_GHZ)
        out["summary"] = {
            "mode": "eager launches on one stream (no hipGraph replay, no side stream), rocprofv3 --pmc, sdeo kernels only",
            "device_ms_per_ddim_step": round(ms, 3), "dispatches_per_ddim_step": m["dispatches"]["per_ddim_step"],
            "l2_fabric_read_GB_per_step": round(rd / 1e9, 3), "l2_fabric_write_GB_per_step": round(wr / 1e9, 3),
            "fabric_GBps": round((rd + wr) / 1e9 / (ms / 1e3), 1), "fabric_frac_of_8TBps": round((rd + wr) / 1e12 / (ms / 1e3) / 8.0, 4),
            "note_bytes": "FETCH_SIZE x 2 + WRITE_SIZE: L2 <-> fabric bytes, i.e. HBM plus Infinity-Cache hits (MI355X_MICROARCH.md section HBM)",
            "mfma_flop_executed_per_step": mops, "mfma_flop_algorithmic_per_step": 2.173e12,
            "mfma_executed_TFLOPs": round(mops / 1e12 / (ms / 1e3), 1), "mfma_executed_frac_of_2500": round(mops / 1e12 / (ms / 1e3) / 2500.0, 4),
            "mfma_algorithmic_frac_of_2500": round(2.173 / (ms / 1e3) / 2500.0, 4),
            "mfma_busy_cycles_per_step": busy,
            "mfma_busy_frac": round(busy / (1024 * ms * 1e-3 * clk * 1e9), 4),
            "mfma_busy_note": f"SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x summed kernel duration x {clk} GHz); GRBM_GUI_ACTIVE reads high on "
                              "dispatches this short (MI355X_MICROARCH.md DVFS), so the denominator uses the dispatch durations"}
        if "vae_decode" in out["mfma"]:
            vm = out["mfma"]["vae_decode"]; vf = out["fetch"]["vae_decode"]; vw = out["write"]["vae_decode"]
            vms = vm["duration_ns"]["total"] / 1e6
            out["summary"]["vae_decode"] = {"device_ms": round(vms, 3),
                                            "l2_fabric_read_GB": round(vf["FETCH_SIZE"]["total"] * 2 * 1024 / 1e9, 3),
                                            "l2_fabric_write_GB": round(vw["WRITE_SIZE"]["total"] * 1024 / 1e9, 3),
                                            "mfma_executed_TFLOPs": round(vm["SQ_INSTS_VALU_MFMA_MOPS_F16"]["total"] * 512 / 1e12 / (vms / 1e3), 1)}
        print(json.dumps(out["summary"], indent=1))
    except KeyError as e:
        print("summary skipped, missing", e)
    json.dump(out, open(out_path, "w"), indent=1)
    for name in out:
        if name.startswith("_") or name == "summary":
            continue
        for cn, v in out[name]["per_ddim_step"].items():
            print(f"{name:12s} {cn:28s} per DDIM step = {v['per_ddim_step']:.6g}")


if __name__ == "__main__":
    main()
