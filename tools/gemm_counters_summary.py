"""rocpd databases of tools/gemm_counters.py (one per rocprofv3 --pmc pass) -> one JSON, one entry per (kernel, grid):

    python tools/gemm_counters_summary.py <out.json> <db> [<db> ...]

Per entry: dispatches, mean duration, mean of every counter over the dispatches (the first one of each group is dropped:
it sets function attributes and runs cold), and derived ratios.  Units as the guide states them (MI355X_MICROARCH.md,
constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES
cycles summed over SIMDs, FETCH_SIZE raw KB (x2 for bytes of wide streaming reads on gfx950), WRITE_SIZE KB."""
import json
import re
import sqlite3
import sys
from collections import defaultdict

CLK_GHZ = 2.1


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:100]


def main():
    out_path = sys.argv[1]
    groups = defaultdict(lambda: {"counters": defaultdict(list), "dur": {}, "meta": None})
    for db in sys.argv[2:]:
        c = sqlite3.connect(db)
        rows = c.execute("select dispatch_id, kernel_name, grid_size_x, grid_size_y, grid_size_z, workgroup_size, lds_block_size, vgpr_count, "
                         "accum_vgpr_count, sgpr_count, counter_name, value, duration from counters_collection "
                         "where kernel_name like '%sdeo::%' order by dispatch_id").fetchall()
        first = {}
        for did, kn, gx, gy, gz, wg, lds, vg, ag, sg, cn, v, dur in rows:
            key = f"{short(kn)} grid=({gx // max(wg, 1)},{gy},{gz}) wg={wg}"
            if key not in first:
                first[key] = did
            if did == first[key]:
                continue                                  # cold launch
            g = groups[key]
            g["counters"][cn].append(v)
            g["dur"][(db, did)] = dur
            g["meta"] = {"lds_bytes": lds, "vgpr": vg, "agpr": ag, "sgpr": sg, "workgroups": (gx // max(wg, 1)) * gy * gz}
    out = {"_how": "rocprofv3 --pmc <set> --kernel-include-regex sdeo -- python3 tools/gemm_counters.py; per (kernel, grid) means; "
                   "first dispatch of each group dropped; clock for derived fractions %.1f GHz" % CLK_GHZ}
    for key, g in sorted(groups.items()):
        durs = list(g["dur"].values())
        e = {"dispatches": len(durs), "avg_ns": sum(durs) / max(len(durs), 1), **(g["meta"] or {})}
        cm = {cn: sum(v) / len(v) for cn, v in g["counters"].items()}
        e["counters"] = cm
        d = {}
        wc = cm.get("SQ_WAVE_CYCLES")
        if wc:
            for cn in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if cn in cm:
                    d[cn + "/WAVE_CYCLES"] = cm[cn] / wc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cm:
            d["mfma_busy_frac"] = cm["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * e["avg_ns"] * CLK_GHZ)
        if "SQ_LDS_BANK_CONFLICT" in cm and cm.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict/idx_active"] = cm["SQ_LDS_BANK_CONFLICT"] / cm["SQ_LDS_IDX_ACTIVE"]
        if "SQ_LDS_BANK_CONFLICT" in cm and cm.get("SQ_INSTS_LDS"):
            d["lds_conflict_cycles_per_lds_inst"] = cm["SQ_LDS_BANK_CONFLICT"] / cm["SQ_INSTS_LDS"]
        if "FETCH_SIZE" in cm:
            d["hbm_side_read_MB"] = cm["FETCH_SIZE"] * 2 * 1024 / 1e6
        if "WRITE_SIZE" in cm:
            d["hbm_side_write_MB"] = cm["WRITE_SIZE"] * 1024 / 1e6
        e["derived"] = d
        out[key] = e
    json.dump(out, open(out_path, "w"), indent=1)
    for k, e in out.items():
        if k.startswith("_"):
            continue
        print(f"{k}\n    n={e['dispatches']} avg={e['avg_ns'] / 1e3:.1f} us  " + "  ".join(f"{a}={b:.3g}" for a, b in e["derived"].items()))


if __name__ == "__main__":
    main()
