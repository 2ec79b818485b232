"""Summarise rocprofv3 SQ counter passes of the pair tile kernel (tools/pmc_pairs.sh) per matrix size.

    python tools/pmc_sq_summary.py gpurun_out/r2/pmc_base profiles/r2_pmc_pairs_base.json

Counter units (MI355X_MICROARCH.md, cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles (4 shader cycles), summed over waves (or SEs for
BUSY); SQ_INSTS_* count wave-instructions.  Derived here:
  valu_issue_share  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES      (share of resident-wave time spent issuing VALU)
  valu_slots_per_simd_cycle = SQ_INSTS_VALU * 2 cyc / (kernel cycles * SIMDs)  (a plain wave64 VALU op holds a SIMD-32 for 2 cycles)
  fma_share         = SQ_INSTS_VALU_FMA_F32 / SQ_INSTS_VALU
  executed_flop_frac = 2*64*SQ_INSTS_VALU_FMA_F32 / (kernel time * 157.3 TF)   (FMA flops only)
"""
import collections
import csv
import glob
import json
import re
import sys


def load(dirpath):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(dirpath + "/*/*_counter_collection.csv"):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "pair_tile_kernel" not in name:
                    continue
                mm = re.search(r"PairCfg<(\w+), (\d+)", name)
                key = f"{mm.group(1)}_MR{mm.group(2)}" if mm else name[:60]
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                acc[key]["VGPR"].append(float(row.get("VGPR_Count", 0) or 0))
                acc[key]["LDS"].append(float(row.get("LDS_Block_Size", 0) or 0))
    dur = collections.defaultdict(list)
    for path in glob.glob(dirpath + "/s/*_kernel_trace.csv"):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "pair_tile_kernel" not in name:
                    continue
                mm = re.search(r"PairCfg<(\w+), (\d+)", name)
                key = f"{mm.group(1)}_MR{mm.group(2)}" if mm else name[:60]
                dur[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    return acc, dur


def main(dirpath, out):
    acc, dur = load(dirpath)
    doc = {"source": f"rocprofv3 --kernel-trace --pmc passes of tools/pmc_pairs.sh ({dirpath}), MI355X; "
                     "last 4 of 6 launches per size averaged; SQ cycle counters are quad-cycles summed over waves",
           "kernels": {}}
    for key in sorted(acc):
        c = {k: sum(v[2:]) / max(len(v[2:]), 1) for k, v in acc[key].items()}
        ms = sorted(dur[key])[len(dur[key]) // 2] if dur.get(key) else None
        d = dict(c)
        d["kernel_ms_unprofiled_trace_median"] = ms
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                      "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM"):
                if k in c:
                    d[k + "_over_WAVE_CYCLES"] = c[k] / wc
        if "SQ_INSTS_VALU" in c:
            for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32"):
                if k in c:
                    d[k + "_over_INSTS_VALU"] = c[k] / c["SQ_INSTS_VALU"]
            if ms:
                cycles = ms * 1e-3 * 2.4e9
                d["valu_issue_cycles_over_simd_cycles_at_2p4GHz"] = c["SQ_INSTS_VALU"] * 2 / (cycles * 1024)
                d["lds_insts_x2p3cyc_over_cu_cycles_at_2p4GHz"] = c.get("SQ_INSTS_LDS", 0) * 2.3 / (cycles * 256)
                if "SQ_INSTS_VALU_FMA_F32" in c:
                    d["executed_fma_flop_frac_of_157TF"] = 128 * c["SQ_INSTS_VALU_FMA_F32"] / (ms * 1e-3) / 157.3e12
        if "GRBM_GUI_ACTIVE" in c and ms:
            d["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9
        doc["kernels"][key] = d
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    for key, d in doc["kernels"].items():
        print(key, {k: (round(v, 4) if isinstance(v, float) and v < 100 else v) for k, v in d.items() if "over" in k or "frac" in k or "clock" in k or k.startswith("kernel_ms") or k in ("VGPR", "LDS")})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
