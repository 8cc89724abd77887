"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_{F32|BF16} SQ_INSTS_MFMA SQ_WAVE_CYCLES
 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE).
MfmaUtil follows the gfx94x derived-metric formula rocprof falls back to on gfx950
(MI355X_MICROARCH.md, counters section): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CUs * 4), with
GRBM_GUI_ACTIVE divided by 8 because rocprofv3 reports it summed over the 8 XCDs.  mops_rate is
MFMA_MOPS (512 FLOP each) per XCD-cycle against the 16 per cycle an XCD can retire (32 CUs x 4 SIMDs x
64 FLOP/clk for f32; x16 that for bf16).  usage: pmc_mfma.py <pmc_dir> <out.json> [f32|bf16]"""
import csv, glob, sys, json, collections

d, out = sys.argv[1], sys.argv[2]
kind = sys.argv[3] if len(sys.argv) > 3 else "f32"
mops = "SQ_INSTS_VALU_MFMA_MOPS_" + kind.upper()
per_cycle = 16.0 if kind == "f32" else 256.0
f = (glob.glob(d + '/*counter_collection.csv') + glob.glob(d + '/*/*counter_collection.csv'))[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'crnn::' not in r['Kernel_Name'] or 'crnn::warm_' in r['Kernel_Name']:
        continue
    key = (r['Kernel_Name'].split('(')[0].replace('crnn::', ''), int(r['Grid_Size']))
    acc.setdefault(key, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
res = []
for (name, grid), c in acc.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    gui = m.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    row = {"kernel": name, "grid_threads": grid, "launches": len(next(iter(c.values()))),
           "counters_mean": {k: round(v, 1) for k, v in m.items()}}
    if gui > 0 and m.get('SQ_INSTS_MFMA', 0) > 0:
        row["MfmaUtil"] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 256 * 4), 4)
        row["mops_rate"] = round(m.get(mops, 0.0) / (gui * 8) / per_cycle, 4)
        row["lds_bank_conflict_frac"] = round(m.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(1.0, m.get('SQ_ACTIVE_INST_LDS', 1.0)), 4)
        print("%-28s grid %8d  MfmaUtil %.3f  mops/peak %.3f  wait_any/wave_cycles %.2f" %
              (name, grid, row["MfmaUtil"], row["mops_rate"], m.get('SQ_WAIT_INST_ANY', 0) / max(1.0, m.get('SQ_WAVE_CYCLES', 1))))
    res.append(row)
json.dump(res, open(out, 'w'), indent=1)
