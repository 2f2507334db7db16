"""MFMA utilisation per kernel from one rocprofv3 PMC pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
usage: python profiles/summarize_mfma.py m_results.db out.json
SQ_VALU_MFMA_BUSY_CYCLES is reported per shader-engine slice (32 records per dispatch on gfx950) and sums to
(number of MFMA instructions) x (cycles each occupies its SIMD's matrix pipe: 64 for v_mfma_f32_32x32x2_f32); GRBM_GUI_ACTIVE
(8 records per dispatch, one per XCD) is the dispatch's duration in shader-clock cycles.  MfmaUtil = busy cycles /
(duration x 256 CUs x 4 SIMDs) -- the gfx94x formula of rocprof's derived counters, which ROCm 7.2 lacks for gfx950."""
import json
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, counter_name, count(*), sum(counter_value), avg(counter_value) from pmc_events "
                 "group by name, counter_name").fetchall()
k = {}
for name, ctr, n, tot, avg in rows:
    k.setdefault(name, {})[ctr] = (n, tot, avg)
out = {'_about': __doc__.replace('\n', ' '), 'kernels': {}}
for name, v in k.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in v or 'GRBM_GUI_ACTIVE' not in v:
        continue
    n_disp = v['GRBM_GUI_ACTIVE'][0] // 8
    busy = v['SQ_VALU_MFMA_BUSY_CYCLES'][1] / max(n_disp, 1)
    dur = v['GRBM_GUI_ACTIVE'][2]
    if busy <= 0:
        continue
    short = name.split('(')[0].replace('void ', '')
    out['kernels'][short] = {'dispatches': n_disp, 'mfma_busy_cycles_per_dispatch': round(busy),
                             'duration_cycles': round(dur), 'mfma_util': round(busy / (dur * 1024.0), 4)}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for name, v in sorted(out['kernels'].items(), key=lambda kv: -kv[1]['mfma_busy_cycles_per_dispatch'] * kv[1]['dispatches']):
    print(f"{name[:64]:64s} {v['dispatches']:4d} dispatches  MFMA busy {v['mfma_busy_cycles_per_dispatch'] / 1e6:8.2f} Mcyc  "
          f"duration {v['duration_cycles'] / 1e3:8.1f} kcyc  util {v['mfma_util']:.3f}")
