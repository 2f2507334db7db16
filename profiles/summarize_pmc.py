"""HBM traffic per launch from two rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs).
usage: python profiles/summarize_pmc.py fetch.db write.db out.json
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the
bytes of wide (16 B/lane) coalesced reads -> the read bytes of kernels that read with 16-byte loads are doubled
(WIDE below); WRITE_SIZE matched known byte counts (conv1 writes exactly its 636 MB output) and is used as is."""
import json
import sqlite3
import sys

WIDE = ('gemm_f32_kernel', 'ffn_pc_kernel', 'sqz_stage_kernel', 'ffn_fused_kernel', 'rowgemm_kernel', 'attention_kernel', 'dwconv_ln_silu_kernel',
        'layernorm256_kernel')


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), avg(counter_value) from pmc_events where counter_name = ? group by name",
                     (counter,)).fetchall()
    return {r[0]: (r[1], r[2] * 1024.0) for r in rows}


fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {'_about': __doc__.replace('\n', ' '), 'kernels': {}}
for name in fetch:
    n, rd = fetch[name]
    wr = write.get(name, (0, 0.0))[1]
    short = name.split('(')[0].replace('void ', '')
    wide = any(w in name for w in WIDE)
    rdc = rd * 2 if wide else rd
    out['kernels'][short] = {'launches_profiled': n, 'fetch_size_bytes_raw': round(rd), 'read_bytes_corrected': round(rdc),
                             'write_bytes': round(wr), 'hbm_bytes': round(rdc + wr)}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
for k, v in sorted(out['kernels'].items(), key=lambda kv: -kv[1]['hbm_bytes'])[:12]:
    print(f"{k[:70]:70s} {v['launches_profiled']:5d} launches  read {v['read_bytes_corrected'] / 1e6:9.2f} MB  write {v['write_bytes'] / 1e6:9.2f} MB")
