#!/usr/bin/env python
"""Turn rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected in separate `--pmc` runs as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes) into profiles/knn_traffic.json, the `roofline.traffic`
that bench.py reports for the dominant kernel.

    python tools/pmc_traffic.py <fetch_results.db> <write_results.db> [kernel-substring] > profiles/knn_traffic.json

Corrections applied (and recorded in the JSON): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide (16 B per lane) coalesced read stream -- the kNN kernel's loads are 16 B per lane in 512-B
group segments, so the read side is doubled; WRITE_SIZE is taken as reported (uncalibrated, small here)."""
import json
import sqlite3
import sys


def avg_counter(db, counter, kernel):
    c = sqlite3.connect(db)
    rows = list(c.execute("select avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like ?", (counter, f"%{kernel}%")))
    return (rows[0][0] or 0.0), rows[0][1]


def main():
    fetch_db, write_db = sys.argv[1], sys.argv[2]
    kernel = sys.argv[3] if len(sys.argv) > 3 else "knn_kernel"
    valu_db = sys.argv[4] if len(sys.argv) > 4 else None  # a third pass with --pmc SQ_WAVES SQ_INSTS_VALU: VALU wave-instructions per wave
    f_kib, nf = avg_counter(fetch_db, "FETCH_SIZE", kernel)
    w_kib, nw = avg_counter(write_db, "WRITE_SIZE", kernel)
    out = {
        "kernel": kernel,
        "fetch_size_kib_per_launch_raw": f_kib, "write_size_kib_per_launch_raw": w_kib, "launches_sampled": [nf, nw],
        "fetch_correction": 2.0,
        "hbm_bytes_per_launch": int(2.0 * f_kib * 1024 + w_kib * 1024),
        "note": "FETCH_SIZE x2 (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; "
                "the fabric-side counters include Infinity-Cache hits, so this is memory-side traffic, an upper bound on HBM bytes",
    }
    if valu_db:
        valu, nv = avg_counter(valu_db, "SQ_INSTS_VALU", kernel)
        waves, _ = avg_counter(valu_db, "SQ_WAVES", kernel)
        out.update(valu_wave_instructions_per_launch=valu, waves_per_launch=waves, valu_insts_per_wave=(valu / waves if waves else None), valu_launches_sampled=nv)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
