#!/usr/bin/env python3
"""Summarise the HBM-traffic counter passes of tools/profile.sh into one JSON file.

usage: tools/pmc_traffic.py gpurun_out/prof_<tag> profiles/<round>/<name>.json

FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  On gfx950 FETCH_SIZE tallies 128-byte read
requests at 64 bytes, so read bytes = 2 * FETCH_SIZE * 1024 for wide coalesced streams; WRITE_SIZE is exact for wide
stores.  Both passes run the same command, so dispatches are matched by kernel name and averaged per launch.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from indexr_amd import provenance  # noqa: E402


PASSES = 3  # decode passes in one counter run of tools/profile.sh


def short(name):
    m = re.match(r"(?:void )?(?:ixr::)?(k_\w+)(<[^>]*>)?", name)
    if not m:
        return None
    if m.group(1) in ("k_parse", "k_parse_head", "k_parse_waves"):  # the framing walk, whatever its launch form and window
        return "k_parse"
    t = m.group(2) or ""
    t = t.replace("unsigned int", "u32").replace("unsigned long", "u64")
    t = re.sub(r",\s*\d+u?", "", t)  # k_delta<u64, 256> and k_delta<u64, 1024> are one kernel to the bench
    if m.group(1) == "k_dict_lds":  # the lookups of one pack class, split over several kernels by dictionary size
        return "k_dict_num" + t
    if m.group(1) == "k_str_bases":  # two builds by the size of the length table
        return "k_str_bases"
    return m.group(1) + t


def collect(prof_dir, sub, counter):
    acc = defaultdict(lambda: [0.0, 0])
    files = glob.glob("%s/%s/*/*_counter_collection.csv" % (prof_dir, sub))
    if not files:
        raise SystemExit("no counter csv under %s/%s" % (prof_dir, sub))
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                a = acc[k]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def main():
    prof_dir, out = sys.argv[1], sys.argv[2]
    fetch = collect(prof_dir, "pmc_fetch", "FETCH_SIZE")
    write = collect(prof_dir, "pmc_write", "WRITE_SIZE")
    bench = None
    try:
        with open(prof_dir + "/bench_stats.json") as fh:
            bench = json.loads(fh.read().strip().splitlines()[-1])
    except Exception:
        pass
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0, 0]), write.get(k, [0.0, 0])
        kernels[k] = {
            "launches_counted": max(f[1], w[1]),
            "fetch_size_KiB_per_launch_raw": f[0] / f[1] if f[1] else None,
            "read_bytes_per_launch": 2.0 * 1024.0 * f[0] / f[1] if f[1] else None,
            "write_bytes_per_launch": 1024.0 * w[0] / w[1] if w[1] else None,
        }
        r, wv = kernels[k]["read_bytes_per_launch"], kernels[k]["write_bytes_per_launch"]
        kernels[k]["hbm_bytes_per_launch"] = (r or 0.0) + (wv or 0.0)
        # tools/profile.sh runs the counter passes with --steps 2 --warmup 1: three passes over the table
        kernels[k]["hbm_bytes_per_pass"] = kernels[k]["hbm_bytes_per_launch"] * kernels[k]["launches_counted"] / PASSES
    doc = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); read = 2 * FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB",
        "workload": bench["config"]["workload"] if bench else None,
        "packs": bench["config"]["packs_on_rank0"] if bench else None,
        # the kernels these counters belong to: bench.py quotes the file only while indexr_amd/csrc still hashes to this
        "csrc_hash": provenance.csrc_hash(),
        "launches_per_pass": {k: v["launches_counted"] / PASSES for k, v in kernels.items()},
        "hbm_bytes_per_pass_all_kernels": sum(v["hbm_bytes_per_pass"] for v in kernels.values()),
        "kernels": kernels,
    }
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, v in kernels.items():
        print("%-26s launches %5d  read %10.1f MB  write %10.1f MB" % (
            k, v["launches_counted"], (v["read_bytes_per_launch"] or 0) / 1e6, (v["write_bytes_per_launch"] or 0) / 1e6))


if __name__ == "__main__":
    main()
