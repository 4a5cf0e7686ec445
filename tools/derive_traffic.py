#!/usr/bin/env python3
"""profiles/k_gt_hbm_traffic.json from the four FETCH_SIZE / WRITE_SIZE rocprofv3 csv files.
usage: python tools/derive_traffic.py <dir with pmc_{fetch,write}_size_{streaming,census}.csv> [rows per dispatch] > profiles/k_gt_hbm_traffic.json"""
import csv, json, sys, collections

d = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 311296  # rows per dispatch (bench.py SHAPES["c3"])
raw = collections.defaultdict(dict)
for ctr, tagc in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    for path, kernels in (("streaming", ("k_stream",)), ("census", ("k_gt", "k_count_eol"))):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open("%s/pmc_%s_size_%s.csv" % (d, tagc, path))):
            for k in kernels:
                if "bvcf_dev::%s(" % k in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                    acc[k].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            raw[k][ctr] = sum(v) / len(v)
out = {
    "_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with nothing but --output-format csv) on "
                "`python bench.py --steps 1 --warmup 0 --blocks 3 --no-cpu-baseline --no-e2e --path {1,2}` (c3, %d rows per dispatch), " % rows +
                "tools/refresh_profiles.sh + tools/derive_traffic.py. Raw counters are KiB per dispatch, mean over dispatches. "
                "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly half of a 16 B/lane streaming read, "
                "so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.",
    "rows_per_dispatch": rows,
    "raw_KiB_per_dispatch": raw,
    "c3": {},
}
c3 = out["c3"]
for k, v in raw.items():
    rd, wr = 2 * v["FETCH_SIZE"] * 1024, v["WRITE_SIZE"] * 1024
    c3[k + "_read_bytes_per_launch"] = rd
    c3[k + "_write_bytes_per_launch"] = wr
    c3[k + "_traffic_bytes_per_launch"] = rd + wr
    c3[k + "_traffic_bytes_per_launch_per_row"] = (rd + wr) / rows
c3["traffic_bytes_per_launch_per_row"] = c3["k_gt_traffic_bytes_per_launch_per_row"]
c3["algorithmic"] = {"k_stream_read_bytes_per_row": "the whole line, ~10166 B", "k_gt_read_bytes_per_row": 10016,
                     "write_bytes_per_row": 672}
# the general-format profile (c5), k_stream_gen: optional pair of files from the same script
import os
if os.path.exists("%s/pmc_fetch_size_c5.csv" % d):
    rows5 = 98304  # bench.py SHAPES["c5"]
    v = {}
    for ctr, tagc in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open("%s/pmc_%s_size_c5.csv" % (d, tagc)))
                if "bvcf_dev::k_stream_gen(" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        v[ctr] = sum(vals) / len(vals)
    rd, wr = 2 * v["FETCH_SIZE"] * 1024, v["WRITE_SIZE"] * 1024
    out["raw_KiB_per_dispatch"]["k_stream_gen (c5, %d rows per dispatch)" % rows5] = v
    out["c5"] = {"rows_per_dispatch": rows5, "k_stream_gen_read_bytes_per_launch": rd, "k_stream_gen_write_bytes_per_launch": wr,
                 "k_stream_gen_traffic_bytes_per_launch": rd + wr,
                 # (the key bench.py looks up for the streaming path's dominant kernel)
                 "k_stream_traffic_bytes_per_launch_per_row": (rd + wr) / rows5,
                 "algorithmic": {"k_stream_gen_read_bytes_per_row": "the whole line, ~24361 B"}}
# sites-only (c2: the census + k_sites2 chain) and configs[3] (c4: k_stream + followers): every kernel of the chain
for prof, rows_p in (("c2", 1000000), ("c4", 262144)):
    if not os.path.exists("%s/pmc_fetch_size_%s.csv" % (d, prof)):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr, tagc in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        for r in csv.DictReader(open("%s/pmc_%s_size_%s.csv" % (d, tagc, prof))):
            if "bvcf_dev::" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                acc[r["Kernel_Name"].split("(")[0].split("::")[-1]][ctr].append(float(r["Counter_Value"]))
    sec = {"rows_per_dispatch": rows_p, "kernels": {}}
    tot_r = tot_w = 0.0
    for k, v in acc.items():
        rd = 2 * sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1) * 1024
        wr = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1) * 1024
        sec["kernels"][k] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr}
        tot_r += rd
        tot_w += wr
    sec["chain_read_bytes_per_block"] = tot_r
    sec["chain_write_bytes_per_block"] = tot_w
    dom = ("k_sites2p" if "k_sites2p" in sec["kernels"] else "k_sites2") if prof == "c2" else "k_stream"
    if dom in sec["kernels"]:
        t = sec["kernels"][dom]["read_bytes_per_launch"] + sec["kernels"][dom]["write_bytes_per_launch"]
        sec["traffic_bytes_per_launch_per_row" if prof == "c2" else "k_stream_traffic_bytes_per_launch_per_row"] = t / rows_p
    out[prof] = sec
print(json.dumps(out, indent=1))
